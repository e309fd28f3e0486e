"""Elasticity PCG per-iteration time through the tool's solver: single-reduction form (el_cg_variant = 1: update pass + apply) vs classic (p-update, apply, K2),
interleaved; config 5 and config 5 doubled per axis.  usage: python tools/r3_el_iter.py"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L

lib = L.load_library()
ctx = L.context(0)
el = L.Elasticity()
el.E, el.nu = 210e9, 0.3
el.body[2] = -76518.0
el.rtol = 1e-8
for n, Ls in (((256, 64, 64), (1.0, 0.25, 0.25)), ((512, 128, 128), (1.0, 0.25, 0.25))):
    g = L.make_grid(3, n, Ls)
    N = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    field = np.empty(N)
    ref = None
    for rnd in range(2 if n[0] == 256 else 1):
        for variant in (1, 0):
            L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_cg_variant", variant))
            st = L.Stats()
            L.check(ctx, lib.pdegpu_elasticity_solve(ctx, g, el, L.dptr(field), None, st))
            if ref is None:
                ref = field.copy()
            print(f"{n} variant {variant}: {st.cg_iterations} iterations, {st.cg_ms:.1f} ms in the PCG = {st.cg_ms / st.cg_iterations * 1e3:.1f} us per iteration, "
                  f"rel residual {st.final_rel_residual:.2e}, field vs first run {np.abs(field - ref).max() / np.abs(ref).max():.2e}", flush=True)
L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_cg_variant", 1))
