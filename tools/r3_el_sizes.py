"""Elasticity block apply (k_el_fast) by grid size: non-temporal stores on / off (el_fast_nt) x wavefront target (el_fast_waves), interleaved rounds, medians;
fraction of the 8 TB/s HBM peak at 48 B per node.  usage: python tools/r3_el_sizes.py"""
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
sizes = [(512, 128, 128), (640, 160, 160), (256, 256, 256), (320, 320, 320), (768, 192, 192)]
variants = [(nt, w) for nt in (1, 0) for w in (8192, 16384, 32768, 65536, 131072)]
for n in sizes:
    g = L.make_grid(3, n, (1.0, n[1] / n[0], n[2] / n[0]))
    nodes = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    res = {v: [] for v in variants}
    for rnd in range(3):
        for v in variants:
            for k, x in zip((b"el_fast_nt", b"el_fast_waves"), v):
                L.check(ctx, lib.pdegpu_set_tunable(ctx, k, x))
            ms = C.c_double()
            L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 20, C.byref(ms)))
            res[v].append(ms.value)
    print(f"{n}: " + "  ".join(f"nt{v[0]} w{v[1]}: {np.median(res[v])*1e3:.1f} us ({48*nodes/np.median(res[v])/8e9:.3f})" for v in variants), flush=True)
