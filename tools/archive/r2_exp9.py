"""Round-2 experiment 9 (one GPU): which part of k_el_fast takes the time — marching workgroups only, boundary workgroups only, both."""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
el = L.Elasticity(); el.E, el.nu = 210e9, 0.3
for n in ((256, 64, 64), (512, 128, 128)):
    g = L.make_grid(3, n, (1.0, 0.25, 0.25))
    nodes = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    res = {}
    for rnd in range(3):
        for part in (0, 1, 2, 3):
            L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_debug_part", part % 3))
            L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_bnd_last", 1 if part == 3 else 0))
            ms = C.c_double()
            L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 30, C.byref(ms)))
            res.setdefault(part, []).append(ms.value * 1e3)
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_debug_part", 0))
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_bnd_last", 0))
    print(n, {("both", "marching only", "boundary only", "both, boundary workgroups last")[k]: round(float(np.median(v)), 1) for k, v in res.items()}, "us", flush=True)
