"""Round-2 experiment 12 (one GPU): elasticity apply with the y- / z-face nodes inside the marching kernel (el_fast_yzface) against the class-table
pass over all boundary nodes, whole launch, interleaved rounds."""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
el = L.Elasticity(); el.E, el.nu = 210e9, 0.3
for n in ((256, 64, 64), (512, 128, 128), (500, 100, 100), (128, 128, 128)):
    g = L.make_grid(3, n, (1.0, 0.25, 0.25))
    res = {}
    for rnd in range(3):
        for npl, pf, ry in ((1, 0, 2), (1, 1, 2), (1, 1, 1), (2, 0, 1), (2, 1, 1), (2, 1, 2)):
            for k, v in (("el_fast_npl", npl), ("el_fast_nt", pf), ("el_fast_ry", ry)):
                L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), v))
            ms = C.c_double()
            L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 30, C.byref(ms)))
            res.setdefault((npl, pf, ry), []).append(ms.value * 1e3)
    print(n, {f"nodes/lane={k[0]} nt-store={k[1]} ry={k[2]}": round(float(np.median(v)), 1) for k, v in res.items()}, "us", flush=True)
