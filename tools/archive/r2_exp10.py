"""Round-2 experiment 10 (one GPU): k_el_fast marching part only at 512 x 128 x 128 and config 5 — non-temporal stores / loads, the y hand-off,
rows per wave and the wavefront target, interleaved rounds in one process."""
import ctypes as C, sys, itertools
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
el = L.Elasticity(); el.E, el.nu = 210e9, 0.3
part = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variants = [dict(el_fast_nt=nt, el_fast_yx=yx, el_fast_ry=ry, el_fast_waves=w)
            for nt, yx, ry, w in ((0, 1, 2, 4096), (1, 1, 2, 4096), (2, 1, 2, 4096), (3, 1, 2, 4096), (0, 0, 2, 4096), (3, 0, 2, 4096),
                                  (3, 1, 2, 2048), (3, 1, 2, 3072), (3, 1, 4, 2048), (3, 1, 1, 6144))]
for n in ((512, 128, 128), (256, 64, 64)):
    g = L.make_grid(3, n, (1.0, 0.25, 0.25))
    res = {}
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_debug_part", part))
    for rnd in range(3):
        for k, v in enumerate(variants):
            for key, val in v.items():
                L.check(ctx, lib.pdegpu_set_tunable(ctx, key.encode(), val))
            ms = C.c_double()
            L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 30, C.byref(ms)))
            res.setdefault(k, []).append(ms.value * 1e3)
    for k, v in enumerate(variants):
        print(n, "part", part, v, round(float(np.median(res[k])), 1), "us", flush=True)
