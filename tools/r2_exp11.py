import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path("/root/repo")
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
el = L.Elasticity(); el.E, el.nu = 210e9, 0.3
for n in ((512, 128, 128), (256, 64, 64), (511, 128, 128)):
    g = L.make_grid(3, n, (1.0, 0.25, 0.25))
    res = {}
    for rnd in range(3):
        for part in (1, 4):
            L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_debug_part", part))
            ms = C.c_double()
            L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 30, C.byref(ms)))
            res.setdefault(part, []).append(ms.value * 1e3)
    print(n, {k: round(float(np.median(v)), 1) for k, v in res.items()}, flush=True)
