"""Round-2 experiment 17 (one GPU): elasticity block apply (k_el_fast) by grid size and wavefront target — does the wave-count rule of
el_fast_geometry hold away from config 5 and 512x128x128?  usage: python tools/r2_exp17.py"""
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
sizes = [(256, 64, 64), (384, 96, 96), (500, 100, 100), (512, 128, 128), (640, 160, 160), (128, 128, 128), (200, 200, 200), (256, 256, 256), (320, 320, 320), (400, 200, 100)]
targets = [0, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
for n in sizes:
    g = L.make_grid(3, n, (1.0, n[1] / n[0], n[2] / n[0]))
    nodes = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    res = {t: [] for t in targets}
    for rnd in range(3):
        for t in targets:
            L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_fast_waves", t))
            ms = C.c_double()
            L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 20, C.byref(ms)))
            res[t].append(ms.value)
    print(f"{n}: " + "  ".join(f"{t}: {np.median(res[t])*1e3:.1f} us ({48*nodes/np.median(res[t])/8e9:.3f})" for t in targets), flush=True)
L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_fast_waves", 0))
