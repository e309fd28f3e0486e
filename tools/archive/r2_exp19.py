"""Round-2 experiment 19 (one GPU): elasticity block apply at HBM-only sizes, fine sweep of the wavefront target (= z-chunk count): is there a
fit structure like the heat kernels' (workgroups per resident round)?  usage: python tools/r2_exp19.py"""
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
for n in ((512, 128, 128), (640, 160, 160), (256, 256, 256), (320, 320, 320)):
    g = L.make_grid(3, n, (1.0, n[1] / n[0], n[2] / n[0]))
    nodes = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    ntx = (n[0] + 1 - 1 + 61) // 62
    nyg = ((n[1] + 1 - 2 + 1) // 2 + 3) // 4
    cols = ntx * nyg * 4
    planes = n[2] + 1 - 2
    seen = {}
    for nzc in list(range(1, 41)) + [48, 56, 64, 80, 96, 128]:
        zlen = (planes + nzc - 1) // nzc
        c = (planes + zlen - 1) // zlen
        if c in seen or zlen < 2:
            continue
        target = cols * (nzc - 1) + 1              # smallest target that gives this chunk count
        L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_fast_waves", target))
        t = []
        for rnd in range(2):
            ms = C.c_double()
            L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 10, C.byref(ms)))
            t.append(ms.value)
        seen[c] = min(t)
        print(f"{n} chunks {c:3d} zlen {zlen:3d} workgroups {cols * c // 4:6d}: {min(t)*1e3:7.1f} us ({48*nodes/min(t)/8e9:.3f})", flush=True)
L.check(ctx, lib.pdegpu_set_tunable(ctx, b"el_fast_waves", 0))
