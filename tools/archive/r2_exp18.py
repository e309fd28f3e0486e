"""Round-2 experiment 18 (one GPU): elasticity block apply beyond the Infinity Cache — rows per column and the LDS y-hand-off change the register
footprint (RY 1 without the hand-off: 126 VGPRs = 4 waves per SIMD; the default RY 2 with it: 162 = 3).  usage: python tools/r2_exp18.py"""
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
cfgs = [dict(el_fast_ry=2, el_fast_yx=1), dict(el_fast_ry=1, el_fast_yx=0), dict(el_fast_ry=1, el_fast_yx=1), dict(el_fast_ry=2, el_fast_yx=0)]
waves = [0, 16384, 32768]
for n in ((256, 64, 64), (512, 128, 128), (640, 160, 160), (256, 256, 256), (320, 320, 320)):
    g = L.make_grid(3, n, (1.0, n[1] / n[0], n[2] / n[0]))
    nodes = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    res = {}
    for rnd in range(3):
        for i, cfg in enumerate(cfgs):
            for w in waves:
                for k, v in dict(cfg, el_fast_waves=w).items():
                    L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), v))
                ms = C.c_double()
                L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 12, C.byref(ms)))
                res.setdefault((i, w), []).append(ms.value)
    for i, cfg in enumerate(cfgs):
        print(f"{n} {cfg}: " + "  ".join(f"waves {w}: {np.median(res[(i, w)])*1e3:.1f} us ({48*nodes/np.median(res[(i, w)])/8e9:.3f})" for w in waves), flush=True)
for k, v in dict(el_fast_ry=2, el_fast_yx=1, el_fast_waves=0).items():
    L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), v))
