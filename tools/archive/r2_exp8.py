"""Round-2 experiment 8 (one GPU): wavefront target of the elasticity marching kernel (el_fast_waves), rows per column.
usage: python tools/r2_exp8.py"""
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
cfgs = [dict(el_fast_waves=w, el_fast_ry=ry) for ry in (2, 4) for w in (1152, 2048, 2304, 2880, 4096, 6144)]
for n in ((256, 64, 64), (512, 128, 128)):
    g = L.make_grid(3, n, (1.0, 0.25, 0.25))
    nodes = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    res = {i: [] for i in range(len(cfgs))}
    for rnd in range(3):
        for i, cfg in enumerate(cfgs):
            for k, v in cfg.items():
                L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), v))
            ms = C.c_double()
            L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 30, C.byref(ms)))
            res[i].append(ms.value)
    for i, cfg in enumerate(cfgs):
        m = float(np.median(res[i]))
        print(f"{n} {cfg}: {m*1e3:.1f} us/apply, {48*nodes/m/1e6:.0f} GB/s alg ({48*nodes/m/8e9:.3f})", flush=True)
