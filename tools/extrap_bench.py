"""Scratch: CG iterations per transient solve with / without the time-extrapolated initial guess."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
from pdegpu import heat
import numpy as np
lib = L.load_library(); ctx = L.context(0)
outs = {}
for ex in (0, 1):
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"extrapolate", ex))
    for name, fn in (("config 2 (40 of 200 steps)", lambda: heat._solve_heat_2d_raw(1.0, 1.0, 1024, 1024, 1.0, 0.0, 20.0, 0.0025, 40)),
                     ("3-D 64^3, 40 steps", lambda: heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 64, 64, 64, 1.0, 0.0, 20.0, 0.0025, 40)),
                     ("3-D 128^3 rod T_left/T_right, 20 steps", lambda: heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 128, 128, 128, 1.0, 0.0, 20.0, 0.0025, 20, T_left=100.0, T_right=0.0))):
        t0 = time.perf_counter(); f = fn(); dt = time.perf_counter() - t0
        print(f"extrapolate={ex}: {name}: {dt:.3f} s, {f.meta['cg_iterations']} its", flush=True)
        if ex == 0: outs[name] = f.values
        else: print(f"     max rel diff vs extrapolate=0: {np.abs(f.values - outs[name]).max() / np.abs(outs[name]).max():.2e}", flush=True)
