"""Time-to-solution of the largest single-GPU problems: config 4's grid (1024 x 512 x 512, steady, rtol 1e-8) and a 512 x 128 x 128
elasticity cantilever (config 5 scaled by 2 per axis, rtol 1e-8)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import numpy as np
from pdegpu import heat, elasticity
t0 = time.perf_counter()
f = heat._solve_heat_3d_raw(2.0, 1.0, 1.0, 1024, 512, 512, 1.0, 0.0, 0.0, 0.1, 1, steady=True, source_type="constant", source_value=1.0, rtol=1e-8)
dt = time.perf_counter() - t0
print(f"config-4 grid 1024x512x512 steady: {dt:.2f} s wall, {f.meta['cg_iterations']} CG iterations, max u = {f.values.max():.6f}, "
      f"{(1023*511*511)*f.meta['cg_iterations']/dt/1e9:.1f} GDoF/s incl. set-up and the 2.2 GB copy-out", flush=True)
del f
t0 = time.perf_counter()
g = elasticity._solve_elasticity_3d_static(1.0, 0.25, 0.25, 512, 128, 128, 210e9, 0.3, 0.0, 0.0, -76518.0, "stress", rtol=1e-8)
dt = time.perf_counter() - t0
print(f"elasticity 512x128x128 (25.8 M DoF): {dt:.2f} s wall, {g.meta['cg_iterations']} PCG iterations, max von Mises = {g.values.max():.6e}", flush=True)
