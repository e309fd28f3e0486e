"""Scratch: end-to-end wall time of BASELINE configs 1 and 2 through the tool-level solvers."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import heat
t0 = time.perf_counter()
f = heat._solve_heat_1d_raw(2.0, 1000, 1.0, 20.0, 0.0, 0.0, 0.01, 200)
t1 = time.perf_counter()
print(f"config1 1D 1000 el, 200 steps: {t1-t0:.2f} s, cg_iterations={f.meta['cg_iterations']}, u(1.0,t=2)={f.values[-1][500]:.9f}")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t0 = time.perf_counter()
f = heat._solve_heat_2d_raw(1.0, 1.0, 1024, 1024, 1.0, 0.0, 20.0, 0.0025, steps)
t1 = time.perf_counter()
print(f"config2 2D 1024^2, {steps} steps: {t1-t0:.2f} s, cg_iterations={f.meta['cg_iterations']} ({f.meta['cg_iterations']/steps:.0f}/step), "
      f"{(t1-t0)/max(1,f.meta['cg_iterations'])*1e6:.1f} us/iteration, values {f.values.shape}")
