"""Scratch: BASELINE config 5 (3-D elasticity 256x64x64) timing on one GPU."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import elasticity
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
t0 = time.perf_counter()
f = elasticity._solve_elasticity_3d_static(1.0, 0.25, 0.25, nx, nx // 4, nx // 4, 210e9, 0.3, 0.0, 0.0, -76518.0, "stress", rtol=1e-8)
dt = time.perf_counter() - t0
print(f"elasticity {nx}x{nx//4}x{nx//4}: {dt:.2f} s wall, cg_iterations={f.meta['cg_iterations']}, max vm={f.values.max():.6e}")
