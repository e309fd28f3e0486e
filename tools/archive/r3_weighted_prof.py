"""one composite-core solve at 192^3 and one spherical solve, for rocprofv3 --kernel-trace --stats"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import curvilinear as cv, heat as H
n = 192
f = H._solve_heat_3d_raw(1.0, 1.0, 1.0, n, n, n, 1.0, 0.0, 20.0, 0.01, 2, core_radius=0.3, core_diffusivity=5.0, rtol=1e-8)
print(f.meta["cg_iterations"], f.meta["solve_ms"])
from pdegpu import _lib
_lib.close_contexts()
