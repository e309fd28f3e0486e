"""one spherical and one composite-core solve at 192^3 for rocprofv3 --kernel-trace --stats / --pmc"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import curvilinear as cv, heat as H, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
f = cv._solve_heat_3d_spherical_raw(0.1, 1.0, n, n, n, 1.0, 0.0, 20.0, 0.01, 2, rtol=1e-8)
print("spherical", f.meta["cg_iterations"], f.meta["solve_ms"], f.meta["solver_path"])
f = H._solve_heat_3d_raw(1.0, 1.0, 1.0, n, n, n, 1.0, 0.0, 20.0, 0.01, 2, core_radius=0.3, core_diffusivity=5.0, rtol=1e-8)
print("composite core", f.meta["cg_iterations"], f.meta["solve_ms"], f.meta["solver_path"])
_lib.close_contexts()
