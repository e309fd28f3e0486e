"""Wall time of the tools at the reference's default arguments (what a user of the reference calls most), second call."""
import sys, time, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import server
d = tempfile.mkdtemp()
for name in ("solve_heat_1D", "solve_heat_2D", "solve_heat_3D", "solve_heat_1D_cylindrical", "solve_heat_2D_cylindrical", "solve_heat_3D_spherical",
             "solve_elasticity_1D_static", "solve_elasticity_2D_static", "solve_elasticity_3D_static"):
    fn = server.TOOLS[name]
    fn(data_dir=d)
    t0 = time.perf_counter(); r = fn(data_dir=d); dt = time.perf_counter() - t0
    print(f"{name:32s} {dt*1e3:8.1f} ms   cg_iterations={r.meta.get('cg_iterations')} path={r.meta.get('solver_path')}", flush=True)
