"""CG rate of the weighted (curvilinear) engine: _solve_heat_3d_spherical_raw and the composite-core box at a few sizes — DoF/s per CG iteration from meta."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import curvilinear as cv, heat as H
for n in (64, 128, 192, 256):
    t0 = time.perf_counter()
    f = cv._solve_heat_3d_spherical_raw(0.1, 1.0, n, n, n, 1.0, 0.0, 20.0, 0.01, 2, rtol=1e-8)
    wall = time.perf_counter() - t0
    m = f.meta
    it = m.get("cg_iterations", 0)
    unk = (n - 1) ** 3
    print(f"spherical {n}^3: wall {wall:.3f} s, solve_ms {m.get('solve_ms')}, iterations {it}, path {m.get('solver_path')}, "
          f"{unk * it / (m.get('solve_ms', 1e9) * 1e-3) / 1e9:.2f} GDoF/s over the whole call", flush=True)
for n in (128, 192):
    t0 = time.perf_counter()
    f = H._solve_heat_3d_raw(1.0, 1.0, 1.0, n, n, n, 1.0, 0.0, 20.0, 0.01, 2, core_radius=0.3, core_diffusivity=5.0, rtol=1e-8)
    wall = time.perf_counter() - t0
    m = f.meta
    it = m.get("cg_iterations", 0)
    unk = (n - 1) ** 3
    print(f"composite core {n}^3: wall {wall:.3f} s, solve_ms {m.get('solve_ms')}, iterations {it}, path {m.get('solver_path')}, "
          f"{unk * it / (m.get('solve_ms', 1e9) * 1e-3) / 1e9:.2f} GDoF/s over the whole call", flush=True)
    f = H._solve_heat_3d_raw(1.0, 1.0, 1.0, n, n, n, 1.0, 0.0, 20.0, 0.01, 2, rtol=1e-8)
    m = f.meta
    it = m.get("cg_iterations", 0)
    print(f"plain box {n}^3 (same call, marching kernels): solve_ms {m.get('solve_ms')}, iterations {it}, path {m.get('solver_path')}, "
          f"{unk * it / (m.get('solve_ms', 1e9) * 1e-3) / 1e9:.2f} GDoF/s over the whole call", flush=True)
