"""Rates of the 2-D tools at large sizes: elasticity 2-D (thread-per-node block kernel + PCG), curvilinear 2-D (weighted class-table kernel), heat 2-D (persistent kernel)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import curvilinear as cv, heat as H, elasticity as E
for n in (256, 512, 1024):
    t0 = time.perf_counter()
    f = E._solve_elasticity_2d_static(1.0, 0.25, n, n // 4, 210e9, 0.3, 0.0, -7.6e4, "stress", True, rtol=1e-8)
    wall = time.perf_counter() - t0
    m = f.meta
    dof = 2 * (n + 1) * (n // 4 + 1)
    print(f"elasticity 2-D {n}x{n//4}: wall {wall:.3f} s, meta {dict((k, m[k]) for k in m if k in ('cg_iterations', 'solve_ms', 'solver_path'))}, "
          f"{dof * m.get('cg_iterations', 0) / max(m.get('solve_ms', 1e9), 1e-9) / 1e6:.2f} GDoF/s", flush=True)
for n in (512, 1024):
    t0 = time.perf_counter()
    f = cv._solve_heat_2d_cylindrical_raw(0.1, 1.0, 1.0, n, n, 1.0, 0.0, 20.0, 0.0025, 5, rtol=1e-8)
    wall = time.perf_counter() - t0
    m = f.meta
    print(f"cylindrical 2-D {n}^2: wall {wall:.3f} s, iterations {m['cg_iterations']}, solve_ms {m['solve_ms']:.1f}, "
          f"{(n - 1) ** 2 * m['cg_iterations'] / m['solve_ms'] / 1e6:.2f} GDoF/s", flush=True)
    t0 = time.perf_counter()
    f = H._solve_heat_2d_raw(1.0, 1.0, n, n, 1.0, 0.0, 20.0, 0.0025, 5, rtol=1e-8)
    wall = time.perf_counter() - t0
    m = f.meta
    print(f"heat 2-D {n}^2: wall {wall:.3f} s, iterations {m['cg_iterations']}, solve_ms {m['solve_ms']:.1f}, path {m['solver_path']}, "
          f"{(n - 1) ** 2 * m['cg_iterations'] / m['solve_ms'] / 1e6:.2f} GDoF/s", flush=True)
