"""CG rate of the weighted (curvilinear / composite) 3-D engine: _solve_heat_3d_spherical_raw, the composite-core box and the cylinder fallback at a few
sizes — unknowns x PCG iterations / solve_ms (the whole C call: assembly of the rows, right-hand sides, polls and snapshots included), the plain
box beside it.  argv: sizes (default 128 192 256); PDEGPU_WFAST=0 in the environment gives the class-table kernels."""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import curvilinear as cv, heat as H

sizes = [int(v) for v in sys.argv[1:]] or [128, 192, 256]


def line(tag, n, f, wall):
    m = f.meta
    it, unk = m.get("cg_iterations", 0), (n - 1) ** 3
    print(f"{tag} {n}^3: wall {wall:.3f} s, solve_ms {m.get('solve_ms'):.1f}, iterations {it}, path {m.get('solver_path')}, "
          f"{unk * it / (m.get('solve_ms', 1e9) * 1e-3) / 1e9:.2f} GDoF/s over the whole call", flush=True)


for n in sizes:
    for rep in range(2):
        t0 = time.perf_counter()
        f = cv._solve_heat_3d_spherical_raw(0.1, 1.0, n, n, n, 1.0, 0.0, 20.0, 0.01, 2, rtol=1e-8)
        line("spherical", n, f, time.perf_counter() - t0)
    t0 = time.perf_counter()
    f = H._solve_heat_3d_raw(1.0, 1.0, 1.0, n, n, n, 1.0, 0.0, 20.0, 0.01, 2, core_radius=0.3, core_diffusivity=5.0, rtol=1e-8)
    line("composite core", n, f, time.perf_counter() - t0)
    t0 = time.perf_counter()
    f = H._solve_heat_3d_raw(1.0, 1.0, 1.0, n, n, n, 1.0, 0.0, 20.0, 0.01, 2, geometry_type="cylinder", cylinder_radius=0.5, rtol=1e-8)
    line("cylinder fallback", n, f, time.perf_counter() - t0)
    t0 = time.perf_counter()
    f = H._solve_heat_3d_raw(1.0, 1.0, 1.0, n, n, n, 1.0, 0.0, 20.0, 0.01, 2, rtol=1e-8)
    line("plain box (marching kernels)", n, f, time.perf_counter() - t0)
