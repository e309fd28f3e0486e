"""One spherical and one composite-core solve per size (argv), rate over the whole call — for sweeps of the PDEGPU_WFAST_* environment overrides
(tools/gpu_w_sweep.sh runs one process per setting: the overrides are read when the context is created)."""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import curvilinear as cv, heat as H

tag = " ".join(f"{k[7:].lower()}={v}" for k, v in sorted(os.environ.items()) if k.startswith("PDEGPU_")) or "defaults"
for n in [int(v) for v in sys.argv[1:]] or [256]:
    out = []
    for rep in range(2):
        f = cv._solve_heat_3d_spherical_raw(0.1, 1.0, n, n, n, 1.0, 0.0, 20.0, 0.01, 2, rtol=1e-8)
    out.append(("spherical", f.meta))
    f = H._solve_heat_3d_raw(1.0, 1.0, 1.0, n, n, n, 1.0, 0.0, 20.0, 0.01, 2, core_radius=0.3, core_diffusivity=5.0, rtol=1e-8)
    out.append(("composite core", f.meta))
    print(f"[{tag}] {n}^3: " + ", ".join(f"{t} {(n - 1) ** 3 * m['cg_iterations'] / (m['solve_ms'] * 1e-3) / 1e9:.2f} GDoF/s ({m['cg_iterations']} it)" for t, m in out), flush=True)
