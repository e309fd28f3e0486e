"""Config 1 (1-D heat, 2 m / 1000 elements / 200 steps) wall time through the raw solver, direct vs CG path."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import heat, _lib
lib = _lib.load_library(); ctx = _lib.context(0)
for direct in (1, 0, 1):
    _lib.check(ctx, lib.pdegpu_set_tunable(ctx, b"direct_1d", direct))
    for rep in range(2):
        t0 = time.perf_counter()
        f = heat._solve_heat_1d_raw(2.0, 1000, 1.0, 20.0, 0.0, 0.0, 0.01, 200)
        dt = time.perf_counter() - t0
    print(f"direct_1d={direct}: {dt*1e3:.1f} ms wall, solve_ms={f.meta.get('solve_ms')}, path={f.meta['solver_path']}, "
          f"cg_iterations={f.meta['cg_iterations']}, u(1.0, t=2)={f.values[200, 500]:.9f}", flush=True)
