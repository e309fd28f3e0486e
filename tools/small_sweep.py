"""Scratch: z-chunk policy of the marching kernels on small grids (fast_minlen x fast_small_mult)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
from pdegpu import heat
lib = L.load_library(); ctx = L.context(0)
cases = (("2-D 1024^2 x10 steps", lambda: heat._solve_heat_2d_raw(1.0, 1.0, 1024, 1024, 1.0, 0.0, 20.0, 0.0025, 10)),
         ("2-D 256^2 x20", lambda: heat._solve_heat_2d_raw(1.0, 1.0, 256, 256, 1.0, 0.0, 20.0, 0.0025, 20)),
         ("3-D 64^3 x20", lambda: heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 64, 64, 64, 1.0, 0.0, 20.0, 0.0025, 20)),
         ("3-D 128^3 x10", lambda: heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 128, 128, 128, 1.0, 0.0, 20.0, 0.0025, 10)),
         ("3-D 32^3 x20", lambda: heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 32, 32, 32, 1.0, 0.0, 20.0, 0.0025, 20)))
for minlen, mult in ((2, 1), (1, 1), (2, 2), (1, 2), (2, 4), (1, 4), (3, 2), (4, 4)):
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"fast_minlen", minlen))
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"fast_small_mult", mult))
    line = f"minlen={minlen} mult={mult}:"
    for name, fn in cases:
        fn()
        t0 = time.perf_counter(); f = fn(); dt = time.perf_counter() - t0
        line += f"  {name}: {dt/f.meta['cg_iterations']*1e6:5.1f} us/it"
    print(line, flush=True)
