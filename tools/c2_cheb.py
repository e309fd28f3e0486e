"""Config 2 and other 2-D grids: the persistent single-reduction CG plain (persist2d_cheb = 0) and with the Chebyshev polynomial preconditioner
(persist2d_cheb = k terms on [lambda_max / ratio, lambda_max]).  usage: python tools/c2_cheb.py [steps] [n ...]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import numpy as np
from pdegpu import heat, _lib as L
lib = L.load_library(); ctx = L.context(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sizes = [int(v) for v in sys.argv[2:]] or [1024, 512]
def run(n, k, ratio, **kw):
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"persist2d_cheb", k))
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"persist2d_cheb_ratio", ratio))
    return heat._solve_heat_2d_raw(1.0, 1.0, n, n, 1.0, 0.0, 20.0, 0.0025, steps, **kw)
try:
    for n in sizes:
        run(n, 0, 30)
        base = run(n, 0, 30)
        print(f"{n}^2 plain: {base.meta['cg_iterations']} iterations, solve {base.meta['solve_ms']:.1f} ms", flush=True)
        for k, ratio in ((2, 10), (2, 30), (3, 10), (3, 30), (4, 10), (4, 30), (4, 100), (6, 30), (6, 100), (8, 30), (8, 100), (12, 100), (12, 300)):
            f = run(n, k, ratio)
            it = f.meta["cg_iterations"]
            err = np.abs(f.values - base.values).max() / np.abs(base.values).max()
            print(f"{n}^2 k={k:2d} ratio={ratio:3d}: {it} outer iterations ({it * k} stencil applications), solve {f.meta['solve_ms']:.1f} ms "
                  f"({base.meta['solve_ms'] / f.meta['solve_ms']:.2f}x), {f.meta['solve_ms'] * 1e3 / it:.2f} us per outer iteration, max rel diff to plain {err:.2e}", flush=True)
    # steady problem with a source (the other solve type of the tool), 1024^2
    kw = dict(steady=True, source_type="constant", source_value=3.0)
    b = heat._solve_heat_2d_raw(1.0, 1.0, 1024, 1024, 1.0, 0.0, 20.0, 0.0025, 1, **kw)
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"persist2d_cheb", 0))
    b = heat._solve_heat_2d_raw(1.0, 1.0, 1024, 1024, 1.0, 0.0, 20.0, 0.0025, 1, **kw)
    print(f"steady 1024^2 plain: {b.meta['cg_iterations']} iterations, {b.meta['solve_ms']:.1f} ms", flush=True)
    for k, ratio in ((4, 30), (6, 100), (8, 100), (12, 300)):
        L.check(ctx, lib.pdegpu_set_tunable(ctx, b"persist2d_cheb", k))
        L.check(ctx, lib.pdegpu_set_tunable(ctx, b"persist2d_cheb_ratio", ratio))
        f = heat._solve_heat_2d_raw(1.0, 1.0, 1024, 1024, 1.0, 0.0, 20.0, 0.0025, 1, **kw)
        err = np.abs(f.values - b.values).max() / np.abs(b.values).max()
        print(f"steady 1024^2 k={k} ratio={ratio}: {f.meta['cg_iterations']} outer iterations, {f.meta['solve_ms']:.1f} ms, max rel diff {err:.2e}", flush=True)
finally:
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"persist2d_cheb", 0))
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"persist2d_cheb_ratio", 30))
