"""Config 2 (1024^2 cells, 200 backward-Euler steps at the dispatcher's defaults) and other 2-D grids: the register-resident cooperative CG
(persist2d = 1) against the launch-per-kernel loop.  usage: python tools/c2_persist.py [steps]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import heat, _lib as L
lib = L.load_library(); ctx = L.context(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for n in (1024, 512, 256):
    if n == 2048 and steps > 10:
        st = 10
    else:
        st = steps
    for persist in (1, 11, 0, 1, 11, 0):        # 1: classic CG, two XCD-hierarchical barriers; 11: single-reduction CG (one barrier + neighbour hand-off); 0: one launch per kernel
        L.check(ctx, lib.pdegpu_set_tunable(ctx, b"persist2d", persist % 10))
        L.check(ctx, lib.pdegpu_set_tunable(ctx, b"persist2d_sr", persist // 10))
        t0 = time.perf_counter()
        f = heat._solve_heat_2d_raw(1.0, 1.0, n, n, 1.0, 0.0, 20.0, 0.0025, st)
        dt = time.perf_counter() - t0
        it = f.meta["cg_iterations"]
        print(f"{n}^2 persist2d={persist}: {st} steps {dt:.3f} s wall, {it} CG iterations ({it/st:.0f}/step), {dt/it*1e6:.2f} us per iteration incl. everything, "
              f"u_center(t_end) = {f.values[-1][(n//2)*(n+1)+n//2]:.9f}", flush=True)
L.check(ctx, lib.pdegpu_set_tunable(ctx, b"persist2d", 1))
L.check(ctx, lib.pdegpu_set_tunable(ctx, b"persist2d_sr", 1))
