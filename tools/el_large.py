"""Scratch: elasticity PCG iteration time at an HBM-resident size, rows per column."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L, elasticity
lib = L.load_library(); ctx = L.context(0)
n = (512, 128, 128)
for cfg in (dict(el_fast_ry=2), dict(el_fast_ry=1), dict(el_fast_ry=2)):
    for k, v in cfg.items():
        L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), v))
    try:
        t0 = time.perf_counter()
        f = elasticity._solve_elasticity_3d_static(1.0, 0.25, 0.25, *n, 210e9, 0.3, 0.0, 0.0, -76518.0, "stress", rtol=1e-2, max_iter=400)
        its = f.meta["cg_iterations"]
    except Exception as e:
        its = 400
    dt = time.perf_counter() - t0
    print(f"{n} {cfg}: {dt:.3f} s wall for {its} iterations -> <= {dt/its*1e6:.0f} us/it (incl. setup + postprocess)", flush=True)
