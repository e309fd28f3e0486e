"""Scratch: per-iteration time of the small / medium configs with and without reduce launches (lazy_dot)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
from pdegpu import heat, elasticity
lib = L.load_library(); ctx = L.context(0)
for lazy in (1,):
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"lazy_dot", lazy))
    for rep in range(2):
        t0 = time.perf_counter()
        f = heat._solve_heat_2d_raw(1.0, 1.0, 1024, 1024, 1.0, 0.0, 20.0, 0.0025, 20)
        dt = time.perf_counter() - t0
    print(f"lazy_dot={lazy}: config 2 (20 of 200 steps) {dt:.3f} s, {f.meta['cg_iterations']} its, {dt/f.meta['cg_iterations']*1e6:.1f} us/it", flush=True)
    for rep in range(2):
        t0 = time.perf_counter()
        f = heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 64, 64, 64, 1.0, 0.0, 20.0, 0.0025, 20)
        dt = time.perf_counter() - t0
    print(f"lazy_dot={lazy}: 3-D 64^3 transient 20 steps {dt:.3f} s, {f.meta['cg_iterations']} its, {dt/f.meta['cg_iterations']*1e6:.1f} us/it", flush=True)
    for rep in range(2):
        t0 = time.perf_counter()
        f = elasticity._solve_elasticity_3d_static(1.0, 0.25, 0.25, 256, 64, 64, 210e9, 0.3, 0.0, 0.0, -76518.0, "stress", rtol=1e-8)
        dt = time.perf_counter() - t0
    print(f"lazy_dot={lazy}: config 5 {dt:.3f} s, {f.meta['cg_iterations']} its, {dt/f.meta['cg_iterations']*1e6:.1f} us/it", flush=True)
