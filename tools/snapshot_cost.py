"""Scratch: cost of copying every step's field to the host (the reference's default) vs only the last one."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import heat
for name, fn in (("2-D 1024^2, 20 steps", lambda **o: heat._solve_heat_2d_raw(1.0, 1.0, 1024, 1024, 1.0, 0.0, 20.0, 0.0025, 20, **o)),
                 ("3-D 128^3, 20 steps", lambda **o: heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 128, 128, 128, 1.0, 0.0, 20.0, 0.0025, 20, **o)),
                 ("3-D 256^3, 10 steps", lambda **o: heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 256, 256, 256, 1.0, 0.0, 20.0, 0.0025, 10, **o))):
    for stride in (1, 1000):
        fn(snapshot_stride=stride)
        t0 = time.perf_counter(); f = fn(snapshot_stride=stride); dt = time.perf_counter() - t0
        print(f"{name} stride={stride}: {dt*1e3:.1f} ms, {f.values.shape[0]} snapshots of {f.values.shape[1]*8/1e6:.1f} MB, {f.meta['cg_iterations']} its", flush=True)
