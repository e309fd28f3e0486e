"""Transient solves with the snapshot copies in line (snapshot_async = 0) and beside the next step (1): wall of the whole tool-level call.
usage: python tools/snapshot_overlap.py [steps2d]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import numpy as np
from pdegpu import heat, _lib as L
lib = L.load_library(); ctx = L.context(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ref = {}
for name, call in (("config 2 (1024^2)", lambda: heat._solve_heat_2d_raw(1.0, 1.0, 1024, 1024, 1.0, 0.0, 20.0, 0.0025, steps)),
                   ("3-D 256^3, 6 steps", lambda: heat._solve_heat_3d_raw(Lx=1.0, Ly=1.0, Lz=1.0, nx=256, ny=256, nz=256, diffusivity=1.0, T_boundary=0.0,
                                                                           T_initial=20.0, dt=0.001, num_steps=6)),
                   ("1024^2, stride = all steps (one snapshot)", lambda: heat._solve_heat_2d_raw(1.0, 1.0, 1024, 1024, 1.0, 0.0, 20.0, 0.0025, steps,
                                                                                                   snapshot_stride=steps))):
    for mode in (0, 1, 0, 1):
        L.check(ctx, lib.pdegpu_set_tunable(ctx, b"snapshot_async", mode))
        t0 = time.perf_counter()
        f = call()
        dt = time.perf_counter() - t0
        key = name
        if key not in ref:
            ref[key] = f.values.copy()
        same = bool(np.array_equal(ref[key], f.values))
        print(f"{name} snapshot_async={mode}: wall {dt:.3f} s, solve_ms {f.meta.get('solve_ms', float('nan')):.1f}, {f.meta['cg_iterations']} CG iterations, "
              f"{f.values.shape[0]} snapshots, identical to the first run: {same}", flush=True)
L.check(ctx, lib.pdegpu_set_tunable(ctx, b"snapshot_async", 1))
