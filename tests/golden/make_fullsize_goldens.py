"""Generates tests/golden/fullsize_samples.npz: samples of the CPU oracle's solution at sizes the GPU suite cannot afford to
re-run through a sparse LU at test time —
  * BASELINE config 2 at FULL size (2-D heat, 1024 x 1024 cells, kappa = 1, T_boundary = 0, T_initial = 20, dt = 0.0025), the
    first 60 of its 200 steps, sampled at every 16th node per axis at steps 1, 20 and 60;
  * a 3-D heat run at 40^3 cells (the largest the LU handles in about a minute), 6 steps, every 4th node, steps 1 and 6;
  * config 5 at a quarter of its resolution (64 x 16 x 16 cells, gravity, von Mises stress), every (4, 2, 2)-th node.
One factorisation, then triangular solves per step (the matrix does not change between steps; the reference's
re-factorisation every step gives the same numbers).  Oracle output, not reference output.
Takes a few minutes on 8 cores.  Run from the repo root: python tests/golden/make_fullsize_goldens.py"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import fem_p1 as o  # noqa: E402

t0 = time.time()
f2 = o.solve_heat_2d(1.0, 1.0, 1024, 1024, 1.0, 0.0, 20.0, 0.0025, 60, refactor_each_step=False)
print("2-D done", time.time() - t0, flush=True)
v2 = np.asarray(f2.values)[[1, 20, 60]].reshape(3, 1025, 1025)[:, ::16, ::16].copy()
f3 = o.solve_heat_3d(1.0, 1.0, 1.0, 40, 40, 40, 1.0, 0.0, 20.0, 0.0025, 6, refactor_each_step=False)
print("3-D done", time.time() - t0, flush=True)
v3 = np.asarray(f3.values)[[1, 6]].reshape(2, 41, 41, 41)[:, ::4, ::4, ::4].copy()
fe = o.solve_elasticity_3d(1.0, 0.25, 0.25, 64, 16, 16, 210e9, 0.3, 0.0, 0.0, -76518.0, "stress")     # config 5 at 1/4 scale
ve = np.asarray(fe.values)[0].reshape(17, 17, 65)[::2, ::2, ::4].copy()
np.savez_compressed(ROOT / "tests" / "golden" / "fullsize_samples.npz", config2_1024_steps_1_20_60_every16=v2,
                    heat3d_40_steps_1_6_every4=v3, elasticity3d_64x16x16_gravity_vm_every_4_2_2=ve)
print(v2.shape, v3.shape, ve.shape, v2.max(), v3.max(), ve.max())
