"""Round-2 experiment 6 (one GPU): how many resident workgroups / z-chunks the marching kernels want, by grid size.
usage: python tools/r2_exp6.py"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L

lib = L.load_library()
ctx = L.context(0)


def measure(n, cfgs, rounds=2, iters=15):
    grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
    N = (n - 1) ** 3
    res = {c: [] for c in cfgs}
    for rnd in range(rounds):
        for c in cfgs:
            ry, wg, ex, pf = c
            v = ry + 16 * wg + 256 * ex + 512 * pf
            for key in (b"fast_cfg_plain", b"fast_cfg_res", b"fast_cfg_k1", b"fast_cfg_k2"):
                L.check(ctx, lib.pdegpu_set_tunable(ctx, key, v))
            h = L.Heat()
            h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 1, 1.0
            h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
            h.bc_present[0] = 1
            s = C.c_void_p()
            L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
            ms = C.c_double()
            L.check(ctx, lib.pdegpu_session_time_apply(s, iters, C.byref(ms)))
            L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
            L.check(ctx, lib.pdegpu_session_residual(s))
            L.check(ctx, lib.pdegpu_session_reduce_local(s, 1, 1))
            L.check(ctx, lib.pdegpu_session_iterate(s, 3, 0, None))
            st = L.Stats()
            L.check(ctx, lib.pdegpu_session_iterate(s, iters, 1, st))
            res[c].append((ms.value, st.kernel_ms[0] / iters, st.kernel_ms[1] / iters))
            lib.pdegpu_session_destroy(s)
    print(f"=== n = {n}  (ry, wg, ex, pf): SpMV | K1 | K2 | K1+K2   [ms, median of {rounds}]", flush=True)
    for c in cfgs:
        m = np.median(np.array(res[c]), axis=0)
        print(f"    {c}: {m[0]:.4f} ({16*N/m[0]/8e9:.3f}) | {m[1]:.4f} ({40*N/m[1]/8e9:.3f}) | {m[2]:.4f} ({24*N/m[2]/8e9:.3f}) | {m[1]+m[2]:.4f} -> {N/(m[1]+m[2])/1e6:.1f} GDoF/s", flush=True)


L.check(ctx, lib.pdegpu_set_tunable(ctx, b"tile3d_max_nodes", 0))          # marching kernels at every size
measure(512, [(ry, wg, ex, pf) for ry in (2, 4, 8) for wg in (1, 2) for ex in (0, 1) for pf in (0, 1)])
for n in (160, 256, 384):
    measure(n, [(2, wg, 1, 0) for wg in (0, 2, 4, 8)] + [(4, 2, 1, 0), (2, 2, 0, 1)])
measure(1024, [(2, 0, 1, 0), (2, 2, 1, 0), (4, 2, 1, 0), (4, 2, 0, 1), (8, 2, 1, 0), (2, 4, 1, 0)], rounds=1, iters=8)
