"""Round-2 experiment 4 (one GPU): marching / tile kernels after the loads became unconditional.  Interleaved rounds, medians.
usage: python tools/r2_exp4.py [n] [rounds]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib = L.load_library()
ctx = L.context(0)
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
N = (n - 1) ** 3
base = dict(fast_ex=1, fast_ex_all=0, fast_pf=0, fast_ry=2, lazy_dot=1, cg_variant=1, tile3d_max_nodes=3000000, fast_wg_per_cu=0, fast_wg_per_cu_k2=0)
variants = {"march": dict(), "ex_all": dict(fast_ex_all=1), "pf": dict(fast_pf=1), "ex_all+pf": dict(fast_ex_all=1, fast_pf=1),
            "ex0": dict(fast_ex=0), "ry4": dict(fast_ry=4), "ry4+ex_all": dict(fast_ry=4, fast_ex_all=1), "march72": dict(cg_variant=0),
            "tile": dict(tile3d_max_nodes=2000000000, lazy_dot=0), "wg4": dict(fast_wg_per_cu=4), "wg3": dict(fast_wg_per_cu=3), "wg6": dict(fast_wg_per_cu=6)}
res = {k: [] for k in variants}
for rnd in range(rounds):
    for name, ov in variants.items():
        cfg = dict(base, **ov)
        for k, v in cfg.items():
            L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), v))
        h = L.Heat()
        h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 1, 1.0
        h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
        h.bc_present[0] = 1
        s = C.c_void_p()
        L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
        ms = C.c_double()
        L.check(ctx, lib.pdegpu_session_time_apply(s, 20, C.byref(ms)))
        L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
        L.check(ctx, lib.pdegpu_session_residual(s))
        L.check(ctx, lib.pdegpu_session_reduce_local(s, 1, 1))
        L.check(ctx, lib.pdegpu_session_iterate(s, 4, 0, None))
        st = L.Stats()
        L.check(ctx, lib.pdegpu_session_iterate(s, 20, 1, st))
        rr = C.c_double()
        L.check(ctx, lib.pdegpu_session_rr(s, C.byref(rr)))
        k1, k2 = st.kernel_ms[0] / 20, st.kernel_ms[1] / 20
        res[name].append((ms.value, k1, k2, rr.value))
        print(f"round {rnd} {name:10s}: SpMV {ms.value:.4f} ms ({16*N/ms.value/8e9:.3f}) | K1 {k1:.4f} ms | K2 {k2:.4f} ms ({24*N/k2/8e9:.3f}) | iter {k1+k2:.4f} | rr {rr.value:.12e}", flush=True)
        lib.pdegpu_session_destroy(s)
for name, r in res.items():
    a = np.array(r)
    md = np.median(a, axis=0)
    b1 = 48 if name.endswith("72") else 40
    print(f"MEDIAN {name:10s}: SpMV {md[0]:.4f} ms ({16*N/md[0]/8e9:.3f} of 8 TB/s) | K1 {md[1]:.4f} ({b1*N/md[1]/8e9:.3f}) | K2 {md[2]:.4f} ({24*N/md[2]/8e9:.3f}) | iter {md[1]+md[2]:.4f} -> {N/(md[1]+md[2])/1e6:.1f} GDoF/s")
