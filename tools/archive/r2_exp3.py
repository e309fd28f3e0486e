"""Round-2 experiment 3 (one GPU): the one-round tile form with the XCD-aware order on grids beyond L2, against the z-marching
kernels.  Variants are measured in interleaved rounds (the devices drift by ~10 % within a minute).  usage: python tools/r2_exp3.py [n]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = L.load_library()
ctx = L.context(0)

# parity of the XCD-aware tile form first (a grid beyond 4 M nodes), against the class-table kernel
g2 = L.make_grid(3, (170, 150, 161), (1.0, 0.9, 1.1))
N2 = 171 * 151 * 162
rng = np.random.default_rng(5)
x = rng.standard_normal(N2)
for tm in (0, 2000000000):
    L.check(ctx, lib.pdegpu_set_tunable(ctx, b"tile3d_max_nodes", tm))
    y1, y0 = np.empty(N2), np.empty(N2)
    L.check(ctx, lib.pdegpu_heat_apply(ctx, g2, 1.0, 0.3, L.dptr(x), L.dptr(y1), None, 1))
    L.check(ctx, lib.pdegpu_heat_apply(ctx, g2, 1.0, 0.3, L.dptr(x), L.dptr(y0), None, 0))
    X = x.reshape(162, 151, 171)
    m = np.zeros_like(X, bool); m[1:-1, 1:-1, 1:-1] = True
    err = np.abs(y1.reshape(X.shape)[m] - y0.reshape(X.shape)[m]).max() / np.abs(y0).max()
    print(f"tile3d_max_nodes={tm}: apply (15-point) vs class-table kernel, interior rel err {err:.2e}", flush=True)
    assert err < 1e-13

grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
N = (n - 1) ** 3
base = dict(fast_ex=1, lazy_dot=1, cg_variant=1, tile3d_max_nodes=3000000, tile3d_plain=1)
variants = {"march": dict(), "tile": dict(tile3d_max_nodes=2000000000, lazy_dot=0), "tile72": dict(tile3d_max_nodes=2000000000, lazy_dot=0, cg_variant=0),
            "march72": dict(cg_variant=0)}
res = {k: [] for k in variants}
for rnd in range(3):
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
        print(f"round {rnd} {name:8s}: SpMV {ms.value:.4f} ms ({16*N/ms.value/8e9:.3f}) | K1 {k1:.4f} ms | K2 {k2:.4f} ms ({24*N/k2/8e9:.3f}) | iter {k1+k2:.4f} | rr {rr.value:.12e}", flush=True)
        lib.pdegpu_session_destroy(s)
for name, r in res.items():
    a = np.array(r)
    md = np.median(a, axis=0)
    print(f"MEDIAN {name:8s}: SpMV {md[0]:.4f} ms ({16*N/md[0]/8e9:.3f} of 8 TB/s) | K1 {md[1]:.4f} | K2 {md[2]:.4f} ({24*N/md[2]/8e9:.3f}) | iter {md[1]+md[2]:.4f}")
