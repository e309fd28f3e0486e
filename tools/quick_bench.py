"""Scratch timing of the CG session on one GPU: python tools/quick_bench.py [n] [iters]"""
import ctypes as C
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import numpy as np
from pdegpu import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
lib = L.load_library()
ctx = L.context(0)
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
for fast in (1, 0):
    for steady in (1, 0):
        h = L.Heat()
        h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, steady, 1.0
        h.bc_mode, h.fast_path = L.BC_UNIFORM, fast
        h.bc_present[0] = 1
        s = C.c_void_p()
        L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
        L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
        L.check(ctx, lib.pdegpu_session_residual(s))
        L.check(ctx, lib.pdegpu_session_reduce_local(s, 1, 1))
        st = L.Stats()
        L.check(ctx, lib.pdegpu_session_iterate(s, 10, 0, None))
        L.check(ctx, lib.pdegpu_session_sync(s))
        t0 = time.perf_counter()
        L.check(ctx, lib.pdegpu_session_iterate(s, iters, 1, st))
        L.check(ctx, lib.pdegpu_session_sync(s))
        dt = time.perf_counter() - t0
        rr = C.c_double()
        L.check(ctx, lib.pdegpu_session_rr(s, C.byref(rr)))
        N = (n + 1) ** 3
        k1 = st.kernel_ms[0] / iters
        k2 = st.kernel_ms[1] / iters
        print(f"n={n} fast={fast} steady={steady}: {dt/iters*1e3:.3f} ms/iter  K1 {k1:.3f} ms ({48*N/k1/1e9:.0f} GB/s alg)  "
              f"K2 {k2:.3f} ms ({24*N/k2/1e9:.0f} GB/s)  CG {N*iters/dt/1e9:.2f} GDoF/s  rr={rr.value:.6e}", flush=True)
        lib.pdegpu_session_destroy(s)
