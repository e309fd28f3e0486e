"""A/B of one context tunable on the CG kernels of the steady 3-D heat path over grid sizes (one GPU, interleaved rounds, medians).
usage: python tools/ab_sizes.py KEY V0 V1 [n ...]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L

lib = L.load_library()
ctx = L.context(0)
key, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sizes = [int(v) for v in sys.argv[4:]] or [320, 384, 448, 512, 576, 640, 700, 768, 832, 896]
for n in sizes:
    grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
    N = (n - 1) ** 3
    h = L.Heat()
    h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 1, 1.0
    h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
    h.bc_present[0] = 1
    s = C.c_void_p()
    L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
    res = {v0: [], v1: []}
    for rnd in range(4):
        for v in (v0, v1):
            L.check(ctx, lib.pdegpu_set_tunable(ctx, key.encode(), v))
            ms = C.c_double()
            L.check(ctx, lib.pdegpu_session_time_apply(s, 10, C.byref(ms)))
            L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
            L.check(ctx, lib.pdegpu_session_residual(s))
            L.check(ctx, lib.pdegpu_session_iterate(s, 2, 0, None))
            st = L.Stats()
            L.check(ctx, lib.pdegpu_session_iterate(s, 10, 1, st))
            res[v].append((st.kernel_ms[0] / 10, st.kernel_ms[1] / 10, ms.value))
    out = []
    for v in (v0, v1):
        k1, k2, sp = np.median(np.array(res[v]), axis=0)
        out.append(f"{key}={v}: K1 {40*N/k1/8e9:.3f} K2 {24*N/k2/8e9:.3f} SpMV {16*N/sp/8e9:.3f} -> {N/(k1+k2)/1e6:.1f} GDoF/s")
    print(f"n={n:4d}  " + "  |  ".join(out), flush=True)
    lib.pdegpu_session_destroy(s)
L.check(ctx, lib.pdegpu_set_tunable(ctx, key.encode(), v1))
