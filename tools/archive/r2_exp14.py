"""Round-2 experiment 14 (one GPU): non-temporal loads of the input in the one-vector marching modes (bare SpMV, stencil-K2) at 512^3,
8 interleaved rounds.  fast_nt_in: 0 plain loads, 1 non-temporal for the rows no other workgroup reads, 2 for every row."""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
n = 512
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
N = (n - 1) ** 3
variants = [dict(fast_nt_in=0), dict(fast_nt_in=1), dict(fast_nt_in=2), dict(fast_nt_in=2, fast_ry=2)]
res = [[] for _ in variants]
for rnd in range(8):
    for i, v in enumerate(variants):
        L.check(ctx, lib.pdegpu_set_tunable(ctx, b"fast_ry", 0))
        for k, val in v.items():
            L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), val))
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
        L.check(ctx, lib.pdegpu_session_iterate(s, 3, 0, None))
        st = L.Stats()
        L.check(ctx, lib.pdegpu_session_iterate(s, 20, 1, st))
        rr = C.c_double(); L.check(ctx, lib.pdegpu_session_rr(s, C.byref(rr)))
        res[i].append((ms.value, st.kernel_ms[0] / 20, st.kernel_ms[1] / 20, rr.value))
        lib.pdegpu_session_destroy(s)
for i, v in enumerate(variants):
    a = np.array(res[i])
    md, mn = np.median(a, axis=0), a.min(axis=0)
    print(f"{v}: SpMV median {md[0]:.4f} ms ({16*N/md[0]/8e9:.3f}) min {mn[0]:.4f} | K1 {md[1]:.4f} ({40*N/md[1]/8e9:.3f}) | "
          f"K2 median {md[2]:.4f} ({24*N/md[2]/8e9:.3f}) min {mn[2]:.4f} | rr {md[3]:.9e}", flush=True)
