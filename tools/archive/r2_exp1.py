"""Round-2 experiment 1 (one GPU): edge hand-off and deeper prefetch in the one-vector modes of the marching kernel.
usage: python tools/r2_exp1.py [n]"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = L.load_library()
ctx = L.context(0)
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
N = (n - 1) ** 3
base = dict(fast_ex=1, fast_ex_all=0, fast_pf=0, fast_wg_per_cu=0, fast_wg_per_cu_k2=0)
variants = [dict(), dict(fast_ex_all=1), dict(fast_pf=1), dict(fast_ex_all=1, fast_pf=1), dict(fast_ex=0, fast_pf=1),
            dict(fast_ex_all=1, fast_pf=1, fast_wg_per_cu=3), dict(fast_ex_all=1, fast_pf=1, fast_wg_per_cu=5),
            dict(fast_ex_all=1, fast_pf=1, fast_wg_per_cu=6), dict(fast_ex_all=1, fast_pf=1, fast_wg_per_cu=8)]
for ov in variants:
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
    L.check(ctx, lib.pdegpu_session_time_apply(s, 30, C.byref(ms)))
    L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
    L.check(ctx, lib.pdegpu_session_residual(s))
    L.check(ctx, lib.pdegpu_session_reduce_local(s, 1, 1))
    L.check(ctx, lib.pdegpu_session_iterate(s, 4, 0, None))
    st = L.Stats()
    L.check(ctx, lib.pdegpu_session_iterate(s, 30, 1, st))
    rr = C.c_double()
    L.check(ctx, lib.pdegpu_session_rr(s, C.byref(rr)))
    k1, k2 = st.kernel_ms[0] / 30, st.kernel_ms[1] / 30
    print(f"{ov}: SpMV {ms.value:.4f} ms {16*N/ms.value/1e9:6.3f} TB/s ({16*N/ms.value/8e9:.3f}) | K1 {k1:.4f} ms {40*N/k1/1e9:6.3f} TB/s | "
          f"K2 {k2:.4f} ms {24*N/k2/1e9:6.3f} TB/s ({24*N/k2/8e9:.3f}) | iter {k1+k2:.4f} | rr {rr.value:.12e}", flush=True)
    lib.pdegpu_session_destroy(s)
