"""Scratch tuning sweep (one process, one GPU): K1 / K2 / bare-SpMV time per launch for a list of tunable
settings.  usage: python tools/sweep.py [n] [iters]"""
import ctypes as C
import itertools
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = L.load_library()
ctx = L.context(0)
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
N = (n - 1) ** 3

configs = []
base = dict(fast_ry=2, fast_nt=-1, fast_wg_per_cu=0, fast_wg_per_cu_k2=0, grid_mult=8, k2_grid_mult=2, cg_variant=1, fast_ex=1)
for w1 in (0, 2, 3, 4, 6):
    configs.append(dict(base, fast_wg_per_cu=w1))
for w2 in (1, 2, 3, 4, 6, 8):
    configs.append(dict(base, fast_wg_per_cu_k2=w2))
if len(sys.argv) > 3:
    configs = [dict(base, **eval(sys.argv[3]))]

for steady in (1,):
    for cfg in configs:
        for k, v in cfg.items():
            L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), v))
        h = L.Heat()
        h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, steady, 1.0
        h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
        h.bc_present[0] = 1
        s = C.c_void_p()
        L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
        L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
        L.check(ctx, lib.pdegpu_session_residual(s))
        L.check(ctx, lib.pdegpu_session_reduce_local(s, 1, 1))
        L.check(ctx, lib.pdegpu_session_iterate(s, 4, 0, None))
        st = L.Stats()
        L.check(ctx, lib.pdegpu_session_iterate(s, 20, 1, st)); iters = 20
        rr = C.c_double()
        L.check(ctx, lib.pdegpu_session_rr(s, C.byref(rr)))
        ms = C.c_double()
        L.check(ctx, lib.pdegpu_session_time_apply(s, 30, C.byref(ms)))
        k1, k2 = st.kernel_ms[0] / iters, st.kernel_ms[1] / iters
        tot = (k1 + k2)
        print(f"steady={steady} wg={cfg['fast_wg_per_cu']} wg_k2={cfg['fast_wg_per_cu_k2']}: K1+K2 {tot:.3f} ms -> {N/tot/1e6:.1f} GDoF/s | K1 {k1:.3f} ms {48*N/k1/1e6:6.0f} GB/s | K2 {k2:.3f} ms {24*N/k2/1e6:6.0f} GB/s | "
              f"SpMV {ms.value:.3f} ms {16*N/ms.value/1e6:6.0f} GB/s | rr {rr.value:.9e}", flush=True)
        if False:
            names = ["copy 1R+1W", "read 1R", "write 1W", "triad 2R+1W", "inplace 2R+1W"]
            nb = [2, 1, 1, 3, 3]
            vec_bytes = lib.pdegpu_session_plane_doubles(s) * (n + 1) * 8
            for w in range(5):
                L.check(ctx, lib.pdegpu_session_membench(s, w, 20, C.byref(ms)))
                print(f"   membench {names[w]:16s}: {ms.value:.3f} ms  {nb[w]*vec_bytes/ms.value/1e6:6.0f} GB/s", flush=True)
        lib.pdegpu_session_destroy(s)
