"""Round-2 experiment 7 (one GPU): (a) elasticity block apply with the x-face rows inside the marching kernel, wavefront target sweep,
config-5 solve; (b) finalists of the heat marching kernels in many interleaved rounds.  usage: python tools/r2_exp7.py"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
from pdegpu import elasticity

lib = L.load_library()
ctx = L.context(0)


def tun(**kw):
    for k, v in kw.items():
        L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), v))


el = L.Elasticity()
el.E, el.nu = 210e9, 0.3
cfgs = [dict(el_fast_xface=xf, el_fast_waves=w, el_fast_ry=2) for xf in (0, 1) for w in (1024, 2048, 4096)]
for n in ((256, 64, 64), (512, 128, 128)):
    g = L.make_grid(3, n, (1.0, 0.25, 0.25))
    nodes = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    res = {i: [] for i in range(len(cfgs))}
    for rnd in range(3):
        for i, cfg in enumerate(cfgs):
            tun(**cfg)
            ms = C.c_double()
            L.check(ctx, lib.pdegpu_elasticity_time_apply(ctx, g, el, 0, 30, C.byref(ms)))
            res[i].append(ms.value)
    for i, cfg in enumerate(cfgs):
        m = float(np.median(res[i]))
        print(f"{n} {cfg}: {m*1e3:.1f} us/apply, {48*nodes/m/1e6:.0f} GB/s alg ({48*nodes/m/8e9:.3f})", flush=True)
for cfg in (dict(el_fast_xface=0, el_fast_waves=4096), dict(el_fast_xface=1, el_fast_waves=4096), dict(el_fast_xface=1, el_fast_waves=2048)):
    tun(**cfg)
    for rep in range(2):
        t0 = time.perf_counter()
        f = elasticity._solve_elasticity_3d_static(1.0, 0.25, 0.25, 256, 64, 64, 210e9, 0.3, 0.0, 0.0, -76518.0, "stress", rtol=1e-8)
        dt = time.perf_counter() - t0
    print(f"{cfg}: config-5 solve {dt:.3f} s, {f.meta['cg_iterations']} its, {dt/f.meta['cg_iterations']*1e6:.1f} us/it, max vm {f.values.max():.9e}", flush=True)
tun(el_fast_xface=1, el_fast_waves=4096)

# ---- (b) heat finalists, 8 interleaved rounds
n = 512
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
N = (n - 1) ** 3
enc = lambda c: c[0] + 16 * c[1] + 256 * c[2] + 512 * c[3]
plain = [(2, 0, 0, 0), (4, 2, 0, 1), (4, 0, 0, 1), (4, 2, 0, 0), (2, 2, 0, 1)]
k1 = [(2, 0, 1, 0), (2, 2, 1, 0), (4, 0, 1, 0), (4, 2, 1, 0), (2, 2, 0, 0)]
k2 = [(2, 0, 1, 0), (2, 2, 1, 0), (2, 3, 1, 0), (2, 2, 0, 0), (4, 2, 1, 0)]
res = [[] for _ in range(5)]
for rnd in range(8):
    for i in range(5):
        tun(fast_cfg_plain=enc(plain[i]), fast_cfg_res=enc(plain[i]), fast_cfg_k1=enc(k1[i]), fast_cfg_k2=enc(k2[i]))
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
        res[i].append((ms.value, st.kernel_ms[0] / 20, st.kernel_ms[1] / 20))
        lib.pdegpu_session_destroy(s)
for i in range(5):
    a = np.array(res[i])
    md, mn = np.median(a, axis=0), a.min(axis=0)
    print(f"PLAIN {plain[i]}: median {md[0]:.4f} ({16*N/md[0]/8e9:.3f}) min {mn[0]:.4f} | K1 {k1[i]}: median {md[1]:.4f} ({40*N/md[1]/8e9:.3f}) min {mn[1]:.4f} | "
          f"K2 {k2[i]}: median {md[2]:.4f} ({24*N/md[2]/8e9:.3f}) min {mn[2]:.4f}", flush=True)
