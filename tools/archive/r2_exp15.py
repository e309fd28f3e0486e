"""Round-2 experiment 15 (one GPU): the heat marching kernels away from 512^3 — forced z-chunk lengths and rows per thread, interleaved
rounds, to see what the automatic geometry leaves on the table at 5 / 6 tiles per row.  usage: python tools/r2_exp15.py [n ...]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L

lib = L.load_library()
ctx = L.context(0)


def tun(**kw):
    for k, v in kw.items():
        L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), v))


enc = lambda ry, wg, ex, pf: ry + 16 * wg + 256 * ex + 512 * pf
sizes = [int(v) for v in sys.argv[1:]] or [640, 768]
for n in sizes:
    cfgs = [("auto", dict(fast_zlen=0, fast_cfg_k1=-1, fast_cfg_k2=-1, fast_cfg_plain=-1))]
    for z in (24, 32, 48, 64, 96, 128, 192, 256, 384, n):
        cfgs.append((f"zlen {z}", dict(fast_zlen=z, fast_cfg_k1=-1, fast_cfg_k2=-1, fast_cfg_plain=-1)))
    for ry in (4,):
        for z in (0, 64, 128, 256):
            cfgs.append((f"ry {ry} zlen {z}", dict(fast_zlen=z, fast_cfg_k1=enc(ry, 0, 1, 0), fast_cfg_k2=enc(ry, 0, 1, 0), fast_cfg_plain=enc(ry, 0, 0, 1))))
    grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
    N = (n - 1) ** 3
    h = L.Heat()
    h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 1, 1.0
    h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
    h.bc_present[0] = 1
    s = C.c_void_p()
    L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
    res = [[] for _ in cfgs]
    for rnd in range(3):
        for i, (_, cfg) in enumerate(cfgs):
            tun(**cfg)
            ms = C.c_double()
            L.check(ctx, lib.pdegpu_session_time_apply(s, 10, C.byref(ms)))
            L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
            L.check(ctx, lib.pdegpu_session_residual(s))
            L.check(ctx, lib.pdegpu_session_iterate(s, 2, 0, None))
            st = L.Stats()
            L.check(ctx, lib.pdegpu_session_iterate(s, 10, 1, st))
            res[i].append((st.kernel_ms[0] / 10, st.kernel_ms[1] / 10, ms.value))
    for i, (name, _) in enumerate(cfgs):
        k1, k2, sp = np.median(np.array(res[i]), axis=0)
        print(f"n={n:4d} {name:16s}: K1 {k1:.4f} ms ({40*N/k1/8e9:.3f})  K2 {k2:.4f} ms ({24*N/k2/8e9:.3f})  SpMV {sp:.4f} ms ({16*N/sp/8e9:.3f})  -> {N/(k1+k2)/1e6:.1f} GDoF/s", flush=True)
    lib.pdegpu_session_destroy(s)
    tun(fast_zlen=0, fast_cfg_k1=-1, fast_cfg_k2=-1, fast_cfg_plain=-1)
