"""Round-2 experiment 16 (one GPU): z-chunk COUNT sweep of the heat marching kernels over many grid sizes — the data the chunk policy of
fast_geom is fitted to.  One line per (n, chunks): workgroups of the 2-row kernels, fractions of the 8 TB/s peak of K1 / K2 / bare SpMV.
usage: python tools/r2_exp16.py [n ...]"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L

lib = L.load_library()
ctx = L.context(0)


def tun(**kw):
    for k, v in kw.items():
        L.check(ctx, lib.pdegpu_set_tunable(ctx, k.encode(), v))


def xy_of(n, ry):
    ntx = (n + 127) // 128
    wxt = 4 if ntx % 4 == 0 else (2 if ntx % 2 == 0 else 1)
    wyt = 4 // wxt
    nty = (n - 1 + ry - 1) // ry
    return ((ntx + wxt - 1) // wxt) * ((nty + wyt - 1) // wyt)


sizes = [int(v) for v in sys.argv[1:]] or [320, 352, 384, 416, 448, 480, 512, 544, 576, 608, 640, 672, 700, 736, 768, 832, 896, 960, 1024]
for n in sizes:
    grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
    N = (n - 1) ** 3
    planes = n - 1
    h = L.Heat()
    h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 1, 1.0
    h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
    h.bc_present[0] = 1
    s = C.c_void_p()
    L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
    seen = set()
    for c in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 20, 24, 32, 48):
        z = 0 if c == 0 else (planes + c - 1) // c
        cc = 0 if c == 0 else (planes + z - 1) // z
        if (cc, z) in seen or (c and z < 8):
            continue
        seen.add((cc, z))
        tun(fast_zlen=z)
        ms = C.c_double()
        L.check(ctx, lib.pdegpu_session_time_apply(s, 10, C.byref(ms)))
        L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
        L.check(ctx, lib.pdegpu_session_residual(s))
        L.check(ctx, lib.pdegpu_session_iterate(s, 2, 0, None))
        st = L.Stats()
        L.check(ctx, lib.pdegpu_session_iterate(s, 12, 1, st))
        k1, k2, sp = st.kernel_ms[0] / 12, st.kernel_ms[1] / 12, ms.value
        print(f"n={n:4d} chunks={cc:3d} zlen={z:4d} W2={xy_of(n, 2) * cc:6d} W4={xy_of(n, 4) * cc:6d}  K1 {40*N/k1/8e9:.3f}  K2 {24*N/k2/8e9:.3f}  SpMV {16*N/sp/8e9:.3f}  "
              f"-> {N/(k1+k2)/1e6:.1f} GDoF/s", flush=True)
    lib.pdegpu_session_destroy(s)
    tun(fast_zlen=0)
