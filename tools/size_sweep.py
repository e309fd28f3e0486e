"""CG kernels of the steady 3-D heat path by grid size (one GPU): catches geometry pathologies away from 512^3 (partly filled rounds, short
z-chunks).  usage: python tools/size_sweep.py [n ...]"""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
sizes = [int(v) for v in sys.argv[1:]] or [192, 256, 320, 384, 400, 448, 512, 576, 640, 700, 768, 896]
for n in sizes:
    grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
    N = (n - 1) ** 3
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
    L.check(ctx, lib.pdegpu_session_iterate(s, 4, 0, None))
    st = L.Stats()
    L.check(ctx, lib.pdegpu_session_iterate(s, 20, 1, st))
    k1, k2 = st.kernel_ms[0] / 20, st.kernel_ms[1] / 20
    nodes = (n + 1) ** 3
    b1, b2 = st.alg_bytes[0] / 20 / nodes, st.alg_bytes[1] / 20 / nodes      # algorithmic B/DoF of K1 / of K2 (average over its two forms where K2 updates x)
    print(f"n={n:4d}: K1 {k1:.4f} ms ({b1:.0f} B/DoF, {b1*N/k1/8e9:.3f})  K2 {k2:.4f} ms ({b2:.0f} B/DoF, {b2*N/k2/8e9:.3f})  SpMV {ms.value:.4f} ms ({16*N/ms.value/8e9:.3f})  -> {N/(k1+k2)/1e6:.1f} GDoF/s", flush=True)
    lib.pdegpu_session_destroy(s)
