"""Scratch: 512^3 heat CG in Q1 element mode (27-point): marching kernel vs class-table kernel; steady and transient."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = (n - 1) ** 3
for em, name in ((L.ELEM_Q1, "Q1 27-point"), (L.ELEM_P1SPLIT, "P1 split")):
    grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0), em)
    for steady in (1, 0):
        for fast in ((1, 0) if em == L.ELEM_Q1 else (1,)):
            h = L.Heat()
            h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, steady, 1.0
            h.bc_mode, h.fast_path = L.BC_UNIFORM, fast
            h.bc_present[0] = 1
            s = C.c_void_p()
            L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
            L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
            L.check(ctx, lib.pdegpu_session_residual(s))
            L.check(ctx, lib.pdegpu_session_iterate(s, 4, 0, None))
            st = L.Stats()
            L.check(ctx, lib.pdegpu_session_iterate(s, 20, 1, st))
            ms = C.c_double(); L.check(ctx, lib.pdegpu_session_time_apply(s, 20, C.byref(ms)))
            k1, k2 = st.kernel_ms[0] / 20, st.kernel_ms[1] / 20
            per = (st.alg_bytes[0] + st.alg_bytes[1]) / 20 / ((n + 1) ** 3)      # K1 + K2 (the average of K2's two forms where it carries the x update)
            print(f"{name} steady={steady} fast={fast}: K1 {k1:.3f} ms  K2 {k2:.3f} ms -> {N/(k1+k2)/1e6:.1f} GDoF/s ({per:.0f} B/DoF: {per*N/(k1+k2)/1e6:.0f} GB/s); "
                  f"SpMV {ms.value:.3f} ms = {16*N/ms.value/1e6:.0f} GB/s", flush=True)
            lib.pdegpu_session_destroy(s)
