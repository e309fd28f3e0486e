"""512^3 with natural faces: K1 / K2 with the column x = NX-1 inside the marching kernel (hybrid_xcol = 1, default) and in the class-table pass (0),
interleaved, steady (64 B/DoF form) and transient; T_left only | T_side only.  usage: python tools/r3_hybrid_ab.py [n] [side]"""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
N = (n + 1) ** 3
for steady in (1, 0):
    for rnd in range(3):
        for xcol in (1, 0):
            L.check(ctx, lib.pdegpu_set_tunable(ctx, b"hybrid_xcol", xcol))
            h = L.Heat()
            h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, steady, 1.0
            h.bc_mode, h.fast_path = L.BC_DIRECTIONAL, 1
            h.bc_present[0], h.bc_value[0] = 1, 10.0
            if len(sys.argv) > 2 and sys.argv[2] == "side":
                h.bc_present[0], h.bc_present[2], h.bc_value[2] = 0, 1, 3.0
            s = C.c_void_p()
            L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
            L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
            L.check(ctx, lib.pdegpu_session_residual(s))
            L.check(ctx, lib.pdegpu_session_iterate(s, 4, 0, None))
            st = L.Stats()
            L.check(ctx, lib.pdegpu_session_iterate(s, 20, 1, st))
            rr = C.c_double(); L.check(ctx, lib.pdegpu_session_rr(s, C.byref(rr)))
            k1, k2 = st.kernel_ms[0] / 20, st.kernel_ms[1] / 20
            print(f"steady={steady} hybrid_xcol={xcol}: K1 {k1:.3f} ms  K2 {k2:.3f} ms  -> {N/(k1+k2)/1e6:.1f} GDoF/s   rr {rr.value:.9e}", flush=True)
            lib.pdegpu_session_destroy(s)
