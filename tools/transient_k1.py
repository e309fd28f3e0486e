"""K1 / K2 of the 512^3 TRANSIENT session (15-point operator, K1 stores q: three written streams), for placement A/B across processes.
usage: python tools/transient_k1.py [lr]   (lr: natural sides, T_left + T_right)"""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
n = 512
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
N = (n - 1) ** 3
h = L.Heat()
h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 0, 1.0
h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
h.bc_present[0] = 1
if len(sys.argv) > 1 and sys.argv[1] == "lr":
    h.bc_mode = L.BC_DIRECTIONAL
    h.bc_present[0], h.bc_value[0], h.bc_present[1], h.bc_value[1] = 1, 10.0, 1, 0.0
s = C.c_void_p()
L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
L.check(ctx, lib.pdegpu_session_residual(s))
L.check(ctx, lib.pdegpu_session_iterate(s, 5, 0, None))
st = L.Stats()
L.check(ctx, lib.pdegpu_session_iterate(s, 40, 1, st))
k1, k2 = st.kernel_ms[0] / 40, st.kernel_ms[1] / 40
print(f"transient 512^3: K1 {k1:.4f} ms  K2 {k2:.4f} ms  -> {N/(k1+k2)/1e6:.1f} GDoF/s")
lib.pdegpu_session_destroy(s)
