"""rocprofv3 input: 512^3 steady heat with natural faces (T_left only), 30 iterations of the hybrid path."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
n = 512
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
h = L.Heat()
h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 1, 1.0
h.bc_mode, h.fast_path = L.BC_DIRECTIONAL, 1
h.bc_present[0], h.bc_value[0] = 1, 10.0
if len(sys.argv) > 1 and sys.argv[1] == "lr":
    h.bc_present[1], h.bc_value[1] = 1, 0.0
s = C.c_void_p()
L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
L.check(ctx, lib.pdegpu_session_residual(s))
L.check(ctx, lib.pdegpu_session_iterate(s, 30, 0, None))
lib.pdegpu_session_destroy(s)
