"""PMC input: K1 / K2 / bare SpMV at 512^3 under four geometries, a few launches each (run under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE).
Variants are told apart by kernel name (rows per thread) and grid size (workgroups)."""
import ctypes as C
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library()
ctx = L.context(0)
n = 512
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
for ry, wg in ((2, 2), (2, 6), (4, 2), (4, 6)):
    v = ry + 16 * wg + 256 * 1
    vp = ry + 16 * wg
    for key, val in ((b"fast_cfg_plain", vp), (b"fast_cfg_res", vp), (b"fast_cfg_k1", v), (b"fast_cfg_k2", v)):
        L.check(ctx, lib.pdegpu_set_tunable(ctx, key, val))
    h = L.Heat()
    h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 1, 1.0
    h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
    h.bc_present[0] = 1
    s = C.c_void_p()
    L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
    ms = C.c_double()
    L.check(ctx, lib.pdegpu_session_time_apply(s, 3, C.byref(ms)))
    L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
    L.check(ctx, lib.pdegpu_session_residual(s))
    L.check(ctx, lib.pdegpu_session_iterate(s, 6, 0, None))
    lib.pdegpu_session_destroy(s)
    print("done", ry, wg, flush=True)
