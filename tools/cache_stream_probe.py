"""Streaming rates of a session whose vectors are the size of config 5's (27.6 MB each, all of them inside the 256 MiB Infinity Cache): what a flat
pass over cache-resident vectors can reach — the ceiling of the elasticity update pass (7 vector passes, 34 us at config 5 = 5.3 TB/s).
which: 0 copy (1R+1W), 1 read (1R), 2 write (1W), 3 triad (2R+1W), 4 in-place update (2R+1W)."""
import ctypes as C
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L
lib = L.load_library(); ctx = L.context(0)
for n in ((256, 64, 194), (256, 64, 64), (512, 128, 128)):
    grid = L.make_grid(3, n, (1.0, 1.0, 1.0))
    h = L.Heat()
    h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 1, 1.0
    h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
    h.bc_present[0] = 1
    s = C.c_void_p()
    L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n[2] + 1, C.byref(s)))
    plane = int(lib.pdegpu_session_plane_doubles(s))
    vec_bytes = plane * (n[2] + 1) * 8
    out = []
    for which, name, nvec in ((1, "read", 1), (2, "write", 1), (0, "copy", 2), (3, "triad", 3), (4, "update", 3)):
        ms = C.c_double()
        L.check(ctx, lib.pdegpu_session_membench(s, which, 200, C.byref(ms)))
        out.append(f"{name} {nvec * vec_bytes / (ms.value * 1e-3) / 1e12:.2f} TB/s ({ms.value * 1e3:.1f} us)")
    print(f"{n}: vector {vec_bytes / 1e6:.1f} MB: " + ", ".join(out), flush=True)
    lib.pdegpu_session_destroy(s)
