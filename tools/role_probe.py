"""Does K1's mode change when the roles (x, r, p_old, p_new) are played by OTHER buffers of the same session?  One 512^3 session, its six vectors
allocated once; K1 timed (pdegpu_session_debug_time_k1) for every ordered pair (buffer as x, buffer as p_new) — the two written streams — with r and
p_old taken from the rest in order, and for every pair (x, r) likewise.  usage: python tools/role_probe.py [sessions]"""
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
h = L.Heat()
h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 1, 1.0
h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
h.bc_present[0] = 1
names = "x r p0 p1 q b".split()


def t(s, roles):
    ms = C.c_double()
    L.check(ctx, lib.pdegpu_session_debug_time_k1(s, (C.c_int32 * 4)(*roles), 4, C.byref(ms)))
    return ms.value


for sess in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    s = C.c_void_p()
    L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
    L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
    L.check(ctx, lib.pdegpu_session_residual(s))
    L.check(ctx, lib.pdegpu_session_iterate(s, 3, 0, None))
    print(f"session {sess}: rows = buffer as x, columns = buffer as p_new (r, p_old: the first two of the rest)")
    print("          " + "  ".join(f"{nm:>5s}" for nm in names))
    for xb in range(6):
        row = []
        for pn in range(6):
            if pn == xb:
                row.append("    -")
                continue
            rest = [i for i in range(6) if i not in (xb, pn)]
            row.append(f"{t(s, (xb, rest[0], rest[1], pn)):.3f}")
        print(f"    {names[xb]:>4s}  " + "  ".join(row), flush=True)
    print("  rows = buffer as x, columns = buffer as r (p_old, p_new: the first two of the rest)")
    for xb in range(6):
        row = []
        for rb in range(6):
            if rb == xb:
                row.append("    -")
                continue
            rest = [i for i in range(6) if i not in (xb, rb)]
            row.append(f"{t(s, (xb, rb, rest[0], rest[1])):.3f}")
        print(f"    {names[xb]:>4s}  " + "  ".join(row), flush=True)
    lib.pdegpu_session_destroy(s)
