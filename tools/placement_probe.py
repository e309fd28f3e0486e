"""Is K1's slow / fast mode a property of WHERE a session's vectors landed?  One process creates the 512^3 session several times, with dummy
device allocations of varying size in between (so that the allocator hands out different memory), and times K1 / K2 each time.
usage: python tools/placement_probe.py [sessions]"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L

lib = L.load_library()
ctx = L.context(0)
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
n = 512
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
N = (n - 1) ** 3
h = L.Heat()
h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 1, 1.0
h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
h.bc_present[0] = 1
held = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    # perturb: keep one more dummy block of an odd size alive each round
    p = C.c_void_p()
    size = (37 + 61 * it) * (1 << 20) + 4096 * (it % 7)
    assert hip.hipMalloc(C.byref(p), size) == 0
    held.append(p)
    s = C.c_void_p()
    L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
    L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
    L.check(ctx, lib.pdegpu_session_residual(s))
    L.check(ctx, lib.pdegpu_session_iterate(s, 5, 0, None))
    out = []
    for rep in range(3):
        st = L.Stats()
        L.check(ctx, lib.pdegpu_session_iterate(s, 20, 1, st))
        out.append((st.kernel_ms[0] / 20, st.kernel_ms[1] / 20))
    ptrs = [lib.pdegpu_session_plane_ptr(s, v, 0) for v in range(4)]
    print(f"session {it}: K1 " + " ".join(f"{a:.4f}" for a, _ in out) + "  K2 " + " ".join(f"{b:.4f}" for _, b in out) +
          "  x r p q at " + " ".join(hex(q or 0) for q in ptrs), flush=True)
    lib.pdegpu_session_destroy(s)
for p in held:
    hip.hipFree(p)
