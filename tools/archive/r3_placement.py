"""One fresh process = one set of allocations: 512^3 session with the vectors allocated by mode PDEGPU_VEC_CONTIG (0 hipMalloc per vector,
1 hipExtMallocWithFlags contiguous, 2 / 3 one VMM handle of minimum / recommended granularity, 4 / 5 one hipMalloc arena carved at 2 MiB / 1 GiB),
placement tuner off; prints K1 / K2 / bare SpMV times.  Run from tools/r3_placement.sh in alternation.  usage: python tools/r3_placement.py"""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
os.environ.setdefault("PDEGPU_NO_TORCH_PRELOAD", "1")
os.environ.setdefault("PDEGPU_PLACE_TUNE", "0")
from pdegpu import _lib as L

lib = L.load_library()
ctx = L.context(0)
n = 512
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
h = L.Heat()
h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 1, 1.0
h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
h.bc_present[0] = 1
s = C.c_void_p()
L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
L.check(ctx, lib.pdegpu_session_residual(s))
L.check(ctx, lib.pdegpu_session_iterate(s, 6, 0, None))
k1, k2 = [], []
for rep in range(3):
    st = L.Stats()
    L.check(ctx, lib.pdegpu_session_iterate(s, 20, 1, st))
    k1.append(st.kernel_ms[0] / 20)
    k2.append(st.kernel_ms[1] / 20)
ms = C.c_double()
L.check(ctx, lib.pdegpu_session_time_apply(s, 20, C.byref(ms)))
ptrs = [lib.pdegpu_session_plane_ptr(s, v, 0) for v in range(4)]
print(f"mode {os.environ.get('PDEGPU_VEC_CONTIG', '0')} stagger {os.environ.get('PDEGPU_VEC_STAGGER_KB', '-')} tune {os.environ.get('PDEGPU_PLACE_TUNE')}: K1 " + " ".join(f"{a:.4f}" for a in k1) + "  K2 " + " ".join(f"{b:.4f}" for b in k2) +
      f"  spmv {ms.value:.4f}  x r p q at " + " ".join(hex(q or 0) for q in ptrs), flush=True)
lib.pdegpu_session_destroy(s)
