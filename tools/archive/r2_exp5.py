"""Round-2 experiment 5 (one GPU): per-mode sweep of the marching kernels — rows per thread, resident workgroups per CU, LDS
edge hand-off, deeper prefetch — for the bare SpMV, K1 and K2 at 512^3.  Interleaved rounds, medians.
usage: python tools/r2_exp5.py [n] [rounds]"""
import ctypes as C
import itertools
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
from pdegpu import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib = L.load_library()
ctx = L.context(0)
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
N = (n - 1) ** 3
combos = [(ry, wg, ex, pf) for ry in (2, 4) for wg in (0, 2, 3, 4) for ex in (0, 1) for pf in (0, 1)]
res = {c: [] for c in combos}
for rnd in range(rounds):
    for c in combos:
        ry, wg, ex, pf = c
        v = ry + 16 * wg + 256 * ex + 512 * pf
        for key in (b"fast_cfg_plain", b"fast_cfg_res", b"fast_cfg_k1", b"fast_cfg_k2"):
            L.check(ctx, lib.pdegpu_set_tunable(ctx, key, v))
        h = L.Heat()
        h.kappa, h.dt, h.steady, h.source = 1.0, 0.0025, 1, 1.0
        h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
        h.bc_present[0] = 1
        s = C.c_void_p()
        L.check(ctx, lib.pdegpu_session_create(ctx, grid, h, 0, n + 1, C.byref(s)))
        ms = C.c_double()
        L.check(ctx, lib.pdegpu_session_time_apply(s, 15, C.byref(ms)))
        L.check(ctx, lib.pdegpu_session_begin(s, 0.0))
        L.check(ctx, lib.pdegpu_session_residual(s))
        L.check(ctx, lib.pdegpu_session_reduce_local(s, 1, 1))
        L.check(ctx, lib.pdegpu_session_iterate(s, 3, 0, None))
        st = L.Stats()
        L.check(ctx, lib.pdegpu_session_iterate(s, 15, 1, st))
        rr = C.c_double()
        L.check(ctx, lib.pdegpu_session_rr(s, C.byref(rr)))
        k1, k2 = st.kernel_ms[0] / 15, st.kernel_ms[1] / 15
        res[c].append((ms.value, k1, k2))
        print(f"round {rnd} ry={ry} wg={wg} ex={ex} pf={pf}: SpMV {ms.value:.4f} | K1 {k1:.4f} | K2 {k2:.4f} | rr {rr.value:.12e}", flush=True)
        lib.pdegpu_session_destroy(s)
med = {c: np.median(np.array(r), axis=0) for c, r in res.items()}
for col, name, b in ((0, "SpMV", 16), (1, "K1", 40), (2, "K2", 24)):
    order = sorted(combos, key=lambda c: med[c][col])
    print(f"--- {name}: best five (ry, wg, ex, pf)")
    for c in order[:5]:
        print(f"    {c}: {med[c][col]:.4f} ms = {b*N/med[c][col]/8e9:.3f} of 8 TB/s")
    print(f"    default-like (2, 0, {1 if col else 0}, 0): {med[(2, 0, 1 if col else 0, 0)][col]:.4f} ms")
