"""15-point bare SpMV (A = M + dt kappa K, 512^3) under the per-mode configuration word of the marching kernels: rows per thread x deeper prefetch.
usage: python tools/archive/r3_spmv15_cfg.py"""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import torch
from pdegpu import _lib as L
from pdegpu.dist import GpuSlabEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
stream = torch.cuda.Stream(device=0)
unk = (n - 1) ** 3
with torch.cuda.stream(stream):
    for dt, label in ((-1.0, "27-point (Q1)"), (0.0025, "15-point"), (0.0, "7-point")):
        if dt < 0:
            grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0), L.ELEM_Q1)
            dt = 0.0
        else:
            grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
        h = L.Heat(); h.kappa, h.dt, h.steady, h.source = 1.0, dt, 0 if dt else 1, 1.0
        h.bc_mode, h.fast_path = L.BC_UNIFORM, 1; h.bc_present[0] = 1
        for rep in range(2):
            for word in (-1, 4, 516, 2, 514):
                eng = GpuSlabEngine(grid, h, 0, n + 1, 0, stream)
                L.check(eng.ctx, eng.lib.pdegpu_set_tunable(eng.ctx, b"fast_cfg_plain", word))
                ms = C.c_double()
                L.check(eng.ctx, eng.lib.pdegpu_session_time_apply(eng.s, 60, C.byref(ms)))
                print(f"{label} cfg {word:4d} (ry {word & 15 if word >= 0 else 'dflt'}, pf {(word >> 9) & 1 if word >= 0 else 'dflt'}): {ms.value:.4f} ms = {16.0 * unk / ms.value / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
                eng.close()
