"""K1 with the deeper prefetch (plane k+2 requested before plane k is computed; p_{n-1} and x of the plane in flight travel with it): A/B on ONE session
(same allocations: K1's placement modes cancel), alternating the per-mode word of K1 between launches.  512^3 steady, ms per CG iteration and per K1."""
import ctypes as C, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import torch
from pdegpu import _lib as L
from pdegpu.dist import GpuSlabEngine, SlabCG
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
h = L.Heat(); h.kappa, h.dt, h.steady, h.source = 1.0, 0.0, 1, 1.0
h.bc_mode, h.fast_path = L.BC_UNIFORM, 1; h.bc_present[0] = 1
stream = torch.cuda.Stream(device=0)
with torch.cuda.stream(stream):
    eng = GpuSlabEngine(grid, h, 0, n + 1, 0, stream)
    cg = SlabCG(eng, 0, 1)
    cg.start(0.0)
    cg.iterate(30)
    for rep in range(4):
        for word in (-1, 770, 258):
            L.check(eng.ctx, eng.lib.pdegpu_set_tunable(eng.ctx, b"fast_cfg_k1", word))
            cg.iterate(10)
            torch.cuda.synchronize()
            eng.set_timing(1)
            t0 = time.perf_counter()
            cg.iterate(100)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 100 * 1e3
            st = L.Stats(); eng.get_stats(st); eng.set_timing(0)
            print(f"fast_cfg_k1 {word:4d}: iteration {wall:.4f} ms, K1 {st.kernel_ms[0] / max(1, st.kernel_launches[0]):.4f} ms, K2 {st.kernel_ms[1] / max(1, st.kernel_launches[1]):.4f} ms, rr {cg.rr():.3e}", flush=True)
    eng.close()
