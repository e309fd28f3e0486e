"""The launches ONE interior rank of an 8-way slab decomposition of config 3 (512^3, 64 planes) issues per CG iteration, for a kernel trace:
rocprofv3 --kernel-trace --stats -- python3 tools/r3_rank_trace.py [world] [variant] [comm].  variant: classic | sr (single reduction);
comm: none | library (the rank's messages put back, addressed to itself, issued by csrc/comm.hip from C: pdegpu_comm_set_self_loop)"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import torch
from pdegpu import _lib as L
from pdegpu.dist import GpuSlabEngine, NativeComm, SlabCG, slab_range

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variant = sys.argv[2] if len(sys.argv) > 2 else "classic"
with_comm = len(sys.argv) > 3 and sys.argv[3] == "library"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 100
cells = (512, 512, 512)
rank = world // 2 if world > 2 else (1 if world == 2 else 0)
k0, nk = slab_range(cells[2] + 1, world, rank)
grid = L.make_grid(3, cells, (1.0, 1.0, 1.0))
h = L.Heat()
h.kappa, h.steady, h.source = 1.0, 1, 1.0
h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
h.bc_present[0] = 1
stream = torch.cuda.Stream(device=0)
with torch.cuda.stream(stream):
    eng = GpuSlabEngine(grid, h, k0, nk, 0, stream)
    try:
        cg = SlabCG(eng, 0, 1, variant=variant)
    except TypeError:
        cg = SlabCG(eng, 0, 1)
    cg.start(0.0)
    comm = None
    if with_comm:
        comm = NativeComm(eng, 0, 1)
        comm.set_self_loop(True)
        cg.comm = comm
    cg.iterate(10)
    torch.cuda.synchronize()
    cg.iterate(iters)
    torch.cuda.synchronize()
    if comm:
        comm.close()
    eng.close()
print("done", world, variant, nk)
