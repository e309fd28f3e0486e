"""The multi-process rendezvous of the library's own RCCL communicator (csrc/comm.hip) as far as ONE GPU lets it go: 2 processes (gloo group for the
bootstrap) both on cuda:0 make the unique id on rank 0, carry it through the group and call pdegpu_comm_create(rank, 2).  RCCL's bootstrap (sockets,
peer-info all-gather) runs for real; it then either refuses the duplicate device — the expected end on this pool: the error must come back as a
PdeGpuError on EVERY rank, not as a hang — or, where it accepts it, the slab CG runs over it and must reproduce the gloo-staged run.
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 tools/rehearse_native_comm.py"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import torch, torch.distributed as dist
from pdegpu import _lib as L
from pdegpu.dist import GpuSlabEngine, NativeComm, SlabCG, slab_range

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
n = 24
nzg = n * world
grid = L.make_grid(3, (n, n, nzg), (1.0, 1.0, float(world)))
heat = L.Heat(); heat.kappa, heat.dt, heat.steady, heat.source = 1.0, 0.0, 1, 1.0
heat.bc_mode, heat.fast_path = L.BC_UNIFORM, 1; heat.bc_present[0] = 1
k0, nk = slab_range(nzg + 1, world, rank)
stream = torch.cuda.Stream(device=0)
with torch.cuda.stream(stream):
    eng = GpuSlabEngine(grid, heat, k0, nk, 0, stream)
    try:
        comm = NativeComm(eng, rank, world)
        outcome = "created"
    except L.PdeGpuError as e:
        comm, outcome = None, f"refused: {e}"
    print(f"rank {rank}: pdegpu_comm_create -> {outcome}", flush=True)
    flags = [None] * world
    dist.all_gather_object(flags, comm is not None)
    assert all(f == flags[0] for f in flags), flags            # every rank saw the same end
    if comm is not None:
        hist = {}
        for label, c in (("staged", None), ("library", comm)):
            cg = SlabCG(eng, rank, world, comm=c)
            cg.start(0.0)
            cg.iterate(30)
            hist[label] = cg.rr()
        print(f"rank {rank}: rr after 30 iterations: {hist}", flush=True)
        assert abs(hist["library"] - hist["staged"]) <= 1e-9 * hist["staged"]
        comm.close()
    eng.close()
dist.barrier()
if rank == 0:
    print("RENDEZVOUS " + ("RAN" if flags[0] else "REFUSED (duplicate device)") + " OK", flush=True)
dist.destroy_process_group()
