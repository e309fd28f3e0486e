"""Host-side cost per CG iteration of the Python-driven multi-GPU loop (SlabCG.iterate, single-reduction form) on a grid whose GPU time is negligible, with a
one-rank nccl group: (a) the three C-ABI calls alone, (b) + the ONE scalar all-reduce, (c) + an exchange of two planes as a batch of point-to-point messages
(sent to self: the call path of the real exchange); then the same three through the library's own RCCL communicator (csrc/comm.hip): (d) whole chunks from C
(pdegpu_session_iterate_slab; RCCL completes the in-place one-rank all-reduce without a kernel: the call path only), (e) a slab with both neighbours = itself: exchange of its first / last plane posted
through pdegpu_session_exchange_begin / _end + all-reduce, per iteration, from Python.  usage: python tools/dist_overhead.py [cells]"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import torch, torch.distributed as dist
from pdegpu import _lib as L
from pdegpu.dist import GpuSlabEngine, NativeComm, SlabCG, VEC_R
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
grid = L.make_grid(3, (n, n, n), (1.0, 1.0, 1.0))
heat = L.Heat(); heat.kappa, heat.dt, heat.steady, heat.source = 1.0, 0.0, 1, 1.0
heat.bc_mode, heat.fast_path = L.BC_UNIFORM, 1; heat.bc_present[0] = 1
stream = torch.cuda.Stream(device=0)
with torch.cuda.stream(stream):
    eng = GpuSlabEngine(grid, heat, 0, n + 1, 0, stream)
    cg = SlabCG(eng, 0, 1)
    cg.start(0.0)
    cg.iterate(50); torch.cuda.synchronize()
    scratch = eng.plane_tensor(3, 2), eng.plane_tensor(3, 3)          # two planes of q as receive targets (q is not used by the 64-byte form)

    def self_exchange():
        ops = [dist.P2POp(dist.isend, eng.plane_tensor(VEC_R, 1), 0), dist.P2POp(dist.irecv, scratch[0], 0),
               dist.P2POp(dist.isend, eng.plane_tensor(VEC_R, n - 1), 0), dist.P2POp(dist.irecv, scratch[1], 0)]
        return ("planes", dist.batch_isend_irecv(ops))

    for label, force, exch in (("C-ABI calls only", False, False), ("+ one all-reduce", True, False), ("+ all-reduce + plane exchange", True, True)):
        cg.force_collectives = force
        cg.world = 2 if exch else 1
        cg.exchange_begin = (lambda halo=None: self_exchange()) if exch else (lambda halo=None: None)
        for reps in (300, 1000):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); cg.iterate(reps); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            print(f"{label}: {reps} iterations: host enqueue {1e6*(t1-t0)/reps:.1f} us/it, incl. drain {1e6*(t2-t0)/reps:.1f} us/it", flush=True)
    comm = NativeComm(eng, 0, 1)
    cg.force_collectives, cg.world, cg.comm = False, 1, comm
    del cg.exchange_begin
    for reps in (300, 1000):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); cg.iterate(reps); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"library RCCL, iterate_slab (all-reduce only): {reps} iterations: host enqueue {1e6*(t1-t0)/reps:.1f} us/it, incl. drain {1e6*(t2-t0)/reps:.1f} us/it", flush=True)
    comm.close()
    eng.close()
    # a slab with a neighbour on both sides, both of them rank 0 itself (the numbers mean nothing; the calls and the messages are the real ones)
    mid = GpuSlabEngine(grid, heat, 1, n - 1, 0, stream)
    comm = NativeComm(mid, 0, 1)
    mid.set_variant(True); mid.begin(0.0); mid.residual()
    lib, SC_PQ = mid.lib, 2
    for reps in (300, 1000):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            comm.group_begin(); mid.comm_exchange_post(comm, 3, False, 0, 0); comm.group_end()
            mid.k1_part(0); mid.comm_exchange_end(comm); mid.k1_part(1); mid.comm_allreduce(comm, SC_PQ, 3); mid.k2()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"library RCCL, exchange with self + all-reduce, Python loop of C-ABI calls: {reps} iterations: host enqueue {1e6*(t1-t0)/reps:.1f} us/it, "
              f"incl. drain {1e6*(t2-t0)/reps:.1f} us/it", flush=True)
    comm.close()
    mid.close()
dist.destroy_process_group()
