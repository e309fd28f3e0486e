"""Scratch: host-side cost per CG iteration of the Python-driven multi-GPU loop (SlabCG.iterate) on a grid whose GPU time
is negligible, with a one-rank nccl group so that the two all-reduces per iteration are really issued."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import torch, torch.distributed as dist
from pdegpu import _lib as L
from pdegpu.dist import GpuSlabEngine, SlabCG
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
    cg.world = 2            # force the all-reduce calls; halos have no neighbour at rank 0 of "2"... keep exchange off
    cg.exchange_halos = lambda: None
    cg.start(0.0)
    cg.iterate(50); torch.cuda.synchronize()
    for reps in ((200, 1000) if n < 200 else (100, 200)):
        t0 = time.perf_counter(); cg.iterate(reps); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{reps} iterations: host enqueue {1e6*(t1-t0)/reps:.1f} us/it, incl. drain {1e6*(t2-t0)/reps:.1f} us/it", flush=True)
    eng.close()
dist.destroy_process_group()
