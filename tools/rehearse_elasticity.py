"""Rehearsal of solve_elasticity_slabs (3-D elasticity over z-slabs, one process per slab) on ONE GPU under gloo: rank 0 compares the
von Mises field and the iteration count with the single-process solve of the same problem.
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29621 tools/rehearse_elasticity.py"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import numpy as np
import torch, torch.distributed as dist
from pdegpu import _lib as L
from pdegpu.dist import solve_elasticity_slabs

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
n, Ls = (40, 8, 9), (1.0, 0.25, 0.25)
grid = L.make_grid(3, n, Ls)
el = L.Elasticity()
el.E, el.nu, el.quantity = 210e9, 0.3, L.QUANTITY_STRESS if hasattr(L, "QUANTITY_STRESS") else 0
el.body[0], el.body[1], el.body[2] = 0.0, 0.0, -76518.0
out = solve_elasticity_slabs(grid, el, n, rtol=1e-10)
if rank == 0:
    from pdegpu import elasticity
    one = elasticity._solve_elasticity_3d_static(*Ls, *n, 210e9, 0.3, 0.0, 0.0, -76518.0, "stress", rtol=1e-10)
    rel = np.abs(out["field"] - one.values[0]).max() / np.abs(one.values[0]).max()
    print(f"elasticity rehearsal world={world}: {out['iterations']} iterations (single process {one.meta['cg_iterations']}), "
          f"von Mises max rel diff {rel:.2e}", flush=True)
    assert rel < 1e-7
    print("REHEARSAL OK", flush=True)
dist.destroy_process_group()
