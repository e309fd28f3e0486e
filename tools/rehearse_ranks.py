"""Rehearsal of the multi-rank driver with the real HIP engine on ONE GPU: N processes (gloo, planes staged through the host —
RCCL refuses several ranks on one device) each own a z-slab on cuda:0 and run pdegpu.dist.SlabCG exactly as bench.py does at
N > 1; rank 0 checks the gathered solution and residual history against a single-slab session of the same global grid.
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/rehearse_ranks.py [cells] [iters] [sr|classic]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import numpy as np
import torch, torch.distributed as dist
from pdegpu import _lib as L
from pdegpu.dist import GpuSlabEngine, SlabCG, gather_slabs, slab_range, solve_heat_slabs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 120
variant = sys.argv[3] if len(sys.argv) > 3 else "sr"      # sr: ONE 24-byte scalar message per iteration; classic: two 8-byte ones
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
lib = L.load_library()
nzg = n * world
grid = L.make_grid(3, (n, n, nzg), (1.0, 1.0, float(world)))
heat = L.Heat(); heat.kappa, heat.dt, heat.steady, heat.source = 1.0, 0.0, 1, 1.0
heat.bc_mode, heat.fast_path = L.BC_UNIFORM, 1; heat.bc_present[0] = 1
k0, nk = slab_range(nzg + 1, world, rank)
stream = torch.cuda.Stream(device=0)
with torch.cuda.stream(stream):
    eng = GpuSlabEngine(grid, heat, k0, nk, 0, stream)
    cg = SlabCG(eng, rank, world, variant=variant)
    assert cg.sr == (variant == "sr")
    cg.start(0.0)
    hist = [cg.rr()]
    for _ in range(iters // 10):
        cg.iterate(10)
        hist.append(cg.rr())
    cg.finish()
    assert eng.can_split_k1() and cg.overlap          # the overlapped order: K1's bulk runs before the ghost planes arrive
    x = eng.owned_x()
    eng.close()
xs = gather_slabs(x, nzg + 1, rank, world)             # fp64 tensors to rank 0 (no pickling)
if rank == 0:
    xs = xs.ravel()
    with torch.cuda.stream(stream):
        one = GpuSlabEngine(grid, heat, 0, nzg + 1, 0, stream)
        c1 = SlabCG(one, 0, 1, variant="classic")       # the reference run always uses the classic recurrences
        c1.start(0.0)
        h1 = [c1.rr()]
        for _ in range(iters // 10):
            c1.iterate(10)
            h1.append(c1.rr())
        c1.finish()
        x1 = np.empty((nzg + 1) * (n + 1) * (n + 1))
        L.check(one.ctx, lib.pdegpu_session_get_x(one.s, L.dptr(x1)))
        one.close()
    rel = np.abs(xs - x1).max() / np.abs(x1).max()
    hrel = max(abs(a - b) / b for a, b in zip(hist, h1))
    print(f"rehearsal world={world} variant={variant} cells={n}x{n}x{nzg}: max rel diff of x vs single slab {rel:.2e}, residual history rel diff {hrel:.2e}, rr {hist[-1]:.6e}", flush=True)
    assert rel < 1e-11 and hrel < 1e-9
# the tool-level distributed solve: three backward-Euler steps of a box whose longest axis is x (the slabs must cut it), against
# the single-GPU tool
m = (3 * n // 2, n // 2, n // 2 + 1)
out = solve_heat_slabs(m, (1.5, 0.5, 0.6), 1.3, 0.5, T_initial=20.0, dt=0.01, num_steps=3, source=2.0, rtol=1e-12)
if rank == 0:
    from pdegpu import heat as H
    one = H._solve_heat_3d_raw(1.5, 0.5, 0.6, *m, 1.3, 0.5, 20.0, 0.01, 3, source_type="constant", source_value=2.0, rtol=1e-12)
    rel = np.abs(out["values"].reshape(4, -1) - one.values).max() / np.abs(one.values).max()
    print(f"solve_heat_slabs world={world} cells={m}: 4 states vs the single-GPU tool, max rel diff {rel:.2e}, iterations {out['iterations']}", flush=True)
    assert rel < 1e-9 and np.allclose(out["times"], one.times)
    print("REHEARSAL OK", flush=True)
dist.destroy_process_group()
