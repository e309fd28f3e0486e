"""Per-rank surrogate of the multi-GPU runs on ONE GPU: what one rank of an N-way slab decomposition computes per CG iteration
— K1 (split: bulk + ONE launch for the edge planes), K2 and the reduce launch(es) of the stepwise protocol, in the single-reduction form
("sr": one reduce launch, 4 launches per iteration) and the classic one (two) — timed on an interior slab of nk planes of
the real global grid (ghost planes on both sides, no exchange: a rank's own work).  With the single-GPU iteration time t_1 and the
surrogate t_N, strong-scaling efficiency is bounded by t_1 / (N t_N) before any communication; adding the measured or estimated
exchange + all-reduce time gives the projection committed in profiles/.  Two more walls per row put the messages of an interior rank back in, sent
to the rank itself (real RCCL point-to-point messages of the real plane size + the scalar all-reduce call; device-local copies instead of xGMI, so
the wire time is a lower bound, the launch / stream / host path is the real one): issued by the library on its own communicator from C
(pdegpu_session_iterate_slab, csrc/comm.hip) and by pdegpu.dist.SlabCG through torch.distributed.  usage: python tools/slab_surrogate.py [out.json]"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "pde-solver_amd")]
import os
import torch
import torch.distributed as dist
from pdegpu import _lib as L
from pdegpu.dist import GpuSlabEngine, GpuElasticitySlabEngine, NativeComm, SlabCG, slab_range

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))

lib = L.load_library()
out = {"heat": [], "elasticity": []}
stream = torch.cuda.Stream(device=0)


def walls_with_messages(eng, cg, iters, planes):
    """ms per iteration of the same loop with an interior rank's messages (to self): {"library_rccl": ..., "torch_distributed": ...}"""
    def timed():
        cg.iterate(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cg.iterate(iters)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    walls = {}
    comm = NativeComm(eng, 0, 1)
    comm.set_self_loop(True)
    cg.comm = comm
    walls["library_rccl"] = timed()
    cg.comm = None
    comm.close()
    nk = eng.nk

    def self_exchange(halo=None, x_planes=False):
        ops = []
        for own, ghost in (((0, -1),) if eng.k0 > 0 else ()) + (((nk - 1, nk),) if eng.k0 + nk < planes else ()):
            for snd, rcv in zip(eng.halo_tensors(own), eng.halo_tensors(ghost)):
                ops += [dist.P2POp(dist.isend, snd, 0), dist.P2POp(dist.irecv, rcv, 0)]
        return ("planes", dist.batch_isend_irecv(ops))

    cg.force_collectives, cg.world, cg.exchange_begin = True, 2, self_exchange
    walls["torch_distributed"] = timed()
    cg.force_collectives, cg.world = False, 1
    del cg.exchange_begin
    return walls


def heat_case(cells, Ls, world, label, iters=60, variant="sr"):
    nz = cells[2]
    rank = world // 2 if world > 2 else (1 if world == 2 else 0)           # an interior rank where there is one
    k0, nk = slab_range(nz + 1, world, rank)
    grid = L.make_grid(3, cells, Ls)
    h = L.Heat()
    h.kappa, h.steady, h.source = 1.0, 1, 1.0
    h.bc_mode, h.fast_path = L.BC_UNIFORM, 1
    h.bc_present[0] = 1
    with torch.cuda.stream(stream):
        eng = GpuSlabEngine(grid, h, k0, nk, 0, stream)
        cg = SlabCG(eng, 0, 1, variant=variant)                             # world 1: no exchange, the rank's own launches
        cg.start(0.0)
        cg.iterate(10)
        torch.cuda.synchronize()
        # wall time WITHOUT the per-launch HIP events (two event records per launch cost ~2 us of stream time each at these launch rates) ...
        t0 = time.perf_counter()
        cg.iterate(iters)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / iters * 1e3
        # ... and the kernel times from a second, event-bracketed pass
        eng.set_timing(True)
        t0 = time.perf_counter()
        cg.iterate(iters)
        torch.cuda.synchronize()
        wall_timed = (time.perf_counter() - t0) / iters * 1e3
        st = L.Stats()
        eng.get_stats(st)
        eng.set_timing(False)
        msg = walls_with_messages(eng, cg, iters, nz + 1) if world > 1 else None
        eng.close()
    k1, k2 = st.kernel_ms[0] / iters, st.kernel_ms[1] / iters
    unk = (cells[0] - 1) * (cells[1] - 1) * max(0, min(k0 + nk, nz) - max(k0, 1))
    row = {"workload": label, "variant": variant, "world": world, "rank": rank, "planes": nk, "unknowns_of_rank": unk, "k1_ms": k1, "k2_ms": k2,
           "iteration_wall_ms": wall, "iteration_wall_ms_with_event_pairs": wall_timed, "iteration_wall_ms_with_messages_to_self": msg, "k1_frac_of_8TBs": 40.0 * unk / (k1 * 1e-3) / 8e12, "k2_frac_of_8TBs": 24.0 * unk / (k2 * 1e-3) / 8e12}
    out["heat"].append(row)
    print(row, flush=True)


def el_case(cells, Ls, world, iters=200):
    nz = cells[2]
    rank = world // 2 if world > 2 else (1 if world == 2 else 0)
    k0, nk = slab_range(nz + 1, world, rank)
    grid = L.make_grid(3, cells, Ls)
    el = L.Elasticity()
    el.E, el.nu = 210e9, 0.3
    el.body[2] = -76518.0
    with torch.cuda.stream(stream):
        eng = GpuElasticitySlabEngine(grid, el, k0, nk, 0, stream)
        cg = SlabCG(eng, 0, 1)
        cg.start(0.0)
        cg.iterate(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cg.iterate(iters)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / iters * 1e6
        msg = walls_with_messages(eng, cg, iters, nz + 1) if world > 1 else None
        eng.close()
    row = {"workload": "config 5 (256x64x64 elasticity)", "world": world, "rank": rank, "planes": nk, "iteration_wall_us": wall,
           "iteration_wall_us_with_messages_to_self": {k: 1e3 * v for k, v in msg.items()} if msg else None}
    out["elasticity"].append(row)
    print(row, flush=True)


for variant in ("sr", "classic"):
    for world in (1, 2, 4, 8):
        heat_case((512, 512, 512), (1.0, 1.0, 1.0), world, "config 3 strong (512^3 split)", variant=variant)
    for world in (1, 2, 4, 8):
        heat_case((512, 512, 1024), (1.0, 1.0, 2.0), world, "config 4 (1024x512x512, slabs across the 1024-cell axis)", variant=variant)
for world in (1, 2, 4, 8):
    el_case((256, 64, 64), (1.0, 0.25, 0.25), world)
t1 = {(r["workload"], r["variant"]): r["iteration_wall_ms"] for r in out["heat"] if r["world"] == 1}
for r in out["heat"]:
    r["compute_bound_on_efficiency"] = t1[(r["workload"], r["variant"])] / (r["world"] * r["iteration_wall_ms"])
    if r["iteration_wall_ms_with_messages_to_self"]:
        r["efficiency_with_messages_to_self"] = {k: t1[(r["workload"], r["variant"])] / (r["world"] * v) for k, v in r["iteration_wall_ms_with_messages_to_self"].items()}
e1 = [r for r in out["elasticity"] if r["world"] == 1][0]["iteration_wall_us"]
for r in out["elasticity"]:
    r["compute_bound_on_efficiency"] = e1 / (r["world"] * r["iteration_wall_us"])
    if r["iteration_wall_us_with_messages_to_self"]:
        r["efficiency_with_messages_to_self"] = {k: e1 / (r["world"] * v) for k, v in r["iteration_wall_us_with_messages_to_self"].items()}
out["note"] = ("one MI355X; each row = the launches ONE rank of an N-way slab decomposition issues per CG iteration through pdegpu.dist.SlabCG "
               "(sr: K1 bulk + K1 edge planes + reduce + K2 = 4 launches; classic: + a second reduce), on an interior slab of the global grid, "
               "without the exchange itself; compute_bound_on_efficiency = t(1) / (N t(N)) within one variant is an UPPER bound of the "
               "strong-scaling efficiency — the halo exchange (2.2 MB planes per neighbour, overlapped with K1's bulk) and ONE 24-byte "
               "all-reduce per iteration (classic: two 8-byte ones) come on top.  *_with_messages_to_self: the same loop with an interior rank's messages "
               "put back, addressed to the rank itself (real RCCL point-to-point messages of the real size between the rank's own planes + the all-reduce "
               "call of a one-rank communicator, which RCCL completes without a kernel): library_rccl = issued by libpdegpu from C on its own communicator and "
               "stream, torch_distributed = issued by pdegpu.dist.SlabCG through torch.distributed; wire time is a device-local copy, not xGMI")
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps(out, indent=1))
dist.destroy_process_group()
