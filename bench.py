#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json): matrix-free CG on the 3-D heat system,
512 x 512 x 512 cells per GPU, fp64, on N GPUs of one node (one process per GPU, slabs along z, RCCL halo
exchange + dot-product all-reduce).

A "step" is ONE CG iteration (fused K1: x-update + p-update + 7-point SpMV + p.q; K2: r-update + r.r) of the
steady system  kappa K u = f m  with f = 1, T_boundary = 0 on synthetic data of that
shape.  value = free unknowns x K / wall time of the K timed iterations (max over ranks).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  At N = 1 it carries `roofline` (dominant kernel timed live with HIP events on the
library's stream) and `cpu_baseline` (the oracle's C/OpenMP port of the same CG timed on this host's cores on a
bounded sample)."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "pde-solver_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(n_cells: int, budget_s: float = 20.0):
    """same-algorithm CPU baseline: oracle/cg_port.c (classic matrix-free CG, OpenMP) on this host's cores.
    Bounded sample: the same n^3 grid, as many iterations as fit in ~budget_s (at least 3)."""
    from oracle import cg_port
    threads = min(cg_port.max_threads(), len(os.sched_getaffinity(0)))
    nodes = n_cells + 1
    h = 1.0 / n_cells
    w = cg_port.p1_weights((h, h, h), 0.0, 1.0)
    unknowns = (nodes - 2) ** 3
    secs, _, _ = cg_port.run((nodes,) * 3, w, h ** 3, 0.0, 2, threads)          # calibration (2 iterations)
    per_iter = max(secs / 2, 1e-6)
    iters = int(max(3, min(200, budget_s / per_iter)))
    secs, rr, _ = cg_port.run((nodes,) * 3, w, h ** 3, 0.0, iters, threads)
    return {"value": unknowns * iters / secs, "unit": "DoF/s", "cores": threads, "kind": "port",
            "sample": f"{iters} CG iterations of the same {n_cells}^3 steady heat system (oracle/cg_port.c, "
                      f"classic unfused CG, OpenMP static, {threads} threads); the reference's own algorithm "
                      f"(sparse LU per step, FEniCS 2019.1.0) is not installable offline",
            "ms_per_step": secs / iters * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cells", type=int, default=512, help="cells per axis per GPU (BASELINE config 3: 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--general", action="store_true", help="force the class-table kernels (diagnostic)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pdegpu import _lib as L
    from pdegpu.dist import GpuSlabEngine, SlabCG, slab_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # PDEGPU_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, planes are
    # staged through the host by pdegpu.dist); the measured configuration is always nccl = RCCL, one rank per GPU
    backend = os.environ.get("PDEGPU_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n = args.cells
    # weak scaling: every GPU owns a 512-cell-thick slab; the global grid is n x n x (n*N) cells
    nz_global = n * world
    planes = nz_global + 1
    k0, nk = slab_range(planes, world, rank)
    grid = L.make_grid(3, (n, n, nz_global), (1.0, 1.0, float(world)))
    heat = L.Heat()
    heat.kappa, heat.dt, heat.steady, heat.source = 1.0, 0.0, 1, 1.0
    heat.bc_mode, heat.fast_path = L.BC_UNIFORM, 0 if args.general else 1
    heat.bc_present[0] = 1
    unknowns = (n - 1) * (n - 1) * (nz_global - 1)

    stream = torch.cuda.Stream(device=local_rank)
    with torch.cuda.stream(stream):
        eng = GpuSlabEngine(grid, heat, k0, nk, local_rank, stream)
        cg = SlabCG(eng, rank, world)
        # in-run streaming ceilings and the bare SpMV are measured FIRST (they use r, q, b as scratch, so they must precede the
        # CG set-up; they also bring the device to its sustained clocks before the warm-up steps)
        ceil, spmv_ms = {}, None
        if world == 1:
            vec_bytes = eng.plane * nk * 8
            for which, name, nvec in ((1, "read_1R", 1), (2, "write_1W", 1), (0, "copy_1R1W", 2), (3, "triad_2R1W", 3)):
                mb = C.c_double()
                L.check(eng.ctx, eng.lib.pdegpu_session_membench(eng.s, which, 20, C.byref(mb)))
                ceil[name] = nvec * vec_bytes / (mb.value * 1e-3) / 1e9
            ms = C.c_double()
            L.check(eng.ctx, eng.lib.pdegpu_session_time_apply(eng.s, 50, C.byref(ms)))
            spmv_ms = ms.value
        cg.start(0.0)
        stats = L.Stats()

        def run(iters, timed):
            if world == 1:
                eng.iterate_native(iters, timed, stats if timed else None)
            else:
                cg.iterate(iters)

        def fence():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()

        run(args.warmup, False)
        fence()
        t0 = time.perf_counter()
        run(args.steps, True)
        fence()
        elapsed = time.perf_counter() - t0
        rr = eng.rr()

        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        out = {
            "metric": "CG DoF/s (3D heat, matrix-free CG)", "value": unknowns * args.steps / elapsed, "unit": "DoF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"3D heat {n}x{n}x{n} cells per GPU (BASELINE configs[2]: 512^3 P1 grid, steady "
                                   f"kappa*K u = f*m, f=1, T_boundary=0), one CG iteration per step",
                       "global_cells": [n, n, nz_global], "unknowns": unknowns, "element_mode": "p1split",
                       "parallelism": (f"z-slabs x{world}" + ("" if backend == "nccl" else f" (REHEARSAL over {backend}, shared GPU: not a measurement)")) if world > 1 else "single GPU",
                       "kernel_path": "general" if args.general else "fused z-marching",
                       "rr_after": rr},
        }

        if world > 1:
            # per-GPU view of the whole iteration (kernels + halo exchange + two all-reduces); the per-kernel roofline is the N=1 line's
            per_gpu = 64.0 * unknowns / world / (elapsed / args.steps) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": per_gpu, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": per_gpu / HBM_PEAK_GBS,
                               "traffic": None, "kernel": "whole CG iteration per GPU: K1 + K2 (64 B/DoF algorithmic) + r-plane halo exchange + "
                                                          "two 8-byte all-reduces; kernel-level figures are in the n_gpus=1 line"}
        if world == 1:
            # ---- roofline of the dominant kernel (K1 = the SpMV-bearing fused kernel), HIP events ----------
            k1_ms = stats.kernel_ms[0] / max(1, stats.kernel_launches[0])
            k2_ms = stats.kernel_ms[1] / max(1, stats.kernel_launches[1])
            # K1 reads r, p, x and writes p, x (+ q in the 72-B variant); K2 reads r and p (or q), writes r
            nodes_local = (n + 1) * (n + 1) * nk
            k1_per_dof = round(stats.alg_bytes[0] / max(1, stats.kernel_launches[0]) / nodes_local)      # 40 or 48
            k1_bytes = float(k1_per_dof) * unknowns
            k2_bytes = 24.0 * unknowns
            ach = k1_bytes / (k1_ms * 1e-3) / 1e9
            traffic = None
            pmc = ROOT / "profiles" / "pmc_latest.json"
            if pmc.exists():
                try:
                    traffic = json.loads(pmc.read_text()).get("k1_hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "kernel": "k_fast<7,2,CG_K1,nt,ex> (x+=a*p; p=r+b*p; q=A p [not stored]; p.q)" if not args.general else "k_general<CG_K1>",
                               "alg_bytes_per_launch": k1_bytes, "avg_ms": k1_ms,
                               "other_kernels": [{"kernel": "k_fast<7,2,CG_K2,nt,ex> (r-=a*(A p); r.r)" if k1_per_dof == 40 else "k_k2 (r-=a*q; r.r)", "alg_bytes_per_launch": k2_bytes,
                                                  "avg_ms": k2_ms, "achieved": k2_bytes / (k2_ms * 1e-3) / 1e9,
                                                  "frac": k2_bytes / (k2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}],
                               "cg_iteration_alg_bytes_per_dof": k1_per_dof + 24}
            out["roofline"]["in_run_ceiling_GBps"] = ceil
            # SpMV alone (16 B/DoF) for the north-star statement
            out["roofline"]["spmv_only"] = {"kernel": "k_fast<7,2,PLAIN,nt> (q = A p, 16 B/DoF)", "avg_ms": spmv_ms,
                                            "achieved": 16.0 * unknowns / (spmv_ms * 1e-3) / 1e9,
                                            "frac": 16.0 * unknowns / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        eng.close()

    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(n)
        except Exception as e:                                   # the baseline must never void the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "DoF/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
