"""N > 1 with the real HIP engine: two and three processes (one z-slab each) share this box's single GPU and run the product's
multi-rank driver (pdegpu.dist.SlabCG + GpuSlabEngine) under `gloo` with the planes staged through the host — RCCL refuses
several ranks on one device, and the pool gives one GPU per box; the nccl transport itself is covered by
test_rccl_backend_on_library_owned_memory.  Rank 0 compares solution and residual history with a single-slab session."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("variant", ["sr", "classic"])
@pytest.mark.parametrize("world,cells", [(2, 24), (3, 16)])
def test_multi_process_slab_cg_matches_single_slab(lib, gpu_ctx, world, cells, variant):
    """both recurrences of the driver: "sr" = single reduction (ONE 24-byte scalar message per iteration, the default), "classic" = two"""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tools" / "rehearse_ranks.py"), str(cells), "80", variant]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "REHEARSAL OK" in out.stdout


def test_bench_line_of_the_multi_rank_path(lib, gpu_ctx):
    """bench.py under the launcher the driver uses for N > 1, here 2 ranks sharing the GPU over gloo (rehearsal mode): one JSON
    line from rank 0 with the contract's keys and the whole-job unknown count"""
    import json
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PDEGPU_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--cells", "48"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["config"]["unknowns"] == 47 * 47 * 95
    assert d["value"] > 0 and abs(d["value"] - d["config"]["unknowns"] * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]
    # the strong-scaling extra carries its own parity check: the same system, the same iterations, on one GPU alone
    chk = d["extras"]["config3_strong"]["single_gpu_check"]
    assert chk["rr_single_gpu"] > 0 and chk["rel_diff"] < 1e-9, chk
    assert set(d["extras"]["config3_strong"]["overlap_ab_ms_per_step"]) == {"overlapped", "exchange_first", "overlapped_2", "exchange_first_2"}
    chk5 = d["extras"]["config5_elasticity"]["single_gpu_check"]                    # BASELINE config 5 on two slabs against one
    assert chk5["rr_single_gpu"] > 0 and chk5["rel_diff"] < 1e-8, chk5


def test_bench_line_at_four_ranks_on_one_gpu(lib, gpu_ctx):
    """the widest rehearsal this pool allows on a card (its process guard admits six GPU processes: this test's own context, the launcher and 4 ranks;
    world size 8 itself runs on the CPU, tests/test_dist_gloo.py::test_slab_cg_at_the_world_size_of_the_scaling_run): bench.py under the driver's
    launcher, 49 node planes over 4 ranks (slabs of 13 and 12 planes, config 5's 65 planes as 17 + 3 x 16), JSON contract and the three single-GPU
    parity figures"""
    import json
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PDEGPU_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "4", "--steps", "6", "--warmup", "2", "--cells", "48", "--extras-timeout", "500"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["scaling"] == "weak" and d["config"]["unknowns"] == 47 * 47 * (4 * 48 - 1) and d["value"] > 0
    assert "aborted" not in d["extras"], d["extras"]
    for key, bound in (("config3_strong", 1e-9), ("config4_strong", 1e-9), ("config5_elasticity", 1e-8)):
        chk = d["extras"][key]["single_gpu_check"]
        assert chk["rr_single_gpu"] > 0 and chk["rel_diff"] < bound, (key, chk)


def test_multi_process_elasticity_slabs(lib, gpu_ctx):
    """solve_elasticity_slabs with 2 ranks sharing the GPU (gloo rehearsal): von Mises field == the single-process solve; k1 split
    around the exchange (forced on these 5-plane slabs: the default splits slabs of >= 32 MiB per vector)"""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PDEGPU_EL_SPLIT_MIN_MB="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tools" / "rehearse_elasticity.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "REHEARSAL OK" in out.stdout


def test_library_communicator_rendezvous_across_processes(lib, gpu_ctx):
    """the multi-process side of csrc/comm.hip as far as one GPU lets it go: two processes make / carry the unique id and call pdegpu_comm_create(rank, 2)
    on the same device — RCCL's bootstrap runs for real, then refuses the duplicate device: the refusal must come back as an error on BOTH ranks (no hang);
    where a RCCL build accepts it, the slab CG runs over it and must reproduce the staged run (tools/rehearse_native_comm.py decides)"""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tools" / "rehearse_native_comm.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "RENDEZVOUS REFUSED (duplicate device) OK" in out.stdout or "RENDEZVOUS RAN OK" in out.stdout
