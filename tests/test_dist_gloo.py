"""N>1 path on CPU: pdegpu.dist.SlabCG (the product's multi-GPU driver) under torch.distributed `gloo`,
world sizes 1, 2 and 3, with the numpy slab engine of tests/slab_engine_numpy.py standing in for the HIP
session.  Checks that the slab-decomposed CG reproduces (a) the single-rank iterates and (b) the oracle's
solution of the same system."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
N_CELLS, LS, KAPPA, SRC, TB = (7, 6, 11), (1.0, 0.8, 1.5), 1.3, 2.0, 0.5
T_CELLS, T_LS = (9, 5, 6), (1.5, 0.7, 1.0)          # longest axis first: solve_heat_slabs must cut it (axis permutation)
ITERS = 120


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker_transient(rank, world, port, outdir):
    """solve_heat_slabs (the backward-Euler loop over slabs, axis permutation included) with the numpy engine under gloo"""
    for p in (ROOT, ROOT / "pde-solver_amd", ROOT / "tests"):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pdegpu.dist import solve_heat_slabs
        from slab_engine_numpy import NumpySlabEngine

        def factory(n, Ls, hk, k0, nk):
            return NumpySlabEngine(n, Ls, hk["kappa"], hk["source"], hk["T_boundary"], k0, nk, dt=None if hk["steady"] else hk["dt"])
        out = solve_heat_slabs(T_CELLS, T_LS, KAPPA, TB, T_initial=3.0, dt=0.02, num_steps=3, source=SRC, rtol=1e-12,
                               engine_factory=factory)
        if rank == 0:
            np.savez(Path(outdir) / "transient.npz", values=out["values"], times=out["times"], iters=np.array(out["iterations"]))
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, outdir, variant="sr"):
    for p in (ROOT, ROOT / "pde-solver_amd", ROOT / "tests"):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pdegpu.dist import SlabCG, slab_range
        from slab_engine_numpy import NumpySlabEngine
        k0, nk = slab_range(N_CELLS[2] + 1, world, rank)
        eng = NumpySlabEngine(N_CELLS, LS, KAPPA, SRC, TB, k0, nk)
        cg = SlabCG(eng, rank, world, variant=variant)
        assert cg.sr == (variant == "sr")
        cg.start(0.0)
        rr0 = cg.rr()
        cg.iterate(5)
        rr5 = cg.rr()
        cg.iterate(ITERS - 5)
        rr_end = cg.rr()
        cg.finish()
        np.savez(Path(outdir) / f"rank{rank}.npz", x=eng.owned_x(), k0=k0, nk=nk, rr=np.array([rr0, rr5, rr_end]))
    finally:
        dist.destroy_process_group()


def _run(world, tmp_path, variant="sr"):
    out = tmp_path / f"w{world}{variant}"
    out.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), str(out), variant), nprocs=world, join=True)
    parts = [np.load(out / f"rank{r}.npz") for r in range(world)]
    assert sum(int(p["nk"]) for p in parts) == N_CELLS[2] + 1
    x = np.concatenate([p["x"] for p in parts], axis=0)
    for p in parts[1:]:
        assert np.array_equal(p["rr"], parts[0]["rr"])            # every rank sees the same all-reduced scalars
    return x, parts[0]["rr"]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("variant", ["sr", "classic"])
def test_slab_cg_world_1_2_3_agree_and_match_oracle(tmp_path, variant):
    """both recurrences of the driver — single reduction (ONE 24-byte message per iteration, the default) and classic (two 8-byte ones)"""
    from oracle import fem_p1 as o
    x1, rr1 = _run(1, tmp_path, variant)
    ref = o.solve_heat_3d(*LS, *N_CELLS, KAPPA, TB, 0.0, 0.1, 1, steady=True, source_type="constant", source_value=SRC)
    ref = ref.values[0].reshape(x1.shape)
    assert np.abs(x1 - ref).max() < 1e-9 * np.abs(ref).max()
    assert rr1[2] < 1e-20 * rr1[0]
    if variant == "sr":                                           # the same iterates as the classic recurrences (exact arithmetic: identical)
        xc, rrc = _run(1, tmp_path, "classic")
        assert np.allclose(rr1[:2], rrc[:2], rtol=1e-10) and np.abs(x1 - xc).max() < 1e-11 * np.abs(ref).max()
    for world in (2, 3):
        xw, rrw = _run(world, tmp_path, variant)
        assert np.allclose(rrw[:2], rr1[:2], rtol=1e-11)           # same iterates up to summation order
        assert np.abs(xw - ref).max() < 1e-9 * np.abs(ref).max()
        assert np.abs(xw - x1).max() < 1e-11 * np.abs(ref).max()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("variant", ["sr", "classic"])
def test_slab_cg_at_the_world_size_of_the_scaling_run(tmp_path, variant):
    """world size 8 — what the driver's scaling bench launches: 12 node planes over 8 ranks, i.e. slabs of two planes and of ONE plane (its only plane
    is next to both ghosts and is first and last owned plane at once), the first and the last slab holding a Dirichlet plane; partition, ownership of
    the dot products and the thin-slab paths of SlabCG must reproduce the one-slab iterates"""
    x1, rr1 = _run(1, tmp_path, variant)
    x8, rr8 = _run(8, tmp_path, variant)
    assert np.allclose(rr8[:2], rr1[:2], rtol=1e-11)
    assert np.abs(x8 - x1).max() < 1e-11 * np.abs(x1).max()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [1, 2, 3])
def test_transient_heat_over_slabs_matches_oracle(tmp_path, world):
    """the distributed time loop (x-halo exchange per step, step right-hand side, reference norm, CG, gather to rank 0, axes
    permuted so that the slabs cut the longest axis) against the oracle's assembled backward-Euler solve"""
    from oracle import fem_p1 as o
    mp.spawn(_worker_transient, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "transient.npz")
    ref = o.solve_heat_3d(*T_LS, *T_CELLS, KAPPA, TB, 3.0, 0.02, 3, source_type="constant", source_value=SRC)
    assert got["values"].shape == (4, T_CELLS[2] + 1, T_CELLS[1] + 1, T_CELLS[0] + 1)
    assert np.allclose(got["times"], ref.times)
    err = np.abs(got["values"].reshape(4, -1) - ref.values).max()
    assert err < 1e-9 * np.abs(ref.values).max(), err


def test_longest_axis_last():
    from pdegpu.dist import longest_axis_last
    assert longest_axis_last((1024, 512, 512), (2.0, 1.0, 1.0)) == ([1, 2, 0], (512, 512, 1024), (1.0, 1.0, 2.0))
    assert longest_axis_last((512, 512, 512), (1.0, 1.0, 1.0))[0] == [0, 1, 2]
    assert longest_axis_last((256, 64, 64), (1.0, 0.25, 0.25))[1] == (64, 64, 256)


def test_slab_engine_matches_c_port():
    """the numpy engine itself is checked against the oracle's C port (same CG, rr history)"""
    for p in (ROOT / "tests",):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    from oracle import cg_port
    from pdegpu.dist import SlabCG
    from slab_engine_numpy import NumpySlabEngine
    eng = NumpySlabEngine(N_CELLS, LS, KAPPA, SRC, TB, 0, N_CELLS[2] + 1)
    cg = SlabCG(eng, 0, 1, variant="classic")
    cg.start(0.0)
    hist = [float(eng.scalars()[3])]
    for _ in range(10):
        cg.iterate(1)
        hist.append(float(eng.scalars()[3]))
    eng2 = NumpySlabEngine(N_CELLS, LS, KAPPA, SRC, TB, 0, N_CELLS[2] + 1)
    cg2 = SlabCG(eng2, 0, 1, variant="sr")                       # single-reduction form: the same residual history
    cg2.start(0.0)
    hist2 = [cg2.rr()]
    for _ in range(10):
        cg2.iterate(1)
        hist2.append(cg2.rr())
    assert np.allclose(hist2, hist, rtol=1e-10)
    h = [LS[r] / N_CELLS[r] for r in range(3)]
    _, rr, _ = cg_port.run(tuple(n + 1 for n in N_CELLS), cg_port.p1_weights(h, 0.0, KAPPA), SRC * h[0] * h[1] * h[2], TB, 10)
    assert np.allclose(hist, rr, rtol=1e-10)
