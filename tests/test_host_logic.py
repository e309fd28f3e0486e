"""CPU-only checks of the product's host logic and of the C-ABI library itself (no compute on a GPU):
the library loads and exports every symbol include/pdegpu.h declares, the device's macro-element
integration (run through its host twin, same source) agrees with the oracle's assembled matrices for
every boundary class, the slab partition is a partition, and the host-side load vector of
project(Expression(degree=2)) agrees with the oracle."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import fem_p1 as o

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol(lib):
    from pdegpu import _lib
    per_header = {}
    for name in ("pdegpu.h", "pdegpu_stepwise.h"):              # the tool-level / session API and the driver / rehearsal API
        header = (ROOT / "include" / name).read_text()
        per_header[name] = set(re.findall(r"^[a-z][a-z0-9_ *]*?[ *](pdegpu_[a-z0-9_]+)\s*\(", header, re.M)) - {"pdegpu_ctx", "pdegpu_session"}
        assert per_header[name], f"no declarations found in {name}"
    assert not (per_header["pdegpu.h"] & per_header["pdegpu_stepwise.h"])
    # the single steps of an iteration live in the second header only
    assert {"pdegpu_session_k1", "pdegpu_session_k2_part", "pdegpu_session_exchange_post", "pdegpu_el_session_k1_pre"} <= per_header["pdegpu_stepwise.h"]
    assert {"pdegpu_heat_solve", "pdegpu_elasticity_solve", "pdegpu_session_iterate", "pdegpu_session_iterate_slab", "pdegpu_comm_create"} <= per_header["pdegpu.h"]
    declared = per_header["pdegpu.h"] | per_header["pdegpu_stepwise.h"]
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in the headers but not exported by libpdegpu.so: {missing}"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert b"gfx950" in lib.pdegpu_version()


def test_struct_layout_matches_header(lib):
    from pdegpu import _lib
    assert C.sizeof(_lib.Grid) == 48
    assert C.sizeof(_lib.Heat) == 8 * 2 + 8 + 8 + 8 + 8 + 24 + 12 + 4 + 8 + 8
    assert C.sizeof(_lib.Stats) == 8 * 3 + 8 + 8 * 3 + 32 + 32 + 32 + 8


def _class_node(c, n):
    return {0: 0, 1: 1, 2: n}[c]


@pytest.mark.parametrize("dim,n,Ls", [(1, (3,), (2.0,)), (2, (3, 2), (1.0, 0.5)), (3, (2, 3, 2), (1.0, 0.6, 0.25))])
def test_device_tables_match_oracle_assembly(lib, dim, n, Ls):
    """every (boundary class, offset) weight of the class tables == the assembled matrix entry"""
    from pdegpu import _lib
    g = _lib.make_grid(dim, n, Ls, _lib.ELEM_P1SPLIT)
    tK, tM, ld = np.zeros(729), np.zeros(729), np.zeros(27)
    assert lib.pdegpu_heat_tables_host(g, _lib.dptr(tK), _lib.dptr(tM), _lib.dptr(ld)) == 0
    tK, tM = tK.reshape(27, 27), tM.reshape(27, 27)
    if dim == 1:
        c, cells = o.interval_mesh(n[0], 0.0, Ls[0])
    elif dim == 2:
        c, cells = o.rectangle_mesh(Ls[0], Ls[1], n[0], n[1])
    else:
        c, cells = o.box_mesh(*Ls, *n)
    K, M, load = o.assemble_scalar(c, cells)
    K, M = K.toarray(), M.toarray()
    nn = [n[r] + 1 if r < dim else 1 for r in range(3)]
    scaleK, scaleM = np.abs(K).max(), np.abs(M).max()
    checked = 0
    for cls in range(27):
        cc = (cls % 3, (cls // 3) % 3, cls // 9)
        if any(cc[r] != 1 for r in range(dim, 3)):
            continue                                    # classes of unused axes are never looked up
        idx3 = [_class_node(cc[r], n[r]) if r < dim else 0 for r in range(3)]
        if any(r < dim and n[r] < 2 and cc[r] == 1 for r in range(3)):
            continue
        row = idx3[0] + nn[0] * (idx3[1] + nn[1] * idx3[2])
        assert ld[cls] == pytest.approx(load[row], rel=1e-13)
        for off in range(27):
            d = (off % 3 - 1, (off // 3) % 3 - 1, off // 9 - 1)
            nb = [idx3[r] + d[r] for r in range(3)]
            inside = all(0 <= nb[r] < nn[r] for r in range(3))
            if not inside:
                assert tK[cls, off] == 0.0 and tM[cls, off] == 0.0
                continue
            col = nb[0] + nn[0] * (nb[1] + nn[1] * nb[2])
            assert tK[cls, off] == pytest.approx(K[row, col], abs=1e-13 * scaleK)
            assert tM[cls, off] == pytest.approx(M[row, col], abs=1e-13 * scaleM)
            checked += 1
    assert checked >= 3 ** dim * 2


def test_q1_tables_are_consistent(lib):
    from pdegpu import _lib
    g = _lib.make_grid(3, (4, 4, 4), (1.0, 1.0, 1.0), _lib.ELEM_Q1)
    tK, tM, ld = np.zeros(729), np.zeros(729), np.zeros(27)
    assert lib.pdegpu_heat_tables_host(g, _lib.dptr(tK), _lib.dptr(tM), _lib.dptr(ld)) == 0
    h = 0.25
    k = tK.reshape(27, 27)[13] / h
    # classic trilinear stencil: centre 8/3, faces 0, edges -1/6, corners -1/12
    for off in range(27):
        d = (off % 3 - 1, (off // 3) % 3 - 1, off // 9 - 1)
        nnz = sum(abs(v) for v in d)
        assert k[off] == pytest.approx({0: 8 / 3, 1: 0.0, 2: -1 / 6, 3: -1 / 12}[nnz], abs=1e-13)
    m = tM.reshape(27, 27)[13] / h ** 3 * 216
    for off in range(27):
        d = (off % 3 - 1, (off // 3) % 3 - 1, off // 9 - 1)
        assert m[off] == pytest.approx({0: 64, 1: 16, 2: 4, 3: 1}[sum(abs(v) for v in d)], abs=1e-10)
    assert ld[13] == pytest.approx(h ** 3) and ld[0] == pytest.approx(h ** 3 / 8)
    assert abs(tK.reshape(27, 27).sum(axis=1)).max() < 1e-12          # constants in the null space, every class


def test_tables_host_rejects_bad_grid(lib):
    from pdegpu import _lib
    g = _lib.make_grid(3, (4, 0, 4), (1.0, 1.0, 1.0))
    buf = np.zeros(729)
    assert lib.pdegpu_heat_tables_host(g, _lib.dptr(buf), _lib.dptr(buf.copy()), _lib.dptr(np.zeros(27))) == _lib.EINVAL
    assert lib.pdegpu_grid_num_nodes(g) == -1
    assert lib.pdegpu_grid_num_nodes(_lib.make_grid(3, (512, 512, 512), (1, 1, 1))) == 513 ** 3


@pytest.mark.parametrize("planes,ranks", [(513, 1), (513, 2), (513, 8), (1025, 8), (7, 3), (8, 8)])
def test_slab_partition_is_a_partition(lib, planes, ranks):
    k0, nk = C.c_int64(), C.c_int64()
    nxt = 0
    sizes = []
    for r in range(ranks):
        assert lib.pdegpu_slab_partition(planes, ranks, r, C.byref(k0), C.byref(nk)) == 0
        assert k0.value == nxt and nk.value >= 1
        nxt += nk.value
        sizes.append(nk.value)
    assert nxt == planes and max(sizes) - min(sizes) <= 1
    assert lib.pdegpu_slab_partition(planes, ranks, ranks, C.byref(k0), C.byref(nk)) != 0


def test_num_snapshots(lib):
    from pdegpu import _lib
    h = _lib.Heat()
    h.num_steps, h.snapshot_stride = 200, 1
    assert lib.pdegpu_heat_num_snapshots(h) == 201
    h.snapshot_stride = 50
    assert lib.pdegpu_heat_num_snapshots(h) == 5
    h.snapshot_stride = 64
    assert lib.pdegpu_heat_num_snapshots(h) == 5          # 0, 64, 128, 192 and the last state 200
    h.steady = 1
    assert lib.pdegpu_heat_num_snapshots(h) == 1


@pytest.mark.parametrize("dim,n,Ls", [(1, (7,), (2.0,)), (2, (5, 4), (1.0, 2.0)), (3, (3, 4, 2), (1.0, 0.5, 2.0))])
@pytest.mark.parametrize("kind", ["cosine", "sine"])
def test_p2_expression_load_matches_oracle(dim, n, Ls, kind):
    from pdegpu.project import p2_expression_load, trig_product
    import scipy.sparse.linalg as spla
    A, k = 1.7, 2.3
    b = p2_expression_load(dim, n, Ls, trig_product(kind, A, k, dim))
    if dim == 1:
        c, cells = o.interval_mesh(n[0], 0.0, Ls[0])
    elif dim == 2:
        c, cells = o.rectangle_mesh(Ls[0], Ls[1], n[0], n[1])
    else:
        c, cells = o.box_mesh(*Ls, *n)
    _, M, _ = o.assemble_scalar(c, cells)
    f = np.cos if kind == "cosine" else np.sin
    ref = o._project_p2_expression(c, cells, lambda X: A * np.prod(f(k * X), axis=1), M)
    got = spla.splu(M.tocsc()).solve(b)
    assert np.abs(got - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())


def test_no_gpu_fails_loudly(lib):
    """without a device the product path raises — there is no CPU fallback"""
    import torch
    from pdegpu import _lib, heat
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.PdeGpuError):
        heat._solve_heat_3d_raw(1, 1, 1, 4, 4, 4, 1.0, 0.0, 20.0, 0.01, 1)


@pytest.mark.parametrize("dim,n,Ls", [(2, (3, 2), (1.0, 0.5)), (3, (2, 3, 2), (1.0, 0.6, 0.25))])
def test_elasticity_block_tables_match_oracle_assembly(lib, dim, n, Ls):
    """d x d block class table of the elasticity operator (host twin of the device fold) == assembled matrix"""
    from pdegpu import _lib
    lam, mu = o.lame(210e9, 0.3)
    g = _lib.make_grid(dim, n, Ls)
    tab = np.zeros(27 * 15 * 9)
    assert lib.pdegpu_elasticity_tables_host(g, lam, mu, _lib.dptr(tab)) == 0
    tab = tab.reshape(27, 15, 3, 3)
    off15 = [13, 12, 14, 10, 16, 4, 22, 9, 17, 3, 23, 1, 25, 0, 26]
    c, cells = o.rectangle_mesh(Ls[0], Ls[1], n[0], n[1]) if dim == 2 else o.box_mesh(*Ls, *n)
    K = o.assemble_elasticity(c, cells, lam, mu).toarray()
    scale = np.abs(K).max()
    nn = [n[r] + 1 if r < dim else 1 for r in range(3)]
    seen = np.zeros_like(K, dtype=bool)
    for cls in range(27):
        cc = (cls % 3, (cls // 3) % 3, cls // 9)
        if any(cc[r] != 1 for r in range(dim, 3)) or any(r < dim and n[r] < 2 and cc[r] == 1 for r in range(3)):
            continue
        idx3 = [_class_node(cc[r], n[r]) if r < dim else 0 for r in range(3)]
        row = idx3[0] + nn[0] * (idx3[1] + nn[1] * idx3[2])
        for q, off in enumerate(off15):
            d = (off % 3 - 1, (off // 3) % 3 - 1, off // 9 - 1)
            nb = [idx3[r] + d[r] for r in range(3)]
            if not all(0 <= nb[r] < nn[r] for r in range(3)):
                assert np.all(tab[cls, q] == 0.0)
                continue
            col = nb[0] + nn[0] * (nb[1] + nn[1] * nb[2])
            blk = K[row * dim:(row + 1) * dim, col * dim:(col + 1) * dim]
            assert np.abs(tab[cls, q, :dim, :dim] - blk).max() < 1e-12 * scale
            seen[row * dim:(row + 1) * dim, col * dim:(col + 1) * dim] = True
        # the 15 offsets are ALL the couplings of this row
        assert np.abs(K[row * dim:(row + 1) * dim][~seen[row * dim:(row + 1) * dim]]).max() < 1e-12 * scale


def test_loader_binds_one_hip_runtime(lib):
    """after load_library() the process maps exactly one libamdhip64 (torch's copy when torch is installed), so
    importing torch before or after pdegpu cannot end with two HIP runtimes"""
    maps = open("/proc/self/maps").read()
    paths = {line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line}
    assert len(paths) == 1, paths


@pytest.mark.parametrize("dim,n,Ls,origin,kind", [
    (1, (6,), (1.5,), (0.5, 0, 0), "r"), (1, (5,), (2.0,), (0.0, 0, 0), "r2"),
    (2, (4, 3), (1.0, 2.0), (0.25, 0, 0), "r"), (2, (3, 4), (0.8, np.pi), (0.2, 0, 0), "r2sin"),
    (3, (3, 2, 3), (0.8, np.pi, 2 * np.pi), (0.2, 0, 0), "r2sin"), (3, (2, 2, 2), (1.0, 1.0, 1.0), (0.0, 0, 0), "r2"),
    (3, (2, 4, 3), (1.0, 1.0, 1.0), (0.0, -0.5, -0.5), "rcyl")])
def test_weighted_rows_match_oracle_assembly(lib, dim, n, Ls, origin, kind):
    """per-node coupling weights of the curvilinear forms (host twin of k_weighted_assemble, same source) ==
    the oracle's weighted matrices, every node (boundary truncation included)"""
    from pdegpu import _lib
    g = _lib.make_grid(dim, n, Ls)
    N = int(lib.pdegpu_grid_num_nodes(g))
    rK, rM, ld = np.zeros((N, 15)), np.zeros((N, 15)), np.zeros(N)
    org = np.array(origin, dtype=float)
    w = _lib.Weight()
    w.kind = {"r": _lib.WEIGHT_R, "r2": _lib.WEIGHT_R2, "r2sin": _lib.WEIGHT_R2SIN, "rcyl": _lib.WEIGHT_RCYL}[kind]
    for r in range(3):
        w.origin[r] = org[r]
    assert lib.pdegpu_weighted_rows_host(g, w, _lib.dptr(rK), _lib.dptr(rM), _lib.dptr(ld)) == 0
    if dim == 1:
        c, cells = o.interval_mesh(n[0], origin[0], origin[0] + Ls[0])
    elif dim == 2:
        c, cells = o.shifted_rectangle_mesh(origin[0], origin[0] + Ls[0], 0.0, Ls[1], n[0], n[1])
    else:
        c, cells = o.box_mesh(*Ls, *n)
        c = c + org
    K, M, load = o.assemble_weighted(c, cells, kind)
    K, M = K.toarray(), M.toarray()
    assert np.abs(ld - load).max() < 1e-13 * np.abs(load).max()
    off15 = [13, 12, 14, 10, 16, 4, 22, 9, 17, 3, 23, 1, 25, 0, 26]
    nn = [n[r] + 1 if r < dim else 1 for r in range(3)]
    Kd, Md = np.zeros_like(K), np.zeros_like(M)
    for row in range(N):
        i3 = (row % nn[0], (row // nn[0]) % nn[1], row // (nn[0] * nn[1]))
        for q, off in enumerate(off15):
            d = (off % 3 - 1, (off // 3) % 3 - 1, off // 9 - 1)
            nb = [i3[r] + d[r] for r in range(3)]
            if all(0 <= nb[r] < nn[r] for r in range(3)):
                col = nb[0] + nn[0] * (nb[1] + nn[1] * nb[2])
                Kd[row, col], Md[row, col] = rK[row, q], rM[row, q]
            else:
                assert rK[row, q] == 0.0 and rM[row, q] == 0.0
    assert np.abs(Kd - K).max() < 1e-12 * np.abs(K).max()
    assert np.abs(Md - M).max() < 1e-12 * np.abs(M).max()


def test_weighted_rows_with_composite_core_match_oracle(lib):
    """per-simplex core diffusivity (DG0, all vertices + midpoint inside the core radius) in the shared row math"""
    from pdegpu import _lib
    n, R, rc, ratio = (3, 6, 6), 0.5, 0.3, 40.0
    g = _lib.make_grid(3, n, (1.0, 2 * R, 2 * R))
    N = int(lib.pdegpu_grid_num_nodes(g))
    w = _lib.Weight()
    w.kind, w.core_radius, w.core_ratio = _lib.WEIGHT_RCYL, rc, ratio
    w.origin[1] = w.origin[2] = -R
    rK, rM, ld = np.zeros((N, 15)), np.zeros((N, 15)), np.zeros(N)
    assert lib.pdegpu_weighted_rows_host(g, w, _lib.dptr(rK), _lib.dptr(rM), _lib.dptr(ld)) == 0
    c, cells = o.box_mesh(1.0, 2 * R, 2 * R, *n)
    c = c + np.array([0.0, -R, -R])
    P = c[cells]
    core = (np.sqrt(P[:, :, 1] ** 2 + P[:, :, 2] ** 2) < rc).all(axis=1) & (np.hypot(*P.mean(axis=1)[:, 1:].T) < rc)
    assert core.any() and not core.all()
    K, M, load = o.assemble_weighted(c, cells, "rcyl", np.where(core, ratio, 1.0))
    assert np.allclose(rK[:, 0], K.diagonal(), rtol=1e-12) and np.allclose(rM[:, 0], M.diagonal(), rtol=1e-12)
    assert np.allclose(rK.sum(axis=1), 0.0, atol=1e-12 * np.abs(rK).max())


# ---- output-volume policy (the reference stores every step: fenics_mcp_server.py:694-696, 711-713) --------------------------
def test_snapshot_policy_keeps_a_request_that_fits(monkeypatch):
    from pdegpu import _lib as L
    monkeypatch.setenv("PDEGPU_MAX_OUTPUT_GB", "1")
    assert L._memory_available_bytes() is None or L._memory_available_bytes() > 0          # the real probe (MemAvailable / cgroup) on this host
    assert L.snapshot_policy(200, 1001, 1, False) == (1, {"snapshot_stride": 1})
    assert L.snapshot_policy(200, 10 ** 9, 7, True) == (7, {"snapshot_stride": 7})          # steady: one state


def test_snapshot_policy_auto_strides_config3_defaults(monkeypatch):
    """config 3 at the dispatcher's defaults: 201 states x 513^3 nodes = 217 GB -> strided to fit, recorded for meta"""
    from pdegpu import _lib as L
    monkeypatch.setattr(L, "_memory_available_bytes", lambda: None)        # (a host that can hold the budget: the fixed number decides)
    monkeypatch.setenv("PDEGPU_MAX_OUTPUT_GB", "64")
    monkeypatch.delenv("PDEGPU_OUTPUT_POLICY", raising=False)
    N = 513 ** 3
    stride, meta = L.snapshot_policy(200, N, 1, False)
    assert L.count_snapshots(200, stride) * N * 8 <= 64e9 < L.count_snapshots(200, stride - 1) * N * 8
    assert meta["snapshot_stride"] == stride and meta["snapshot_stride_requested"] == 1 and "217" in meta["snapshot_policy"]
    assert L.count_snapshots(200, 1) == 201 and L.count_snapshots(10, 4) == 4            # 0, 4, 8, 10
    # the DEFAULT budget is a fixed number (the same array shapes on every machine): config 3's defaults -> every 8th state
    monkeypatch.delenv("PDEGPU_MAX_OUTPUT_GB")
    assert L.output_budget_bytes() == 32_000_000_000
    assert L.snapshot_policy(200, N, 1, False)[0] == 8 and L.count_snapshots(200, 8) == 26
    # closed form == enumeration, and a long run is strided without walking its steps
    for steps, st in ((0, 1), (1, 1), (10, 3), (200, 7), (201, 200), (5, 9)):
        assert L.count_snapshots(steps, st) == sum(1 for q in range(steps + 1) if q % st == 0 or q == steps)
    big, _ = L.snapshot_policy(10 ** 6, 10 ** 6, 1, False)
    assert L.count_snapshots(10 ** 6, big) * 8e6 <= 32e9 < L.count_snapshots(10 ** 6, big - 1) * 8e6


def test_output_budget_is_capped_by_what_the_host_can_still_give(monkeypatch):
    """a container with 10 GB to spare must stride where the fixed 32 GB would be allocated and pickled; meta says which bound decided"""
    from pdegpu import _lib as L
    monkeypatch.delenv("PDEGPU_MAX_OUTPUT_GB", raising=False)
    monkeypatch.delenv("PDEGPU_OUTPUT_POLICY", raising=False)
    monkeypatch.setattr(L, "_memory_available_bytes", lambda: 10_000_000_000)
    assert L.output_budget_bytes(explain=True) == (5_000_000_000, "half of the host memory still available")
    N = 257 ** 3
    stride, meta = L.snapshot_policy(200, N, 1, False)
    assert L.count_snapshots(200, stride) * N * 8 <= 5e9 < L.count_snapshots(200, stride - 1) * N * 8 and "host memory" in meta["snapshot_policy"]
    monkeypatch.setattr(L, "_memory_available_bytes", lambda: 10 ** 12)
    assert L.output_budget_bytes(explain=True) == (32_000_000_000, "default")


def test_snapshot_policy_error_mode_and_hopeless_requests(monkeypatch):
    from pdegpu import _lib as L
    monkeypatch.setattr(L, "_memory_available_bytes", lambda: None)
    monkeypatch.setenv("PDEGPU_MAX_OUTPUT_GB", "8")
    monkeypatch.setenv("PDEGPU_OUTPUT_POLICY", "error")
    with pytest.raises(L.PdeGpuError, match="snapshot_stride"):
        L.snapshot_policy(200, 513 ** 3, 1, False)
    monkeypatch.setenv("PDEGPU_OUTPUT_POLICY", "auto")
    monkeypatch.setenv("PDEGPU_MAX_OUTPUT_GB", "1")
    with pytest.raises(L.PdeGpuError, match="first and the last"):
        L.snapshot_policy(200, 513 ** 3, 1, False)


def _geom(lib, nodes, mode, occ={0: 2, 1: 2, 2: 3, 3: 4}, cus=256):
    out = (C.c_int32 * 4)()
    assert lib.pdegpu_marching_geometry(cus, nodes[0], nodes[1], nodes[2], mode, occ[mode], out) == 0
    return dict(ry=out[0], wg_per_chunk=out[1], chunks=out[2], planes=out[3])


def test_marching_geometry_of_the_baseline_grids_is_pinned(lib):
    """The z-chunk policy of the marching kernels is FITTED to sweeps (profiles/r02_exp16_chunk_count_sweep.log); this pins what it
    must give on BASELINE's grids so that a re-fit cannot silently move the headline geometry (host-only entry point, no GPU):
    configs[2] 512^3: one resident round of 512 workgroups in both CG kernels — 2 x 256 (2 rows per thread) in the stencil K2, 4 x 128 (4 rows
    per thread) in K1, which leaves the solution vector to K2 in the default configuration, and in the bare SpMV;
    configs[3] 1024x512x512 on one GPU and on 2 / 4 / 8 slabs across the long axis: the same 512 workgroups, longer or shorter chunks."""
    g = _geom(lib, (513, 513, 513), 3)
    assert (g["ry"], g["wg_per_chunk"], g["chunks"], g["planes"]) == (2, 256, 2, 256), g
    g = _geom(lib, (513, 513, 513), 2)
    assert (g["ry"], g["wg_per_chunk"], g["chunks"], g["planes"]) == (4, 128, 4, 128), g
    g = _geom(lib, (513, 513, 513), 0)
    assert (g["ry"], g["wg_per_chunk"], g["chunks"]) == (4, 128, 4), g
    for planes in (1025, 513, 257, 129):
        g = _geom(lib, (513, 513, planes), 2)
        assert (g["ry"], g["wg_per_chunk"], g["chunks"]) == (4, 128, 4), (planes, g)
        assert g["chunks"] * g["planes"] >= planes - 2 > (g["chunks"] - 1) * g["planes"]
        g = _geom(lib, (513, 513, planes), 3)
        assert (g["ry"], g["wg_per_chunk"], g["chunks"]) == (2, 256, 2), (planes, g)


def test_marching_geometry_away_from_512_uses_many_chunks_or_a_fitting_round(lib):
    """what the sweeps showed: a single chunk of 525 / 576 workgroups (700^3 / 768^3, the old rule) runs at 0.49-0.56 of the HBM peak, many
    dispatcher-scheduled chunks at 0.70; a round that fits 2 workgroups per CU is taken where one exists"""
    for n in (576, 640, 700, 768, 896):
        g = _geom(lib, (n + 1,) * 3, 2)
        W = g["wg_per_chunk"] * g["chunks"]
        assert W >= 4 * 512 and 24 <= g["planes"] <= 160, (n, g)
    for n, fit in ((320, 480), (384, 504), (448, 448), (480, 480)):      # K1 (4 rows per thread): a round of at most 2 workgroups per CU
        g = _geom(lib, (n + 1,) * 3, 2)
        assert g["wg_per_chunk"] * g["chunks"] == fit, (n, g)
    g = _geom(lib, (1025,) * 3, 3)                  # stencil-K2 at 8 tiles per row: 4 rows make the one fitting round
    assert (g["ry"], g["wg_per_chunk"], g["chunks"]) == (4, 512, 1), g
    g = _geom(lib, (897,) * 3, 3)                   # ... at 7 they do not: 2 rows, many chunks
    assert g["ry"] == 2 and g["wg_per_chunk"] * g["chunks"] > 2048, g
    with pytest.raises(Exception):
        assert lib.pdegpu_marching_geometry(0, 10, 10, 10, 2, 2, (C.c_int32 * 4)()) == 0


def test_dirichlet_set_of_the_library_is_the_oracles(lib):
    """the index predicate the kernels use (pg_dirichlet + the facet rule of make_grid_dev) against oracle.box_dirichlet — itself pinned to
    the facet rule by enumeration in test_oracle_known_answers.py — for every boundary-argument combination of the box branch"""
    from pdegpu import _lib
    for nx, ny, nz in ((1, 1, 1), (2, 2, 2), (2, 3, 4), (3, 2, 2), (4, 3, 2), (5, 4, 3)):
        n = (nx + 1) * (ny + 1) * (nz + 1)
        grid = _lib.make_grid(3, (nx, ny, nz), (1.0, 1.0, 1.0))
        combos = [(None, None, None)] + [(a, b, c) for a in (None, 1.0) for b in (None, 2.0) for c in (None, 3.0) if (a, b, c) != (None, None, None)]
        for T_left, T_right, T_side in combos:
            h = _lib.Heat()
            if (T_left, T_right, T_side) == (None, None, None):
                h.bc_mode = _lib.BC_UNIFORM
                h.bc_present[0], h.bc_value[0] = 1, 7.0
                ref_m, ref_v = o.box_dirichlet(nx, ny, nz, T_boundary=7.0)
            else:
                h.bc_mode = _lib.BC_DIRECTIONAL
                for r, v in enumerate((T_left, T_right, T_side)):
                    h.bc_present[r], h.bc_value[r] = (0, 0.0) if v is None else (1, v)
                ref_m, ref_v = o.box_dirichlet(nx, ny, nz, T_left=T_left, T_right=T_right, T_side=T_side)
            mask = np.zeros(n, np.uint8)
            vals = np.zeros(n)
            assert lib.pdegpu_dirichlet_set_host(grid, h, mask.ctypes.data_as(C.POINTER(C.c_uint8)), _lib.dptr(vals)) == 0
            assert (mask.astype(bool) == ref_m).all(), (nx, ny, nz, T_left, T_right, T_side)
            assert np.array_equal(vals[ref_m], ref_v[ref_m])


def test_bench_extras_watchdog_reports_a_hang_as_a_failure(tmp_path):
    """bench.py's extras watchdog: an extra leg that never returns gives exit code 3 AND the main JSON line (from a snapshot, with
    extras.aborted and the cpu_baseline block intact) — a hang must not read as 'status ok'"""
    import json
    import subprocess
    import sys
    script = tmp_path / "drive.py"
    script.write_text(f"""
import sys, time, types
sys.path.insert(0, {str(ROOT)!r})
import bench
class FakeRunner:
    world, rank, local_rank = 1, 0, 0
    def run_case(self, case, steps, warmup, extras_n1=False):
        time.sleep(30)                       # a collective that never completes
args = types.SimpleNamespace(extras_timeout=1.0, cells=8, steps=4, warmup=1)
out = {{"metric": "m", "value": 1.0, "cpu_baseline": {{"value": 2.0}}, "roofline": {{"frac": 0.5}}}}
case = bench.make_case("config3", "weak", 8, 1)
bench.run_extras_guarded(FakeRunner(), args, case, out, 1.0)
print("NOT REACHED")
""")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3, (r.returncode, r.stderr[-400:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and "NOT REACHED" not in r.stdout
    line = json.loads(lines[0])
    assert "aborted" in line["extras"] and line["cpu_baseline"] == {"value": 2.0} and line["roofline"] == {"frac": 0.5}
