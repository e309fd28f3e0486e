"""Parity at BASELINE.json's full size (config 3: 512^3 cells, 513^3 nodes), where the oracle cannot run:
size-independent properties of the operator and an independent residual check of a real solve."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N1 = 513


def _apply(lib, ctx, cM, cK, x, fast=1):
    """fast = 1: the fused z-marching kernels (Dirichlet rows give 0); fast = 0: the class-table kernel, full operator on every node"""
    from pdegpu import _lib
    g = _lib.make_grid(3, (512, 512, 512), (1.0, 1.0, 1.0))
    y = np.empty_like(x)
    _lib.check(ctx, lib.pdegpu_heat_apply(ctx, g, cM, cK, _lib.dptr(x), _lib.dptr(y), None, fast))
    return y


def test_operator_properties_at_512_cubed(lib, gpu_ctx):
    """fused z-marching kernels at full size: K annihilates constants, <Ax,y> == <x,Ay>, A(ax+by) == aAx+bAy,
    Dirichlet rows are exactly zero, and the interior row of a delta is the closed-form 7-point stencil"""
    rng = np.random.default_rng(7)
    n = N1 ** 3
    interior = np.zeros((N1, N1, N1), bool)
    interior[1:-1, 1:-1, 1:-1] = True
    interior = interior.ravel()
    x = rng.uniform(-1, 1, n)
    y = rng.uniform(-1, 1, n)
    # the operator under test is P A P: symmetric on vectors that vanish on the Dirichlet nodes
    x[~interior] = 0.0
    y[~interior] = 0.0
    Ax = _apply(lib, gpu_ctx, 0.0, 1.0, x)
    Ay = _apply(lib, gpu_ctx, 0.0, 1.0, y)
    assert np.all(Ax[~interior] == 0.0)
    lhs, rhs = float(Ax @ y), float(x @ Ay)
    assert abs(lhs - rhs) <= 1e-11 * (np.linalg.norm(Ax) * np.linalg.norm(y))
    assert float(x @ Ax) > 0.0                                              # positive definite on the free nodes
    comb = _apply(lib, gpu_ctx, 0.0, 1.0, 0.7 * x - 1.3 * y)
    assert np.abs(comb - (0.7 * Ax - 1.3 * Ay)).max() <= 1e-12 * np.abs(Ax).max()
    ones = _apply(lib, gpu_ctx, 0.0, 1.0, np.ones(n))
    h = 1.0 / 512
    assert np.abs(ones).max() <= 1e-12 * 6 * h                              # K 1 = 0 on every free row
    delta = np.zeros(n)
    c = (256 * N1 + 300) * N1 + 123
    delta[c] = 1.0
    col = _apply(lib, gpu_ctx, 1.0, 0.25, delta)                            # M + 0.25 K: 15-point kernel
    v = h ** 3
    expect = {0: 0.4 * v + 0.25 * 6 * h, 1: v / 20 - 0.25 * h, N1: v / 20 - 0.25 * h, N1 * N1: v / 20 - 0.25 * h,
              1 + N1: v / 30, 1 + N1 * N1: v / 30, N1 + N1 * N1: v / 30, 1 + N1 + N1 * N1: v / 20}
    for off, w in expect.items():
        for sgn in ((1, -1) if off else (1,)):
            assert col[c + sgn * off] == pytest.approx(w, rel=1e-12)
    assert np.count_nonzero(col) == 15


def test_steady_solve_at_512_cubed_satisfies_its_equation(lib, gpu_ctx):
    """config 3(ii): steady kappa K u = f m, f = 1, T_b = 0, rtol 1e-8; the returned field is checked by an
    independent application of the operator — the class-table kernel (fast_path = 0), NOT the marching family that produced the
    solution — (||b - A u|| <= 2e-8 ||b||), by the maximum principle, by the
    symmetry of the solution and against the known maximum of the continuous problem"""
    from pdegpu import heat
    f = heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 512, 512, 512, 1.0, 0.0, 0.0, 0.1, 1, steady=True,
                                source_type="constant", source_value=1.0, rtol=1e-8)
    u = f.values[0]
    assert f.meta["solver_path"] == "fast" and f.values.shape == (1, N1 ** 3)
    Au = _apply(lib, gpu_ctx, 0.0, 1.0, u, fast=0)
    h = 1.0 / 512
    U = u.reshape(N1, N1, N1)
    r = (h ** 3 - Au.reshape(N1, N1, N1))[1:-1, 1:-1, 1:-1]
    bnorm = h ** 3 * 511 ** 1.5
    assert np.linalg.norm(r.ravel()) <= 2e-8 * bnorm
    assert U.min() >= 0.0 and np.all(U[0] == 0.0) and np.all(U[:, :, -1] == 0.0)
    assert U.max() == pytest.approx(0.0562128, rel=2e-4)                    # max of -Lap u = 1 on the unit cube
    assert np.abs(U - U[::-1]).max() <= 1e-6 * U.max() and np.abs(U - U.transpose(2, 1, 0)).max() <= 1e-6 * U.max()
    print(f"\n512^3 steady solve: {f.meta['cg_iterations']} CG iterations")


def test_repeated_solves_do_not_leak_device_memory(lib, gpu_ctx):
    """every handle a solve creates dies with it (INTEGRATION.md ownership rule): free device memory after 40 mixed solves
    is what it was after the first round"""
    import torch
    from pdegpu import heat, elasticity, curvilinear as cv

    def round_():
        heat._solve_heat_1d_raw(2.0, 300, 1.0, 20.0, 0.0, 0.0, 0.01, 5)
        heat._solve_heat_2d_raw(1.0, 1.0, 96, 64, 1.0, 0.0, 20.0, 0.003, 2)
        heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 40, 24, 20, 1.0, 0.0, 20.0, 0.003, 2)
        heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 40, 24, 20, 1.0, 0.0, 20.0, 0.003, 2, T_left=5.0, T_right=1.0)
        heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 20, 12, 10, 1.0, 0.0, 20.0, 0.003, 1, initial_type="sine")
        elasticity._solve_elasticity_3d_static(1.0, 0.3, 0.2, 24, 6, 5, 210e9, 0.3, 0.0, 0.0, -7.0e4, "stress", rtol=1e-8)
        elasticity._solve_elasticity_2d_static(1.0, 0.3, 24, 9, 210e9, 0.3, 0.0, -5.0e4, "strain", True, rtol=1e-8)
        cv._solve_heat_2d_cylindrical_raw(r_inner=0.1, r_outer=1.0, z_length=2.0, nr=20, nz=16, diffusivity=1.0, T_boundary=0.0,
                                          T_initial=20.0, dt=0.01, num_steps=2, steady=False, source_type="none", source_value=0.0,
                                          initial_type="constant", initial_amplitude=1.0)

    round_()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(5):
        round_()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, f"device memory shrank by {(free0 - free1) / 2**20:.1f} MiB over 40 solves"


def test_config4_grid_one_slab_vs_two_slabs_to_convergence(lib, gpu_ctx):
    """BASELINE config 4's grid (1024 x 512 x 512 cells, 2.7e8 nodes, 2.2 GB per vector) on ONE GPU, relabelled as the multi-GPU
    driver does (the slabs cut the 1024-cell axis): two slab sessions — halo planes and scalar sums moved by device copies in the
    order of SlabCG.iterate, K1 split around the exchange — run to rtol 1e-8 and must give the residual history AND the field of
    the single-slab session, beyond what the oracle can reach"""
    import torch
    from pdegpu import _lib
    from pdegpu.dist import GpuSlabEngine, longest_axis_last, slab_range, VEC_R, SC_PQ, SC_RR
    perm, n, Ls = longest_axis_last((1024, 512, 512), (2.0, 1.0, 1.0))
    assert n == (512, 512, 1024) and perm == [1, 2, 0]
    grid = _lib.make_grid(3, n, Ls)
    h = _lib.Heat()
    h.kappa, h.steady, h.source = 1.0, 1, 1.0
    h.bc_mode, h.fast_path = _lib.BC_UNIFORM, 1
    h.bc_present[0] = 1
    check_at = (0, 1, 2, 6, 50, 400)

    def run(ns):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            engs = [GpuSlabEngine(grid, h, *slab_range(n[2] + 1, ns, r), 0, stream) for r in range(ns)]
            assert not engs[0].needs_p_halo() and engs[0].can_split_k1()

            def allreduce(which):
                slot = SC_PQ if which == 0 else SC_RR
                tot = sum(e.scalars()[slot:slot + 1].clone() for e in engs)
                for e in engs:
                    e.scalars()[slot:slot + 1].copy_(tot)

            for e in engs:
                e.begin(0.0)
                e.residual()
            allreduce(1)
            rr0 = engs[0].rr()
            hist, it, rr = {0: rr0}, 0, rr0
            while rr > 1e-16 * rr0 and it < 6000:
                for _ in range(50 if it >= 50 else 1):
                    for e in engs:
                        e.k1_part(0)
                    for r in range(ns - 1):
                        lo, hi = engs[r], engs[r + 1]
                        hi.plane_tensor(VEC_R, -1).copy_(lo.plane_tensor(VEC_R, lo.nk - 1))
                        lo.plane_tensor(VEC_R, lo.nk).copy_(hi.plane_tensor(VEC_R, 0))
                    for e in engs:
                        e.k1_part(1)
                    allreduce(0)
                    for e in engs:
                        e.k2()
                    allreduce(1)
                    it += 1
                rr = engs[0].rr()
                if it in check_at:
                    hist[it] = rr
            for e in engs:
                e.flush()
            x = np.concatenate([e.owned_x() for e in engs], axis=0)
            for e in engs:
                e.close()
        return x, hist, it

    x1, h1, it1 = run(1)
    x2, h2, it2 = run(2)
    assert it1 == it2 and 1000 < it1 < 4000
    assert set(h1) == set(h2) and all(np.isclose(h2[k], h1[k], rtol=1e-9) for k in h1)
    assert np.abs(x2 - x1).max() <= 1e-9 * np.abs(x1).max()
    # the field itself: maximum principle and the mirror symmetries of the box problem (z is the relabelled 1024-cell axis)
    assert x1.min() >= 0.0 and np.abs(x1 - x1[::-1]).max() <= 1e-6 * x1.max() and np.abs(x1 - x1[:, ::-1]).max() <= 1e-6 * x1.max()


def test_config3_transient_step_at_512_cubed(lib, gpu_ctx):
    """config 3's transient leg: ONE backward-Euler step (M + dt kappa K) u^1 = M u^0 at 512^3 through the tool-level solver
    (15-point operator, 72-B variant, rtol 1e-10), checked by an independent application of the operator with the class-table
    kernel: || (M + dt K) u^1 - M u^0 || <= 1e-8 || M u^0 || on the free rows; u^1 keeps the boundary values and the symmetries"""
    from pdegpu import _lib, heat
    dt = 0.0025
    f = heat._solve_heat_3d_raw(1.0, 1.0, 1.0, 512, 512, 512, 1.0, 0.0, 20.0, dt, 1, rtol=1e-10)
    assert f.values.shape == (2, N1 ** 3) and np.allclose(f.times, [0.0, dt]) and f.meta["snapshot_stride"] == 1
    u0, u1 = f.values[0], f.values[1]
    g = _lib.make_grid(3, (512, 512, 512), (1.0, 1.0, 1.0))

    def apply_general(cM, cK, x):
        y = np.empty_like(x)
        _lib.check(gpu_ctx, lib.pdegpu_heat_apply(gpu_ctx, g, cM, cK, _lib.dptr(x), _lib.dptr(y), None, 0))
        return y.reshape(N1, N1, N1)[1:-1, 1:-1, 1:-1]

    lhs = apply_general(1.0, dt, u1)
    rhs = apply_general(1.0, 0.0, u0)
    res = np.linalg.norm((lhs - rhs).ravel()) / np.linalg.norm(rhs.ravel())
    assert res <= 1e-8, res
    U1 = u1.reshape(N1, N1, N1)
    U0 = u0.reshape(N1, N1, N1)
    assert np.all(U0[1:-1, 1:-1, 1:-1] == 20.0) and np.all(U1[0] == 0.0) and np.all(U1[:, :, -1] == 0.0)
    assert np.isfinite(U1).all() and U1[1:-1, 1:-1, 1:-1].max() <= 20.0 + 1e-6
    # symmetries of the DISCRETE problem: the Kuhn split (and with it the 15-point mass stencil) is invariant under the point
    # reflection through the centre and under any relabelling of the axes — not under the mirror image of a single axis
    assert np.abs(U1 - U1[::-1, ::-1, ::-1]).max() <= 1e-7 * 20.0
    assert np.abs(U1 - U1.transpose(2, 1, 0)).max() <= 1e-7 * 20.0 and np.abs(U1 - U1.transpose(0, 2, 1)).max() <= 1e-7 * 20.0
    assert f.meta["cg_iterations"] > 10 and f.meta["solver_path"] == "fast"


def test_config5_at_full_size(lib, gpu_ctx):
    """BASELINE config 5 at its full size (256 x 64 x 64 cells, E = 210 GPa, nu = 0.3, gravity: 3.26 M unknowns, kappa ~ 1e6):
    the tool-level solve, its displacement checked by an independent application of the block operator
    (|| b - A u || <= 1e-7 || b || on the unclamped rows), the iteration count of the Jacobi-PCG against round 1's 4009 (+-5 %),
    the clamp, and the mirror symmetry of the cantilever in y"""
    from pdegpu import _lib, elasticity
    n, Ls = (256, 64, 64), (1.0, 0.25, 0.25)
    f = elasticity._solve_elasticity_3d_static(*Ls, *n, 210e9, 0.3, 0.0, 0.0, -76518.0, "stress", want_displacement=True, rtol=1e-8)
    it = f.meta["cg_iterations"]
    assert abs(it - 4009) <= 0.05 * 4009, it
    NX, NY, NZ = (c + 1 for c in n)
    u = np.ascontiguousarray(f.displacement)
    assert u.shape == (NX * NY * NZ, 3)
    grid = _lib.make_grid(3, n, Ls)
    el = _lib.Elasticity()
    el.E, el.nu = 210e9, 0.3
    y = np.empty_like(u)
    _lib.check(gpu_ctx, lib.pdegpu_elasticity_apply(gpu_ctx, grid, el, _lib.dptr(u), _lib.dptr(y), 0))
    # right-hand side: body_z * int phi_i; the load vector comes from the heat operator's class-table kernel (on the Kuhn split the
    # boundary entries are NOT h^3 / 2^faces: a node's share depends on how many of the six tetrahedra of each cell touch it)
    m = np.empty(NX * NY * NZ)
    dummy = np.empty_like(m)
    _lib.check(gpu_ctx, lib.pdegpu_heat_apply(gpu_ctx, grid, 1.0, 0.0, _lib.dptr(np.zeros_like(m)), _lib.dptr(dummy), _lib.dptr(m), 0))
    hx, hy, hz = (Ls[r] / n[r] for r in range(3))
    assert m.sum() == pytest.approx(Ls[0] * Ls[1] * Ls[2], rel=1e-12) and m.reshape(NZ, NY, NX)[5, 5, 5] == pytest.approx(hx * hy * hz, rel=1e-12)
    b = np.zeros((NZ, NY, NX, 3))
    b[..., 2] = -76518.0 * m.reshape(NZ, NY, NX)
    Y, U = y.reshape(NZ, NY, NX, 3), u.reshape(NZ, NY, NX, 3)
    free = np.s_[:, :, 1:]                                                  # clamped face x = 0 excluded
    res = np.linalg.norm((b - Y)[free].ravel()) / np.linalg.norm(b[free].ravel())
    assert res <= 1e-7, res
    assert np.all(U[:, :, 0] == 0.0)
    # mirror image in y (u_x, u_z even, u_y odd: load along z, clamp at x = 0).  The Kuhn split is not mirror symmetric, so the
    # discrete field has this symmetry only up to the discretisation error, O(h^2) of the 64-cell cross-section
    scale = np.abs(U).max()
    assert np.abs(U[..., 0] - U[:, ::-1, :, 0]).max() <= 5e-3 * scale
    assert np.abs(U[..., 2] - U[:, ::-1, :, 2]).max() <= 5e-3 * scale
    assert np.abs(U[..., 1] + U[:, ::-1, :, 1]).max() <= 5e-3 * scale
    assert U[..., 2].min() < 0.0 and np.isfinite(f.values).all() and f.values.min() >= 0.0


def test_random_tiny_grids_all_paths_match_the_oracle(lib, gpu_ctx):
    """edge shapes: 1-5 cells per axis (grids with no interior node, a single interior node, one-row tiles), random boundary
    specifications, steady and transient — whatever path the library picks (marching, tile, hybrid, class-table, Thomas) must
    reproduce the oracle"""
    from oracle import fem_p1 as o
    from pdegpu import heat, elasticity
    rng = np.random.default_rng(20251004)
    rel = lambda a, b: np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)
    paths = set()
    for case in range(36):
        dim = int(rng.integers(1, 4))
        n = [int(rng.integers(1, 6)) for _ in range(dim)]
        Ls = [float(rng.uniform(0.3, 2.0)) for _ in range(dim)]
        steady = bool(rng.integers(0, 2))
        common = dict(diffusivity=float(rng.uniform(0.2, 2.0)), T_initial=float(rng.uniform(-5, 20)), dt=0.01, num_steps=2, steady=steady,
                      source_type="constant", source_value=float(rng.uniform(-3, 3)))
        if dim == 1:
            kw = dict(length=Ls[0], nx=n[0], T_left=float(rng.uniform(-5, 5)), T_right=float(rng.uniform(-5, 5)), **common)
            got, ref = heat._solve_heat_1d_raw(**kw, rtol=1e-13), o.solve_heat_1d(**kw)
        elif dim == 2:
            kw = dict(Lx=Ls[0], Ly=Ls[1], nx=n[0], ny=n[1], T_boundary=float(rng.uniform(-5, 5)), **common)
            got, ref = heat._solve_heat_2d_raw(**kw, rtol=1e-13), o.solve_heat_2d(**kw)
        else:
            kw = dict(Lx=Ls[0], Ly=Ls[1], Lz=Ls[2], nx=n[0], ny=n[1], nz=n[2], T_boundary=float(rng.uniform(-5, 5)), **common)
            mode = int(rng.integers(0, 3))
            if mode == 1:
                kw.update(T_left=float(rng.uniform(-5, 5)), T_right=None, T_side=None)
            elif mode == 2:
                kw.update(T_left=None, T_right=float(rng.uniform(-5, 5)), T_side=float(rng.uniform(-5, 5)))
            got, ref = heat._solve_heat_3d_raw(**kw, rtol=1e-13), o.solve_heat_3d(**kw)
        paths.add(got.meta["solver_path"])
        assert rel(got.values, ref.values) < 1e-9, (case, dim, n, steady, got.meta["solver_path"])
    assert {"tridiagonal-direct", "fast", "general"} <= paths
    for case in range(8):
        n = [int(rng.integers(1, 6)) for _ in range(3)]
        body = [float(rng.uniform(-1e4, 1e4)) for _ in range(3)]
        kw = dict(Lx=1.0, Ly=0.4, Lz=0.3, nx=n[0], ny=n[1], nz=n[2], E=210e9, nu=float(rng.uniform(0.0, 0.45)), body_fx=body[0], body_fy=body[1],
                  body_fz=body[2], quantity="stress" if case % 2 else "strain")
        got, ref = elasticity._solve_elasticity_3d_static(**kw, rtol=1e-13), o.solve_elasticity_3d(**kw)
        assert rel(got.values, ref.values) < 1e-8, (case, n)


def test_random_mid_grids_fast_paths_match_the_oracle(lib, gpu_ctx):
    """3-18 cells per axis, anisotropic spacing, random boundary specification: the 2-D tile kernel, the marching kernels and
    the hybrid (natural faces) path against the oracle"""
    from oracle import fem_p1 as o
    from pdegpu import heat
    rng = np.random.default_rng(7771)
    rel = lambda a, b: np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)
    paths = []
    for case in range(24):
        dim = 2 + case % 2
        n = [int(rng.integers(3, 19)) for _ in range(dim)]
        Ls = [float(rng.uniform(0.3, 2.0)) for _ in range(dim)]
        common = dict(diffusivity=float(rng.uniform(0.2, 2.0)), T_initial=float(rng.uniform(-5, 20)), dt=float(rng.uniform(0.002, 0.02)), num_steps=2,
                      steady=bool(rng.integers(0, 2)), source_type="constant", source_value=float(rng.uniform(-3, 3)))
        if dim == 2:
            kw = dict(Lx=Ls[0], Ly=Ls[1], nx=n[0], ny=n[1], T_boundary=float(rng.uniform(-5, 5)), **common)
            got, ref = heat._solve_heat_2d_raw(**kw, rtol=1e-13), o.solve_heat_2d(**kw)
        else:
            kw = dict(Lx=Ls[0], Ly=Ls[1], Lz=Ls[2], nx=n[0], ny=n[1], nz=n[2], T_boundary=float(rng.uniform(-5, 5)), **common)
            if case % 4 == 1:
                kw.update(T_left=float(rng.uniform(-5, 5)), T_right=float(rng.uniform(-5, 5)) if rng.integers(0, 2) else None,
                          T_side=float(rng.uniform(-5, 5)) if rng.integers(0, 2) else None)
            got, ref = heat._solve_heat_3d_raw(**kw, rtol=1e-13), o.solve_heat_3d(**kw)
        paths.append(got.meta["solver_path"])
        assert rel(got.values, ref.values) < 1e-9, (case, dim, n, got.meta["solver_path"])
    assert paths.count("fast") >= 12 and "fast+boundary" in paths and "general" not in paths


def test_wide_planes_take_four_rows_per_thread_in_k2_and_agree(lib, gpu_ctx):
    """planes of 1025 x 1025 nodes (8 tiles per row): the stencil-K2 with its automatic rows per thread (4 where they make the one fitting round of
    workgroups on HBM-resident grids, stencil.hip fast_rows) and with 2 rows forced must give the same solve — same iteration count, fields to
    round-off — and the field must satisfy its equation (independent residual through the bare operator)"""
    from pdegpu import heat, _lib
    kw = dict(Lx=1.0, Ly=1.0, Lz=0.01, nx=1024, ny=1024, nz=6, diffusivity=1.0, T_boundary=0.0, T_initial=0.0, dt=0.01, num_steps=1,
              source_type="constant", source_value=3.0, steady=True)
    a = heat._solve_heat_3d_raw(**kw, rtol=1e-10)
    _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"fast_cfg_k2", 2 + 16 * 2 + 256))
    try:
        b = heat._solve_heat_3d_raw(**kw, rtol=1e-10)
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"fast_cfg_k2", -1))
    assert a.meta["solver_path"] == "fast" and a.meta["cg_iterations"] == b.meta["cg_iterations"] > 0
    assert np.abs(a.values - b.values).max() <= 1e-9 * np.abs(a.values).max()
    # K u = f m on the free nodes: residual through the unmasked operator, load from the same call
    g = _lib.make_grid(3, (1024, 1024, 6), (1.0, 1.0, 0.01))
    u = np.ascontiguousarray(a.values[0])
    y = np.empty_like(u)
    load = np.empty_like(u)
    _lib.check(gpu_ctx, lib.pdegpu_heat_apply(gpu_ctx, g, 0.0, 1.0, _lib.dptr(u), _lib.dptr(y), _lib.dptr(load), 0))
    free = np.zeros((7, 1025, 1025), bool)
    free[1:-1, 1:-1, 1:-1] = True
    free = free.ravel()
    r = 3.0 * load[free] - y[free]
    assert np.linalg.norm(r) <= 1e-8 * np.linalg.norm(3.0 * load[free])


@pytest.mark.gpu
@pytest.mark.parametrize("cells", [(300, 260, 200), (700, 90, 150), (384, 200, 130)])
def test_marching_kernels_on_hbm_resident_odd_grids_match_the_class_table_kernels(lib, gpu_ctx, cells):
    """HBM-resident grids away from 512^3 take the fitted z-chunk policy (many dispatcher-scheduled chunks or a fitting round) and the row-major
    workgroup order with ragged last workgroups (3 and 6 tiles per row here): the CG iterates of the marching kernels must be those of the
    class-table kernels, which share nothing with them but the weights"""
    from pdegpu import _lib
    grid = _lib.make_grid(3, cells, (1.0, 0.9, 0.7))
    nodes = (cells[0] + 1) * (cells[1] + 1) * (cells[2] + 1)
    assert nodes * 8 > (64 << 20)
    out = []
    for fast in (1, 0):
        h = _lib.Heat()
        h.kappa, h.dt, h.steady, h.source = 1.0, 1.0, 1, 1.0
        h.bc_mode, h.fast_path = _lib.BC_UNIFORM, fast
        h.bc_present[0] = 1
        s = C.c_void_p()
        _lib.check(gpu_ctx, lib.pdegpu_session_create(gpu_ctx, grid, h, 0, cells[2] + 1, C.byref(s)))
        try:
            _lib.check(gpu_ctx, lib.pdegpu_session_begin(s, 0.0))
            _lib.check(gpu_ctx, lib.pdegpu_session_residual(s))
            _lib.check(gpu_ctx, lib.pdegpu_session_reduce_local(s, 1, 1))
            st = _lib.Stats()
            _lib.check(gpu_ctx, lib.pdegpu_session_iterate(s, 6, 1, st))
            rr = C.c_double()
            _lib.check(gpu_ctx, lib.pdegpu_session_rr(s, C.byref(rr)))
            _lib.check(gpu_ctx, lib.pdegpu_session_flush(s))
            x = np.empty(nodes)
            _lib.check(gpu_ctx, lib.pdegpu_session_get_x(s, _lib.dptr(x)))
            assert st.path == fast
            out.append((rr.value, x))
        finally:
            lib.pdegpu_session_destroy(s)
    assert out[0][0] == pytest.approx(out[1][0], rel=1e-10)
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-11 * np.abs(out[1][1]).max()


@pytest.mark.parametrize("cells", [(3000, 3000), (4096, 2200)])
def test_flat_view_of_hbm_resident_2d_grids_matches_the_class_table_kernel(lib, gpu_ctx, cells):
    """2-D grids beyond 64 MiB per vector take the flat view of the marching kernels under the fitted chunk policy (one 'row tile' per plane): the
    bare operator on the free nodes must be the class-table kernel's"""
    from pdegpu import _lib
    g = _lib.make_grid(2, cells, (1.0, 0.7))
    N = (cells[0] + 1) * (cells[1] + 1)
    x = np.random.default_rng(1).standard_normal(N)
    out = []
    for fast in (1, 0):
        y, load = np.empty(N), np.empty(N)
        _lib.check(gpu_ctx, lib.pdegpu_heat_apply(gpu_ctx, g, 0.3, 1.0, _lib.dptr(x), _lib.dptr(y), _lib.dptr(load), fast))
        out.append(y.reshape(cells[1] + 1, cells[0] + 1)[1:-1, 1:-1])
    assert np.abs(out[0] - out[1]).max() <= 1e-13 * np.abs(out[1]).max()


@pytest.mark.parametrize("bcs", [(1, 1, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)])
def test_natural_face_sessions_on_an_hbm_resident_grid_match_the_class_table_kernels(lib, gpu_ctx, bcs):
    """257 x 201 x 171 nodes (70 MB per vector; 257 = 2 * 128 + 1 nodes per row, so the column x = NX-1 is outside the tiles): the hybrid path —
    marching kernel with the x-face rows, class-table pass for the rest, the Dirichlet column left out of the CG passes, placement tuner on — against
    the class-table kernels everywhere, for T_left + T_right, T_left only, T_right only and T_side only"""
    from pdegpu import _lib
    cells = (256, 200, 170)
    grid = _lib.make_grid(3, cells, (1.0, 0.8, 0.7))
    nodes = (cells[0] + 1) * (cells[1] + 1) * (cells[2] + 1)
    out = []
    for fast in (1, 0):
        h = _lib.Heat()
        h.kappa, h.dt, h.steady, h.source = 1.0, 1.0, 1, 1.0
        h.bc_mode, h.fast_path = _lib.BC_DIRECTIONAL, fast
        for f in range(3):
            h.bc_present[f], h.bc_value[f] = bcs[f], 2.0 + f
        s = C.c_void_p()
        _lib.check(gpu_ctx, lib.pdegpu_session_create(gpu_ctx, grid, h, 0, cells[2] + 1, C.byref(s)))
        try:
            _lib.check(gpu_ctx, lib.pdegpu_session_begin(s, 0.0))
            _lib.check(gpu_ctx, lib.pdegpu_session_residual(s))
            _lib.check(gpu_ctx, lib.pdegpu_session_reduce_local(s, 1, 1))
            st = _lib.Stats()
            _lib.check(gpu_ctx, lib.pdegpu_session_iterate(s, 8, 1, st))
            rr = C.c_double()
            _lib.check(gpu_ctx, lib.pdegpu_session_rr(s, C.byref(rr)))
            _lib.check(gpu_ctx, lib.pdegpu_session_flush(s))
            x = np.empty(nodes)
            _lib.check(gpu_ctx, lib.pdegpu_session_get_x(s, _lib.dptr(x)))
            assert st.path == (3 if fast else 0)
            out.append((rr.value, x))
        finally:
            lib.pdegpu_session_destroy(s)
    assert out[0][0] == pytest.approx(out[1][0], rel=1e-10)
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-11 * np.abs(out[1][1]).max()
