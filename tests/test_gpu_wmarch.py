"""The z-marching kernels of the weighted 3-D operators (csrc/wmarch.hip) against the CPU oracle: the operator itself against the oracle's assembled
matrices `assemble_weighted` (<= 1e-13 of the row scale), whole solves of the tools that take this path against the oracle's solves and against the
class-table kernels they replace.  fp64."""
import math

import numpy as np
import pytest

from oracle import fem_p1 as o

pytestmark = pytest.mark.gpu


def _weight(L, kind, origin, core_radius=0.0, core_ratio=1.0):
    w = L.Weight()
    w.kind = kind
    for r in range(3):
        w.origin[r] = float(origin[r])
    w.core_radius, w.core_ratio = float(core_radius), float(core_ratio)
    return w


CASES = {
    # name: (oracle kind, library kind, cells, lengths, origin, core_radius, core_ratio)
    "spherical_two_x_tiles": ("r2sin", "WEIGHT_R2SIN", (140, 9, 7), (0.9, math.pi, 2 * math.pi), (0.1, 0.0, 0.0), 0.0, 1.0),
    "spherical_solid": ("r2sin", "WEIGHT_R2SIN", (12, 11, 10), (1.0, math.pi, 2 * math.pi), (0.0, 0.0, 0.0), 0.0, 1.0),
    "spherical_row_of_129_nodes": ("r2sin", "WEIGHT_R2SIN", (128, 4, 5), (0.9, math.pi, 2 * math.pi), (0.1, 0.0, 0.0), 0.0, 1.0),
    "cylinder_fallback": ("rcyl", "WEIGHT_RCYL", (131, 8, 9), (1.0, 1.0, 1.0), (0.0, -0.5, -0.5), 0.0, 1.0),
    "cylinder_fallback_core": ("rcyl", "WEIGHT_RCYL", (20, 12, 10), (1.0, 1.0, 1.0), (0.0, -0.5, -0.5), 0.3, 40.0),
    "box_core": ("one", "WEIGHT_ONE", (17, 11, 12), (1.0, 0.8, 0.9), (0.0, 0.0, 0.0), 0.45, 35.0),
    "box_core_two_cells": ("one", "WEIGHT_ONE", (2, 2, 2), (1.0, 0.8, 0.9), (0.0, 0.0, 0.0), 0.45, 35.0),
}


@pytest.mark.parametrize("ry", [1, 2])
@pytest.mark.parametrize("case", sorted(CASES))
def test_marching_weighted_operator_matches_the_assembled_matrix(lib, gpu_ctx, case, ry):
    """interior rows of cM M_w + cK K_w applied by k_wmarch == the oracle's assembled matrices (exact weight integrals), x arbitrary on EVERY node
    (the boundary values enter the interior rows: that is the lifting)"""
    from pdegpu import _lib as L
    okind, lkind, n, lengths, origin, core_radius, core_ratio = CASES[case]
    coords, cells = o.box_mesh(*lengths, *n)
    coords = coords + np.array(origin)
    kcell = np.where(o.core_cells(coords, cells, core_radius), core_ratio, 1.0) if core_radius > 0 else None
    K, M, _ = o.assemble_weighted(coords, cells, okind, kcell)
    cM, cK = 0.7, 0.013
    A = (cM * M + cK * K).tocsr()
    N = coords.shape[0]
    rng = np.random.default_rng(5)
    x = rng.uniform(-1.0, 1.0, N)
    interior = ~o.box_dirichlet(*n, T_boundary=0.0)[0]
    ref = np.where(interior, A @ x, 0.0)
    grid = L.make_grid(3, n, lengths, L.ELEM_P1SPLIT)
    w = _weight(L, getattr(L, lkind), origin, core_radius, core_ratio)
    y = np.empty(N)
    L.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast_ry", ry))
    try:
        L.check(gpu_ctx, lib.pdegpu_heat_apply_weighted(gpu_ctx, grid, w, cM, cK, L.dptr(x), L.dptr(y), 1))
    finally:
        L.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast_ry", 2))
    scale = np.abs(A).dot(np.abs(x)).max()
    assert np.abs(y - ref).max() <= 1e-13 * scale
    assert np.abs(ref).max() > 1e-3 * scale                                  # (the comparison is not one of zeros)
    # ... and the class-table kernels on every row, same entry point
    L.check(gpu_ctx, lib.pdegpu_heat_apply_weighted(gpu_ctx, grid, w, cM, cK, L.dptr(x), L.dptr(y), 0))
    assert np.abs(y - A @ x).max() <= 1e-13 * scale


@pytest.mark.parametrize("ry", [1, 2])
@pytest.mark.parametrize("faces", [2, 3, 4])
@pytest.mark.parametrize("case", ["cylinder_fallback", "cylinder_fallback_core", "box_core", "box_core_two_cells"])
def test_marching_weighted_operator_with_natural_faces(lib, gpu_ctx, case, faces, ry):
    """x-independent weights with directional conditions (:582-604, :606-623) — faces = 2: x = 0 and x = Lx constrained (the cylinder tool's T_left /
    T_right), 3: x = 0 alone, 4: x = 0 and the side faces (x = Lx natural; T_side constrains 0 < x < Lx only, so the edges of the face x = Lx stay free):
    the marching kernels also compute the truncated rows of the nodes on the natural faces, their edges and corners == the oracle's assembled matrices"""
    from pdegpu import _lib as L
    okind, lkind, n, lengths, origin, core_radius, core_ratio = CASES[case]
    coords, cells = o.box_mesh(*lengths, *n)
    coords = coords + np.array(origin)
    kcell = np.where(o.core_cells(coords, cells, core_radius), core_ratio, 1.0) if core_radius > 0 else None
    K, M, _ = o.assemble_weighted(coords, cells, okind, kcell)
    cM, cK = 1.0, 0.021
    A = (cM * M + cK * K).tocsr()
    N = coords.shape[0]
    x = np.random.default_rng(7).uniform(-1.0, 1.0, N)
    if faces == 4 and n[0] <= 2:
        pytest.skip("T_side constrains nothing with two cells in x (facet rule): the same case as faces = 3")
    dir_mask = o.box_dirichlet(*n, T_left=1.0, T_right=1.0 if faces == 2 else None, T_side=1.0 if faces == 4 else None)[0]
    free = ~dir_mask
    ref = np.where(free, A @ x, 0.0)
    grid = L.make_grid(3, n, lengths, L.ELEM_P1SPLIT)
    w = _weight(L, getattr(L, lkind), origin, core_radius, core_ratio)
    y = np.empty(N)
    L.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast_ry", ry))
    try:
        L.check(gpu_ctx, lib.pdegpu_heat_apply_weighted(gpu_ctx, grid, w, cM, cK, L.dptr(x), L.dptr(y), faces))
    finally:
        L.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast_ry", 2))
    assert np.abs(y - ref).max() <= 1e-13 * np.abs(A).dot(np.abs(x)).max()


def test_cylinder_tool_with_left_and_right_values_marches(lib, gpu_ctx):
    """the reference's typical cylinder call — T_left / T_right forwarded, T_side matching no facet (:590-594): marching kernels, natural side faces"""
    from pdegpu import heat
    for extra in (dict(), dict(core_radius=0.3, core_diffusivity=40.0, steady=True, source_type="constant", source_value=5.0)):
        kw = dict(Lx=1.0, nx=9, ny=8, nz=7, diffusivity=1.0, T_boundary=20.0, T_initial=60.0, dt=0.01, num_steps=3, cylinder_radius=0.5, T_left=100.0,
                  T_right=20.0, T_side=55.0)
        kw.update(extra)
        ref = o.solve_heat_3d_cylinder_fallback(**kw)
        got = heat._solve_heat_3d_raw(Ly=1.0, Lz=1.0, geometry_type="cylinder", rtol=1e-12, **kw)
        assert got.meta["solver_path"] == "weighted-marching"
        assert np.abs(got.values - ref.values).max() <= 1e-9 * np.abs(ref.values).max()
    # the composite core on the plain box with all three directional values: every boundary node constrained, three different values
    kw = dict(Lx=1.0, Ly=0.8, Lz=0.9, nx=9, ny=8, nz=10, diffusivity=1.0, T_boundary=20.0, T_initial=70.0, dt=0.005, num_steps=3, core_radius=0.45,
              core_diffusivity=35.0, T_left=90.0, T_right=-5.0, T_side=10.0)
    got = heat._solve_heat_3d_raw(rtol=1e-12, **kw)
    assert got.meta["solver_path"] == "weighted-marching"
    assert np.abs(got.values - o.solve_heat_3d(**kw).values).max() <= 1e-9 * 90.0


@pytest.mark.parametrize("case", ["cylinder_left_only", "core_left_only_steady", "core_left_and_side"])
def test_natural_x_face_solves_march_and_match_the_oracle(lib, gpu_ctx, case):
    from pdegpu import heat
    if case == "cylinder_left_only":
        kw = dict(Lx=1.0, nx=9, ny=8, nz=7, diffusivity=1.0, T_boundary=20.0, T_initial=60.0, dt=0.01, num_steps=3, cylinder_radius=0.5, T_left=100.0)
        ref = o.solve_heat_3d_cylinder_fallback(**kw)
        got = heat._solve_heat_3d_raw(Ly=1.0, Lz=1.0, geometry_type="cylinder", rtol=1e-12, **kw)
    else:
        kw = dict(Lx=1.0, Ly=0.8, Lz=0.9, nx=9, ny=8, nz=10, diffusivity=1.0, T_boundary=20.0, T_initial=70.0, dt=0.005, num_steps=3, core_radius=0.45,
                  core_diffusivity=35.0, T_left=50.0)
        if case == "core_left_only_steady":
            kw.update(steady=True, source_type="constant", source_value=4.0)
        else:
            kw.update(T_side=10.0, initial_type="zero")
        ref = o.solve_heat_3d(**kw)
        got = heat._solve_heat_3d_raw(rtol=1e-12, **kw)
    assert got.meta["solver_path"] == "weighted-marching"
    assert np.abs(got.values - ref.values).max() <= 1e-9 * np.abs(ref.values).max()


@pytest.mark.parametrize("case,faces", [("spherical_two_x_tiles", 1), ("cylinder_fallback_core", 1), ("cylinder_fallback_core", 2), ("box_core", 3), ("box_core", 4)])
def test_marching_weighted_operator_is_symmetric_positive_on_the_free_nodes(lib, gpu_ctx, case, faces):
    """what PCG needs of the operator the kernels apply: <A x, y> = <x, A y> and <A x, x> > 0 for vectors that vanish on the constrained nodes —
    a property of the kernels themselves, no oracle involved (symmetry / SPD property tests)"""
    from pdegpu import _lib as L
    okind, lkind, n, lengths, origin, core_radius, core_ratio = CASES[case]
    N = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    if faces == 1:
        dir_mask = o.box_dirichlet(*n, T_boundary=0.0)[0]
    else:
        dir_mask = o.box_dirichlet(*n, T_left=1.0, T_right=1.0 if faces == 2 else None, T_side=1.0 if faces == 4 else None)[0]
    rng = np.random.default_rng(11)
    x, y = (np.where(dir_mask, 0.0, rng.uniform(-1.0, 1.0, N)) for _ in range(2))
    grid = L.make_grid(3, n, lengths, L.ELEM_P1SPLIT)
    w = _weight(L, getattr(L, lkind), origin, core_radius, core_ratio)
    Ax, Ay = np.empty(N), np.empty(N)
    L.check(gpu_ctx, lib.pdegpu_heat_apply_weighted(gpu_ctx, grid, w, 1.0, 0.05, L.dptr(x), L.dptr(Ax), faces))
    L.check(gpu_ctx, lib.pdegpu_heat_apply_weighted(gpu_ctx, grid, w, 1.0, 0.05, L.dptr(y), L.dptr(Ay), faces))
    assert np.all(Ax[dir_mask] == 0.0)
    assert abs(Ax @ y - x @ Ay) <= 1e-12 * (np.abs(Ax) @ np.abs(y))
    assert Ax @ x > 0 and Ay @ y > 0


def test_marching_weighted_apply_refuses_what_it_does_not_cover(lib, gpu_ctx):
    from pdegpu import _lib as L
    grid = L.make_grid(2, (8, 9), (1.0, math.pi), L.ELEM_P1SPLIT)
    w = _weight(L, L.WEIGHT_R2SIN, (0.1, 0.0, 0.0))
    x = np.ones(9 * 10)
    y = np.empty_like(x)
    assert lib.pdegpu_heat_apply_weighted(gpu_ctx, grid, w, 1.0, 1.0, L.dptr(x), L.dptr(y), 1) == L.EUNSUPPORTED
    grid3 = L.make_grid(3, (1, 4, 4), (1.0, 1.0, 1.0), L.ELEM_P1SPLIT)            # one cell in x: no interior column
    w1 = _weight(L, L.WEIGHT_ONE, (0.0, 0.0, 0.0), 0.4, 3.0)
    x = np.ones(2 * 5 * 5)
    y = np.empty_like(x)
    assert lib.pdegpu_heat_apply_weighted(gpu_ctx, grid3, w1, 1.0, 1.0, L.dptr(x), L.dptr(y), 1) == L.EUNSUPPORTED


def _spherical(cv, n, **kw):
    return cv._solve_heat_3d_spherical_raw(0.1, 1.0, n[0], n[1], n[2], 0.9, 20.0, 60.0, 0.004, 3, source_type="constant", source_value=10.0, **kw)


@pytest.mark.parametrize("steady", [False, True])
def test_spherical_tool_marches_and_matches_the_oracle(lib, gpu_ctx, steady):
    from pdegpu import curvilinear as cv
    kw = dict(r_inner=0.1, r_outer=1.0, nr=13, ntheta=9, nphi=11, diffusivity=0.9, T_boundary=20.0, T_initial=60.0, dt=0.004, num_steps=3, steady=steady,
              source_type="constant", source_value=10.0)
    got = cv._solve_heat_3d_spherical_raw(**kw, rtol=1e-12)
    ref = o.solve_heat_3d_spherical(**kw)
    assert got.meta["solver_path"] == "weighted-marching"
    assert np.abs(got.values - ref.values).max() <= 1e-9 * np.abs(ref.values).max()


@pytest.mark.parametrize("ry", [1, 2])
@pytest.mark.parametrize("case", ["cylinder", "cylinder_core", "box_core", "box_core_steady"])
def test_x_independent_weights_march_and_match_the_oracle(lib, gpu_ctx, case, ry):
    from pdegpu import _lib as L, heat
    L.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast_ry", ry))
    try:
        if case.startswith("cylinder"):
            kw = dict(Lx=1.0, nx=9, ny=8, nz=7, diffusivity=1.0, T_boundary=20.0, T_initial=60.0, dt=0.01, num_steps=3, cylinder_radius=0.5)
            if case.endswith("core"):
                kw.update(core_radius=0.3, core_diffusivity=40.0)
            ref = o.solve_heat_3d_cylinder_fallback(**kw)
            got = heat._solve_heat_3d_raw(Ly=1.0, Lz=1.0, geometry_type="cylinder", rtol=1e-12, **kw)
        else:
            kw = dict(Lx=1.0, Ly=0.8, Lz=0.9, nx=9, ny=8, nz=10, diffusivity=1.0, T_boundary=20.0, T_initial=70.0, dt=0.005, num_steps=3, core_radius=0.45,
                      core_diffusivity=35.0)
            if case.endswith("steady"):
                kw.update(steady=True, source_type="constant", source_value=30.0)
            ref = o.solve_heat_3d(**kw)
            got = heat._solve_heat_3d_raw(rtol=1e-12, **kw)
    finally:
        L.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast_ry", 2))
    assert got.meta.get("solver_path", "weighted-marching") == "weighted-marching"
    assert np.abs(got.values - ref.values).max() <= 1e-9 * np.abs(ref.values).max()


@pytest.mark.parametrize("case", ["spherical", "box_core"])
def test_marching_and_class_table_pcg_agree(lib, gpu_ctx, case):
    """the same Jacobi-PCG through both kernel families at a size with several z-chunks and x-tiles: fields to 1e-9, iteration counts within one"""
    from pdegpu import _lib as L, curvilinear as cv, heat
    outs = []
    try:
        for wfast in (1, 0):
            L.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast", wfast))
            if case == "spherical":
                f = _spherical(cv, (150, 37, 41), rtol=1e-11)
            else:
                f = heat._solve_heat_3d_raw(Lx=1.0, Ly=1.0, Lz=1.0, nx=150, ny=38, nz=40, diffusivity=1.0, T_boundary=5.0, T_initial=70.0, dt=0.002, num_steps=2,
                                            core_radius=0.4, core_diffusivity=20.0, rtol=1e-11)
            outs.append(f)
    finally:
        L.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast", 1))
    assert outs[0].meta["solver_path"] == "weighted-marching" and outs[1].meta["solver_path"] != "weighted-marching"
    assert abs(outs[0].meta["cg_iterations"] - outs[1].meta["cg_iterations"]) <= 2 and outs[0].meta["cg_iterations"] > 10
    assert np.abs(outs[0].values - outs[1].values).max() <= 1e-9 * np.abs(outs[1].values).max()


@pytest.mark.parametrize("case", ["spherical", "box_core", "cylinder_lr"])
def test_stencil_k2_with_deferred_x_update_gives_the_flat_k2_solve(lib, gpu_ctx, case):
    """wfast_xdefer: K2 of the weighted marching PCG applies the weighted stencil to the stored p again (z -= alpha D^-1 A p: K1 stores no q) and updates
    x for two iterations at once, every other iteration, with p_(n-1) = (p_n - z_n) / beta_n; the flat K2 on a stored q (wfast_xdefer = 0) must give
    the same solve: fields to 1e-9, iteration counts within two — per-lane weights, wave-uniform weights, and natural faces (the cylinder tool's
    T_left / T_right call)"""
    from pdegpu import _lib as L, curvilinear as cv, heat
    outs = []
    try:
        for xd in (1, 0):
            L.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast_xdefer", xd))
            if case == "spherical":
                f = _spherical(cv, (150, 37, 41), rtol=1e-11)
            elif case == "box_core":
                f = heat._solve_heat_3d_raw(Lx=1.0, Ly=1.0, Lz=1.0, nx=150, ny=38, nz=40, diffusivity=1.0, T_boundary=5.0, T_initial=70.0, dt=0.002, num_steps=2,
                                            core_radius=0.4, core_diffusivity=20.0, rtol=1e-11)
            else:
                f = heat._solve_heat_3d_raw(Lx=1.0, Ly=1.0, Lz=1.0, nx=140, ny=20, nz=22, diffusivity=1.0, T_boundary=0.0, T_initial=50.0, dt=0.004, num_steps=2,
                                            geometry_type="cylinder", cylinder_radius=0.5, T_left=10.0, T_right=30.0, T_side=None, rtol=1e-11)
            assert f.meta.get("solver_path", "weighted-marching") == "weighted-marching"
            outs.append(f)
    finally:
        L.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast_xdefer", -1))
    assert abs(outs[0].meta["cg_iterations"] - outs[1].meta["cg_iterations"]) <= 2 and outs[0].meta["cg_iterations"] > 10
    assert np.abs(outs[0].values - outs[1].values).max() <= 1e-9 * np.abs(outs[1].values).max()
