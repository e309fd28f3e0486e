"""The Q1 element mode (north_star's "27-point stencil": PDEGPU_ELEM_Q1, the 27-point marching kernel k_fast<27, ...>, the truncated 27-point
classes on natural faces) against the Q1 oracle (oracle/fem_q1.py: element-wise Gauss assembly, pinned by tests/test_oracle_q1.py) — the operator
against the assembled matrices at <= 1e-13, whole solves at <= 1e-9.  fp64.  (The reference itself is P1, fenics_mcp_server.py:230, 370, 535: the
reference-exact mode is the P1 split; Q1 differs from it at O(h^2), test_gpu_heat.py::test_q1_mode_differs_by_h2.)"""
import numpy as np
import pytest

from oracle import fem_q1 as q

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("n,Ls", [((19, 23), (1.0, 0.5)), ((9, 7, 6), (1.0, 0.6, 0.25)), ((130, 5, 4), (2.0, 0.3, 0.2)), ((2, 2, 2), (1.0, 1.0, 1.0)),
                                  ((260, 6, 3), (1.0, 0.2, 0.1))])
@pytest.mark.parametrize("cM,cK", [(0.0, 1.0), (1.0, 0.0), (1.0, 0.37)])
def test_q1_operator_on_every_row_matches_the_assembled_matrix(lib, gpu_ctx, n, Ls, cM, cK):
    """class-table kernel, natural boundary on every face (all 27 boundary classes of the truncated 27-point stencil) == CSR SpMV of the oracle"""
    from pdegpu import _lib
    K, M, load = q.assemble_gauss(n, Ls)
    x = np.random.default_rng(1).uniform(-1, 1, load.size)
    y, ld = np.empty_like(x), np.empty_like(x)
    g = _lib.make_grid(len(n), n, Ls, _lib.ELEM_Q1)
    _lib.check(gpu_ctx, lib.pdegpu_heat_apply(gpu_ctx, g, cM, cK, _lib.dptr(x), _lib.dptr(y), _lib.dptr(ld), 0))
    A = (cM * M + cK * K).tocsr()
    assert np.abs(y - A @ x).max() <= 1e-13 * abs(A).dot(np.abs(x)).max()
    assert rel_err(ld, load) < 1e-13


@pytest.mark.parametrize("n,Ls", [((19, 23), (1.0, 0.5)), ((9, 7, 6), (1.0, 0.6, 0.25)), ((130, 5, 4), (2.0, 0.3, 0.2)), ((260, 6, 3), (1.0, 0.2, 0.1)),
                                  ((40, 37, 33), (1.0, 0.9, 1.1))])
@pytest.mark.parametrize("cM,cK", [(0.0, 1.0), (1.0, 0.37)])
def test_q1_marching_kernel_matches_the_assembled_matrix(lib, gpu_ctx, n, Ls, cM, cK):
    """k_fast<27, ...> (3-D; 2-D through its flat view / the plane kernels): the free rows of the all-Dirichlet operator, x arbitrary on every node"""
    from pdegpu import _lib
    K, M, load = q.assemble_gauss(n, Ls)
    x = np.random.default_rng(2).uniform(-1, 1, load.size)
    y = np.empty_like(x)
    g = _lib.make_grid(len(n), n, Ls, _lib.ELEM_Q1)
    _lib.check(gpu_ctx, lib.pdegpu_heat_apply(gpu_ctx, g, cM, cK, _lib.dptr(x), _lib.dptr(y), None, 1))
    nn = [v + 1 for v in n]
    idx = np.unravel_index(np.arange(load.size), nn[::-1])[::-1]                         # idx[r] = index along axis r (x fastest)
    interior = np.ones(load.size, bool)
    for r in range(len(n)):
        interior &= (idx[r] > 0) & (idx[r] < n[r])
    A = (cM * M + cK * K).tocsr()
    ref = np.where(interior, A @ x, 0.0)
    assert np.abs(y - ref).max() <= 1e-13 * abs(A).dot(np.abs(x)).max()


@pytest.mark.parametrize("case", ["3d_transient", "3d_steady", "3d_left_only", "3d_left_right", "3d_sides_only", "3d_wide", "3d_tiles"])
def test_q1_3d_solves_match_the_q1_oracle(lib, gpu_ctx, case):
    """marching kernel alone (uniform boundary value), marching + class-table boundary pass (natural faces), one-round tile kernels (small grids)"""
    from pdegpu import heat, _lib
    n = {"3d_wide": (140, 9, 7), "3d_tiles": (12, 10, 9)}.get(case, (30, 17, 12))
    kw = dict(Lx=1.0, Ly=0.8, Lz=0.3, nx=n[0], ny=n[1], nz=n[2], diffusivity=0.9, T_boundary=1.5, T_initial=20.0, dt=0.003, num_steps=3,
              steady=(case == "3d_steady"), source_type="constant", source_value=2.0)
    if case == "3d_left_only":
        kw.update(T_left=10.0)
    elif case == "3d_left_right":
        kw.update(T_left=10.0, T_right=-3.0)
    elif case == "3d_sides_only":
        kw.update(T_side=4.0)
    got = heat._solve_heat_3d_raw(**kw, rtol=1e-12, element_mode=_lib.ELEM_Q1)
    ref = q.solve_heat_3d(**kw)
    natural = case in ("3d_left_only", "3d_left_right", "3d_sides_only")
    assert got.meta["element_mode"] == "q1" and got.meta["solver_path"] == ("fast+boundary" if natural else "fast")
    assert np.allclose(got.coords, ref.coords, atol=1e-14) and got.values.shape == ref.values.shape
    assert rel_err(got.values, ref.values) < 1e-9


@pytest.mark.parametrize("fast", [True, False])
def test_q1_3d_class_table_solver_matches_the_q1_oracle(lib, gpu_ctx, fast):
    from pdegpu import heat, _lib
    kw = dict(Lx=1.0, Ly=1.0, Lz=1.0, nx=9, ny=8, nz=7, diffusivity=1.0, T_boundary=0.0, T_initial=20.0, dt=0.01, num_steps=3, T_left=50.0, T_side=5.0)
    got = heat._solve_heat_3d_raw(**kw, rtol=1e-12, element_mode=_lib.ELEM_Q1, fast_path=fast)
    assert rel_err(got.values, q.solve_heat_3d(**kw).values) < 1e-9


@pytest.mark.parametrize("case", ["transient", "steady", "wide"])
def test_q1_2d_solves_match_the_q1_oracle(lib, gpu_ctx, case):
    """the 9-point form: persistent cooperative CG / plane kernels"""
    from pdegpu import heat, _lib
    n = (150, 40) if case == "wide" else (23, 19)
    kw = dict(Lx=1.0, Ly=0.7, nx=n[0], ny=n[1], diffusivity=0.8, T_boundary=3.0, T_initial=20.0, dt=0.004, num_steps=3, steady=(case == "steady"),
              source_type="constant", source_value=6.0)
    got = heat._solve_heat_2d_raw(**kw, rtol=1e-12, element_mode=_lib.ELEM_Q1)
    ref = q.solve_heat_2d(**kw)
    assert got.meta["element_mode"] == "q1" and rel_err(got.values, ref.values) < 1e-9
