"""Known answers for the Q1 (multilinear) oracle, oracle/fem_q1.py — the checker of the build's Q1 element mode (north_star's "27-point stencil").
The reference has no Q1 path (it is P1, fenics_mcp_server.py:230, 370, 535), so nothing of the reference pins this; the module is pinned by two
independent assemblies, the closed-form stencils of SURVEY.md App. A.3 and O(h^2) convergence to the P1 oracle."""
import numpy as np
import pytest

from oracle import fem_p1 as o, fem_q1 as q


@pytest.mark.parametrize("n,L", [((5,), (2.0,)), ((4, 3), (1.0, 0.5)), ((3, 4, 2), (1.0, 0.6, 0.25)), ((1, 1, 1), (0.3, 0.2, 0.1)), ((2, 1, 3), (1.0, 1.0, 1.0))])
def test_gauss_assembly_equals_the_tensor_product_of_1d_matrices(n, L):
    Kg, Mg, lg = q.assemble_gauss(n, L)
    Kk, Mk, lk = q.assemble_kron(n, L)
    assert abs(Kg - Kk).max() <= 1e-13 * abs(Kk).max() and abs(Mg - Mk).max() <= 1e-14 * abs(Mk).max()
    assert np.abs(lg - lk).max() <= 1e-15 * lk.max()
    assert abs(Kg - Kg.T).max() <= 1e-14 * abs(Kg).max() and abs(Mg - Mg.T).max() == 0.0
    assert np.abs(Kg @ np.ones(lg.size)).max() <= 1e-12 * abs(Kg).max()               # constants in the null space
    assert np.allclose(Mg @ np.ones(lg.size), lg, rtol=1e-13) and lg.sum() == pytest.approx(np.prod(L), rel=1e-13)


def test_closed_form_interior_stencils():
    """3-D stiffness (x 1/h) centre 8/3, faces 0, edges -1/6, corners -1/12; mass (x 216/h^3) 64, 16, 4, 1; 2-D stiffness
    8/3, -1/3, -1/3; 2-D mass (x 36/h^2) 16, 4, 1"""
    h = 0.25
    K, M, load = q.assemble_gauss((4, 4, 4), (1.0, 1.0, 1.0))
    c = 2 + 5 * (2 + 5 * 2)
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                nnz = abs(dx) + abs(dy) + abs(dz)
                col = c + dx + 5 * (dy + 5 * dz)
                assert K[c, col] / h == pytest.approx({0: 8 / 3, 1: 0.0, 2: -1 / 6, 3: -1 / 12}[nnz], abs=1e-13)
                assert M[c, col] * 216 / h ** 3 == pytest.approx({0: 64, 1: 16, 2: 4, 3: 1}[nnz], rel=1e-13)
    assert load[c] == pytest.approx(h ** 3) and load[0] == pytest.approx(h ** 3 / 8)
    K2, M2, _ = q.assemble_gauss((4, 4), (1.0, 1.0))
    c2 = 2 + 5 * 2
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            nnz = abs(dx) + abs(dy)
            col = c2 + dx + 5 * dy
            assert K2[c2, col] == pytest.approx({0: 8 / 3, 1: -1 / 3, 2: -1 / 3}[nnz], abs=1e-13)
            assert M2[c2, col] * 36 / h ** 2 == pytest.approx({0: 16, 1: 4, 2: 1}[nnz], rel=1e-13)


def test_1d_q1_is_p1():
    K, M, load = q.assemble_gauss((7,), (2.0,))
    c, cells = o.interval_mesh(7, 0.0, 2.0)
    Kp, Mp, lp = o.assemble_scalar(c, cells)
    assert abs(K - Kp).max() < 1e-13 and abs(M - Mp).max() < 1e-15 and np.abs(load - lp).max() < 1e-15


def test_coordinates_are_those_of_the_p1_meshes():
    assert np.array_equal(q.grid_coords((3, 4, 2), (1.0, 0.6, 0.25)), o.box_mesh(1.0, 0.6, 0.25, 3, 4, 2)[0])
    assert np.array_equal(q.grid_coords((3, 4), (1.0, 0.6)), o.rectangle_mesh(1.0, 0.6, 3, 4)[0])


def test_q1_and_p1_solutions_approach_each_other_at_second_order():
    errs = []
    for n in (4, 8, 16):
        kw = dict(Lx=1.0, Ly=1.0, Lz=1.0, nx=n, ny=n, nz=n, diffusivity=1.0, T_boundary=0.0, T_initial=20.0, dt=0.01, num_steps=5)
        a, b = q.solve_heat_3d(**kw), o.solve_heat_3d(**kw)
        assert np.array_equal(a.coords, b.coords) and a.values.shape == b.values.shape
        errs.append(np.abs(a.values[-1] - b.values[-1]).max() / np.abs(b.values[-1]).max())
    assert errs[0] / errs[1] > 2.5 and errs[1] / errs[2] > 3.0 and errs[2] < 0.02
    errs2 = []
    for n in (8, 16, 32):
        kw = dict(Lx=1.0, Ly=0.7, nx=n, ny=n, diffusivity=0.8, T_boundary=3.0, T_initial=20.0, dt=0.004, num_steps=4)
        a, b = q.solve_heat_2d(**kw), o.solve_heat_2d(**kw)
        errs2.append(np.abs(a.values[-1] - b.values[-1]).max() / np.abs(b.values[-1]).max())
    assert errs2[0] / errs2[1] > 2.5 and errs2[1] / errs2[2] > 3.0


def test_steady_states_with_nodally_exact_answers():
    # f = 0, T_left / T_right only: the linear profile lies in the Q1 space and solves the discrete problem exactly
    f = q.solve_heat_3d(2.0, 1.0, 0.5, 6, 3, 2, 1.3, 0.0, 0.0, 0.01, 1, steady=True, T_left=20.0, T_right=5.0)
    assert np.abs(f.values[0] - (20.0 + (5.0 - 20.0) * f.coords[:, 0] / 2.0)).max() < 1e-11
    # all-Dirichlet, no source: u = T_boundary
    g = q.solve_heat_2d(1.0, 1.0, 5, 4, 1.0, 7.0, 20.0, 0.01, 1, steady=True)
    assert np.abs(g.values[0] - 7.0).max() < 1e-12


@pytest.mark.parametrize("dim,n,Ls", [(2, (3, 2), (1.0, 0.5)), (3, (2, 3, 2), (1.0, 0.6, 0.25)), (3, (3, 3, 3), (1.0, 1.0, 1.0))])
def test_the_librarys_q1_class_tables_are_rows_of_the_assembled_matrices(lib, dim, n, Ls):
    """every (boundary class, offset) weight of the build's Q1 class tables (host twin of the device integration, natural faces included: the
    truncated 27-point classes) == the entry of the oracle's assembled Q1 matrices"""
    from pdegpu import _lib
    g = _lib.make_grid(dim, n, Ls, _lib.ELEM_Q1)
    tK, tM, ld = np.zeros(729), np.zeros(729), np.zeros(27)
    assert lib.pdegpu_heat_tables_host(g, _lib.dptr(tK), _lib.dptr(tM), _lib.dptr(ld)) == 0
    tK, tM = tK.reshape(27, 27), tM.reshape(27, 27)
    K, M, load = q.assemble_gauss(n, Ls)
    K, M = K.toarray(), M.toarray()
    nn = [n[r] + 1 if r < dim else 1 for r in range(3)]
    scaleK, scaleM = np.abs(K).max(), np.abs(M).max()
    checked = 0
    for cls in range(27):
        cc = (cls % 3, (cls // 3) % 3, cls // 9)
        if any(cc[r] != 1 for r in range(dim, 3)) or any(r < dim and n[r] < 2 and cc[r] == 1 for r in range(3)):
            continue
        idx3 = [{0: 0, 1: 1, 2: n[r]}[cc[r]] if r < dim else 0 for r in range(3)]
        row = idx3[0] + nn[0] * (idx3[1] + nn[1] * idx3[2])
        assert ld[cls] == pytest.approx(load[row], rel=1e-13)
        for off in range(27):
            d = (off % 3 - 1, (off // 3) % 3 - 1, off // 9 - 1)
            nb = [idx3[r] + d[r] for r in range(3)]
            if not all(0 <= nb[r] < nn[r] for r in range(3)):
                assert tK[cls, off] == 0.0 and tM[cls, off] == 0.0
                continue
            col = nb[0] + nn[0] * (nb[1] + nn[1] * nb[2])
            assert tK[cls, off] == pytest.approx(K[row, col], abs=1e-13 * scaleK)
            assert tM[cls, off] == pytest.approx(M[row, col], abs=1e-13 * scaleM)
            checked += 1
    assert checked >= 3 ** dim * 4
