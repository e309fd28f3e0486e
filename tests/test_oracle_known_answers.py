"""Pins the CPU oracle (oracle/fem_p1.py).  The reference ships no tests, fixtures or stored outputs for
this path and FEniCS 2019.1.0 cannot be imported here, so parity is *unpinned by the reference*; these
are the known answers that can be established offline: nodally exact 1-D solutions,
closed-form P1 stencils and macro-element matrices, trivial solutions, patch tests, rigid-body null
spaces, SPD-ness and O(h^2) convergence."""
import math

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle import fem_p1 as o


# ---- (i)/(ii) 1-D steady problems: P1 is nodally exact ------------------------------------------------
def test_1d_steady_linear_is_nodally_exact():
    f = o.solve_heat_1d(2.0, 1000, 1.0, 20.0, 0.0, 0.0, 0.01, 10, steady=True)
    x = f.coords[:, 0]
    assert f.values.shape == (1, 1001) and f.times.tolist() == [0.0]
    assert np.abs(f.values[0] - (20.0 - 10.0 * x)).max() < 1e-9


def test_1d_steady_constant_source_is_nodally_exact():
    Lx, kappa, src = 2.0, 0.7, 3.0
    f = o.solve_heat_1d(Lx, 64, kappa, 5.0, -1.0, 0.0, 0.01, 1, steady=True, source_type="constant", source_value=src)
    x = f.coords[:, 0]
    exact = 5.0 + (-1.0 - 5.0) * x / Lx + src * x * (Lx - x) / (2 * kappa)
    assert np.abs(f.values[0] - exact).max() < 1e-10


# ---- (iii) BASELINE config 1 (dispatcher kwargs): regression values --------------
def test_config1_transient_values():
    f = o.solve_heat_1d(2.0, 1000, 1.0, 20.0, 0.0, 0.0, 0.01, 200)
    assert f.values.shape == (201, 1001)
    assert np.allclose(f.times, np.arange(201) * 0.01)
    x = f.coords[:, 0]
    at = lambda xx, step: f.values[step][np.argmin(np.abs(x - xx))]      # noqa: E731
    assert at(0.5, 200) == pytest.approx(14.931253742, abs=2e-9)
    assert at(1.0, 200) == pytest.approx(9.902778170, abs=2e-9)
    assert at(1.5, 200) == pytest.approx(4.931253827, abs=2e-9)
    assert at(1.0, 50) == pytest.approx(6.236415842, abs=2e-9)
    # u^0 = IC with the boundary nodes overwritten (fenics_mcp_server.py:296-297)
    assert f.values[0][0] == 20.0 and f.values[0][-1] == 0.0 and np.all(f.values[0][1:-1] == 0.0)


# ---- (iv) closed-form stencils and macro-element matrices ---------------------
def test_3d_interior_stencils_closed_form():
    hx, hy, hz = 0.5, 0.25, 0.2
    c, cells = o.box_mesh(4 * hx, 4 * hy, 4 * hz, 4, 4, 4)
    K, M, load = o.assemble_scalar(c, cells)
    mid = 2 * 25 + 2 * 5 + 2
    offs = {(dx, dy, dz): mid + dx + 5 * dy + 25 * dz for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)}
    rowK = K[mid].toarray().ravel()
    rowM = M[mid].toarray().ravel() / (hx * hy * hz)
    expK = {(0, 0, 0): 2 * (hy * hz / hx + hx * hz / hy + hx * hy / hz), (1, 0, 0): -hy * hz / hx,
            (0, 1, 0): -hx * hz / hy, (0, 0, 1): -hx * hy / hz}
    for d, idx in offs.items():
        key = tuple(abs(v) for v in d)
        same_sign = len({v for v in d if v != 0}) <= 1
        kexp = expK.get(key, 0.0) if sum(abs(v) for v in d) <= 1 else 0.0
        assert rowK[idx] == pytest.approx(kexp, abs=1e-12), d                 # classic 7-point, no diagonals
        nnz = sum(abs(v) for v in d)
        mexp = {0: 2 / 5, 1: 1 / 20, 2: 1 / 30, 3: 1 / 20}[nnz] if same_sign else 0.0
        assert rowM[idx] == pytest.approx(mexp, abs=1e-13), d                 # 15-point consistent mass
    assert rowM.sum() == pytest.approx(1.0)
    assert load[mid] == pytest.approx(hx * hy * hz)


def test_3d_macro_element_matrices():
    c, cells = o.box_mesh(1, 1, 1, 1, 1, 1)
    K, M, load = o.assemble_scalar(c, cells)
    K6 = np.array([[6, -2, -2, 0, -2, 0, 0, 0], [-2, 4, 0, -1, 0, -1, 0, 0], [-2, 0, 4, -1, 0, 0, -1, 0],
                   [0, -1, -1, 4, 0, 0, 0, -2], [-2, 0, 0, 0, 4, -1, -1, 0], [0, -1, 0, 0, -1, 4, 0, -2],
                   [0, 0, -1, 0, -1, 0, 4, -2], [0, 0, 0, -2, 0, -2, -2, 6]])
    M120 = np.array([[12, 2, 2, 2, 2, 2, 2, 6], [2, 4, 0, 1, 0, 1, 0, 2], [2, 0, 4, 1, 0, 0, 1, 2],
                     [2, 1, 1, 4, 0, 0, 0, 2], [2, 0, 0, 0, 4, 1, 1, 2], [2, 1, 0, 0, 1, 4, 0, 2],
                     [2, 0, 1, 0, 1, 0, 4, 2], [6, 2, 2, 2, 2, 2, 2, 12]])
    assert np.allclose(6 * K.toarray(), K6, atol=1e-13)
    assert np.allclose(120 * M.toarray(), M120, atol=1e-13)
    assert np.allclose(24 * load, [6, 2, 2, 2, 2, 2, 2, 6])


def test_2d_and_1d_stencils():
    hx, hy = 0.5, 0.25
    c, cells = o.rectangle_mesh(4 * hx, 4 * hy, 4, 4)
    K, M, _ = o.assemble_scalar(c, cells)
    mid = 2 * 5 + 2
    rk, rm = K[mid].toarray().ravel(), M[mid].toarray().ravel() / (hx * hy)
    assert rk[mid] == pytest.approx(2 * (hy / hx + hx / hy)) and rk[mid + 1] == pytest.approx(-hy / hx)
    assert rk[mid + 5] == pytest.approx(-hx / hy) and abs(rk[mid + 6]) < 1e-14
    assert rm[mid] == pytest.approx(0.5)
    for d in (1, -1, 5, -5, 6, -6):
        assert rm[mid + d] == pytest.approx(1 / 12)
    assert abs(rm[mid + 4]) < 1e-15 and abs(rm[mid - 4]) < 1e-15            # opposite diagonal absent
    c1, cells1 = o.interval_mesh(4, 0.0, 2.0)
    K1, M1, _ = o.assemble_scalar(c1, cells1)
    assert np.allclose(K1[2].toarray().ravel()[1:4] * 0.5, [-1, 2, -1])
    assert np.allclose(M1[2].toarray().ravel()[1:4] * 6 / 0.5, [1, 4, 1])


# ---- (v) trivial solutions ------------------------------------------------------------------------
def test_steady_all_dirichlet_no_source_is_constant():
    f = o.solve_heat_3d(1, 2, 0.5, 4, 5, 3, 1.3, 7.5, 0.0, 0.1, 1, steady=True)
    assert np.abs(f.values[0] - 7.5).max() < 1e-12
    f2 = o.solve_heat_2d(1, 1, 6, 5, 1.0, -3.0, 0.0, 0.1, 1, steady=True)
    assert np.abs(f2.values[0] + 3.0).max() < 1e-12


def test_elasticity_zero_load_is_zero():
    e = o.solve_elasticity_3d(1, 0.5, 0.5, 4, 2, 2, 210e9, 0.3)
    assert np.abs(e.values).max() == 0.0
    e2 = o.solve_elasticity_2d(1, 1, 4, 4, 210e9, 0.3, quantity="strain")
    assert np.abs(e2.values).max() == 0.0


# ---- (vi) 1-D bar --------------------------------------------------------------------------------
def test_1d_bar_displacement_and_projection():
    Lb, nx, E, A, bf = 2.0, 40, 210e9, 0.5, 1.0e4
    e = o.solve_elasticity_1d(Lb, nx, E, A, bf, "stress")
    x = e.coords[:, 0]
    u = e.extra["displacement"]
    assert np.abs(u - bf * (Lb * x - x ** 2 / 2) / (E * A)).max() < 1e-12 * np.abs(u).max() + 1e-20   # nodally exact
    sig = e.values[0]
    inner = slice(3, -3)
    assert np.abs(sig[inner] - bf * (Lb - x[inner]) / A).max() < 2e-3 * sig.max()      # L2 projection end effects aside
    eps = o.solve_elasticity_1d(Lb, nx, E, A, bf, "strain").values[0]
    assert np.allclose(eps * E, sig, rtol=1e-12)
    assert e.meta["name"] == "axial_stress" and e.meta["pde"] == "elasticity_1d"


# ---- elasticity: patch test, rigid-body modes, symmetry ---------------------------------------------
@pytest.mark.parametrize("dim", [2, 3])
def test_elasticity_patch_and_rigid_body(dim):
    if dim == 2:
        c, cells = o.rectangle_mesh(1.0, 0.5, 3, 2)
    else:
        c, cells = o.box_mesh(1.0, 0.5, 0.25, 3, 2, 2)
    lam, mu = o.lame(210e9, 0.3)
    K = o.assemble_elasticity(c, cells, lam, mu)
    assert abs(K - K.T).max() < 1e-3 * abs(K).max() * 1e-9
    n = c.shape[0]
    # rigid translations and rotations are in the null space
    modes = [np.tile(np.eye(dim)[a], n) for a in range(dim)]
    if dim == 2:
        modes.append(np.stack([-c[:, 1], c[:, 0]], 1).ravel())
    else:
        for a, b in ((0, 1), (0, 2), (1, 2)):
            r = np.zeros((n, 3))
            r[:, a], r[:, b] = -c[:, b], c[:, a]
            modes.append(r.ravel())
    for m in modes:
        assert np.abs(K @ m).max() < 1e-6 * abs(K).max() * np.abs(m).max() * 1e-3
    # patch test: a linear displacement field gives the exact constant strain in every cell
    G = np.arange(1, dim * dim + 1, dtype=float).reshape(dim, dim) * 1e-4
    u = c @ G.T
    eps = o.cell_strain(c, cells, u)
    assert np.allclose(eps, 0.5 * (G + G.T)[None], atol=1e-15)
    # interior equilibrium: K u has no interior forces for a constant-stress state
    f = (K @ u.ravel()).reshape(n, dim)
    nn = (4, 3) if dim == 2 else (4, 3, 3)
    idx = np.arange(n)
    ix = idx % nn[0]
    iy = (idx // nn[0]) % nn[1]
    interior = (ix > 0) & (ix < nn[0] - 1) & (iy > 0) & (iy < nn[1] - 1)
    if dim == 3:
        iz = idx // (nn[0] * nn[1])
        interior &= (iz > 0) & (iz < nn[2] - 1)
    assert interior.any()
    assert np.abs(f[interior]).max() < 1e-9 * np.abs(f).max()


def test_von_mises_uniaxial():
    # uniaxial stress state: von Mises stress equals the axial stress
    E, nu = 200e9, 0.25
    lam, mu = o.lame(E, nu)
    s = 1.0e6
    eps = np.diag([s / E, -nu * s / E, -nu * s / E])[None]
    assert o.von_mises(eps, lam, mu, "stress")[0] == pytest.approx(s, rel=1e-12)
    assert o.von_mises(eps, lam, mu, "strain")[0] == pytest.approx(2.0 / 3.0 * (1 + nu) * s / E, rel=1e-12)


# ---- (vii)/(viii) SPD, symmetry, constants in the null space of K, O(h^2) convergence -----------------
def test_scalar_matrices_properties():
    c, cells = o.box_mesh(1, 0.7, 0.4, 3, 4, 2)
    K, M, load = o.assemble_scalar(c, cells)
    assert abs(K - K.T).max() < 1e-14 and abs(M - M.T).max() < 1e-15
    assert np.abs(K @ np.ones(K.shape[0])).max() < 1e-13
    assert np.allclose(M @ np.ones(M.shape[0]), load)
    assert load.sum() == pytest.approx(1 * 0.7 * 0.4)
    assert np.linalg.eigvalsh(M.toarray()).min() > 0
    assert np.linalg.eigvalsh(K.toarray()).min() > -1e-12


def test_second_order_convergence_steady_poisson():
    # -Laplace u = 1 on the unit cube, u = 0 on the boundary: max u -> 0.056212... ; check the error ratio
    errs = []
    for n in (4, 8, 16):
        f = o.solve_heat_3d(1, 1, 1, n, n, n, 1.0, 0.0, 0.0, 0.1, 1, steady=True, source_type="constant", source_value=1.0)
        errs.append(abs(f.values[0].max() - 0.0562128))
    assert errs[0] / errs[1] > 3.0 and errs[1] / errs[2] > 3.0


def test_directional_bc_sets():
    mask, vals = o.box_dirichlet(3, 2, 2, T_left=5.0, T_right=None, T_side=1.0)
    n = 4 * 3 * 3
    ix = np.arange(n) % 4
    assert mask[ix == 0].all() and (vals[ix == 0] == 5.0).all()
    assert not mask[ix == 3].any()                                   # right face natural
    side = (ix > 0) & (ix < 3)
    iy = (np.arange(n) // 4) % 3
    iz = np.arange(n) // 12
    onb = (iy == 0) | (iy == 2) | (iz == 0) | (iz == 2)
    assert (mask[side] == onb[side]).all() and (vals[side & onb] == 1.0).all()


def test_projected_sine_initial_condition():
    f = o.solve_heat_1d(math.pi, 32, 1.0, 0.0, 0.0, 0.0, 0.01, 1, initial_type="sine", initial_amplitude=2.0,
                        initial_wavenumber=1.0)
    x = f.coords[:, 0]
    # L2 projection of the P2 interpolant is close to, but not equal to, nodal interpolation
    d = np.abs(f.values[0][1:-1] - 2.0 * np.sin(x[1:-1])).max()
    assert 1e-6 < d < 5e-3
    assert f.values[0][0] == 0.0 and f.values[0][-1] == 0.0           # bc.apply after project


# ---- DirichletBC marks facets (DOLFIN's default method "topological", [3P]) ----------------------------------------------
def _facet_rule_set(nx, ny, nz, T_left, T_right, T_side):
    """The constrained node set by the facet rule itself: boundary facets of the mesh (triangles that belong to ONE
    tetrahedron), a facet is taken when the predicate of fenics_mcp_server.py:606-617 holds at its three vertices and at its
    midpoint, every vertex of a taken facet is constrained; bcs applied in list order (left, right, side)."""
    Lx, Ly, Lz = 1.0, 0.7, 1.3
    coords, cells = o.box_mesh(Lx, Ly, Lz, nx, ny, nz)
    faces = np.concatenate([cells[:, [a, b, c]] for a, b, c in ((0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3))])
    faces = np.sort(faces, axis=1)
    uniq, cnt = np.unique(faces, axis=0, return_counts=True)
    bnd = uniq[cnt == 1]
    near = lambda a, b: np.abs(a - b) < 3e-16 * 16             # noqa: E731  (DOLFIN_EPS)
    preds = {"left": lambda x: near(x[..., 0], 0.0), "right": lambda x: near(x[..., 0], Lx),
             "side": lambda x: ~(near(x[..., 0], 0.0) | near(x[..., 0], Lx))}
    n = coords.shape[0]
    mask, vals = np.zeros(n, bool), np.zeros(n)
    for name, T in (("left", T_left), ("right", T_right), ("side", T_side)):
        if T is None:
            continue
        P = coords[bnd]                                          # [nf, 3, 3]
        ok = preds[name](P).all(axis=1) & preds[name](P.mean(axis=1))
        nodes = np.unique(bnd[ok])
        mask[nodes] = True
        vals[nodes] = T
    return mask, vals


@pytest.mark.parametrize("nx", [1, 2, 3, 4, 5])
def test_box_dirichlet_is_the_facet_rule(nx):
    for ny, nz in ((1, 1), (2, 3), (3, 2)):
        for T_left, T_right, T_side in ((1.0, None, None), (None, 2.0, None), (None, None, 3.0), (1.0, 2.0, 3.0), (1.0, None, 3.0)):
            m0, v0 = _facet_rule_set(nx, ny, nz, T_left, T_right, T_side)
            m1, v1 = o.box_dirichlet(nx, ny, nz, T_left=T_left, T_right=T_right, T_side=T_side)
            assert (m0 == m1).all(), (nx, ny, nz, T_left, T_right, T_side)
            assert np.array_equal(v0[m0], v1[m1])
    if nx <= 2:                                                  # every side facet touches x = 0 or x = Lx: T_side constrains nothing
        m, _ = o.box_dirichlet(nx, 3, 3, T_side=5.0)
        assert not m.any()


def test_side_value_on_a_two_cell_grid_leaves_the_transient_problem_unconstrained():
    """nx = 2, T_side only: no Dirichlet node, so the backward-Euler steps conserve the mean of a constant initial field exactly"""
    f = o.solve_heat_3d(1.0, 1.0, 1.0, 2, 3, 3, 1.0, 0.0, 20.0, 0.01, 3, T_side=5.0)
    assert np.abs(f.values - 20.0).max() < 1e-10
