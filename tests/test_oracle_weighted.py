"""Known answers for the oracle's curvilinear (weighted) heat restatement, fenics_mcp_server.py:769-1464 —
like the Cartesian path it is unpinned by the reference (no tests / fixtures exist, FEniCS not installable)."""
import math

import numpy as np
import pytest

from oracle import fem_p1 as o


def test_weight_integrals_are_exact_for_polynomial_weights():
    # int_a^b r dr, int r^2 dr: the load vector sums to the exact integral (P1 / P2 interpolation is exact)
    c, cells = o.interval_mesh(7, 0.3, 2.0)
    _, M, load = o.assemble_weighted(c, cells, "r")
    assert load.sum() == pytest.approx((2.0 ** 2 - 0.3 ** 2) / 2, rel=1e-14)
    assert np.allclose(M @ np.ones(8), load, rtol=1e-14)
    _, M2, load2 = o.assemble_weighted(c, cells, "r2")
    assert load2.sum() == pytest.approx((2.0 ** 3 - 0.3 ** 3) / 3, rel=1e-14)
    # 2-D: int int r dr dz = (b^2-a^2)/2 * H
    c2, cells2 = o.shifted_rectangle_mesh(0.5, 1.5, 0.0, 2.0, 5, 4)
    K, M, load = o.assemble_weighted(c2, cells2, "r")
    assert load.sum() == pytest.approx((1.5 ** 2 - 0.5 ** 2) / 2 * 2.0, rel=1e-14)
    assert abs(K - K.T).max() < 1e-14 and abs(M - M.T).max() < 1e-15
    assert np.abs(K @ np.ones(K.shape[0])).max() < 1e-13
    # weight 1 reproduces the unweighted matrices: emulate with r on [1,1+eps]? use the identity w=r at r->const
    K0, M0, l0 = o.assemble_scalar(c2, cells2)
    assert load.sum() / l0.sum() == pytest.approx(1.0, rel=1e-12)             # mean radius of [0.5, 1.5] is 1


def test_r2sin_weight_converges_to_the_sphere_volume():
    vols = []
    for n in (4, 8, 16):
        c, cells = o.box_mesh(1.0, np.pi, 2 * np.pi, n, n, n)
        _, _, load = o.assemble_weighted(c, cells, "r2sin")
        vols.append(load.sum())
    exact = 4.0 / 3.0 * np.pi
    errs = [abs(v - exact) for v in vols]
    assert errs[2] < 2e-4 * exact and errs[0] / errs[1] > 8 and errs[1] / errs[2] > 8      # P2 interpolation of sin: O(h^4)


def test_1d_cylindrical_steady_converges_to_the_log_profile():
    errs = []
    for nr in (16, 32, 64):
        f = o.solve_heat_1d_cylindrical(0.5, 2.0, nr, 1.3, 10.0, 2.0, 0.0, 0.01, 1, steady=True)
        r = f.coords[:, 0]
        exact = 10.0 + (2.0 - 10.0) * np.log(r / 0.5) / math.log(2.0 / 0.5)
        errs.append(np.abs(f.values[0] - exact).max())
    assert errs[0] / errs[1] > 3.5 and errs[1] / errs[2] > 3.5 and errs[2] < 2e-4
    assert f.meta["geometry_type"] == "annulus" and f.meta["coordinate_system"] == "cylindrical"


def test_1d_spherical_steady_converges_to_the_1_over_r_profile_and_solid_sphere_source():
    errs = []
    for nr in (16, 32, 64):
        f = o.solve_heat_1d_spherical(0.5, 2.0, nr, 1.0, 10.0, 2.0, 0.0, 0.01, 1, steady=True)
        r = f.coords[:, 0]
        A = (10.0 - 2.0) / (1 / 0.5 - 1 / 2.0)
        exact = 2.0 + A * (1 / r - 1 / 2.0)
        errs.append(np.abs(f.values[0] - exact).max())
    assert errs[0] / errs[1] > 3.5 and errs[1] / errs[2] > 3.5
    # solid sphere (r_inner = 0: no inner BC, :968), uniform source: u = T_out + f (R^2 - r^2) / (6 k)
    g = o.solve_heat_1d_spherical(0.0, 1.5, 64, 2.0, 99.0, 3.0, 0.0, 0.01, 1, steady=True, source_type="constant", source_value=4.0)
    r = g.coords[:, 0]
    assert np.abs(g.values[0] - (3.0 + 4.0 * (1.5 ** 2 - r ** 2) / 12.0)).max() < 2e-3
    assert g.meta["geometry_type"] == "sphere"


def test_transient_weighted_conserves_constants_and_decays():
    f = o.solve_heat_2d_cylindrical(0.0, 1.0, 2.0, 6, 8, 1.0, 5.0, 5.0, 0.01, 3)
    assert np.abs(f.values - 5.0).max() < 1e-12                                # u == T_b is a fixed point
    assert np.allclose(f.coords[:, 1], 0.0) and f.coords[:, 2].max() == pytest.approx(2.0)
    g = o.solve_heat_3d_spherical(0.2, 1.0, 4, 4, 4, 1.0, 0.0, 20.0, 0.01, 3)
    m = np.abs(g.values).max(axis=1)
    assert m[0] == 20.0 and np.all(np.diff(m) < 0) and g.values.min() > -1e-9
    h = o.solve_heat_2d_spherical(0.2, 1.0, 5, 6, 1.0, 1.0, 20.0, 0.01, 2)
    assert h.values.shape == (3, 42) and np.allclose(h.coords[:, 1], 0.0)


def test_cylinder_fallback_restatement():
    # the weight integrates sqrt(y^2+z^2) over the box: converges to the exact value of int int r dy dz on [-R,R]^2
    R = 0.5
    vals = []
    for n in (4, 8, 16):
        c, cells = o.box_mesh(1.0, 2 * R, 2 * R, 2, n, n)
        c = c + np.array([0.0, -R, -R])
        _, _, load = o.assemble_weighted(c, cells, "rcyl")
        vals.append(load.sum())
    a = 2 * R
    exact = a ** 3 / 6 * (math.sqrt(2) + math.log(1 + math.sqrt(2)))            # int over a square of side a centred at 0 of r
    errs = [abs(v - exact) for v in vals]
    assert errs[2] < 1e-3 * exact and errs[0] > errs[1] > errs[2]
    # composite core: the per-cell factor only touches cells entirely inside the core
    f = o.solve_heat_3d_cylinder_fallback(1.0, 6, 8, 8, 1.0, 20.0, 100.0, 0.01, 2, cylinder_radius=0.5, core_radius=0.3,
                                          core_diffusivity=50.0)
    g = o.solve_heat_3d_cylinder_fallback(1.0, 6, 8, 8, 1.0, 20.0, 100.0, 0.01, 2, cylinder_radius=0.5)
    assert f.values.shape == g.values.shape == (3, 7 * 9 * 9) and np.abs(f.values[-1] - g.values[-1]).max() > 1e-3
    assert f.meta["base_diffusivity"] == 1.0 and "diffusivity" not in f.meta and g.meta["diffusivity"] == 1.0
    assert g.meta["Ly"] == 1.0 and g.meta["coordinate_system"] == "cylindrical" and g.coords[:, 1].min() == -0.5
    # directional: T_side constrains nothing on the box fallback; left/right faces are set, the rest is natural
    h = o.solve_heat_3d_cylinder_fallback(1.0, 4, 4, 4, 1.0, 0.0, 5.0, 0.01, 1, cylinder_radius=0.5, T_left=50.0, T_side=99.0)
    x = h.coords[:, 0]
    assert np.all(h.values[:, x == 0.0] == 50.0) and not np.any(h.values == 99.0)
    assert "T_side" in h.meta and "T_boundary" not in h.meta
    # u == T_b fixed point
    k = o.solve_heat_3d_cylinder_fallback(2.0, 3, 4, 4, 1.0, 7.0, 7.0, 0.01, 2, cylinder_radius=0.5)
    assert np.abs(k.values - 7.0).max() < 1e-12


def test_unit_weight_is_the_plain_assembly_and_the_core_factor_scales_whole_cells():
    """the "one" weight kind (composite core on the plain box): K, M, load equal the unweighted assembly; with a per-cell
    factor the stiffness is the cell-wise weighted sum; the host twin of the GPU's per-node rows reproduces both"""
    import ctypes as C
    from pdegpu import _lib as L
    n, Ls = (4, 3, 5), (1.0, 0.6, 0.9)
    coords, cells = o.box_mesh(*Ls, *n)
    K0, M0, l0 = o.assemble_scalar(coords, cells)
    K1, M1, l1 = o.assemble_weighted(coords, cells, "one")
    assert abs(K1 - K0).max() < 1e-14 and abs(M1 - M0).max() < 1e-15 and np.abs(l1 - l0).max() < 1e-15
    core = o.core_cells(coords, cells, 0.5)
    assert 0 < core.sum() < cells.shape[0]
    Kc, _, _ = o.assemble_weighted(coords, cells, "one", np.where(core, 9.0, 1.0))
    Kin, _, _ = o.assemble_weighted(coords, cells[core], "one")
    assert abs((Kc - K0) - 8.0 * Kin).max() < 1e-12
    # host twin of k_weighted_assemble (kind 0 + core) against the oracle's assembled rows
    lib = L.load_library()
    grid = L.make_grid(3, n, Ls)
    w = L.Weight()
    w.kind, w.core_radius, w.core_ratio = L.WEIGHT_ONE, 0.5, 9.0
    N = coords.shape[0]
    rK, rM, ld = np.zeros((N, 15)), np.zeros((N, 15)), np.zeros(N)
    assert lib.pdegpu_weighted_rows_host(grid, w, L.dptr(rK.reshape(-1)), L.dptr(rM.reshape(-1)), L.dptr(ld)) == 0
    offs = [13, 12, 14, 10, 16, 4, 22, 9, 17, 3, 23, 1, 25, 0, 26]
    NX, NY = n[0] + 1, n[1] + 1
    Kd, Md = Kc.toarray(), M0.toarray()
    for q, oid in enumerate(offs):
        dx, dy, dz = oid % 3 - 1, (oid // 3) % 3 - 1, oid // 9 - 1
        for i in range(N):
            ix, iy, iz = i % NX, (i // NX) % NY, i // (NX * NY)
            jx, jy, jz = ix + dx, iy + dy, iz + dz
            inside = 0 <= jx < NX and 0 <= jy < NY and 0 <= jz <= n[2]
            j = (jz * NY + jy) * NX + jx
            assert abs(rK[i, q] - (Kd[i, j] if inside else 0.0)) < 1e-12 and abs(rM[i, q] - (Md[i, j] if inside else 0.0)) < 1e-14
    assert np.abs(ld - l0).max() < 1e-15


def test_box_core_and_cylinder_trig_restatements():
    """oracle-side sanity of the two (f) leftovers: the core changes the box solution and leaves the metadata of :747-753; the
    cylinder branch with a cosine initial field starts from the projected field (not the constant)"""
    a = o.solve_heat_3d(1.0, 0.8, 0.9, 5, 6, 7, 1.0, 20.0, 70.0, 0.01, 2, core_radius=0.45, core_diffusivity=30.0)
    b = o.solve_heat_3d(1.0, 0.8, 0.9, 5, 6, 7, 1.0, 20.0, 70.0, 0.01, 2)
    assert np.abs(a.values[-1] - b.values[-1]).max() > 1e-2 and np.array_equal(a.values[0], b.values[0])
    assert a.meta["base_diffusivity"] == 1.0 and a.meta["core_diffusivity"] == 30.0 and "diffusivity" not in a.meta
    c = o.solve_heat_3d_cylinder_fallback(1.0, 5, 6, 6, 1.0, 0.0, 50.0, 0.01, 1, cylinder_radius=0.5, initial_type="cosine",
                                          initial_amplitude=10.0, initial_wavenumber=1.0)
    ctr = np.argmin(np.linalg.norm(c.coords - np.array([0.0, 0.0, 0.0]), axis=1))
    x0 = np.argmin(np.linalg.norm(c.coords - np.array([0.4, 0.0, 0.0]), axis=1))
    assert c.values[0, ctr] == 0.0                                              # boundary node: bc.apply after the projection
    assert abs(c.values[0, x0] - 10.0 * np.cos(0.4)) < 0.5                      # interior: ~ A cos(kx) cos(0) cos(0)
