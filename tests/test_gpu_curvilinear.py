"""GPU parity of the curvilinear (weighted) heat tools vs the CPU oracle: per-node assembled coefficients
(k_weighted_assemble), ELL mode of the general kernels, Jacobi-PCG.  fp64; tolerance 1e-8 relative to max|u|
with PCG rtol 1e-12 (r^2 sin(theta) weights span orders of magnitude across the grid)."""
import numpy as np
import pytest

from oracle import fem_p1 as o

pytestmark = pytest.mark.gpu
TOL = 1e-8


def rel_err(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def _check(got, ref):
    assert got.values.shape == ref.values.shape and np.allclose(got.times, ref.times)
    assert np.allclose(got.coords, ref.coords, atol=1e-14)
    assert rel_err(got.values, ref.values) < TOL
    for k, v in ref.meta.items():
        assert got.meta[k] == v, k
    assert got.dim == ref.dim


@pytest.mark.parametrize("r_inner", [0.0, 0.3])
@pytest.mark.parametrize("steady", [False, True])
@pytest.mark.parametrize("kind", ["cylindrical", "spherical"])
def test_radial_1d(lib, gpu_ctx, kind, steady, r_inner):
    from pdegpu import curvilinear as cv
    kw = dict(r_inner=r_inner, r_outer=1.5, nr=40, diffusivity=0.8, T_inner=100.0, T_outer=20.0, T_initial=20.0, dt=0.004,
              num_steps=4, steady=steady, source_type="constant", source_value=7.0)
    if kind == "cylindrical":
        _check(cv._solve_heat_1d_cylindrical_raw(**kw, rtol=1e-12), o.solve_heat_1d_cylindrical(**kw))
    else:
        _check(cv._solve_heat_1d_spherical_raw(**kw, rtol=1e-12), o.solve_heat_1d_spherical(**kw))


@pytest.mark.parametrize("r_inner", [0.0, 0.2])
@pytest.mark.parametrize("steady", [False, True])
def test_2d_cylindrical(lib, gpu_ctx, steady, r_inner):
    from pdegpu import curvilinear as cv
    kw = dict(r_inner=r_inner, r_outer=1.0, z_length=2.0, nr=14, nz=19, diffusivity=1.0, T_boundary=20.0, T_initial=80.0,
              dt=0.01, num_steps=3, steady=steady, source_type="constant", source_value=50.0)
    _check(cv._solve_heat_2d_cylindrical_raw(**kw, rtol=1e-12), o.solve_heat_2d_cylindrical(**kw))


@pytest.mark.parametrize("r_inner", [0.0, 0.1])
@pytest.mark.parametrize("steady", [False, True])
def test_2d_spherical(lib, gpu_ctx, steady, r_inner):
    from pdegpu import curvilinear as cv
    kw = dict(r_inner=r_inner, r_outer=1.0, nr=12, ntheta=15, diffusivity=1.0, T_boundary=20.0, T_initial=90.0, dt=0.01,
              num_steps=3, steady=steady, source_type="constant", source_value=30.0)
    _check(cv._solve_heat_2d_spherical_raw(**kw, rtol=1e-12), o.solve_heat_2d_spherical(**kw))


@pytest.mark.parametrize("steady", [False, True])
def test_3d_spherical(lib, gpu_ctx, steady):
    from pdegpu import curvilinear as cv
    kw = dict(r_inner=0.1, r_outer=1.0, nr=8, ntheta=7, nphi=9, diffusivity=1.0, T_boundary=20.0, T_initial=60.0, dt=0.01,
              num_steps=3, steady=steady, source_type="constant", source_value=10.0)
    _check(cv._solve_heat_3d_spherical_raw(**kw, rtol=1e-12), o.solve_heat_3d_spherical(**kw))


def test_curvilinear_tools_through_the_params_boundary(lib, gpu_ctx, tmp_path):
    """PDEParameters dict -> curvilinear tool -> pickle (file prefix of the reference, :2109, 2282, 2357, 2435, 2510)"""
    import pickle
    from pathlib import Path
    from pdegpu import server
    for p, prefix in (({"pde_type": "heat", "dim": 1, "geometry_type": "cylinder", "num_steps": 2}, "heat_1d_cylindrical_"),
                      ({"pde_type": "heat", "dim": 1, "geometry_type": "sphere", "num_steps": 2}, "heat_1d_spherical_"),
                      ({"pde_type": "heat", "dim": 2, "geometry_type": "tube", "nx": 8, "ny": 8, "num_steps": 2}, "heat_2d_cylindrical_"),
                      ({"pde_type": "heat", "dim": 2, "geometry_type": "ball", "nx": 8, "ny": 8, "num_steps": 2}, "heat_2d_spherical_"),
                      ({"pde_type": "heat", "dim": 3, "geometry_type": "sphere", "nx": 5, "ny": 5, "nz": 5, "num_steps": 2}, "heat_3d_spherical_")):
        out = server.solve(p, data_dir=str(tmp_path))
        assert "error" not in out, out
        assert Path(out["data_file"]).name.startswith(prefix)
        f = pickle.load(open(out["data_file"], "rb"))
        assert f.values.shape[0] == 3 and np.isfinite(f.values).all()
        assert f.meta["coordinate_system"] in ("cylindrical", "spherical")


@pytest.mark.parametrize("case", [
    dict(),                                                                      # uniform T_boundary on the whole box boundary
    dict(T_left=100.0, T_right=20.0, T_side=55.0),                               # directional; T_side matches no facet on the fallback mesh
    dict(T_left=100.0),                                                          # everything else natural
    dict(core_radius=0.3, core_diffusivity=40.0),                                # composite core, DG0 diffusivity
    dict(core_radius=0.3, core_diffusivity=40.0, T_left=80.0, T_right=10.0, steady=True, source_type="constant", source_value=5.0),
])
def test_3d_cylinder_boxmesh_fallback(lib, gpu_ctx, case):
    """solve_heat_3D(geometry_type="cylinder") as the reference runs it without mshr (:523-529, :645, :537-604)"""
    from pdegpu import heat
    kw = dict(Lx=1.0, nx=7, ny=8, nz=6, diffusivity=1.0, T_boundary=20.0, T_initial=60.0, dt=0.01, num_steps=3,
              cylinder_radius=0.5)
    kw.update(case)
    ref = o.solve_heat_3d_cylinder_fallback(**kw)
    got = heat._solve_heat_3d_raw(Ly=1.0, Lz=1.0, geometry_type="cylinder", rtol=1e-12, **kw)
    _check(got, ref)


@pytest.mark.parametrize("kind", ["cosine", "sine"])
def test_cylinder_fallback_projected_trig_initial_condition(lib, gpu_ctx, kind):
    """sine / cosine initial conditions on the cylinder branch (:672-687): project(Expression(degree=2), V) on the shifted BoxMesh
    with the plain dx measure, then the weighted backward-Euler steps; with a composite core on top"""
    from pdegpu import heat
    kw = dict(Lx=1.2, nx=6, ny=7, nz=5, diffusivity=0.9, T_boundary=3.0, T_initial=60.0, dt=0.01, num_steps=2, cylinder_radius=0.5,
              initial_type=kind, initial_amplitude=40.0, initial_wavenumber=2.3, core_radius=0.25, core_diffusivity=12.0)
    ref = o.solve_heat_3d_cylinder_fallback(**kw)
    got = heat._solve_heat_3d_raw(Ly=1.0, Lz=1.0, geometry_type="cylinder", rtol=1e-12, **kw)
    _check(got, ref)
    interior = ~o.box_dirichlet(6, 7, 5, T_boundary=3.0)[0]
    assert np.abs(ref.values[0][interior]).max() > 5.0                       # the projected field really is the trigonometric one


@pytest.mark.parametrize("case", [
    dict(),                                                                  # uniform boundary value, constant initial field
    dict(steady=True, source_type="constant", source_value=30.0),
    dict(initial_type="cosine", initial_amplitude=25.0, initial_wavenumber=1.7),
    dict(initial_type="zero", T_left=90.0, T_side=10.0),                     # directional: x = Lx stays natural
    dict(T_left=50.0, steady=True, source_type="constant", source_value=4.0),
])
def test_3d_box_with_composite_core(lib, gpu_ctx, case):
    """the composite core of solve_heat_3D on the PLAIN box (:537-572, unit weight :647): cells within core_radius of the x-axis
    — the box edge y = z = 0 — carry core_diffusivity"""
    from pdegpu import heat
    kw = dict(Lx=1.0, Ly=0.8, Lz=0.9, nx=7, ny=8, nz=9, diffusivity=1.0, T_boundary=20.0, T_initial=70.0, dt=0.005, num_steps=3,
              core_radius=0.45, core_diffusivity=35.0)
    kw.update(case)
    ref = o.solve_heat_3d(**kw)
    plain = o.solve_heat_3d(**{k: v for k, v in kw.items() if not k.startswith("core_")})
    assert rel_err(ref.values[-1], plain.values[-1]) > 1e-4                 # the core matters in this set-up
    got = heat._solve_heat_3d_raw(rtol=1e-12, **kw)
    _check(got, ref)
    assert got.meta["core_radius"] == 0.45 and "diffusivity" not in got.meta and got.meta["geometry_type"] == "box"


def test_cylinder_through_the_tool_and_params(lib, gpu_ctx, tmp_path):
    import pickle
    from pdegpu import server
    out = server.solve({"pde_type": "heat", "dim": 3, "geometry_type": "tube", "domain_size": {"r1": 0.1, "r2": 0.4, "h": 1.5},
                        "nx": 6, "ny": 10, "nz": 10, "bc_values": {"t_left": 100.0, "T_right": 20.0}, "core_diffusivity": 25.0,
                        "initial_value": 20.0, "num_steps": 2, "dt": 0.01}, data_dir=str(tmp_path))
    assert "error" not in out, out
    f = pickle.load(open(out["data_file"], "rb"))
    assert out["solver"] == "solve_heat_3D" and f.meta["geometry_type"] == "cylinder" and f.meta["core_radius"] == 0.1
    assert f.values.shape == (3, 7 * 9 * 9) and np.isfinite(f.values).all()
    kw = dict(out["solver_args"])
    ref = o.solve_heat_3d_cylinder_fallback(kw["Lx"], kw["nx"], kw["ny"], kw["nz"], kw["diffusivity"], kw["T_boundary"], kw["T_initial"],
                                            kw["dt"], kw["num_steps"], kw["cylinder_radius"], T_left=kw["T_left"], T_right=kw["T_right"],
                                            core_radius=kw["core_radius"], core_diffusivity=kw["core_diffusivity"])
    assert rel_err(f.values, ref.values) < TOL


@pytest.mark.parametrize("case", ["spherical", "spherical_two_planes", "cylinder", "core_box", "core_box_one_cell_in_x"])
def test_compact_coefficient_rows_equal_one_row_per_node(lib, gpu_ctx, case):
    """Where the weight does not depend on one coordinate (r^2 sin(theta): not on phi; the cylinder fallback's sqrt(y^2 + z^2), the composite core and the plain
    weight: not on x) the weighted operators store the coefficient rows of three representative planes / columns (ell_compact = 1, the default).
    Same arithmetic on the same numbers: the fields must equal those of the one-row-per-node form bit for bit — incl. grids with only two planes /
    two columns, where no interior representative exists."""
    from pdegpu import _lib, curvilinear as cv, heat
    outs = []
    try:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast", 0))        # (the class-table kernels: the marching ones have tests of their own)
        for compact in (1, 0):
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"ell_compact", compact))
            if case.startswith("spherical"):
                nphi = 1 if case.endswith("two_planes") else 9
                f = cv._solve_heat_3d_spherical_raw(0.1, 1.0, 8, 7, nphi, 1.0, 20.0, 60.0, 0.01, 3, source_type="constant", source_value=10.0, rtol=1e-12)
            elif case == "cylinder":
                f = heat._solve_heat_3d_raw(Lx=1.0, Ly=1.0, Lz=1.0, nx=7, ny=8, nz=6, diffusivity=1.0, T_boundary=20.0, T_initial=60.0, dt=0.01, num_steps=3,
                                            geometry_type="cylinder", cylinder_radius=0.5, core_radius=0.3, core_diffusivity=40.0, T_left=80.0, rtol=1e-12)
            else:
                nx = 1 if case.endswith("one_cell_in_x") else 7
                f = heat._solve_heat_3d_raw(Lx=1.0, Ly=0.8, Lz=0.9, nx=nx, ny=8, nz=9, diffusivity=1.0, T_boundary=20.0, T_initial=70.0, dt=0.005,
                                            num_steps=3, core_radius=0.45, core_diffusivity=35.0, initial_type="cosine", initial_amplitude=25.0,
                                            initial_wavenumber=1.7, rtol=1e-12, **(dict(T_left=50.0) if nx == 1 else {}))   # (one cell in x: only x = 0 clamped, or no unknown is left)
            outs.append(f)
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"ell_compact", 1))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast", 1))
    assert outs[0].meta["cg_iterations"] == outs[1].meta["cg_iterations"]
    assert (outs[0].meta["cg_iterations"] > 0) == (case != "spherical_two_planes")      # (one cell in phi under a uniform boundary value: no unknowns)
    assert np.array_equal(outs[0].values, outs[1].values)


@pytest.mark.parametrize("case", ["spherical", "cylinder"])
@pytest.mark.parametrize("wfast", [0, 1])
def test_weighted_pcg_without_reduce_launches_is_bitwise_the_eager_one(lib, gpu_ctx, case, wfast):
    """lazy_dot = 1 (default): inside a chunk K2's prologue sums K1's p.q partials and the next K1's prologue K2's r.z partials — no reduce launches, the
    same summation order; the weighted Jacobi-PCG (wfast 0: class-table K1 with per-node coefficients, k_k2_pc; 1: the marching kernels) must give the
    fields of the eager loop bit for bit"""
    from pdegpu import _lib, curvilinear as cv, heat
    outs = []
    try:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast", wfast))
        for lazy in (1, 0):
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"lazy_dot", lazy))
            if case == "spherical":
                f = cv._solve_heat_3d_spherical_raw(0.1, 1.0, 10, 9, 11, 1.0, 20.0, 60.0, 0.01, 3, source_type="constant", source_value=10.0, rtol=1e-12)
            else:
                f = heat._solve_heat_3d_raw(Lx=1.0, Ly=1.0, Lz=1.0, nx=9, ny=8, nz=7, diffusivity=1.0, T_boundary=20.0, T_initial=60.0, dt=0.01, num_steps=3,
                                            geometry_type="cylinder", cylinder_radius=0.5, core_radius=0.3, core_diffusivity=40.0, rtol=1e-12)
            outs.append(f)
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"lazy_dot", 1))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"wfast", 1))
    assert outs[0].meta["cg_iterations"] == outs[1].meta["cg_iterations"] > 0
    assert np.array_equal(outs[0].values, outs[1].values)
