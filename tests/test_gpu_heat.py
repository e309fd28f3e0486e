"""GPU parity tests of the heat path: HIP kernels (through the C-ABI) vs the CPU oracle on the same inputs.
Floating point (fp64): the stated tolerance is 1e-9 relative to max|u| with CG rtol 1e-12;
operator applications are compared at 1e-13 relative."""
import ctypes as C

import numpy as np
import pytest

from oracle import fem_p1 as o

pytestmark = pytest.mark.gpu

TOL = 1e-9


@pytest.fixture(params=["tile", "march"])
def cg_kernel_form(request, lib, gpu_ctx):
    """small grids run their two CG kernels as one-round tiles (k_tile3d / k_plane2d) by default; "march" switches both off so
    that the z-marching K1 / K2 (the kernels of the large grids) are exercised on the same small problems"""
    from pdegpu import _lib
    if request.param == "march":
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"tile3d_max_nodes", 0))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"plane2d", 0))
    yield request.param
    _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"tile3d_max_nodes", 3000000))
    _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"plane2d", 1))


def rel_err(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def _mesh(dim, n, Ls):
    if dim == 1:
        return o.interval_mesh(n[0], 0.0, Ls[0])
    if dim == 2:
        return o.rectangle_mesh(Ls[0], Ls[1], n[0], n[1])
    return o.box_mesh(*Ls, *n)


def test_device_name_is_gfx950(lib, gpu_ctx):
    buf = C.create_string_buffer(256)
    assert lib.pdegpu_device_name(gpu_ctx, buf, 256) == 0
    assert b"gfx950" in buf.value, buf.value


@pytest.mark.parametrize("dim,n,Ls", [(1, (9,), (2.0,)), (2, (6, 5), (1.0, 0.5)), (3, (5, 4, 3), (1.0, 0.6, 0.25))])
def test_device_tables_equal_host_twin(lib, gpu_ctx, dim, n, Ls):
    from pdegpu import _lib
    for em in (_lib.ELEM_P1SPLIT, _lib.ELEM_Q1):
        g = _lib.make_grid(dim, n, Ls, em)
        dev = [np.zeros(729), np.zeros(729), np.zeros(27)]
        host = [np.zeros(729), np.zeros(729), np.zeros(27)]
        _lib.check(gpu_ctx, lib.pdegpu_heat_tables(gpu_ctx, g, *[_lib.dptr(a) for a in dev]))
        assert lib.pdegpu_heat_tables_host(g, *[_lib.dptr(a) for a in host]) == 0
        for d, h in zip(dev, host):
            assert np.abs(d - h).max() <= 1e-14 * np.abs(h).max()


@pytest.mark.parametrize("dim,n,Ls", [(1, (37,), (2.0,)), (2, (19, 23), (1.0, 0.5)), (3, (9, 7, 6), (1.0, 0.6, 0.25)),
                                      (3, (130, 5, 4), (2.0, 0.3, 0.2))])
@pytest.mark.parametrize("cM,cK", [(0.0, 1.0), (1.0, 0.0), (1.0, 0.37)])
def test_general_apply_matches_assembled_matrix(lib, gpu_ctx, dim, n, Ls, cM, cK):
    """full-grid operator (natural boundaries everywhere) == scipy CSR SpMV of the oracle's matrices"""
    from pdegpu import _lib
    c, cells = _mesh(dim, n, Ls)
    K, M, load = o.assemble_scalar(c, cells)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, c.shape[0])
    y, ld = np.empty_like(x), np.empty_like(x)
    g = _lib.make_grid(dim, n, Ls)
    _lib.check(gpu_ctx, lib.pdegpu_heat_apply(gpu_ctx, g, cM, cK, _lib.dptr(x), _lib.dptr(y), _lib.dptr(ld), 0))
    ref = cM * (M @ x) + cK * (K @ x)
    assert rel_err(y, ref) < 1e-13
    assert rel_err(ld, load) < 1e-13


@pytest.mark.parametrize("n,Ls", [((9, 7, 6), (1.0, 0.6, 0.25)), ((130, 9, 5), (2.0, 0.3, 0.2)), ((260, 6, 11), (1.0, 1.0, 1.0)),
                                  ((16, 16, 40), (1.0, 1.0, 1.0)), ((2, 2, 2), (1.0, 1.0, 1.0))])
@pytest.mark.parametrize("cM,cK", [(0.0, 1.0), (1.0, 0.0), (1.0, 0.37)])
def test_fast_apply_matches_assembled_matrix_on_free_rows(lib, gpu_ctx, n, Ls, cM, cK):
    """fused z-marching kernel (7-point when cM == 0, 15-point otherwise; DPP x-shifts, tile edges, partial
    tiles in x / y / z) == assembled matrix on the free rows, exactly 0 on the Dirichlet rows"""
    from pdegpu import _lib
    c, cells = o.box_mesh(*Ls, *n)
    K, M, _ = o.assemble_scalar(c, cells)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, c.shape[0])
    y = np.empty_like(x)
    g = _lib.make_grid(3, n, Ls)
    _lib.check(gpu_ctx, lib.pdegpu_heat_apply(gpu_ctx, g, cM, cK, _lib.dptr(x), _lib.dptr(y), None, 1))
    mask, _ = o.box_dirichlet(*n, T_boundary=0.0)
    ref = cM * (M @ x) + cK * (K @ x)
    ref[mask] = 0.0
    assert np.all(y[mask] == 0.0)
    assert rel_err(y, ref) < 1e-13


def test_heat_1d_config1_matches_oracle(lib, gpu_ctx):
    """BASELINE config 1 kwargs, shortened to 25 steps for the GPU suite; plus the
    known steady answers"""
    from pdegpu import heat
    got = heat._solve_heat_1d_raw(2.0, 1000, 1.0, 20.0, 0.0, 0.0, 0.01, 25, rtol=1e-12)
    ref = o.solve_heat_1d(2.0, 1000, 1.0, 20.0, 0.0, 0.0, 0.01, 25)
    assert got.values.shape == (26, 1001) and np.allclose(got.times, ref.times)
    assert np.allclose(got.coords, ref.coords, atol=1e-15)
    assert rel_err(got.values, ref.values) < TOL
    st = heat._solve_heat_1d_raw(2.0, 64, 0.7, 5.0, -1.0, 0.0, 0.01, 3, steady=True, source_type="constant",
                                 source_value=3.0, rtol=1e-13)
    x = st.coords[:, 0]
    assert np.abs(st.values[0] - (5.0 - 3.0 * x + 3.0 * x * (2.0 - x) / 1.4)).max() < 1e-9
    assert st.times.tolist() == [0.0] and st.meta["steady"] is True


def test_heat_1d_direct_tridiagonal_path(lib, gpu_ctx, monkeypatch):
    """1-D grids run the Thomas recurrence inside one launch (csrc/tridiag.hip): same fields as the CG path
    (PDEGPU_DIRECT_1D=0) and the oracle; grids longer than one LDS chunk (1024 nodes); snapshot stride; the config-1
    known answers of SURVEY §8c(iii) at full length"""
    from pdegpu import heat, _lib
    full = heat._solve_heat_1d_raw(2.0, 1000, 1.0, 20.0, 0.0, 0.0, 0.01, 200)
    assert full.meta["solver_path"] == "tridiagonal-direct" and full.meta["cg_iterations"] == 0
    assert full.values.shape == (201, 1001)
    for (t, i), want in {(200, 250): 14.931253742, (200, 500): 9.902778170, (200, 750): 4.931253827, (50, 500): 6.236415842}.items():
        assert abs(full.values[t, i] - want) < 5e-9
    # longer than four chunks, trigonometric IC (project -> 1-D mass solve, also direct), source term
    kw = dict(source_type="constant", source_value=2.5, initial_type="cosine", initial_amplitude=3.0, initial_wavenumber=2.0)
    got = heat._solve_heat_1d_raw(1.5, 4500, 0.3, 1.0, -2.0, 0.0, 0.02, 6, **kw)
    ref = o.solve_heat_1d(1.5, 4500, 0.3, 1.0, -2.0, 0.0, 0.02, 6, **kw)
    assert rel_err(got.values, ref.values) < TOL
    # the CG path on the same problem
    ctx = _lib.context(0)
    _lib.check(ctx, lib.pdegpu_set_tunable(ctx, b"direct_1d", 0))
    try:
        cg = heat._solve_heat_1d_raw(1.5, 4500, 0.3, 1.0, -2.0, 0.0, 0.02, 6, rtol=1e-13, **kw)
    finally:
        _lib.check(ctx, lib.pdegpu_set_tunable(ctx, b"direct_1d", 1))
    assert cg.meta["solver_path"] == "general" and cg.meta["cg_iterations"] > 0
    assert rel_err(got.values, cg.values) < TOL
    # stride 4 over 10 steps: snapshots 0, 4, 8, 10
    monkeypatch.setenv("PDEGPU_SNAPSHOT_STRIDE", "4")
    st = heat._solve_heat_1d_raw(2.0, 50, 1.0, 20.0, 0.0, 5.0, 0.01, 10)
    monkeypatch.delenv("PDEGPU_SNAPSHOT_STRIDE")
    every = heat._solve_heat_1d_raw(2.0, 50, 1.0, 20.0, 0.0, 5.0, 0.01, 10)
    assert np.allclose(st.times, [0.0, 0.04, 0.08, 0.1]) and np.array_equal(st.values, every.values[[0, 4, 8, 10]])


@pytest.mark.parametrize("steady", [False, True])
def test_heat_2d_matches_oracle(lib, gpu_ctx, steady, cg_kernel_form):
    from pdegpu import heat
    kw = dict(Lx=1.0, Ly=0.5, nx=24, ny=17, diffusivity=0.8, T_boundary=3.0, T_initial=20.0, dt=0.004, num_steps=4,
              steady=steady, source_type="constant", source_value=5.0)
    got = heat._solve_heat_2d_raw(**kw, rtol=1e-12)
    ref = o.solve_heat_2d(**kw)
    assert got.values.shape == ref.values.shape
    assert np.allclose(got.coords, ref.coords, atol=1e-15)
    assert rel_err(got.values, ref.values) < TOL
    assert set(ref.meta) <= set(got.meta)


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("steady", [False, True])
def test_heat_3d_uniform_bc_matches_oracle(lib, gpu_ctx, fast, steady, cg_kernel_form):
    from pdegpu import heat
    kw = dict(Lx=1.0, Ly=0.8, Lz=0.5, nx=14, ny=11, nz=9, diffusivity=1.2, T_boundary=2.0, T_initial=20.0, dt=0.003,
              num_steps=3, steady=steady, source_type="constant", source_value=4.0)
    got = heat._solve_heat_3d_raw(**kw, rtol=1e-12, fast_path=fast)
    ref = o.solve_heat_3d(**kw)
    assert got.meta["solver_path"] == ("fast" if fast else "general")
    assert np.allclose(got.coords, ref.coords, atol=1e-15)
    assert rel_err(got.values, ref.values) < TOL
    for k in ("name", "unit", "pde", "coordinate_system", "Lx", "Ly", "Lz", "geometry_type", "T_boundary", "diffusivity"):
        assert got.meta[k] == ref.meta[k]


def test_heat_3d_wide_grid_fast_path(lib, gpu_ctx, cg_kernel_form):
    """several 128-node x-tiles and a ragged last tile; two z-chunks"""
    from pdegpu import heat
    kw = dict(Lx=2.0, Ly=0.3, Lz=0.4, nx=150, ny=6, nz=20, diffusivity=1.0, T_boundary=0.0, T_initial=20.0, dt=0.002,
              num_steps=2)
    got = heat._solve_heat_3d_raw(**kw, rtol=1e-12)
    ref = o.solve_heat_3d(**kw)
    assert got.meta["solver_path"] == "fast"
    assert rel_err(got.values, ref.values) < TOL


@pytest.mark.parametrize("bcs", [dict(T_left=10.0, T_right=1.0, T_side=None), dict(T_left=10.0, T_right=None, T_side=3.0),
                                 dict(T_left=None, T_right=4.0, T_side=None), dict(T_left=5.0, T_right=6.0, T_side=7.0),
                                 dict(T_left=None, T_right=None, T_side=2.5)])      # both x-faces natural: both computed by the marching kernel
def test_heat_3d_directional_bc_natural_faces(lib, gpu_ctx, bcs):
    """faces left natural use the truncated class stencils (fenics_mcp_server.py:606-623)"""
    from pdegpu import heat
    kw = dict(Lx=1.0, Ly=0.5, Lz=0.4, nx=9, ny=6, nz=5, diffusivity=1.0, T_boundary=0.0, T_initial=2.0, dt=0.01,
              num_steps=3, source_type="constant", source_value=1.5, **bcs)
    got = heat._solve_heat_3d_raw(**kw, rtol=1e-12)
    ref = o.solve_heat_3d(**kw)
    assert got.meta["solver_path"] == "fast+boundary"      # marching kernel (interior) + class-table kernel (boundary nodes)
    assert rel_err(got.values, ref.values) < TOL
    gen = heat._solve_heat_3d_raw(**kw, rtol=1e-12, fast_path=False)
    assert gen.meta["solver_path"] == "general" and rel_err(gen.values, ref.values) < TOL
    for k, v in bcs.items():
        assert (k in got.meta) == (v is not None)
    assert "T_boundary" not in got.meta


@pytest.mark.parametrize("nx", [128, 129, 130, 257])
@pytest.mark.parametrize("steady", [False, True])
def test_hybrid_path_tile_edges_and_modes(lib, gpu_ctx, nx, steady):
    """hybrid (natural faces) runs vs the class-table kernels on grids whose last x-node falls on / next to a 128-node tile
    boundary (the marching kernel covers x = NX-1 only when a tile reaches it), steady (7-point) and transient (15-point)"""
    from pdegpu import heat
    kw = dict(Lx=2.0, Ly=0.5, Lz=0.4, nx=nx, ny=5, nz=4, diffusivity=0.7, T_boundary=0.0, T_initial=2.0, dt=0.02,
              num_steps=2, source_type="constant", source_value=1.5, T_left=10.0, T_right=None, T_side=None, steady=steady)
    a = heat._solve_heat_3d_raw(**kw, rtol=1e-12)
    b = heat._solve_heat_3d_raw(**kw, rtol=1e-12, fast_path=False)
    assert a.meta["solver_path"] == "fast+boundary" and b.meta["solver_path"] == "general"
    assert rel_err(a.values, b.values) < TOL
    if nx <= 130:
        assert rel_err(a.values, o.solve_heat_3d(**kw).values) < TOL


@pytest.mark.parametrize("nx", [20, 128, 129, 200])
def test_hybrid_64_byte_variant_matches_stored_q_variant(lib, gpu_ctx, nx):
    """steady natural-face runs: K2 applying the operator again (marching + boundary launch, r updated in place; hybrid_k2 = 1) vs the
    form that stores q (hybrid_k2 = 0), faces free on one or both x-sides, with and without the x-face rows in the marching kernel
    (without them the 64-byte form must switch itself off), and vs the oracle"""
    from pdegpu import heat, _lib
    for bcs in (dict(T_left=3.0, T_right=None, T_side=None), dict(T_left=None, T_right=None, T_side=1.0), dict(T_left=1.0, T_right=2.0, T_side=None)):
        kw = dict(Lx=1.5, Ly=0.5, Lz=0.45, nx=nx, ny=7, nz=6, diffusivity=0.8, T_boundary=0.0, T_initial=2.0, dt=0.02, num_steps=1,
                  source_type="constant", source_value=1.2, steady=True, **bcs)
        runs = {}
        try:
            for k2s, fx in ((1, 1), (0, 1), (1, 0)):
                _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"hybrid_k2", k2s))
                _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"hybrid_fx", fx))
                runs[(k2s, fx)] = heat._solve_heat_3d_raw(**kw, rtol=1e-12)
        finally:
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"hybrid_k2", 1))
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"hybrid_fx", 1))
        a = runs[(1, 1)]
        assert a.meta["solver_path"] == "fast+boundary"
        for r in runs.values():
            assert rel_err(a.values, r.values) < 1e-9
        if nx <= 129:
            assert rel_err(a.values, o.solve_heat_3d(**kw).values) < TOL


@pytest.mark.parametrize("nx", [127, 128, 200])
@pytest.mark.parametrize("steady", [False, True])
def test_hybrid_x_faces_in_the_marching_kernel_q1_and_p1(lib, gpu_ctx, nx, steady):
    """the x-face rows computed by the edge lanes of the marching kernel (halved in-face terms + inward coupling; 7-, 15- and
    27-point forms) vs the class-table kernel everywhere (hybrid_fx = 0 and fast_path = 0), both element modes, faces free on
    one or both sides; hybrid_fx = 1 and 0 must agree to round-off"""
    from pdegpu import heat, _lib
    for em in (_lib.ELEM_P1SPLIT, _lib.ELEM_Q1):
        for bcs in (dict(T_left=3.0, T_right=None, T_side=None), dict(T_left=None, T_right=None, T_side=1.0)):
            kw = dict(Lx=1.5, Ly=0.5, Lz=0.45, nx=nx, ny=6, nz=5, diffusivity=0.8, T_boundary=0.0, T_initial=2.0, dt=0.02, num_steps=2,
                      source_type="constant", source_value=1.2, steady=steady, element_mode=em, **bcs)
            a = heat._solve_heat_3d_raw(**kw, rtol=1e-12)
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"hybrid_fx", 0))
            try:
                b = heat._solve_heat_3d_raw(**kw, rtol=1e-12)
            finally:
                _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"hybrid_fx", 1))
            g = heat._solve_heat_3d_raw(**kw, rtol=1e-12, fast_path=False)
            assert a.meta["solver_path"] == "fast+boundary" and g.meta["solver_path"] == "general"
            assert rel_err(a.values, g.values) < TOL and rel_err(a.values, b.values) < 1e-10
            if em == _lib.ELEM_P1SPLIT and nx <= 128:
                ref = o.solve_heat_3d(**{k: v for k, v in kw.items() if k != "element_mode"})
                assert rel_err(a.values, ref.values) < TOL


@pytest.mark.parametrize("dim", [1, 2, 3])
@pytest.mark.parametrize("kind", ["sine", "cosine"])
def test_projected_trig_initial_condition(lib, gpu_ctx, dim, kind):
    """initial_type sine/cosine = project(Expression(degree=2), V) then bc.apply (:278-297)"""
    from pdegpu import heat
    common = dict(dt=0.005, num_steps=2, initial_type=kind, initial_amplitude=2.0, initial_wavenumber=3.0)
    if dim == 1:
        got = heat._solve_heat_1d_raw(2.0, 40, 1.0, 1.0, 0.5, 0.0, rtol=1e-12, **common)
        ref = o.solve_heat_1d(2.0, 40, 1.0, 1.0, 0.5, 0.0, **common)
    elif dim == 2:
        got = heat._solve_heat_2d_raw(1.0, 1.5, 12, 10, 1.0, 0.25, 0.0, rtol=1e-12, **common)
        ref = o.solve_heat_2d(1.0, 1.5, 12, 10, 1.0, 0.25, 0.0, **common)
    else:
        got = heat._solve_heat_3d_raw(1.0, 1.5, 0.5, 8, 7, 6, 1.0, 0.25, 0.0, rtol=1e-12, **common)
        ref = o.solve_heat_3d(1.0, 1.5, 0.5, 8, 7, 6, 1.0, 0.25, 0.0, **common)
    assert rel_err(got.values[0], ref.values[0]) < TOL          # the projected IC itself
    assert rel_err(got.values, ref.values) < TOL


def test_trivial_and_edge_cases(lib, gpu_ctx):
    from pdegpu import heat, _lib
    # steady, all-Dirichlet, no source: u == T_b
    f = heat._solve_heat_3d_raw(1, 1, 1, 8, 8, 8, 1.0, 7.5, 0.0, 0.1, 1, steady=True, rtol=1e-12)
    assert np.abs(f.values[0] - 7.5).max() < 1e-9
    # a transient step from the exact steady state needs no CG iteration at all
    f = heat._solve_heat_3d_raw(1, 1, 1, 8, 8, 8, 1.0, 7.5, 7.5, 0.1, 2)
    assert np.abs(f.values - 7.5).max() < 1e-12 and f.meta["cg_iterations"] <= 2
    # smallest grids: one cell per axis has no free node
    f = heat._solve_heat_3d_raw(1, 1, 1, 1, 1, 1, 1.0, 2.0, 9.0, 0.1, 2)
    assert f.values.shape == (3, 8) and np.all(f.values == 2.0)
    f = heat._solve_heat_1d_raw(1.0, 1, 1.0, 3.0, 4.0, 0.0, 0.1, 2)
    assert np.array_equal(f.values, [[3.0, 4.0]] * 3)
    # num_steps = 0: only the initial state
    f = heat._solve_heat_2d_raw(1, 1, 4, 4, 1.0, 1.0, 5.0, 0.1, 0)
    assert f.values.shape == (1, 25) and f.times.tolist() == [0.0]
    # snapshot stride keeps state 0, every stride-th and the last state
    full = heat._solve_heat_3d_raw(1, 1, 1, 6, 6, 6, 1.0, 0.0, 20.0, 0.01, 5, rtol=1e-12)
    part = heat._solve_heat_3d_raw(1, 1, 1, 6, 6, 6, 1.0, 0.0, 20.0, 0.01, 5, rtol=1e-12, snapshot_stride=2)
    assert np.allclose(part.times, [0.0, 0.02, 0.04, 0.05])
    assert np.array_equal(part.values, full.values[[0, 2, 4, 5]])
    # invalid arguments come back as errors, not crashes
    with pytest.raises(_lib.PdeGpuError):
        heat._solve_heat_3d_raw(1, 1, 1, 0, 4, 4, 1.0, 0.0, 20.0, 0.01, 1)
    with pytest.raises(_lib.PdeGpuError):
        heat._solve_heat_3d_raw(1, 1, 1, 4, 4, 4, -1.0, 0.0, 20.0, 0.01, 1)
    with pytest.raises(_lib.PdeGpuError):                          # composite core: the ratio core / base diffusivity must be positive
        heat._solve_heat_3d_raw(1, 1, 1, 4, 4, 4, 1.0, 0.0, 20.0, 0.01, 1, core_radius=0.2, core_diffusivity=-5.0)
    # the stride taken from the environment is the one meta reports
    import os
    os.environ["PDEGPU_SNAPSHOT_STRIDE"] = "3"
    try:
        env = heat._solve_heat_3d_raw(1, 1, 1, 6, 6, 6, 1.0, 0.0, 20.0, 0.01, 5, rtol=1e-12)
    finally:
        del os.environ["PDEGPU_SNAPSHOT_STRIDE"]
    assert env.meta["snapshot_stride"] == 3 and np.allclose(env.times, [0.0, 0.03, 0.05]) and full.meta["snapshot_stride"] == 1
    # the output budget: 6 states of 343 nodes do not fit 10 kB -> strided automatically, recorded in meta; or refused
    os.environ["PDEGPU_MAX_OUTPUT_GB"] = "1e-5"
    try:
        auto = heat._solve_heat_3d_raw(1, 1, 1, 6, 6, 6, 1.0, 0.0, 20.0, 0.01, 5, rtol=1e-12)
        assert auto.meta["snapshot_stride"] > 1 and auto.meta["snapshot_stride_requested"] == 1 and "auto-strided" in auto.meta["snapshot_policy"]
        assert auto.values.shape[0] * 343 * 8 <= 1e4 and np.array_equal(auto.values[-1], full.values[-1]) and np.array_equal(auto.values[0], full.values[0])
        os.environ["PDEGPU_OUTPUT_POLICY"] = "error"
        with pytest.raises(_lib.PdeGpuError, match="snapshot_stride"):
            heat._solve_heat_3d_raw(1, 1, 1, 6, 6, 6, 1.0, 0.0, 20.0, 0.01, 5, rtol=1e-12)
    finally:
        del os.environ["PDEGPU_MAX_OUTPUT_GB"]
        os.environ.pop("PDEGPU_OUTPUT_POLICY", None)


def test_q1_mode_differs_by_h2(lib, gpu_ctx):
    """Q1 element mode (BASELINE.json wording) vs the reference's P1 split: O(h^2) apart, not equal"""
    from pdegpu import heat, _lib
    errs = []
    for n in (8, 16):
        kw = dict(Lx=1.0, Ly=1.0, Lz=1.0, nx=n, ny=n, nz=n, diffusivity=1.0, T_boundary=0.0, T_initial=20.0, dt=0.01,
                  num_steps=5)
        q1 = heat._solve_heat_3d_raw(**kw, rtol=1e-12, element_mode=_lib.ELEM_Q1)
        ref = o.solve_heat_3d(**kw)
        assert q1.meta["element_mode"] == "q1" and q1.meta["solver_path"] == "fast"        # 27-point marching kernel
        errs.append(rel_err(q1.values[-1], ref.values[-1]))
    assert 1e-4 < errs[1] < errs[0] < 0.1
    assert errs[0] / errs[1] > 2.5


def test_fast_and_general_sessions_agree(lib, gpu_ctx):
    """same CG iterates from the fused z-marching kernels and from the class-table kernels"""
    from pdegpu import _lib
    grid = _lib.make_grid(3, (40, 21, 18), (1.0, 1.0, 1.0))
    out = []
    for fast in (1, 0):
        h = _lib.Heat()
        h.kappa, h.dt, h.steady, h.source = 1.0, 1.0, 1, 1.0
        h.bc_mode, h.fast_path = _lib.BC_UNIFORM, fast
        h.bc_present[0] = 1
        s = C.c_void_p()
        _lib.check(gpu_ctx, lib.pdegpu_session_create(gpu_ctx, grid, h, 0, 19, C.byref(s)))
        try:
            _lib.check(gpu_ctx, lib.pdegpu_session_begin(s, 0.0))
            _lib.check(gpu_ctx, lib.pdegpu_session_residual(s))
            _lib.check(gpu_ctx, lib.pdegpu_session_reduce_local(s, 1, 1))
            st = _lib.Stats()
            _lib.check(gpu_ctx, lib.pdegpu_session_iterate(s, 12, 1, st))
            rr = C.c_double()
            _lib.check(gpu_ctx, lib.pdegpu_session_rr(s, C.byref(rr)))
            _lib.check(gpu_ctx, lib.pdegpu_session_flush(s))
            x = np.empty(41 * 22 * 19)
            _lib.check(gpu_ctx, lib.pdegpu_session_get_x(s, _lib.dptr(x)))
            assert st.kernel_ms[0] > 0 and st.kernel_ms[1] > 0 and st.kernel_launches[0] == 12
            assert st.path == fast
            out.append((rr.value, x))
        finally:
            lib.pdegpu_session_destroy(s)
    assert out[0][0] == pytest.approx(out[1][0], rel=1e-10)
    assert rel_err(out[0][1], out[1][1]) < 1e-11


@pytest.mark.parametrize("fast,variant", [(1, 1), (1, 0), (0, 0), (2, 1)])
@pytest.mark.parametrize("nslabs", [2, 3])
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("sr", [False, True])
def test_slabs_on_one_gpu_match_single_slab(lib, gpu_ctx, fast, variant, nslabs, split, sr):
    _slab_case(lib, fast, variant, nslabs, split, sr)


@pytest.mark.parametrize("nslabs", [6, 14])
@pytest.mark.parametrize("sr", [False, True])
def test_thin_slabs_of_one_to_three_planes(lib, gpu_ctx, nslabs, sr):
    """14 planes over 6 (3, 3, 2, 2, 2, 2 planes) and 14 slabs (ONE plane each: a plane next to both ghosts): K1 and K2 in two parts where one part is
    empty, the same plane the first and the last of its slab"""
    _slab_case(lib, 1, 1, nslabs, True, sr)


def _slab_case(lib, fast, variant, nslabs, split, sr):
    """slab sessions (k0 > 0, ghost planes, global boundary classes) with the halo exchange / all-reduce done by
    plain device copies on ONE GPU: same order of operations as pdegpu.dist.SlabCG.iterate, no RCCL needed.
    Must reproduce the single-slab CG iterates and the oracle's solution.  sr: the single-reduction recurrences — ONE message
    {p.Ap, r.r, Ap.Ap} per iteration, summed here over the slabs — against the classic single-slab run."""
    import torch
    from pdegpu import _lib
    from pdegpu.dist import GpuSlabEngine, slab_range, VEC_R, VEC_P, SC_PQ, SC_RR, SC_QQ, SC_RR_POLL
    n, Ls = (20, 9, 13), (1.0, 0.7, 1.2)
    grid = _lib.make_grid(3, n, Ls)
    h = _lib.Heat()
    h.kappa, h.steady, h.source = 1.1, 1, 3.0
    h.bc_mode, h.fast_path = _lib.BC_UNIFORM, fast
    h.bc_present[0], h.bc_value[0] = 1, 0.25
    hybrid = fast == 2                       # natural faces except x = 0: marching kernel + boundary kernel per slab
    if hybrid:
        h.bc_mode, h.fast_path = _lib.BC_DIRECTIONAL, 1
    iters = 600 if hybrid else 150

    def solve(ns, sr=False):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            engs = []
            for r in range(ns):
                k0, nk = slab_range(n[2] + 1, ns, r)
                engs.append(GpuSlabEngine(grid, h, k0, nk, 0, stream))
                _lib.check(engs[-1].ctx, lib.pdegpu_set_tunable(engs[-1].ctx, b"cg_variant", variant))
                engs[-1].set_variant(sr)
            # the 64-B variant keeps p on the ghost planes itself: only r is exchanged
            # hybrid slabs with neighbours: q is stored, r and p travel; a hybrid session that owns every plane uses the 64-B form too
            assert engs[0].needs_p_halo() == (not (variant == 1 and (fast == 1 or (hybrid and ns == 1))))
            vecs = (VEC_R, VEC_P) if engs[0].needs_p_halo() else (VEC_R,)

            def exchange():
                for v in vecs:
                    for r in range(ns - 1):
                        lo, hi = engs[r], engs[r + 1]
                        hi.plane_tensor(v, -1).copy_(lo.plane_tensor(v, lo.nk - 1))
                        lo.plane_tensor(v, lo.nk).copy_(hi.plane_tensor(v, 0))

            def allreduce(which, slots=None):
                a, b = slots if slots else ((SC_PQ, SC_PQ + 1) if which == 0 else (SC_RR, SC_RR + 1))
                tot = sum(e.scalars()[a:b].clone() for e in engs)
                for e in engs:
                    e.scalars()[a:b].copy_(tot)

            def rr_now():
                if sr:                                   # the r.r of the last K2 sits in its partials: sum, "all-reduce" scalars[6], read
                    for e in engs:
                        e.sr_poll()
                    allreduce(1, (SC_RR_POLL, SC_RR_POLL + 1))
                return engs[0].rr()

            for e in engs:
                e.begin(0.0)
                e.residual()
            allreduce(1)
            hist = [rr_now()]
            for it in range(iters):
                if split:
                    # the overlapped order of SlabCG.iterate: the bulk of K1 runs BEFORE the ghost planes arrive
                    assert all(e.can_split_k1() == (fast == 1) for e in engs)
                    for e in engs:
                        e.k1_part(0)
                    exchange()
                    for e in engs:
                        e.k1_part(1)
                else:
                    exchange()
                    for e in engs:
                        e.k1()
                allreduce(0, (SC_PQ, SC_QQ + 1) if sr else None)      # sr: {p.Ap, r.r, Ap.Ap} in ONE message
                if split:
                    # K2 in two parts (the next exchange is posted between them): after part 0 the planes that travel are final
                    assert all(e.can_split_k2() == (fast == 1 and variant == 1) for e in engs)
                    for e in engs:
                        e.k2_part(0)
                    sent = [(e, w, e.plane_tensor(VEC_R, w).clone()) for e in engs
                            for w in ((0,) if e.k0 > 0 else ()) + ((e.nk - 1,) if e.k0 + e.nk < n[2] + 1 else ())] if it % 25 == 0 else []
                    for e in engs:
                        e.k2_part(1)
                    assert all(torch.equal(t, e.plane_tensor(VEC_R, w)) for e, w, t in sent)
                else:
                    for e in engs:
                        e.k2()
                if not sr:
                    allreduce(1)
                if it < 6:
                    hist.append(rr_now())
            xs = []
            for e in engs:
                e.flush()
                x = np.empty(e.nk * (n[1] + 1) * (n[0] + 1))
                _lib.check(e.ctx, lib.pdegpu_session_get_x(e.s, _lib.dptr(x)))
                xs.append(x)
                assert e.lib.pdegpu_session_plane_doubles(e.s) == e.plane
            for e in engs:
                e.close()
        return np.concatenate(xs), np.array(hist)

    split, keep = False, split
    x1, h1 = solve(1)
    split = keep
    xs, hs = solve(nslabs, sr)
    ref = o.solve_heat_3d(*Ls, *n, 1.1, 0.25, 0.0, 0.1, 1, steady=True, source_type="constant", source_value=3.0,
                          **(dict(T_left=0.25) if hybrid else {})).values[0]
    assert np.allclose(hs, h1, rtol=1e-10)
    assert rel_err(x1, ref) < TOL and rel_err(xs, ref) < TOL
    assert rel_err(xs, x1) < 1e-11


def test_rccl_backend_on_library_owned_memory(lib, gpu_ctx):
    """one-rank `nccl` (= RCCL) process group on this GPU: the all-reduce of the device scalars and a plane
    send/recv-to-self run on zero-copy tensor views of library-owned memory — the exact objects the multi-GPU
    driver hands to torch.distributed (world sizes > 1 need more GPUs than the test box has)."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from pdegpu import _lib
    from pdegpu.dist import GpuSlabEngine, SlabCG, SC_RR
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        grid = _lib.make_grid(3, (24, 10, 12), (1.0, 1.0, 1.0))
        h = _lib.Heat()
        h.kappa, h.steady, h.source = 1.0, 1, 1.0
        h.bc_mode, h.fast_path = _lib.BC_UNIFORM, 1
        h.bc_present[0] = 1
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            eng = GpuSlabEngine(grid, h, 0, 13, 0, stream)
            cg = SlabCG(eng, 0, 1)
            cg.start(0.0)
            rr0 = eng.rr()
            t = eng.scalars()[SC_RR:SC_RR + 1]
            dist.all_reduce(t, op=dist.ReduceOp.SUM)               # in place on library memory
            torch.cuda.synchronize()
            assert eng.rr() == rr0
            src, dst = eng.plane_tensor(1, 3), eng.plane_tensor(3, 5)
            ops = [dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            torch.cuda.synchronize()
            assert torch.equal(src, dst) and float(src.abs().sum()) > 0
            cg.iterate(20)
            assert eng.rr() < rr0
            eng.close()
    finally:
        dist.destroy_process_group()


def test_two_slabs_exchanging_through_rccl_in_the_overlapped_order(lib, gpu_ctx):
    """The closest thing to N = 2 over RCCL one GPU allows: two slab sessions in ONE process whose ghost planes travel as `nccl` point-to-point
    messages (sends to self, one batch per exchange = one ncclGroup) in exactly the order SlabCG.iterate uses — exchange posted, K1 on the planes
    away from the ghosts while it is in flight, wait (stream-ordered), the edge planes, ONE message {p.Ap, r.r, Ap.Ap} through dist.all_reduce, K2 —
    on zero-copy views of library-owned memory.  A receive that landed late, or early into planes part 0 still reads, would change the iterates:
    the residual history must be the single slab's."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from pdegpu import _lib
    from pdegpu.dist import GpuSlabEngine, SlabCG, slab_range, VEC_R, SC_PQ, SC_QQ, SC_RR, SC_RR_POLL
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        n, Ls = (140, 33, 27), (1.0, 0.7, 1.2)
        grid = _lib.make_grid(3, n, Ls)
        h = _lib.Heat()
        h.kappa, h.steady, h.source = 1.1, 1, 3.0
        h.bc_mode, h.fast_path = _lib.BC_UNIFORM, 1
        h.bc_present[0], h.bc_value[0] = 1, 0.25
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            one = GpuSlabEngine(grid, h, 0, n[2] + 1, 0, stream)
            c1 = SlabCG(one, 0, 1, variant="classic")
            c1.start(0.0)
            ref = [c1.rr()]
            for _ in range(8):
                c1.iterate(5)
                ref.append(c1.rr())
            one.close()
            engs = [GpuSlabEngine(grid, h, *slab_range(n[2] + 1, 2, r), 0, stream) for r in range(2)]
            lo, hi = engs
            for e in engs:
                e.set_variant(True)
                assert e.can_split_k1() and not e.needs_p_halo()
                e.begin(0.0)
                e.residual()

            def sum_over_slabs(a, b):
                tot = lo.scalars()[a:b] + hi.scalars()[a:b]
                for e in engs:
                    e.scalars()[a:b].copy_(tot)
                    dist.all_reduce(e.scalars()[a:b], op=dist.ReduceOp.SUM)      # world 1: the identity, through RCCL, in place on library memory

            def rr_now():
                for e in engs:
                    e.sr_poll()
                sum_over_slabs(SC_RR_POLL, SC_RR_POLL + 1)
                return lo.rr()

            sum_over_slabs(SC_RR, SC_RR + 1)
            hist = [rr_now()]
            for it in range(40):
                ops = [dist.P2POp(dist.isend, lo.plane_tensor(VEC_R, lo.nk - 1), 0), dist.P2POp(dist.irecv, hi.plane_tensor(VEC_R, -1), 0),
                       dist.P2POp(dist.isend, hi.plane_tensor(VEC_R, 0), 0), dist.P2POp(dist.irecv, lo.plane_tensor(VEC_R, lo.nk), 0)]
                works = dist.batch_isend_irecv(ops)
                for e in engs:
                    e.k1_part(0)
                for w in works:
                    w.wait()
                for e in engs:
                    e.k1_part(1)
                sum_over_slabs(SC_PQ, SC_QQ + 1)
                for e in engs:
                    e.k2()
                if it % 5 == 4:
                    hist.append(rr_now())
            for e in engs:
                e.close()
        assert np.allclose(hist, ref, rtol=1e-9), (hist, ref)
        assert hist[-1] < 1e-2 * hist[0]
    finally:
        dist.destroy_process_group()


def test_random_shapes_fast_and_general_apply(lib, gpu_ctx):
    """seeded sweep over ragged 3-D shapes and anisotropic cells: tile edges of the z-marching kernels (x-tiles of
    128, y-tiles of 2 rows, z-chunks) and every boundary class of the general kernels vs the assembled matrices"""
    from pdegpu import _lib
    rng = np.random.default_rng(2026)
    shapes = [(int(rng.integers(2, 300)), int(rng.integers(2, 14)), int(rng.integers(2, 40))) for _ in range(10)]
    shapes += [(127, 3, 9), (128, 4, 9), (129, 5, 8), (256, 2, 17), (257, 3, 3), (3, 3, 97)]
    for n in shapes:
        Ls = tuple(float(v) for v in rng.uniform(0.2, 2.0, 3))
        c, cells = o.box_mesh(*Ls, *n)
        K, M, _ = o.assemble_scalar(c, cells)
        x = rng.uniform(-1, 1, c.shape[0])
        mask, _ = o.box_dirichlet(*n, T_boundary=0.0)
        g = _lib.make_grid(3, n, Ls)
        for cM, cK in ((0.0, 1.0), (1.0, 0.013)):
            ref = cM * (M @ x) + cK * (K @ x)
            y = np.empty_like(x)
            _lib.check(gpu_ctx, lib.pdegpu_heat_apply(gpu_ctx, g, cM, cK, _lib.dptr(x), _lib.dptr(y), None, 0))
            assert rel_err(y, ref) < 1e-13, (n, cM, "general")
            _lib.check(gpu_ctx, lib.pdegpu_heat_apply(gpu_ctx, g, cM, cK, _lib.dptr(x), _lib.dptr(y), None, 1))
            refm = ref.copy()
            refm[mask] = 0.0
            assert np.all(y[mask] == 0.0) and rel_err(y, refm) < 1e-13, (n, cM, "fast")


@pytest.mark.parametrize("n,Ls", [((19, 23), (1.0, 0.5)), ((130, 9), (2.0, 0.3)), ((260, 40), (1.0, 1.0)), ((2, 2), (1.0, 1.0)),
                                  ((1024, 64), (1.0, 0.1))])
@pytest.mark.parametrize("cM,cK", [(0.0, 1.0), (1.0, 0.0), (1.0, 0.37)])
def test_fast_2d_view_matches_assembled_matrix(lib, gpu_ctx, n, Ls, cM, cK):
    """2-D grids through the fused marching kernels (flat view: the 2-D P1 stencil is the 3-D 15-point kernel on a
    single row of the middle axis) == assembled 2-D matrices on the free rows"""
    from pdegpu import _lib
    c, cells = o.rectangle_mesh(Ls[0], Ls[1], n[0], n[1])
    K, M, _ = o.assemble_scalar(c, cells)
    x = np.random.default_rng(11).uniform(-1, 1, c.shape[0])
    y = np.empty_like(x)
    _lib.check(gpu_ctx, lib.pdegpu_heat_apply(gpu_ctx, _lib.make_grid(2, n, Ls), cM, cK, _lib.dptr(x), _lib.dptr(y), None, 1))
    idx = np.arange(c.shape[0])
    ix, iy = idx % (n[0] + 1), idx // (n[0] + 1)
    mask = (ix == 0) | (ix == n[0]) | (iy == 0) | (iy == n[1])
    ref = cM * (M @ x) + cK * (K @ x)
    ref[mask] = 0.0
    assert np.all(y[mask] == 0.0) and rel_err(y, ref) < 1e-13


def test_heat_2d_uses_the_fused_kernels(lib, gpu_ctx):
    from pdegpu import heat
    kw = dict(Lx=1.0, Ly=0.7, nx=150, ny=33, diffusivity=0.9, T_boundary=1.0, T_initial=20.0, dt=0.002, num_steps=3)
    got = heat._solve_heat_2d_raw(**kw, rtol=1e-12)
    slow = heat._solve_heat_2d_raw(**kw, rtol=1e-12, fast_path=False)
    ref = o.solve_heat_2d(**kw)
    assert got.meta["solver_path"] == "fast" and slow.meta["solver_path"] == "general"
    assert rel_err(got.values, ref.values) < TOL and rel_err(slow.values, ref.values) < TOL
    st = heat._solve_heat_2d_raw(**{**kw, "steady": True, "source_type": "constant", "source_value": 3.0}, rtol=1e-12)
    assert st.meta["solver_path"] == "fast"
    assert rel_err(st.values, o.solve_heat_2d(**{**kw, "steady": True, "source_type": "constant", "source_value": 3.0}).values) < TOL


@pytest.mark.parametrize("case", ["3d_fast", "3d_fast_72B", "3d_general", "2d_fast", "3d_transient_15pt"])
def test_lazy_dot_products_are_bitwise_the_eager_ones(lib, gpu_ctx, case, cg_kernel_form):
    """default CG loop (the next kernel's prologue sums the previous launch's partials, no reduce launches) vs
    lazy_dot=0 (k_reduce after every dot-producing kernel): same summation order, so identical iterates bit for bit"""
    from pdegpu import heat, _lib
    def run():
        if case == "2d_fast":
            return heat._solve_heat_2d_raw(1.0, 0.7, 90, 50, 0.8, 3.0, 20.0, 0.004, 3, rtol=1e-11)
        kw = dict(steady=True, source_type="constant", source_value=2.0) if case != "3d_transient_15pt" else {}
        return heat._solve_heat_3d_raw(1.0, 0.8, 0.6, 40, 21, 17, 0.9, 1.5, 20.0, 0.003, 2, rtol=1e-11,
                                       fast_path=(case != "3d_general"), **kw)
    outs = []
    try:
        if case == "3d_fast_72B":
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"cg_variant", 0))
        for lazy in (1, 0):
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"lazy_dot", lazy))
            outs.append(run())
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"lazy_dot", 1))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"cg_variant", 1))
    assert outs[0].meta["cg_iterations"] == outs[1].meta["cg_iterations"] > 0
    assert np.array_equal(outs[0].values, outs[1].values)


@pytest.mark.parametrize("case", ["3d_transient", "3d_steady", "3d_natural_faces", "2d_transient", "3d_wide"])
def test_q1_27_point_marching_kernel_matches_class_table_kernel(lib, gpu_ctx, case, cg_kernel_form):
    """Q1 element mode: the 27-point (2-D: 9-point) form of the marching kernel — two x-shifts per node, weights by
    |dx|,|dy|,|dz| — against the class-table kernel on anisotropic cells; and the bare operator against the host tables"""
    from pdegpu import heat, _lib
    q1 = dict(element_mode=_lib.ELEM_Q1, rtol=1e-12)
    if case == "2d_transient":
        run = lambda fast: heat._solve_heat_2d_raw(1.0, 0.7, 150, 40, 0.8, 3.0, 20.0, 0.004, 3, fast_path=fast, **q1)
    else:
        n = (140, 9, 7) if case == "3d_wide" else (30, 17, 12)
        kw = dict(Lx=1.0, Ly=0.8, Lz=0.3, nx=n[0], ny=n[1], nz=n[2], diffusivity=0.9, T_boundary=1.5, T_initial=20.0, dt=0.003,
                  num_steps=2, steady=(case == "3d_steady"), source_type="constant", source_value=2.0)
        if case == "3d_natural_faces":
            kw.update(T_left=10.0, T_right=None, T_side=None)
        run = lambda fast: heat._solve_heat_3d_raw(**kw, fast_path=fast, **q1)
    a, b = run(True), run(False)
    assert a.meta["solver_path"] == ("fast+boundary" if case == "3d_natural_faces" else "fast") and b.meta["solver_path"] == "general"
    assert a.meta["element_mode"] == "q1" and rel_err(a.values, b.values) < TOL


@pytest.mark.parametrize("steady", [False, True])
@pytest.mark.parametrize("em", ["p1split", "q1"])
def test_2d_tile_kernel_and_flat_view_agree(lib, gpu_ctx, steady, em):
    """2-D CG kernels: one-round tile kernel k_plane2d (default; 2 / 4 / 8 rows per tile) vs the flat view of the marching
    kernels (plane2d = 0) vs the class-table kernels, P1 (7-point incl. the (1,1) diagonal) and Q1 (9-point), steady
    (5-point, stencil K2) and transient; awkward widths around the 128-node tile"""
    from pdegpu import heat, _lib
    emode = _lib.ELEM_Q1 if em == "q1" else _lib.ELEM_P1SPLIT
    outs = []
    try:
        for nx in (129, 257, 100):
            res = []
            for cfg in (dict(plane2d=1, plane2d_ry=4), dict(plane2d=1, plane2d_ry=2), dict(plane2d=1, plane2d_ry=8), dict(plane2d=0), None):
                for k, v in (cfg or {}).items():
                    _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, k.encode(), v))
                res.append(heat._solve_heat_2d_raw(1.3, 0.7, nx, 37, 0.8, 3.0, 20.0, 0.004, 2, steady=steady, source_type="constant",
                                                   source_value=2.0, rtol=1e-12, element_mode=emode, fast_path=cfg is not None))
            assert [r.meta["solver_path"] for r in res] == ["fast"] * 4 + ["general"]
            for r in res[1:]:
                assert rel_err(r.values, res[0].values) < TOL
            outs.append(res[0])
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"plane2d", 1))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"plane2d_ry", 4))
    if em == "p1split":
        ref = o.solve_heat_2d(1.3, 0.7, 100, 37, 0.8, 3.0, 20.0, 0.004, 2, steady=steady, source_type="constant", source_value=2.0)
        assert rel_err(outs[2].values, ref.values) < TOL


@pytest.mark.parametrize("setting", [dict(fast_ry=1), dict(fast_ry=2), dict(fast_ry=4), dict(fast_ry=8), dict(fast_nt=0), dict(fast_nt=3), dict(fast_ex=0),
                                     dict(fast_wg_per_cu=2), dict(fast_wg_per_cu=6), dict(fast_wg_per_cu_k2=3), dict(grid_mult=2), dict(k2_grid_mult=8),
                                     dict(extrapolate=0), dict(hybrid_bc=0), dict(hybrid_fx=0), dict(hybrid_k2=0), dict(fast_minlen=4), dict(fast_small_mult=4),
                                     dict(cg_variant=0), dict(cg_variant=2), dict(cg_variant=0, fast_ex=0, fast_ry=4), dict(fast_pf=1), dict(fast_pf=0),
                                     dict(fast_pf=1, fast_ry=4), dict(fast_zlen=3), dict(tile3d_max_nodes=0),
                                     dict(fast_linear=0), dict(place_tune=0), dict(cg_xdefer=0), dict(cg_xdefer=0, cg_variant=2),
                                     # per-mode configuration words: ry + 16 wg + 256 ex + 512 pf
                                     dict(fast_cfg_plain=4 + 16 * 2 + 512, fast_cfg_res=2 + 256 + 512, fast_cfg_k1=4 + 16 * 3 + 256, fast_cfg_k2=1 + 16 + 512)])
def test_every_tunable_leaves_the_solution_alone(lib, gpu_ctx, setting, cg_kernel_form):
    """'results do not depend on the tunables' (include/pdegpu.h): each one, away from its default, on a 3-D steady (7-point,
    64-B variant), a 3-D transient (15-point) and a natural-face problem, against the default run"""
    from pdegpu import heat, _lib
    defaults = dict(fast_ry=0, fast_nt=-1, fast_ex=1, fast_wg_per_cu=0, fast_wg_per_cu_k2=0, grid_mult=8, k2_grid_mult=2, extrapolate=1,
                    hybrid_bc=1, hybrid_fx=1, hybrid_k2=1, fast_minlen=1, fast_small_mult=1, cg_variant=1, fast_pf=-1, fast_zlen=0, tile3d_max_nodes=3000000,
                    fast_cfg_plain=-1, fast_cfg_res=-1, fast_cfg_k1=-1, fast_cfg_k2=-1, fast_linear=1, place_tune=1, cg_xdefer=1)
    def runs():
        a = heat._solve_heat_3d_raw(1.0, 0.8, 0.6, 140, 13, 11, 0.9, 1.5, 20.0, 0.003, 1, steady=True, source_type="constant", source_value=2.0, rtol=1e-12)
        b = heat._solve_heat_3d_raw(1.0, 0.8, 0.6, 30, 17, 12, 0.9, 1.5, 20.0, 0.003, 3, rtol=1e-12)
        c = heat._solve_heat_3d_raw(1.0, 0.8, 0.6, 30, 17, 12, 0.9, 0.0, 20.0, 0.003, 2, rtol=1e-12, T_left=4.0, T_right=None, T_side=None)
        return a, b, c
    base = runs()
    try:
        for k, v in setting.items():
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, k.encode(), v))
        alt = runs()
    finally:
        for k in setting:
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, k.encode(), defaults[k]))
    for x, y in zip(alt, base):
        assert rel_err(x.values, y.values) < TOL


@pytest.mark.parametrize("n,Ls", [((2, 2), (1.0, 1.0)), ((5, 3), (1.0, 0.4)), ((130, 37), (1.0, 0.6)), ((300, 513), (0.7, 1.3)),
                                  ((1023, 260), (2.0, 0.5)), ((640, 2049), (1.0, 3.0)), ((200, 1022), (0.3, 1.5)), ((129, 1000), (0.3, 1.5))])
@pytest.mark.parametrize("single_reduction", [0, 1, 2])
def test_persistent_2d_cg_matches_the_launch_per_kernel_loop_and_the_oracle(lib, gpu_ctx, n, Ls, single_reduction):
    """the register-resident cooperative CG of the 2-D tools (persist2d.hip: bands of 1 / 2 / 4 / 8 rows, short last bands, fewer
    bands than CUs, 1- and 8-wave-wide grids, a one-node grid) against the launch-per-kernel loop (persist2d = 0) — same
    iteration counts to +-2, fields to 1e-10 — and, where the sparse LU is affordable, against the oracle"""
    from pdegpu import heat, _lib
    kw = dict(Lx=Ls[0], Ly=Ls[1], nx=n[0], ny=n[1], diffusivity=0.9, T_boundary=1.5, T_initial=20.0, dt=0.002, num_steps=2)

    def run(persist, **extra):
        # single_reduction = 1: the Chronopoulos-Gear recurrences (one grid barrier + one neighbour hand-off per iteration); 2: the same with the Chebyshev
        # polynomial preconditioner (the default; bands of <= 4 rows — the 2049-row grid runs form 1); the 260- and 1022-row grids end in a band of ONE row
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d", persist))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d_sr", min(single_reduction, 1) if persist else 0))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d_cheb", 8 if single_reduction == 2 else 0))
        try:
            return heat._solve_heat_2d_raw(**{**kw, **extra}, rtol=1e-12)
        finally:
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d", 1))
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d_sr", 1))
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d_cheb", 8))

    for extra in (dict(), dict(steady=True, source_type="constant", source_value=3.0)):
        a, b = run(1, **extra), run(0, **extra)
        assert a.meta["solver_path"] == "fast"
        # the cooperative launch stops at the first iteration with r.r <= target; the polled loop may run up to a chunk longer
        solves = 1 if extra else kw["num_steps"]
        if single_reduction == 2 and n[1] <= 1025:
            # outer iterations of the preconditioned form: every one applies the stencil 8 times and passes ONE grid barrier
            assert 0 < a.meta["cg_iterations"] <= b.meta["cg_iterations"] // 2 + 2 * solves
        else:
            assert b.meta["cg_iterations"] - 64 * solves <= a.meta["cg_iterations"] <= b.meta["cg_iterations"] + (2 + 0.02 * b.meta["cg_iterations"] * min(single_reduction, 1)) * solves
        assert rel_err(a.values, b.values) < 1e-10
        if (n[0] + 1) * (n[1] + 1) <= 200000:
            assert rel_err(a.values, o.solve_heat_2d(**{**kw, **extra}).values) < TOL


def test_persistent_2d_cg_as_an_ordinary_launch_is_bitwise_the_cooperative_one(lib, gpu_ctx):
    """persist2d_launch = 1 (profiling runs): the same kernel on the same occupancy-sized grid without the cooperative-launch API — same fields, same counts"""
    from pdegpu import heat, _lib
    kw = dict(Lx=1.0, Ly=0.6, nx=300, ny=513, diffusivity=0.9, T_boundary=1.5, T_initial=20.0, dt=0.002, num_steps=2, rtol=1e-12)
    outs = []
    try:
        for mode in (0, 1):
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d_launch", mode))
            outs.append(heat._solve_heat_2d_raw(**kw))
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d_launch", 0))
    assert outs[0].meta["cg_iterations"] == outs[1].meta["cg_iterations"] > 0 and np.array_equal(outs[0].values, outs[1].values)


def test_persistent_2d_cg_at_a_tight_tolerance_single_reduction_vs_classic(lib, gpu_ctx):
    """an ill-conditioned 2-D system (stretched cells, steady) at rtol = 1e-13: the single-reduction recurrences of the cooperative
    launch may lose positivity near the tolerance; the solve then retries with the classic recurrences instead of reporting
    non-convergence, and both forms return the same field"""
    from pdegpu import heat, _lib
    kw = dict(Lx=1.0, Ly=0.02, nx=300, ny=180, diffusivity=1.0, T_boundary=0.0, T_initial=0.0, dt=0.01, num_steps=1, steady=True,
              source_type="constant", source_value=1.0, rtol=1e-13)
    outs = []
    try:
        for sr, cheb in ((1, 0), (0, 0), (1, 8)):
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d_sr", sr))
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d_cheb", cheb))
            outs.append(heat._solve_heat_2d_raw(**kw))
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d_sr", 1))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d_cheb", 8))
    assert all(f.meta["cg_converged"] if "cg_converged" in f.meta else True for f in outs)
    assert rel_err(outs[0].values, outs[1].values) < 1e-10
    # the preconditioned form (default) at the same tolerance: same field, far fewer grid barriers
    assert rel_err(outs[2].values, outs[1].values) < 1e-10 and 0 < outs[2].meta["cg_iterations"] < outs[1].meta["cg_iterations"]
    assert abs(outs[0].meta["cg_iterations"] - outs[1].meta["cg_iterations"]) <= 0.05 * outs[1].meta["cg_iterations"] + 4


def test_persistent_2d_cg_session_iterate_leaves_the_state_of_the_stepwise_loop(lib, gpu_ctx):
    """pdegpu_session_iterate on a 2-D session: N iterations in ONE cooperative launch leave r.r, x (after flush) and the scalars
    where N launches of K1 / K2 leave them (summation orders differ: 1e-12)"""
    import ctypes as C
    from pdegpu import _lib
    grid = _lib.make_grid(2, (257, 91), (1.0, 0.4))
    h = _lib.Heat()
    h.kappa, h.dt, h.steady, h.source = 1.0, 0.0, 1, 2.0
    h.bc_mode, h.fast_path = _lib.BC_UNIFORM, 1
    h.bc_present[0], h.bc_value[0] = 1, 0.5
    out = []
    for persist in (1, 0):
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d", persist))
        s = C.c_void_p()
        _lib.check(gpu_ctx, lib.pdegpu_session_create(gpu_ctx, grid, h, 0, 1, C.byref(s)))
        try:
            _lib.check(gpu_ctx, lib.pdegpu_session_begin(s, 0.0))
            _lib.check(gpu_ctx, lib.pdegpu_session_residual(s))
            hist = []
            for chunk in (1, 2, 7, 30):
                _lib.check(gpu_ctx, lib.pdegpu_session_iterate(s, chunk, 0, None))
                rr = C.c_double()
                _lib.check(gpu_ctx, lib.pdegpu_session_rr(s, C.byref(rr)))
                hist.append(rr.value)
            _lib.check(gpu_ctx, lib.pdegpu_session_flush(s))
            x = np.empty(258 * 92)
            _lib.check(gpu_ctx, lib.pdegpu_session_get_x(s, _lib.dptr(x)))
            out.append((np.array(hist), x))
        finally:
            lib.pdegpu_session_destroy(s)
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"persist2d", 1))
    assert np.allclose(out[0][0], out[1][0], rtol=1e-10) and rel_err(out[0][1], out[1][1]) < 1e-12


@pytest.mark.parametrize("cells", [(140, 13, 11), (300, 40, 37)])
def test_solution_vector_updated_by_every_other_k2_is_the_plain_iterate(lib, gpu_ctx, cells):
    """cg_xdefer (marching sessions in the 64 B/DoF form): K1 leaves x alone; K2 updates it for two iterations at once, every other time, with the
    direction of the iteration before rebuilt from p_n = r_n + beta_n p_(n-1); flush makes up for an odd last iteration.  After n = 1 ... 9
    iterations + flush, x must be the iterate of the plain scheme (x += alpha p in the next K1) — through the native loop, the stepwise classic
    protocol and the stepwise single-reduction protocol"""
    import ctypes as C
    from pdegpu import _lib
    grid = _lib.make_grid(3, cells, (1.0, 0.9, 0.7))
    nodes = (cells[0] + 1) * (cells[1] + 1) * (cells[2] + 1)

    def run(xdefer, n, protocol):
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"cg_xdefer", xdefer))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"tile3d_max_nodes", 0))      # the marching kernels on the small grid too
        h = _lib.Heat()
        h.kappa, h.dt, h.steady, h.source = 1.0, 1.0, 1, 1.0
        h.bc_mode, h.fast_path = _lib.BC_UNIFORM, 1
        h.bc_present[0], h.bc_value[0] = 1, 0.5
        s = C.c_void_p()
        _lib.check(gpu_ctx, lib.pdegpu_session_create(gpu_ctx, grid, h, 0, cells[2] + 1, C.byref(s)))
        try:
            if protocol == "sr":
                _lib.check(gpu_ctx, lib.pdegpu_session_set_cg_variant(s, 1))
            _lib.check(gpu_ctx, lib.pdegpu_session_begin(s, 0.25))
            _lib.check(gpu_ctx, lib.pdegpu_session_residual(s))
            _lib.check(gpu_ctx, lib.pdegpu_session_reduce_local(s, 1, 1))
            if protocol == "native":
                _lib.check(gpu_ctx, lib.pdegpu_session_iterate(s, n, 0, None))
            else:
                for _ in range(n):
                    _lib.check(gpu_ctx, lib.pdegpu_session_k1(s))
                    _lib.check(gpu_ctx, lib.pdegpu_session_k2(s))
            _lib.check(gpu_ctx, lib.pdegpu_session_flush(s))
            x = np.empty(nodes)
            _lib.check(gpu_ctx, lib.pdegpu_session_get_x(s, _lib.dptr(x)))
            return x
        finally:
            lib.pdegpu_session_destroy(s)

    try:
        for protocol in ("native", "classic", "sr"):
            for n in (1, 2, 3, 4, 5, 9):
                a, b = run(1, n, protocol), run(0, n, protocol)
                assert np.isfinite(a).all() and np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (protocol, n)
            assert np.abs(run(0, 9, protocol) - run(0, 3, protocol)).max() > 1e-6      # (the iterations do move x)
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"cg_xdefer", 1))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"tile3d_max_nodes", 3000000))


def test_stepwise_timing_tells_the_two_forms_of_k2_apart(lib, gpu_ctx):
    """pdegpu_stats of a session whose K2 carries the x update of every other iteration (cg_xdefer): [0] K1 at 24 B per node, [1] the K2 launches
    without the update at 24, [2] those with it at 40 — what bench.py's roofline block is computed from; with cg_xdefer = 0: [0] 40, [1] 24, [2] empty"""
    import ctypes as C
    from pdegpu import _lib
    cells = (140, 20, 18)
    nodes = (cells[0] + 1) * (cells[1] + 1) * (cells[2] + 1)
    grid = _lib.make_grid(3, cells, (1.0, 0.9, 0.7))
    try:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"tile3d_max_nodes", 0))
        for xdefer in (1, 0):
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"cg_xdefer", xdefer))
            h = _lib.Heat()
            h.kappa, h.dt, h.steady, h.source = 1.0, 1.0, 1, 1.0
            h.bc_mode, h.fast_path = _lib.BC_UNIFORM, 1
            h.bc_present[0] = 1
            s = C.c_void_p()
            _lib.check(gpu_ctx, lib.pdegpu_session_create(gpu_ctx, grid, h, 0, cells[2] + 1, C.byref(s)))
            try:
                _lib.check(gpu_ctx, lib.pdegpu_session_set_cg_variant(s, 1))
                _lib.check(gpu_ctx, lib.pdegpu_session_begin(s, 0.0))
                _lib.check(gpu_ctx, lib.pdegpu_session_residual(s))
                _lib.check(gpu_ctx, lib.pdegpu_session_set_timing(s, 1))
                for _ in range(7):
                    _lib.check(gpu_ctx, lib.pdegpu_session_k1(s))
                    _lib.check(gpu_ctx, lib.pdegpu_session_k2(s))
                st = _lib.Stats()
                _lib.check(gpu_ctx, lib.pdegpu_session_get_stats(s, st))
            finally:
                lib.pdegpu_session_destroy(s)
            per = [st.alg_bytes[k] / max(1, st.kernel_launches[k]) / nodes for k in range(3)]
            if xdefer:
                assert [st.kernel_launches[k] for k in range(3)] == [7, 4, 3] and per == [24.0, 24.0, 40.0], (list(st.kernel_launches), per)
                assert st.kernel_ms[0] > 0 and st.kernel_ms[1] > 0 and st.kernel_ms[2] > 0
            else:
                assert [st.kernel_launches[k] for k in range(3)] == [7, 7, 0] and per[:2] == [40.0, 24.0], (list(st.kernel_launches), per)
            assert st.cg_iterations == 7
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"cg_xdefer", 1))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"tile3d_max_nodes", 3000000))


@pytest.mark.gpu
@pytest.mark.parametrize("dim, stride", [(2, 1), (3, 1), (3, 2), (1, 1)])
def test_snapshots_copied_beside_the_next_step_are_the_snapshots_copied_in_line(lib, gpu_ctx, dim, stride):
    """snapshot_async (SnapCopier in csrc/heat.hip): the TimeSeriesField of a transient solve (fenics_mcp_server.py:307-318, 436-468, 705-716 append one
    snapshot per step) leaves the device through two device buffers and a helper thread while the next step is solved.  Every snapshot must be in the
    caller's array when the solve returns, bit for bit what the in-line copies deliver — odd step counts, a stride, the CG path in 1-D — and match the oracle."""
    from pdegpu import heat, _lib
    if dim == 2:
        kw = dict(Lx=1.0, Ly=0.7, nx=37, ny=29, diffusivity=0.8, T_boundary=2.0, T_initial=20.0, dt=0.003, num_steps=7)
        solve, ref = heat._solve_heat_2d_raw, o.solve_heat_2d
    elif dim == 3:
        kw = dict(Lx=1.0, Ly=0.8, Lz=0.6, nx=17, ny=13, nz=11, diffusivity=0.8, T_boundary=2.0, T_initial=20.0, dt=0.003, num_steps=7)
        solve, ref = heat._solve_heat_3d_raw, o.solve_heat_3d
    else:
        kw = dict(length=2.0, nx=50, diffusivity=1.0, T_left=20.0, T_right=0.0, T_initial=5.0, dt=0.01, num_steps=7)
        solve, ref = heat._solve_heat_1d_raw, o.solve_heat_1d
    got = {}
    for mode in (0, 1):
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"snapshot_async", mode))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"direct_1d", 0))          # 1-D through the step loop (CG), not the one-launch Thomas path
        try:
            got[mode] = solve(**kw, rtol=1e-12, snapshot_stride=stride)
        finally:
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"snapshot_async", 1))
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"direct_1d", 1))
    assert got[0].values.shape == got[1].values.shape and got[0].values.shape[0] == 1 + (7 + stride - 1) // stride
    assert np.array_equal(got[0].values, got[1].values) and np.array_equal(got[0].times, got[1].times)
    if stride == 1:
        assert rel_err(got[1].values, ref(**kw).values) < TOL
