"""GPU parity tests of the static elasticity path vs the CPU oracle (fp64; tolerance 1e-8 relative to the field
maximum with PCG rtol 1e-12: the clamped cantilever systems have condition numbers of 1e4-1e6)."""
import numpy as np
import pytest

from oracle import fem_p1 as o

pytestmark = pytest.mark.gpu
TOL = 1e-8


def rel_err(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("dim,n,Ls", [(2, (9, 7), (1.0, 0.5)), (3, (6, 5, 4), (1.0, 0.6, 0.25)), (3, (70, 3, 2), (2.0, 0.3, 0.2))])
@pytest.mark.parametrize("plane_stress", [True, False])
def test_block_operator_matches_assembled_matrix(lib, gpu_ctx, dim, n, Ls, plane_stress):
    from pdegpu import _lib
    if dim == 3 and plane_stress:
        pytest.skip("plane stress is a 2-D notion")
    E, nu = 210e9, 0.3
    lam, mu = o.lame(E, nu, plane_stress and dim == 2)
    c, cells = o.rectangle_mesh(Ls[0], Ls[1], n[0], n[1]) if dim == 2 else o.box_mesh(*Ls, *n)
    K = o.assemble_elasticity(c, cells, lam, mu)
    rng = np.random.default_rng(3)
    u = rng.uniform(-1, 1, (c.shape[0], dim))
    y = np.empty_like(u)
    el = _lib.Elasticity()
    el.E, el.nu, el.plane_stress = E, nu, int(plane_stress)
    g = _lib.make_grid(dim, n, Ls)
    _lib.check(gpu_ctx, lib.pdegpu_elasticity_apply(gpu_ctx, g, el, _lib.dptr(u), _lib.dptr(y), 0))
    ref = (K @ u.ravel()).reshape(-1, dim)
    assert rel_err(y, ref) < 1e-12


@pytest.mark.parametrize("n,Ls", [((6, 5, 4), (1.0, 0.6, 0.25)), ((63, 4, 5), (2.0, 0.3, 0.2)), ((64, 3, 3), (2.0, 0.3, 0.2)),
                                  ((65, 2, 9), (1.0, 0.2, 0.5)), ((130, 5, 3), (1.0, 0.2, 0.1)), ((2, 2, 2), (1.0, 1.0, 1.0)),
                                  ((20, 11, 8), (1.0, 0.5, 0.4)), ((9, 23, 6), (0.5, 1.0, 0.3)), ((66, 8, 12), (1.0, 0.2, 0.3))])
def test_fast_block_apply_variants_match_assembled_matrix(lib, gpu_ctx, n, Ls):
    """k_el_fast (z-marching interior + boundary workgroups): rows per column 1 | 2, z-chunk lengths from one plane to the
    whole column, x-tiles ending at 62 / 63 / 64 / 65 nodes, against the assembled matrix and the thread-per-node kernel"""
    from pdegpu import _lib
    E, nu = 210e9, 0.3
    lam, mu = o.lame(E, nu)
    c, cells = o.box_mesh(*Ls, *n)
    K = o.assemble_elasticity(c, cells, lam, mu)
    u = np.random.default_rng(11).uniform(-1, 1, (c.shape[0], 3))
    ref = (K @ u.ravel()).reshape(-1, 3)
    el = _lib.Elasticity()
    el.E, el.nu = E, nu
    g = _lib.make_grid(3, n, Ls)
    outs = {}
    try:
        for fast, ry, waves, yx in ((0, 1, 2048, 1), (1, 1, 2048, 1), (1, 2, 2048, 1), (1, 1, 1, 1), (1, 2, 100000, 1), (1, 1, 2048, 0),
                                    (1, 2, 2048, 0), (1, 2, 1, 0), (1, 4, 2048, 1), (1, 4, 2048, 0), (1, 4, 1, 1)):
            for k, v in (("el_fast", fast), ("el_fast_ry", ry), ("el_fast_waves", waves), ("el_fast_yx", yx)):
                _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, k.encode(), v))
            y = np.full_like(u, np.nan)
            _lib.check(gpu_ctx, lib.pdegpu_elasticity_apply(gpu_ctx, g, el, _lib.dptr(u), _lib.dptr(y), 0))
            assert rel_err(y, ref) < 1e-12, (fast, ry, waves, yx)
            outs[(fast, ry, waves, yx)] = y
    finally:
        for k, v in (("el_fast", 1), ("el_fast_ry", 2), ("el_fast_waves", 0), ("el_fast_yx", 1)):
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, k.encode(), v))
    # z-chunking does not change a single bit (same per-node arithmetic)
    assert np.array_equal(outs[(1, 1, 2048, 1)], outs[(1, 1, 1, 1)])
    # neither does handing the y-halo rows through LDS instead of loading them
    assert np.array_equal(outs[(1, 1, 2048, 1)], outs[(1, 1, 2048, 0)]) and np.array_equal(outs[(1, 2, 2048, 1)], outs[(1, 2, 2048, 0)])


@pytest.mark.parametrize("n,Ls", [((6, 5, 4), (1.0, 0.6, 0.25)), ((64, 3, 3), (2.0, 0.3, 0.2)), ((130, 5, 3), (1.0, 0.2, 0.1)), ((2, 2, 2), (1.0, 1.0, 1.0)),
                                  ((20, 11, 8), (1.0, 0.5, 0.4)), ((62, 14, 17), (1.0, 0.2, 0.3))])
def test_masked_launch_of_the_pcg_loop_matches_assembled_matrix(lib, gpu_ctx, n, Ls):
    """form 2 of pdegpu_elasticity_apply is the launch the PCG loop issues: the rows of the assembled matrix, zero on the clamped face"""
    from pdegpu import _lib
    E, nu = 210e9, 0.3
    lam, mu = o.lame(E, nu)
    c, cells = o.box_mesh(*Ls, *n)
    K = o.assemble_elasticity(c, cells, lam, mu)
    u = np.random.default_rng(12).uniform(-1, 1, (c.shape[0], 3))
    ref = (K @ u.ravel()).reshape(-1, 3)
    ref[c[:, 0] < 1e-12] = 0.0
    el = _lib.Elasticity()
    el.E, el.nu = E, nu
    y = np.full_like(u, np.nan)
    _lib.check(gpu_ctx, lib.pdegpu_elasticity_apply(gpu_ctx, _lib.make_grid(3, n, Ls), el, _lib.dptr(u), _lib.dptr(y), 2))
    assert rel_err(y, ref) < 1e-12


@pytest.mark.parametrize("n", [(256, 64, 64), (250, 60, 44), (186, 70, 90)])
def test_one_round_refit_of_the_marching_apply_changes_no_bit(lib, gpu_ctx, n):
    """el_fast_fit (and el_bnd_gran: 64 instead of 256 boundary nodes per boundary workgroup): a launch whose equal z-chunks would need between one and two resident rounds of workgroups (config 5: 21 chunks of 3 planes,
    846 workgroups on the 768 slots of 256 CUs) is cut into as many chunks as fill one round, of two lengths (19 chunks of 4 / 3 planes).  Same
    per-node arithmetic: the result must be the equal-chunk launch's bit for bit (that launch is checked against the assembled matrix above and,
    at config 5's size, by the independent residual of test_config5_at_full_size)"""
    from pdegpu import _lib
    nodes = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)
    u = np.random.default_rng(4).uniform(-1, 1, (nodes, 3))
    el = _lib.Elasticity()
    el.E, el.nu = 210e9, 0.3
    g = _lib.make_grid(3, n, (1.0, 0.25, 0.25))
    outs = []
    try:
        for fit, gran in ((1, 64), (0, 64), (1, 256)):
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"el_fast_fit", fit))
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"el_bnd_gran", gran))
            y = np.full_like(u, np.nan)
            _lib.check(gpu_ctx, lib.pdegpu_elasticity_apply(gpu_ctx, g, el, _lib.dptr(u), _lib.dptr(y), 2))
            outs.append(y)
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"el_fast_fit", 1))
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"el_bnd_gran", 64))
    assert np.isfinite(outs[0]).all() and np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert np.all(outs[0].reshape(n[2] + 1, n[1] + 1, n[0] + 1, 3)[:, :, 0] == 0.0)


@pytest.mark.parametrize("n,Ls", [((6, 5, 4), (1.0, 0.6, 0.25)), ((37, 3, 2), (2.0, 0.3, 0.2)), ((16, 1, 1), (1.0, 0.1, 0.1))])
def test_mfma_element_by_element_apply_matches_assembled_matrix(lib, gpu_ctx, n, Ls):
    """v_mfma_f64_16x16x4_f64 element-by-element form (24x24 macro element, scatter by fp64 atomics) == matrix;
    ragged 16-element batches included"""
    from pdegpu import _lib
    E, nu = 210e9, 0.3
    lam, mu = o.lame(E, nu)
    c, cells = o.box_mesh(*Ls, *n)
    K = o.assemble_elasticity(c, cells, lam, mu)
    u = np.random.default_rng(5).uniform(-1, 1, (c.shape[0], 3))
    y = np.empty_like(u)
    el = _lib.Elasticity()
    el.E, el.nu = E, nu
    _lib.check(gpu_ctx, lib.pdegpu_elasticity_apply(gpu_ctx, _lib.make_grid(3, n, Ls), el, _lib.dptr(u), _lib.dptr(y), 1))
    ref = (K @ u.ravel()).reshape(-1, 3)
    assert rel_err(y, ref) < 1e-12


@pytest.mark.parametrize("quantity", ["stress", "strain"])
def test_bar_1d_matches_oracle(lib, gpu_ctx, quantity):
    from pdegpu import elasticity
    got = elasticity._solve_elasticity_1d_static(2.0, 40, 210e9, 0.5, 1.0e4, quantity, rtol=1e-13, want_displacement=True)
    ref = o.solve_elasticity_1d(2.0, 40, 210e9, 0.5, 1.0e4, quantity)
    assert got.values.shape == (1, 41) and got.times.tolist() == [0.0]
    assert rel_err(got.displacement[:, 0], ref.extra["displacement"]) < TOL
    assert rel_err(got.values, ref.values) < TOL
    for k, v in ref.meta.items():
        assert got.meta[k] == v


@pytest.mark.parametrize("plane_stress", [True, False])
@pytest.mark.parametrize("quantity", ["stress", "strain"])
def test_elasticity_2d_matches_oracle(lib, gpu_ctx, plane_stress, quantity):
    from pdegpu import elasticity
    kw = dict(Lx=1.0, Ly=0.4, nx=16, ny=7, E=210e9, nu=0.3, body_fx=1.0e4, body_fy=-76518.0, quantity=quantity,
              plane_stress=plane_stress)
    got = elasticity._solve_elasticity_2d_static(**kw, rtol=1e-12, want_displacement=True)
    ref = o.solve_elasticity_2d(**kw)
    assert rel_err(got.displacement, ref.extra["displacement"]) < TOL
    assert rel_err(got.values, ref.values) < TOL
    for k, v in ref.meta.items():
        assert got.meta[k] == v


@pytest.mark.parametrize("quantity", ["stress", "strain"])
def test_elasticity_3d_matches_oracle(lib, gpu_ctx, quantity):
    """scaled-down BASELINE config 5: cantilever under gravity (dispatcher default body_fz = -9.81*7800)"""
    from pdegpu import elasticity
    kw = dict(Lx=1.0, Ly=0.25, Lz=0.25, nx=16, ny=4, nz=4, E=210e9, nu=0.3, body_fx=0.0, body_fy=0.0, body_fz=-76518.0,
              quantity=quantity)
    got = elasticity._solve_elasticity_3d_static(**kw, rtol=1e-12, want_displacement=True)
    ref = o.solve_elasticity_3d(**kw)
    assert np.allclose(got.coords, ref.coords, atol=1e-15)
    assert rel_err(got.displacement, ref.extra["displacement"]) < TOL
    assert rel_err(got.values, ref.values) < TOL
    for k, v in ref.meta.items():
        assert got.meta[k] == v


@pytest.mark.parametrize("dim", [2, 3])
def test_deferred_x_update_of_the_pcg_gives_the_same_solve(lib, gpu_ctx, dim):
    """el_xdefer: every other update pass of the single-reduction PCG leaves x += alpha p out and the next one makes up for it with the
    search direction rebuilt from what it reads anyway (p_old' = (p_old - D^-1 r) / beta').  Same iterates up to round-off: both settings against
    the oracle, against each other, and the iteration counts within 1 %"""
    from pdegpu import elasticity, _lib
    if dim == 3:
        kw = dict(Lx=1.0, Ly=0.25, Lz=0.25, nx=40, ny=9, nz=7, E=210e9, nu=0.3, body_fx=3.0e3, body_fy=0.0, body_fz=-76518.0, quantity="stress")
        solve, ref = elasticity._solve_elasticity_3d_static, o.solve_elasticity_3d(**kw)
    else:
        kw = dict(Lx=1.0, Ly=0.3, nx=48, ny=14, E=210e9, nu=0.3, body_fx=1.0e4, body_fy=-76518.0, quantity="strain", plane_stress=False)
        solve, ref = elasticity._solve_elasticity_2d_static, o.solve_elasticity_2d(**kw)
    got = {}
    try:
        for xd in (1, 0):
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"el_xdefer", xd))
            got[xd] = solve(**kw, rtol=1e-12, want_displacement=True)
            assert rel_err(got[xd].displacement, ref.extra["displacement"]) < TOL and rel_err(got[xd].values, ref.values) < TOL
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"el_xdefer", 1))
    assert rel_err(got[1].displacement, got[0].displacement) < 1e-9
    it1, it0 = got[1].meta["cg_iterations"], got[0].meta["cg_iterations"]
    assert it0 > 50 and abs(it1 - it0) <= max(2, 0.01 * it0), (it1, it0)


def test_stepwise_elasticity_session_with_deferred_x_stops_anywhere(lib, gpu_ctx):
    """the stepwise protocol (the multi-GPU driver's: pdegpu_el_session_k1 = update pass + apply + reduce) alternates the two forms of the update
    pass by itself; flush / get_u apply an update that was left out.  After n = 1 ... 7 iterations the displacement must be the plain form's"""
    import ctypes as C
    from pdegpu import _lib
    n = (40, 9, 7)
    grid = _lib.make_grid(3, n, (1.0, 0.25, 0.2))
    el = _lib.Elasticity()
    el.E, el.nu = 210e9, 0.3
    el.body[0], el.body[2] = 3.0e3, -76518.0
    nodes = (n[0] + 1) * (n[1] + 1) * (n[2] + 1)

    def run(xd, iters, flush):
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"el_xdefer", xd))
        h = C.c_void_p()
        _lib.check(gpu_ctx, lib.pdegpu_el_session_create(gpu_ctx, grid, el, 0, n[2] + 1, C.byref(h)))
        try:
            assert lib.pdegpu_el_session_cg_variant(h) == 1
            _lib.check(gpu_ctx, lib.pdegpu_el_session_begin(h))
            for _ in range(iters):
                _lib.check(gpu_ctx, lib.pdegpu_el_session_k1(h))
                _lib.check(gpu_ctx, lib.pdegpu_el_session_k2(h))
            if flush:
                _lib.check(gpu_ctx, lib.pdegpu_el_session_flush(h))
            u = np.empty((nodes, 3))
            _lib.check(gpu_ctx, lib.pdegpu_el_session_get_u(h, _lib.dptr(u)))
            return u
        finally:
            lib.pdegpu_el_session_destroy(h)

    try:
        for iters in (1, 2, 3, 4, 5, 7):
            ref = run(0, iters, True)
            for flush in (True, False):                 # (get_u applies a pending update itself)
                got = run(1, iters, flush)
                assert np.isfinite(got).all() and np.abs(got - ref).max() <= 1e-11 * max(np.abs(ref).max(), 1e-300), (iters, flush)
        assert np.abs(run(0, 7, True) - run(0, 3, True)).max() > 0.0
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"el_xdefer", 1))


def test_elasticity_zero_load_and_errors(lib, gpu_ctx):
    from pdegpu import elasticity, _lib
    f = elasticity._solve_elasticity_3d_static(1, 1, 1, 4, 4, 4, 210e9, 0.3)          # no load -> u = 0 (:2639-2642)
    assert np.all(f.values == 0.0) and f.meta["cg_iterations"] == 0
    f = elasticity._solve_elasticity_2d_static(1, 1, 4, 4, 210e9, 0.3, quantity="strain")
    assert np.all(f.values == 0.0)
    with pytest.raises(_lib.PdeGpuError):
        elasticity._solve_elasticity_3d_static(1, 1, 1, 4, 4, 4, 210e9, 0.5, body_fz=-1.0)   # incompressible limit
    with pytest.raises(_lib.PdeGpuError):
        elasticity._solve_elasticity_2d_static(1, 1, 0, 4, 210e9, 0.3)


@pytest.mark.parametrize("nslabs", [1, 2, 3])
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("sr", [True, False])
def test_elasticity_slab_sessions_on_one_gpu(lib, gpu_ctx, nslabs, split, sr):
    _el_slab_case(lib, gpu_ctx, nslabs, split, sr)


@pytest.mark.parametrize("nslabs,split", [(7, False), (14, False), (3, True)])
def test_elasticity_thin_slabs(lib, gpu_ctx, nslabs, split):
    """14 planes as 7 slabs of two planes and 14 slabs of ONE plane (a plane next to both ghosts; slabs that own nothing but a face of the box), and the
    split apply on slabs of 4-5 planes (the smallest it accepts)"""
    _el_slab_case(lib, gpu_ctx, nslabs, split, True)


def _el_slab_case(lib, gpu_ctx, nslabs, split, sr):
    """the multi-GPU elasticity path on ONE GPU: slab sessions (ghost planes, global boundary classes), halo exchange of r (3 component
    planes; p on the ghost planes is maintained by k1) and the scalar all-reduces done by device copies in the order pdegpu.dist.SlabCG
    uses — split: k1 in two parts with the exchange in between (the overlapped order), the ghost planes of r poisoned while part 0 runs;
    then von Mises + project() from the gathered displacement.  Must reproduce the oracle.  sr: the single-reduction PCG — the residual update is
    part of k1 (k1_pre -> exchange of r -> k1_part 0 / 1), k2 does nothing, ONE block {p.Ap, r.z, r.r, Ap.D^-1 Ap} is summed over the slabs."""
    import torch
    from pdegpu import _lib
    from pdegpu.dist import GpuElasticitySlabEngine, slab_range
    n, Ls = (14, 4, 13), (1.0, 0.3, 0.6)
    E, nu, body = 210e9, 0.3, (0.0, 2.0e4, -76518.0)
    grid = _lib.make_grid(3, n, Ls)
    el = _lib.Elasticity()
    el.E, el.nu = E, nu
    for c in range(3):
        el.body[c] = body[c]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        engs = [GpuElasticitySlabEngine(grid, el, *slab_range(n[2] + 1, nslabs, r), 0, stream) for r in range(nslabs)]
        for e in engs:                      # (the default splits only slabs whose vectors hold >= 32 MiB: small slabs lose more on the two extra launches)
            _lib.check(e.ctx, lib.pdegpu_set_tunable(e.ctx, b"el_split_min_mb", 0 if split else 1 << 20))
            e.set_variant(sr)

        def exchange():
            for r in range(nslabs - 1):
                lo, hi = engs[r], engs[r + 1]
                for a, b in zip(hi.halo_tensors(-1), lo.halo_tensors(lo.nk - 1)):
                    a.copy_(b)
                for a, b in zip(lo.halo_tensors(lo.nk), hi.halo_tensors(0)):
                    a.copy_(b)

        def allreduce(which):
            a, b = engs[0].global_slots(which)
            tot = sum(e.scalars()[a:b].clone() for e in engs)
            for e in engs:
                e.scalars()[a:b].copy_(tot)

        for e in engs:
            e.begin()
        allreduce(1)
        rr0 = engs[0].rr()
        it = 0
        while it < 6000:
            for _ in range(50):
                assert all(e.can_split_k1() == (split and nslabs > 1 and e.nk >= 4) for e in engs)
                if sr or split:
                    for e in engs:                        # the receives have not landed yet: nothing before the exchange may read a ghost plane of r
                        e.k1_pre()                        # (sr: the update pass of the owned planes — r changes here; classic: nothing)
                        for w in ((-1,) if e.k0 > 0 else ()) + ((e.nk,) if e.k0 + e.nk < n[2] + 1 else ()):
                            for t in e.halo_tensors(w):
                                t.fill_(float("nan"))
                        e.k1_part(0)
                    exchange()
                    for e in engs:
                        e.k1_part(1)
                else:
                    exchange()
                    for e in engs:
                        e.k1()
                allreduce(0)
                for e in engs:
                    e.k2()                                # (sr: a no-op)
                if not sr:
                    allreduce(1)
            it += 50
            if engs[0].rr() <= 1e-24 * rr0:
                break
        for e in engs:
            e.flush()
        u = np.concatenate([e.get_u((n[0] + 1) * (n[1] + 1)) for e in engs], axis=0)
        for e in engs:
            e.close()
    ref = o.solve_elasticity_3d(*Ls, *n, E, nu, *body, "stress")
    assert rel_err(u, ref.extra["displacement"]) < TOL
    field = np.empty(u.shape[0])
    _lib.check(gpu_ctx, lib.pdegpu_elasticity_postprocess(gpu_ctx, grid, el, _lib.dptr(np.ascontiguousarray(u)), _lib.dptr(field)))
    assert rel_err(field, ref.values[0]) < TOL


@pytest.mark.parametrize("transport", ["torch", "library"])
def test_solve_elasticity_slabs_single_rank(lib, gpu_ctx, transport):
    """the rank-level driver (what each process runs under torch.distributed) with world size 1; "library": the PCG chunks are single C calls
    (pdegpu_el_session_iterate_slab) on a one-rank RCCL communicator of the library's own"""
    from pdegpu import _lib
    from pdegpu.dist import solve_elasticity_slabs
    n, Ls = (16, 4, 4), (1.0, 0.25, 0.25)
    grid = _lib.make_grid(3, n, Ls)
    el = _lib.Elasticity()
    el.E, el.nu = 210e9, 0.3
    el.body[2] = -76518.0
    out = solve_elasticity_slabs(grid, el, n, rtol=1e-12, transport=transport)
    ref = o.solve_elasticity_3d(*Ls, *n, 210e9, 0.3, 0.0, 0.0, -76518.0, "stress")
    assert rel_err(out["displacement"], ref.extra["displacement"]) < TOL and rel_err(out["field"], ref.values[0]) < TOL


@pytest.mark.parametrize("dim", [2, 3])
def test_lazy_dot_products_are_bitwise_the_eager_ones(lib, gpu_ctx, dim):
    """elasticity PCG without reduce launches (prologue sums) == with k_el_reduce after every kernel, bit for bit"""
    from pdegpu import elasticity, _lib
    outs = []
    try:
        for lazy in (1, 0):
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"lazy_dot", lazy))
            if dim == 2:
                outs.append(elasticity._solve_elasticity_2d_static(1.0, 0.3, 24, 9, 210e9, 0.3, 0.0, -5.0e4, "stress", True, rtol=1e-10))
            else:
                outs.append(elasticity._solve_elasticity_3d_static(1.0, 0.3, 0.2, 20, 6, 5, 210e9, 0.3, 0.0, 0.0, -7.0e4, "stress", rtol=1e-10))
    finally:
        _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, b"lazy_dot", 1))
    assert outs[0].meta["cg_iterations"] == outs[1].meta["cg_iterations"] > 0
    assert np.array_equal(outs[0].values, outs[1].values)


def test_row_and_kernel_variants_agree(lib, gpu_ctx):
    """3-D PCG with the marching block apply at 1 | 2 rows per column, with / without the LDS y-hand-off, with the
    thread-per-node kernels, eager dot products, non-temporal hints in the apply and the flat kernels, two nodes per lane, faces folded in or
    not: the same solution to solver tolerance (partial sums are grouped differently)"""
    from pdegpu import elasticity, _lib
    def tun(**kw):
        for k, v in kw.items():
            _lib.check(gpu_ctx, lib.pdegpu_set_tunable(gpu_ctx, k.encode(), v))
    outs = []
    try:
        for cfg in (dict(el_fast=1, el_fast_ry=2, el_fast_yx=1), dict(el_fast=1, el_fast_ry=1, el_fast_yx=1), dict(el_fast=1, el_fast_ry=2, el_fast_yx=0),
                    dict(el_fast=0), dict(el_fast=1, el_fast_ry=2, el_fast_yx=1, lazy_dot=0), dict(el_grid_mult=1), dict(el_grid_mult=16, el_fast_waves=300),
                    dict(el_fast_ry=4, el_fast_yx=1), dict(el_fast_ry=4, el_fast_yx=0), dict(el_fast_xface=0), dict(el_fast_xface=0, el_fast_ry=1),
                    dict(el_fast_nt=1), dict(el_fast_nt=1, el_fast_ry=1), dict(el_fast_yzface=1), dict(el_fast_yzface=0, el_fast_nt=0),
                    dict(el_cg_variant=0), dict(el_cg_variant=0, lazy_dot=0), dict(el_cg_variant=0, el_fast=0)):
            tun(**cfg)
            outs.append(elasticity._solve_elasticity_3d_static(1.0, 0.3, 0.2, 70, 6, 5, 210e9, 0.3, 1.0e3, 0.0, -7.0e4, "stress",
                                                               rtol=1e-12))
    finally:
        tun(el_fast=1, el_fast_ry=2, el_fast_yx=1, lazy_dot=1, el_grid_mult=4, el_fast_waves=0, el_fast_xface=1, el_fast_nt=-1, el_fast_yzface=3, el_cg_variant=1)
    ref = o.solve_elasticity_3d(1.0, 0.3, 0.2, 70, 6, 5, 210e9, 0.3, 1.0e3, 0.0, -7.0e4, "stress")
    for f in outs:
        assert rel_err(f.values, ref.values) < TOL
        assert rel_err(f.values, outs[0].values) < 1e-9
    assert np.array_equal(outs[0].values, outs[4].values)          # lazy vs eager dot products: bitwise
