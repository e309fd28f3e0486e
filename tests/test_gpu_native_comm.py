"""The slab loop's communication issued by the library itself (csrc/comm.hip: RCCL resolved at run time, exchange on the communicator's own
stream, all-reduce on the compute stream) — what one GPU allows to be checked of it: a ONE-rank communicator (a real ncclComm: sends to self are
real point-to-point messages inside one ncclGroup; the in-place one-rank all-reduce goes through RCCL's enqueue path, which completes it without a kernel) driving (a) whole iterations from C
(pdegpu_session_iterate_slab) against the host-driven loop, (b) TWO slab sessions of one process whose ghost planes travel through grouped
sends to self in the overlapped order — the residual history must be the single slab's — for the heat CG and the elasticity PCG.
The multi-process rendezvous (unique id through the torch.distributed group, ncclCommInitRank on N GPUs) needs N GPUs: bench.py --gpus N
measures that path in its extras."""
import numpy as np
import pytest

from oracle import fem_p1 as o

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def _heat_case():
    from pdegpu import _lib
    n, Ls = (140, 33, 27), (1.0, 0.7, 1.2)
    grid = _lib.make_grid(3, n, Ls)
    h = _lib.Heat()
    h.kappa, h.steady, h.source = 1.1, 1, 3.0
    h.bc_mode, h.fast_path = _lib.BC_UNIFORM, 1
    h.bc_present[0], h.bc_value[0] = 1, 0.25
    return n, grid, h


@pytest.mark.parametrize("variant", ["sr", "classic"])
def test_iterate_slab_from_c_matches_the_host_driven_loop(lib, gpu_ctx, variant):
    """one rank: pdegpu_session_iterate_slab (k1 parts, the all-reduce call(s) into RCCL, k2 — all enqueued by ONE C call per chunk) against
    SlabCG's Python loop of the same C-ABI calls: same kernels in the same order, so the same residuals to rounding of nothing"""
    import torch
    from pdegpu.dist import GpuSlabEngine, NativeComm, SlabCG
    n, grid, h = _heat_case()
    stream = torch.cuda.Stream()
    hist = {}
    with torch.cuda.stream(stream):
        for native in (False, True):
            eng = GpuSlabEngine(grid, h, 0, n[2] + 1, 0, stream)
            comm = NativeComm(eng, 0, 1) if native else None
            assert comm is None or (lib.pdegpu_comm_rank(comm.h), lib.pdegpu_comm_world(comm.h)) == (0, 1)
            cg = SlabCG(eng, 0, 1, variant=variant, comm=comm)
            cg.start(0.0)
            rr = [cg.rr()]
            for _ in range(8):
                cg.iterate(5)
                rr.append(cg.rr())
            hist[native] = rr
            if comm:
                comm.close()
            eng.close()
    assert hist[True][-1] < 1e-2 * hist[True][0]
    assert np.allclose(hist[True], hist[False], rtol=1e-12), hist


def test_comm_argument_checks(lib, gpu_ctx):
    import ctypes as C
    from pdegpu import _lib
    ident = (C.c_ubyte * 128)()
    assert lib.pdegpu_comm_unique_id(None) == _lib.EINVAL
    assert lib.pdegpu_comm_unique_id(ident) == 0 and any(ident)
    out = C.c_void_p()
    for rank, world in ((1, 1), (-1, 2), (0, 0)):
        assert lib.pdegpu_comm_create(gpu_ctx, rank, world, ident, C.byref(out)) == _lib.EINVAL and not out
    assert lib.pdegpu_comm_rank(None) == -1 and lib.pdegpu_comm_world(None) == -1
    assert lib.pdegpu_session_iterate_slab(None, None, 1) == _lib.EINVAL
    assert lib.pdegpu_el_session_iterate_slab(None, None, 1) == _lib.EINVAL
    lib.pdegpu_comm_destroy(None)


@pytest.mark.parametrize("early", [False, True])
def test_two_heat_slabs_exchanging_through_the_library_communicator(lib, gpu_ctx, early):
    """two slab sessions of ONE process, ghost planes as RCCL sends to self on the communicator's own stream (one ncclGroup per exchange, posted
    through pdegpu_session_exchange_post), the overlapped order of pdegpu_session_iterate_slab: exchange posted, K1 away from the ghosts, wait
    (stream-ordered), edge planes, the 24-byte block through the communicator's all-reduce, K2.  A receive that landed late, or early into planes
    part 0 still reads, would change the iterates: the residual history must be the single slab's.  early: the order inside a chunk of
    pdegpu_session_iterate_slab — K2 in two parts, the NEXT iteration's exchange posted between them (after the planes next to the ghosts, while K2's
    bulk still writes the planes in between), waited for in the middle of the next K1."""
    import torch
    from pdegpu.dist import GpuSlabEngine, NativeComm, SlabCG, slab_range, SC_PQ, SC_QQ, SC_RR, SC_RR_POLL
    n, grid, h = _heat_case()
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
        comm = NativeComm(lo, 0, 1)
        for e in engs:
            e.set_variant(True)
            assert e.can_split_k1() and not e.needs_p_halo()
            e.begin(0.0)
            e.residual()

        def sum_over_slabs(a, b):
            tot = lo.scalars()[a:b] + hi.scalars()[a:b]
            for e in engs:
                e.scalars()[a:b].copy_(tot)
                e.comm_allreduce(comm, a, b - a)                 # world 1: the identity, through RCCL, in place on library memory

        def rr_now():
            for e in engs:
                e.sr_poll()
            sum_over_slabs(SC_RR_POLL, SC_RR_POLL + 1)
            return lo.rr()

        def poison_ghosts():
            lo.plane_tensor(1, lo.nk).fill_(float("nan"))
            hi.plane_tensor(1, -1).fill_(float("nan"))

        def post_exchange():
            poison_ghosts()                                      # only the exchange may make them finite again
            comm.group_begin()
            lo.comm_exchange_post(comm, 1, hi_rank=0)            # send order == receive order: sends to self match first-in first-out
            hi.comm_exchange_post(comm, 2, lo_rank=0)
            hi.comm_exchange_post(comm, 1, lo_rank=0)
            lo.comm_exchange_post(comm, 2, hi_rank=0)
            comm.group_end()

        sum_over_slabs(SC_RR, SC_RR + 1)
        hist = [rr_now()]
        assert all(e.can_split_k2() for e in engs)
        posted = False
        for it in range(40):
            if not posted:
                post_exchange()
            posted = False
            for e in engs:
                e.k1_part(0)
            lo.comm_exchange_end(comm)                           # (both sessions enqueue on the same stream)
            for e in engs:
                e.k1_part(1)
            sum_over_slabs(SC_PQ, SC_QQ + 1)
            if early and it % 5 != 4:                            # (a poll follows every 5th K2: nothing in flight across it, as at the end of a chunk)
                for e in engs:
                    e.k2_part(0)
                post_exchange()
                posted = True
                for e in engs:
                    e.k2_part(1)
            else:
                for e in engs:
                    e.k2()
            if it % 5 == 4:
                hist.append(rr_now())
        comm.close()
        for e in engs:
            e.close()
    assert np.allclose(hist, ref, rtol=1e-9), (hist, ref)
    assert hist[-1] < 1e-2 * hist[0]


@pytest.mark.parametrize("sr", [True, False])
def test_two_elasticity_slabs_exchanging_through_the_library_communicator(lib, gpu_ctx, sr):
    """the elasticity PCG the same way: k1_pre (sr: the update pass — r changes here), the three component planes of r per neighbour as grouped sends to
    self, the split apply around the wait, ONE 32-byte block (classic: 8 + 16 bytes) through the communicator's all-reduce.  Must reproduce the oracle."""
    import torch
    from pdegpu import _lib
    from pdegpu.dist import GpuElasticitySlabEngine, NativeComm, slab_range
    n, Ls = (14, 4, 13), (1.0, 0.3, 0.6)
    E, nu, body = 210e9, 0.3, (0.0, 2.0e4, -76518.0)
    grid = _lib.make_grid(3, n, Ls)
    el = _lib.Elasticity()
    el.E, el.nu = E, nu
    for c in range(3):
        el.body[c] = body[c]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        engs = [GpuElasticitySlabEngine(grid, el, *slab_range(n[2] + 1, 2, r), 0, stream) for r in range(2)]
        lo, hi = engs
        comm = NativeComm(lo, 0, 1)
        for e in engs:
            _lib.check(e.ctx, lib.pdegpu_set_tunable(e.ctx, b"el_split_min_mb", 0))
            e.set_variant(sr)

        def allreduce(which):
            a, b = lo.global_slots(which)
            tot = lo.scalars()[a:b] + hi.scalars()[a:b]
            for e in engs:
                e.scalars()[a:b].copy_(tot)
                e.comm_allreduce(comm, a, b - a)

        for e in engs:
            e.begin()
        allreduce(1)
        rr0 = lo.rr()
        it = 0
        while it < 6000:
            for _ in range(50):
                for e in engs:
                    e.k1_pre()
                for t in lo.halo_tensors(lo.nk) + hi.halo_tensors(-1):
                    t.fill_(float("nan"))
                comm.group_begin()
                lo.comm_exchange_post(comm, 1, hi_rank=0)
                hi.comm_exchange_post(comm, 2, lo_rank=0)
                hi.comm_exchange_post(comm, 1, lo_rank=0)
                lo.comm_exchange_post(comm, 2, hi_rank=0)
                comm.group_end()
                for e in engs:
                    e.k1_part(0)
                lo.comm_exchange_end(comm)
                for e in engs:
                    e.k1_part(1)
                allreduce(0)
                for e in engs:
                    e.k2()
                if not sr:
                    allreduce(1)
            it += 50
            if lo.rr() <= 1e-24 * rr0:
                break
        for e in engs:
            e.flush()
        u = np.concatenate([e.get_u((n[0] + 1) * (n[1] + 1)) for e in engs], axis=0)
        comm.close()
        for e in engs:
            e.close()
    ref = o.solve_elasticity_3d(*Ls, *n, E, nu, *body, "stress")
    assert rel_err(u, ref.extra["displacement"]) < 1e-8


@pytest.mark.parametrize("steady", [True, False])
def test_solve_heat_slabs_with_the_library_transport(lib, gpu_ctx, steady):
    """the tool-level distributed solve on one rank with transport="library": chunks of CG iterations as single C calls, convergence polls and (transient)
    the x-halo step set-up through the communicator; against the single-GPU tool.  The box's longest axis is x: the slabs cut it (axes relabelled)."""
    from pdegpu import heat
    from pdegpu.dist import solve_heat_slabs
    m, Ls = (36, 12, 13), (1.5, 0.5, 0.6)
    kw = dict(steady=True) if steady else dict(T_initial=20.0, dt=0.01, num_steps=3)
    out = solve_heat_slabs(m, Ls, 1.3, 0.5, source=2.0, rtol=1e-12, transport="library", **kw)
    one = heat._solve_heat_3d_raw(*Ls, *m, 1.3, 0.5, 20.0, 0.01, 3, steady=steady, source_type="constant", source_value=2.0, rtol=1e-12)
    got = out["values"].reshape(one.values.shape)
    assert rel_err(got, one.values) < 1e-9 and np.allclose(out["times"], one.times)
    with pytest.raises(ValueError):
        solve_heat_slabs(m, Ls, 1.3, 0.5, steady=True, transport="mpi")


def test_exchange_begin_end_move_the_right_planes(lib, gpu_ctx):
    """pdegpu_session_exchange_begin / _end as pdegpu_session_iterate_slab and SlabCG call them (default peers, one group, both directions), on an interior
    slab in the communicator's rehearsal mode (both neighbours = this rank): the first / last owned plane must arrive in the slab's own lower / upper ghost
    plane — r (CG exchange), x (the step set-up of a transient solve, x_planes) and the three component planes of the elasticity residual."""
    import torch
    from pdegpu import _lib
    from pdegpu.dist import GpuSlabEngine, GpuElasticitySlabEngine, NativeComm, VEC_X, VEC_R
    n, grid, h = _heat_case()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        eng = GpuSlabEngine(grid, h, 5, 9, 0, stream)                      # planes 5 .. 13 of 28: a neighbour on both sides
        comm = NativeComm(eng, 0, 1)
        comm.set_self_loop(True)
        eng.begin(0.0)
        eng.residual()
        for vec, x_planes in ((VEC_R, False), (VEC_X, True)):
            own = [eng.plane_tensor(vec, w) for w in (0, eng.nk - 1)]
            ghost = [eng.plane_tensor(vec, w) for w in (-1, eng.nk)]
            for k, t in enumerate(own):
                t.copy_(torch.arange(t.numel(), dtype=torch.float64, device=t.device) * (k + 1) + 0.5)
            for t in ghost:
                t.fill_(float("nan"))
            eng.comm_exchange_begin(comm, x_planes)
            eng.comm_exchange_end(comm)
            torch.cuda.synchronize()
            assert torch.equal(ghost[0], own[0]) and torch.equal(ghost[1], own[1])
        comm.close()
        eng.close()
        grid_e = _lib.make_grid(3, (14, 4, 13), (1.0, 0.3, 0.6))
        el = _lib.Elasticity()
        el.E, el.nu = 210e9, 0.3
        el.body[2] = -76518.0
        ee = GpuElasticitySlabEngine(grid_e, el, 4, 6, 0, stream)
        comm = NativeComm(ee, 0, 1)
        comm.set_self_loop(True)
        ee.begin()
        for c in range(3):
            for k, w in enumerate((0, ee.nk - 1)):
                t = ee.plane_tensor(1, c, w)
                t.copy_(torch.arange(t.numel(), dtype=torch.float64, device=t.device) + 100.0 * c + 7.0 * k)
            for w in (-1, ee.nk):
                ee.plane_tensor(1, c, w).fill_(float("nan"))
        ee.comm_exchange_begin(comm)
        ee.comm_exchange_end(comm)
        torch.cuda.synchronize()
        for c in range(3):
            assert torch.equal(ee.plane_tensor(1, c, -1), ee.plane_tensor(1, c, 0))
            assert torch.equal(ee.plane_tensor(1, c, ee.nk), ee.plane_tensor(1, c, ee.nk - 1))
        comm.close()
        ee.close()
