"""Slab-decomposed CG across GPUs of one node: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI) for the only two exchanges the path has —

  * halo: before each operator application every rank sends its first / last owned node plane of r and p
    to the lower / upper neighbour and receives their planes into its ghost planes.  Planes are contiguous
    in the pitched layout ((nx+1 padded) x (ny+1) fp64 = 2.2 MB at 513^2), so there is no pack kernel; the
    four point-to-point transfers of one exchange are issued as one batch (one ncclGroup) and use the two
    xGMI links to the slab neighbours;
  * dot products: ONE all-reduce per iteration (single-reduction form, the default where the engine offers it): after K1 the
    24-byte block {p.Ap, r.r, Ap.Ap} — K1 also sums Ap.Ap, the r.r of K2's residual waits in its partials and joins the next
    message, beta comes from the predicted |r_{n+1}|^2 = alpha^2 Ap.Ap - r.r (the same iterates in exact arithmetic).  The classic
    form (variant="classic": 8 bytes after K1, 8 bytes after K2) stays selectable.  The messages are latency-bound, the result
    stays on the device, alpha / beta are formed inside the kernels: the host never synchronises in the loop;
  * overlap: the exchange is posted first, K1 then runs on the owned planes whose stencil touches no ghost plane
    (engine.k1_part(0)) while the planes travel (RCCL works on its own stream), and the one or two planes next to the
    ghosts follow once the transfer has completed (engine.k1_part(1)).  Elasticity works the same way: only r travels (three
    contiguous component planes per neighbour in the one batch), the p-update of the owned planes and the block apply away from
    the ghosts run meanwhile, the ghost planes' p-update and the edge planes follow.

  * who issues them: by default this module, through torch.distributed (P2POp batch + all_reduce: 20 + 90 us of host time per iteration,
    tools/dist_overhead.py — hidden behind 1.4 ms of kernels on a 512^3 slab, not behind the 25-60 us of a thin one).  With
    SlabCG(..., comm=NativeComm(...)) the library issues them itself on its own RCCL communicator (csrc/comm.hip) and a whole chunk of
    iterations is ONE C call (pdegpu_session_iterate_slab): same messages, same order, same stream protocol, no host language in the loop.

The compute engine is abstract so that the same driver runs (a) the HIP session through the C-ABI on GPUs and
(b) under the `gloo` backend on CPU tensors in the test-suite (tests/ supply a numpy engine built on the
oracle) — the N>1 logic (partition, ownership of planes in the dot products, ghost exchange, scalar flow) is
identical in both."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.distributed as dist

from . import _lib as L

VEC_X, VEC_R, VEC_P, VEC_Q = 0, 1, 2, 3
SC_PQ_LOCAL, SC_RR_LOCAL, SC_PQ, SC_RR, SC_RR_PREV = 0, 1, 2, 3, 4
SC_QQ, SC_RR_POLL = 4, 6          # single-reduction form: the message is scalars[2:5] = {p.Ap, r.r, Ap.Ap}; polls read scalars[6]


class _DevArray:
    """zero-copy view of library-owned device memory for torch.as_tensor"""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def slab_range(n_planes: int, world: int, rank: int):
    """owned node planes [k0, k0+nk) of `rank` — C-ABI pdegpu_slab_partition"""
    lib = L.load_library()
    k0, nk = C.c_int64(), C.c_int64()
    rc = lib.pdegpu_slab_partition(n_planes, world, rank, C.byref(k0), C.byref(nk))
    if rc != 0:
        raise L.PdeGpuError(rc, f"cannot split {n_planes} planes over {world} ranks")
    return k0.value, nk.value


class NativeComm:
    """RCCL communicator owned by libpdegpu (csrc/comm.hip) for the context of `engine`: one per rank, created collectively.  The 128-byte
    unique id travels through the torch.distributed group the ranks already share (any backend), or is made locally when world == 1."""

    def __init__(self, engine, rank: int, world: int, group=None):
        self.lib = L.load_library()
        self.rank, self.world, self.ctx = rank, world, engine.ctx
        ident = (C.c_ubyte * 128)()
        rc = self.lib.pdegpu_comm_unique_id(ident) if rank == 0 else 0
        if world > 1:
            # (status, id) travel together: a rank 0 that cannot make an id must not leave the others waiting in the broadcast
            box = [(int(rc), bytes(ident))]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            rc, ident = box[0][0], (C.c_ubyte * 128).from_buffer_copy(box[0][1])
        if rc != 0:
            raise L.PdeGpuError(rc, "pdegpu_comm_unique_id failed on rank 0: librccl.so.1 cannot be loaded or refused to make an id")
        self.h = C.c_void_p()
        L.check(self.ctx, self.lib.pdegpu_comm_create(self.ctx, rank, world, ident, C.byref(self.h)))

    def close(self):
        """before the engine (its context) is closed"""
        if self.h:
            self.lib.pdegpu_comm_destroy(self.h)
            self.h = None

    def set_self_loop(self, on: bool):
        """one-GPU rehearsal: the slab's two neighbours are this rank itself (real messages, meaningless iterates)"""
        L.check(self.ctx, self.lib.pdegpu_comm_set_self_loop(self.h, 1 if on else 0))

    def group_begin(self):
        L.check(self.ctx, self.lib.pdegpu_comm_group_begin(self.h))

    def group_end(self):
        L.check(self.ctx, self.lib.pdegpu_comm_group_end(self.h))


class GpuSlabEngine:
    """pdegpu_session on one GPU; device memory stays owned by the library, torch only sees plane views."""
    supports_sr = True
    sr_rr_lags = True        # single-reduction form: the r.r of the last K2 sits in its partials until the next message (polls: sr_poll)

    def __init__(self, grid, heat, k0: int, nk: int, device: int, stream: Optional[torch.cuda.Stream] = None):
        self.lib = L.load_library()
        self.device = device
        self.stream = stream
        self.grid = grid
        self.k0, self.nk = k0, nk
        self.ctx = C.c_void_p()
        sp = C.c_void_p(stream.cuda_stream) if stream is not None and stream.cuda_stream else None
        rc = self.lib.pdegpu_create(C.byref(self.ctx), device, sp)
        if rc != 0:
            raise L.PdeGpuError(rc, f"pdegpu_create(device={device}) failed: no usable gfx950 device (no CPU fallback)")
        self.s = C.c_void_p()
        L.check(self.ctx, self.lib.pdegpu_session_create(self.ctx, grid, heat, k0, nk, C.byref(self.s)))
        self.plane = int(self.lib.pdegpu_session_plane_doubles(self.s))
        self._planes = {}            # device pointer -> zero-copy tensor view (pointers are fixed for the session's life)
        self._scal = torch.as_tensor(_DevArray(self.lib.pdegpu_session_scalars(self.s), 8), device=f"cuda:{device}")

    def close(self):
        if self.s:
            self.lib.pdegpu_session_destroy(self.s)
            self.s = None
        if self.ctx:
            self.lib.pdegpu_destroy(self.ctx)
            self.ctx = None

    def _ck(self, rc):
        L.check(self.ctx, rc)

    def begin(self, x_fill: float):
        self._ck(self.lib.pdegpu_session_begin(self.s, x_fill))

    def set_variant(self, sr: bool):
        """single-reduction recurrences (one 24-byte message per iteration) or the classic ones; before begin()"""
        self._ck(self.lib.pdegpu_session_set_cg_variant(self.s, 1 if sr else 0))
        self.sr = bool(sr)

    def sr_poll(self):
        self._ck(self.lib.pdegpu_session_sr_poll(self.s))

    def residual(self):
        self._ck(self.lib.pdegpu_session_residual(self.s))

    def k1(self):
        self._ck(self.lib.pdegpu_session_k1(self.s))

    def k1_part(self, part: int):
        """part 0: planes away from the ghosts (halo exchange in flight); part 1: the planes next to them + the p.q sum"""
        self._ck(self.lib.pdegpu_session_k1_part(self.s, part))

    def can_split_k1(self) -> bool:
        return bool(self.lib.pdegpu_session_can_split_k1(self.s))

    def k2(self):
        self._ck(self.lib.pdegpu_session_k2(self.s))

    def k2_part(self, part: int):
        """part 0: the planes next to the ghosts (then the next exchange may be posted); part 1: the rest.  Unsplittable sessions: all in part 0."""
        self._ck(self.lib.pdegpu_session_k2_part(self.s, part))

    def can_split_k2(self) -> bool:
        return bool(self.lib.pdegpu_session_can_split_k2(self.s))

    def flush(self):
        self._ck(self.lib.pdegpu_session_flush(self.s))

    # transient steps (the time loop of _solve_heat_3d_raw distributed over slabs)
    def step_rhs(self):
        self._ck(self.lib.pdegpu_session_step_rhs(self.s))

    def lift_norm(self):
        self._ck(self.lib.pdegpu_session_lift_norm(self.s))

    def reset_cg(self):
        self._ck(self.lib.pdegpu_session_reset_cg(self.s))

    def x_halo_tensors(self, which: int):
        return [self.plane_tensor(VEC_X, which)]

    def set_timing(self, on):
        """True / 1: HIP-event pairs around every K1 / K2 launch; k > 1: around those of every k-th iteration; False / 0: off"""
        self._ck(self.lib.pdegpu_session_set_timing(self.s, int(on)))

    def get_stats(self, stats):
        self._ck(self.lib.pdegpu_session_get_stats(self.s, stats))

    def owned_x(self):
        """solution on the owned planes, numpy [nk, ny+1, nx+1] (synchronises)"""
        import numpy as np
        g = self.grid
        out = np.empty((self.nk, g.n[1] + 1, g.n[0] + 1))
        self._ck(self.lib.pdegpu_session_get_x(self.s, L.dptr(out)))
        return out

    def local_to_global(self, which: int):
        pass            # residual / k1 / k2 already left the slab's sum in the global slot (no extra device op)

    def scalars(self) -> torch.Tensor:
        return self._scal

    def plane_tensor(self, vec: int, which: int) -> torch.Tensor:
        ptr = self.lib.pdegpu_session_plane_ptr(self.s, vec, which)
        if not ptr:
            raise L.PdeGpuError(L.EINVAL, f"no plane {which} of vector {vec}")
        t = self._planes.get(ptr)
        if t is None:
            t = self._planes[ptr] = torch.as_tensor(_DevArray(ptr, self.plane), device=f"cuda:{self.device}")
        return t

    def needs_p_halo(self) -> bool:
        return bool(self.lib.pdegpu_session_needs_p_halo(self.s))

    def halo_tensors(self, which: int):
        """planes `which` of every vector the next k1 needs on the ghost planes"""
        vecs = (VEC_R, VEC_P) if self.needs_p_halo() else (VEC_R,)
        return [self.plane_tensor(v, which) for v in vecs]

    def global_slots(self, which: int):
        if which == 0 and getattr(self, "sr", False):
            return (SC_PQ, SC_QQ + 1)                    # {p.Ap, r.r, Ap.Ap}: one message
        return (SC_PQ, SC_PQ + 1) if which == 0 else (SC_RR, SC_RR + 1)

    # ---- exchange / reduction issued by the library on its own RCCL communicator (NativeComm) ----
    def comm_exchange_begin(self, comm, x_planes: bool = False):
        self._ck(self.lib.pdegpu_session_exchange_begin(self.s, comm.h, 1 if x_planes else 0))

    def comm_exchange_end(self, comm):
        self._ck(self.lib.pdegpu_session_exchange_end(self.s, comm.h))

    def comm_exchange_post(self, comm, what: int, x_planes: bool = False, lo_rank: int = -1, hi_rank: int = -1):
        self._ck(self.lib.pdegpu_session_exchange_post(self.s, comm.h, what, 1 if x_planes else 0, lo_rank, hi_rank))

    def comm_allreduce(self, comm, first: int, count: int):
        self._ck(self.lib.pdegpu_session_allreduce(self.s, comm.h, first, count))

    def iterate_slab(self, comm, iters: int):
        self._ck(self.lib.pdegpu_session_iterate_slab(self.s, comm.h, iters))

    def iterate_native(self, iters: int, time_kernels: bool, stats=None):
        self._ck(self.lib.pdegpu_session_iterate(self.s, iters, 1 if time_kernels else 0, stats))

    def rr(self) -> float:
        v = C.c_double()
        self._ck(self.lib.pdegpu_session_rr(self.s, C.byref(v)))
        return v.value

    def sync(self):
        self._ck(self.lib.pdegpu_session_sync(self.s))


class GpuElasticitySlabEngine:
    """pdegpu_el_session on one GPU (3-D elasticity, Jacobi-PCG): the same driver protocol as the heat engine."""
    supports_sr = True
    sr_rr_lags = False       # single-reduction PCG: the residual update is part of k1, scalars[2] is the r.r of the current residual

    def __init__(self, grid, el, k0: int, nk: int, device: int, stream: Optional[torch.cuda.Stream] = None):
        self.lib = L.load_library()
        self.device, self.k0, self.nk = device, k0, nk
        self.ctx = C.c_void_p()
        sp = C.c_void_p(stream.cuda_stream) if stream is not None and stream.cuda_stream else None
        rc = self.lib.pdegpu_create(C.byref(self.ctx), device, sp)
        if rc != 0:
            raise L.PdeGpuError(rc, f"pdegpu_create(device={device}) failed: no usable gfx950 device (no CPU fallback)")
        self.s = C.c_void_p()
        L.check(self.ctx, self.lib.pdegpu_el_session_create(self.ctx, grid, el, k0, nk, C.byref(self.s)))
        self.plane = int(self.lib.pdegpu_el_session_plane_doubles(self.s))
        self._planes = {}
        self._scal = torch.as_tensor(_DevArray(self.lib.pdegpu_el_session_scalars(self.s), 8), device=f"cuda:{device}")

    def close(self):
        if self.s:
            self.lib.pdegpu_el_session_destroy(self.s)
            self.s = None
        if self.ctx:
            self.lib.pdegpu_destroy(self.ctx)
            self.ctx = None

    def _ck(self, rc):
        L.check(self.ctx, rc)

    def begin(self, x_fill: float = 0.0):
        self._ck(self.lib.pdegpu_el_session_begin(self.s))

    def set_variant(self, sr: bool):
        """single-reduction PCG (r- and p-update fused into k1, ONE 32-byte message per iteration) or the classic recurrences; before begin()"""
        self._ck(self.lib.pdegpu_el_session_set_cg_variant(self.s, 1 if sr else 0))
        self.sr = bool(sr)

    def k1_pre(self):
        """single-reduction form: the update pass of the owned planes (r changes here: its halo exchange follows); classic form: nothing"""
        self._ck(self.lib.pdegpu_el_session_k1_pre(self.s))

    def residual(self):
        pass                                            # begin() already formed r = P b, z and the local sums (x = 0)

    def k1(self):
        self._ck(self.lib.pdegpu_el_session_k1(self.s))

    def k2(self):
        self._ck(self.lib.pdegpu_el_session_k2(self.s))

    def flush(self):
        self._ck(self.lib.pdegpu_el_session_flush(self.s))

    def local_to_global(self, which: int):
        pass

    def scalars(self) -> torch.Tensor:
        return self._scal

    def plane_tensor(self, vec: int, comp: int, which: int) -> torch.Tensor:
        ptr = self.lib.pdegpu_el_session_plane_ptr(self.s, vec, comp, which)
        if not ptr:
            raise L.PdeGpuError(L.EINVAL, f"no plane {which} of vector {vec}/{comp}")
        t = self._planes.get(ptr)
        if t is None:
            t = self._planes[ptr] = torch.as_tensor(_DevArray(ptr, self.plane), device=f"cuda:{self.device}")
        return t

    def halo_tensors(self, which: int):
        # r only (3 component planes, each contiguous: no pack kernel): z = D^-1 r is formed by the p-update, and p on the ghost planes is
        # maintained by k1 itself from the slab's own p_{n-1} there (same formula, same global scalars as the neighbour rank)
        return [self.plane_tensor(1, c, which) for c in range(3)]

    def k1_part(self, part: int):
        """part 0: everything that reads no ghost plane (halo exchange in flight); part 1: ghost p-update, edge planes, the p.q sum"""
        self._ck(self.lib.pdegpu_el_session_k1_part(self.s, part))

    def can_split_k1(self) -> bool:
        return bool(self.lib.pdegpu_el_session_can_split_k1(self.s))

    def global_slots(self, which: int):
        if which == 0 and getattr(self, "sr", False):
            return (0, 4)                                # {p.Ap, r.z, r.r, Ap.D^-1 Ap}: the one message of an iteration
        return (0, 1) if which == 0 else (1, 3)          # p.q | r.z and r.r as one 16-byte message

    def comm_exchange_begin(self, comm, x_planes: bool = False):
        self._ck(self.lib.pdegpu_el_session_exchange_begin(self.s, comm.h))

    def comm_exchange_end(self, comm):
        self._ck(self.lib.pdegpu_el_session_exchange_end(self.s, comm.h))

    def comm_exchange_post(self, comm, what: int, x_planes: bool = False, lo_rank: int = -1, hi_rank: int = -1):
        self._ck(self.lib.pdegpu_el_session_exchange_post(self.s, comm.h, what, lo_rank, hi_rank))

    def comm_allreduce(self, comm, first: int, count: int):
        self._ck(self.lib.pdegpu_el_session_allreduce(self.s, comm.h, first, count))

    def iterate_slab(self, comm, iters: int):
        self._ck(self.lib.pdegpu_el_session_iterate_slab(self.s, comm.h, iters))

    def scalars_host(self):
        out = (C.c_double * 4)()
        self._ck(self.lib.pdegpu_el_session_scalars_host(self.s, out))
        return list(out)                                 # [p.q, r.z, r.r, r.z_prev]

    def rr(self) -> float:
        return self.scalars_host()[2]

    def get_u(self, plane_nodes: int):
        """displacement of the owned planes, numpy [nk * plane_nodes, 3]"""
        import numpy as np
        u = np.empty((self.nk * plane_nodes, 3))
        self._ck(self.lib.pdegpu_el_session_get_u(self.s, L.dptr(u)))
        return u


class SlabCG:
    """CG over a 1-D slab decomposition.  `engine` owns planes [k0, k0+nk) plus one ghost plane per side."""

    def __init__(self, engine, rank: int, world: int, group=None, overlap: bool = True, variant: str = "auto", comm: Optional[NativeComm] = None):
        """variant: "sr" = single-reduction recurrences (ONE scalar message per iteration), "classic" = two; "auto" = "sr" where the
        engine offers it.  comm: a NativeComm of the same rank / world — the library then issues the exchange and the all-reduces itself
        (HIP engines only) and iterate() is one C call per chunk; None: torch.distributed on `group`."""
        self.e, self.rank, self.world, self.group = engine, rank, world, group
        if comm is not None and (comm.rank != rank or comm.world != world or not hasattr(engine, "iterate_slab")):
            raise L.PdeGpuError(L.EINVAL, "comm must be a NativeComm of this rank / world and the engine a HIP engine")
        self.comm = comm
        self.force_collectives = False      # measurement aid: run the scalar all-reduces through the process group at world == 1 too
        self.overlap = overlap
        self.iterations = 0
        if variant not in ("auto", "sr", "classic"):
            raise ValueError(f"variant must be auto, sr or classic, got {variant!r}")
        can = bool(getattr(engine, "supports_sr", False))
        if variant == "sr" and not can:
            raise L.PdeGpuError(L.EUNSUPPORTED, "this engine has no single-reduction form")
        self.sr = can and variant != "classic"
        if can:
            engine.set_variant(self.sr)

    # ---- the two exchanges -------------------------------------------------------------------------
    def _halo(self, which: int):
        """tensors of plane `which` that travel: engine.halo_tensors, or (legacy engines) planes of r and p"""
        if hasattr(self.e, "halo_tensors"):
            return self.e.halo_tensors(which)
        vecs = (VEC_R, VEC_P) if getattr(self.e, "needs_p_halo", lambda: True)() else (VEC_R,)
        return [self.e.plane_tensor(v, which) for v in vecs]

    def _post(self, lo_pairs, hi_pairs):
        """post sends / receives with the slab neighbours as ONE batch (one ncclGroup); returns the pending work"""
        ops: List[dist.P2POp] = []
        lo, hi = self.rank - 1, self.rank + 1
        if lo >= 0:
            for t, (snd, rcv) in enumerate(lo_pairs):
                ops.append(dist.P2POp(dist.isend, snd, lo, self.group, tag=2 * t))
                ops.append(dist.P2POp(dist.irecv, rcv, lo, self.group, tag=2 * t + 1))
        if hi < self.world:
            for t, (snd, rcv) in enumerate(hi_pairs):
                ops.append(dist.P2POp(dist.isend, snd, hi, self.group, tag=2 * t + 1))
                ops.append(dist.P2POp(dist.irecv, rcv, hi, self.group, tag=2 * t))
        if not ops:
            return []
        if ops[0].tensor.is_cuda and dist.get_backend(self.group) == "gloo":
            self._exchange_staged(ops)
            return []
        return dist.batch_isend_irecv(ops)

    def exchange_begin(self, halo=None, x_planes: bool = False):
        """first / last owned plane -> the neighbours' ghost planes.  Returns a token for exchange_end; between the two calls
        the engine may compute on everything that does not read a ghost plane."""
        if self.world == 1:
            return None
        if self.comm is not None:
            self.e.comm_exchange_begin(self.comm, x_planes)
            return ("native", None)
        e, nk = self.e, self.e.nk
        halo = halo or self._halo
        works = self._post(list(zip(halo(0), halo(-1))), list(zip(halo(nk - 1), halo(nk))))
        return ("planes", works)

    def exchange_end(self, token):
        if token is None:
            return
        kind, works = token
        if kind == "native":
            self.e.comm_exchange_end(self.comm)
            return
        for w in works:
            w.wait()          # stream-ordered for NCCL (the compute stream waits, the host does not); completes the copy for gloo

    def exchange_halos(self, halo=None, x_planes: bool = False):
        self.exchange_end(self.exchange_begin(halo, x_planes))

    def _exchange_staged(self, ops):
        """gloo cannot send device memory: stage the planes through host buffers.  Only for rehearsing the multi-rank
        driver with the HIP engine on a box with fewer GPUs than ranks (several ranks share one GPU, which RCCL refuses);
        the production backend is nccl."""
        torch.cuda.current_stream().synchronize()
        host_ops, recvs = [], []
        for op in ops:
            if op.op is dist.isend:
                host_ops.append(dist.P2POp(dist.isend, op.tensor.cpu(), op.peer, self.group, tag=op.tag))
            else:
                buf = torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
                recvs.append((op.tensor, buf))
                host_ops.append(dist.P2POp(dist.irecv, buf, op.peer, self.group, tag=op.tag))
        for w in dist.batch_isend_irecv(host_ops):
            w.wait()
        for dev, buf in recvs:
            dev.copy_(buf)

    def allreduce(self, which: int, slots=None):
        """local partial -> global scalar(s) (sum over ranks), in place; which: 0 after k1, 1 after k2 (classic form) / after a
        residual pass; slots = (a, b): that range of the scalar block instead"""
        if slots is None:
            self.e.local_to_global(which)
        if self.world > 1 or self.force_collectives:
            if slots is not None:
                a, b = slots
            elif hasattr(self.e, "global_slots"):
                a, b = self.e.global_slots(which)
            else:
                a = SC_PQ if which == 0 else SC_RR
                b = a + 1
            if self.comm is not None:
                self.e.comm_allreduce(self.comm, a, b - a)
                return
            t = self.e.scalars()[a:b]
            if t.is_cuda and dist.get_backend(self.group) == "gloo":       # rehearsal transport, see _exchange_staged
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    # ---- CG ---------------------------------------------------------------------------------------
    def start(self, x_fill: float = 0.0):
        """x = Dirichlet values / fill (ghost planes included: they are functions of the index), b = f m,
        r = P(b - A x), rr all-reduced"""
        self.e.begin(x_fill)
        self.e.residual()
        self.allreduce(1)
        self.iterations = 0

    def iterate(self, iters: int):
        # K1 needs r_n (z_n) and p_{n-1} on the ghost planes; engines whose K1 maintains p there itself only exchange r.
        # Engines that can split K1 run its bulk while the planes travel.
        split = self.overlap and hasattr(self.e, "k1_part")
        # engines whose single-reduction form updates r INSIDE k1 (elasticity PCG): k1_pre, then the exchange, then the two parts
        pre = getattr(self.e, "k1_pre", None) if self.sr else None
        if self.comm is not None and self.overlap and not self.force_collectives:
            self.e.iterate_slab(self.comm, iters)      # the same sequence, issued from C
            self.iterations += iters
            return
        # engines that can run K2 in two parts: the NEXT iteration's exchange goes out between them, as soon as the planes it carries are final,
        # and hides behind K2's bulk as well as K1's; nothing stays in flight across calls (the last K2 of a chunk posts nothing)
        early = split and pre is None and self.world > 1 and hasattr(self.e, "k2_part")
        posted = None
        for i in range(iters):
            if pre is not None:
                pre()
                token = self.exchange_begin()
                self.e.k1_part(0)
                self.exchange_end(token)
                self.e.k1_part(1)
            elif split:
                token = posted if posted is not None else self.exchange_begin()
                posted = None
                self.e.k1_part(0)
                self.exchange_end(token)
                self.e.k1_part(1)
            else:
                self.exchange_halos()
                self.e.k1()
            self.allreduce(0)                 # sr: {p.Ap, r.r, Ap.Ap} in one message; classic: p.Ap
            if early and i + 1 < iters:
                self.e.k2_part(0)
                posted = self.exchange_begin()
                self.e.k2_part(1)
            else:
                self.e.k2()
            if not self.sr:
                self.allreduce(1)
        self.iterations += iters

    def rr(self) -> float:
        """global r.r of the current residual (host value: synchronises).  Single-reduction form: the r.r of the last K2 is summed and
        all-reduced here, for the poll only."""
        if self.sr and getattr(self.e, "sr_rr_lags", True):
            self.e.sr_poll()
            self.allreduce(1, slots=(SC_RR_POLL, SC_RR_POLL + 1))
        return self.e.rr()

    # ---- one backward-Euler step on slabs (fenics_mcp_server.py:702-716) ---------------------------
    def begin_step(self) -> float:
        """x holds u^n on the owned planes: exchange its ghost planes, b = M (u^n + dt f), reference norm, r = P (b - A x).
        Returns ||P (b - A g)||^2 (host value; the one synchronisation of the step set-up)."""
        e = self.e
        self.exchange_halos(getattr(e, "x_halo_tensors"), x_planes=True)
        e.step_rhs()
        e.lift_norm()
        self.allreduce(1)
        bnorm2 = e.rr()
        e.reset_cg()
        e.residual()
        self.allreduce(1)
        return bnorm2

    def solve(self, bnorm2: float, rtol: float, max_iter: int = 100000, first_chunk: int = 8):
        """iterate until r.r <= rtol^2 bnorm2 (host polls once per chunk); returns (iterations, relative residual)"""
        target = rtol * rtol * bnorm2
        rr, it, chunk = self.rr(), 0, first_chunk
        while rr > target and it < max_iter:
            n = min(chunk, max_iter - it)
            self.iterate(n)
            it += n
            rr = self.rr()
            chunk = min(2 * chunk, 64)
        self.finish()
        return it, (rr / bnorm2) ** 0.5 if bnorm2 > 0 else rr ** 0.5

    def finish(self):
        self.e.flush()


def gather_slabs(local, n_planes: int, rank: int, world: int, group=None, device: int = 0):
    """owned planes of every rank -> rank 0, as fp64 tensors (no pickling): `local` is numpy [nk, ...]; returns the
    concatenation [n_planes, ...] on rank 0 and None elsewhere.  Slabs differ by at most one plane, so every rank sends a
    buffer of the largest slab (dist.gather needs equal sizes)."""
    import numpy as np
    if world == 1:
        return local
    counts = [slab_range(n_planes, world, r)[1] for r in range(world)]
    nmax = max(counts)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device(f"cuda:{device}") if on_gpu else torch.device("cpu")
    buf = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=torch.float64, device=dev)
    buf[: local.shape[0]].copy_(torch.from_numpy(np.ascontiguousarray(local)))
    recv = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, recv, dst=0, group=group)
    if rank != 0:
        return None
    return np.concatenate([recv[r][: counts[r]].cpu().numpy() for r in range(world)], axis=0)


def longest_axis_last(n_cells, lengths):
    """Permutation that makes the axis with the most cells the slowest one (z): the slabs then cut the longest axis and the halo
    planes are the smallest faces of the box (config 4, 1024 x 512 x 512: 513 x 513 planes instead of 1025 x 513).  The P1 split
    of a box (six Kuhn tetrahedra around the main diagonal) and a uniform Dirichlet boundary are invariant under a relabelling
    of the axes, so only n and L are permuted.  Returns (perm, n', L') with n'[i] = n[perm[i]]."""
    ax = max(range(3), key=lambda a: (n_cells[a], a))
    perm = [a for a in range(3) if a != ax] + [ax]
    return perm, tuple(n_cells[a] for a in perm), tuple(lengths[a] for a in perm)


def solve_heat_slabs(n_cells, lengths, diffusivity: float, T_boundary: float, T_initial: float = 0.0, dt: float = 0.0,
                     num_steps: int = 0, steady: bool = False, source: float = 0.0, rtol: float = 1e-10, max_iter: int = 100000,
                     snapshot_stride: int = 1, group=None, engine_factory=None, permute: bool = True, transport: str = "torch"):
    """`_solve_heat_3d_raw` (box, uniform Dirichlet value; fenics_mcp_server.py:475-762) over the ranks of the default process
    group, one slab per GPU: the backward-Euler loop (:702-716) with one x-halo exchange per step, the CG of every step through
    SlabCG.  Rank 0 returns {"values": [n_snap, nz+1, ny+1, nx+1] in the caller's axis order, "times", "iterations"}; the other
    ranks the iteration counts only.  engine_factory(grid_n, grid_L, heat_kwargs, k0, nk) -> engine replaces the HIP session
    (the CPU test-suite passes the numpy engine).  transport: "torch" — exchange and all-reduces through torch.distributed on `group` — or "library": issued by
    libpdegpu on its own RCCL communicator (NativeComm; HIP engines, backend nccl or a single rank)."""
    import numpy as np
    if transport not in ("torch", "library"):
        raise ValueError(f"transport must be torch or library, got {transport!r}")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    perm = [0, 1, 2]
    n_p, L_p = tuple(n_cells), tuple(lengths)
    if permute:
        perm, n_p, L_p = longest_axis_last(n_cells, lengths)
    k0, nk = slab_range(n_p[2] + 1, world, rank)
    hk = dict(kappa=diffusivity, dt=dt, steady=steady, source=source, T_boundary=T_boundary)
    local = 0
    stream = None
    if engine_factory is not None:
        eng = engine_factory(n_p, L_p, hk, k0, nk)
    else:
        local = int(L.os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0
        if world > 1 and dist.get_backend(group) == "gloo":
            local %= torch.cuda.device_count()
        grid = L.make_grid(3, n_p, L_p)
        heat = L.Heat()
        heat.kappa, heat.dt, heat.steady, heat.source = diffusivity, dt if not steady else 0.0, 1 if steady else 0, source
        heat.bc_mode, heat.fast_path = L.BC_UNIFORM, 1
        heat.bc_present[0], heat.bc_value[0] = 1, T_boundary
        stream = torch.cuda.Stream(device=local)
        torch.cuda.set_device(local)
        eng = GpuSlabEngine(grid, heat, k0, nk, local, stream)

    def run():
        comm = NativeComm(eng, rank, world, group) if transport == "library" else None
        try:
            return run_with(comm)
        finally:
            if comm is not None:
                comm.close()

    def run_with(comm):
        cg = SlabCG(eng, rank, world, group, comm=comm)
        snaps, times, iters = [], [], []

        def snapshot(t):
            part = gather_slabs(eng.owned_x(), n_p[2] + 1, rank, world, group, local)
            if rank == 0:
                # back to the caller's axis order: array axes are (z', y', x') = (perm[2], perm[1], perm[0]) of the original
                src = [2 - perm[2], 2 - perm[1], 2 - perm[0]]          # original array axis held by each current axis
                snaps.append(np.ascontiguousarray(np.transpose(part, np.argsort(src))))
                times.append(t)

        if steady:
            cg.start(0.0)
            bnorm2 = eng.rr()
            it, _ = cg.solve(bnorm2, rtol, max_iter)
            iters.append(it)
            snapshot(0.0)
        else:
            eng.begin(T_initial)
            snapshot(0.0)
            for step in range(1, num_steps + 1):
                bnorm2 = cg.begin_step()
                it, _ = cg.solve(bnorm2, rtol, max_iter)
                iters.append(it)
                if step % snapshot_stride == 0 or step == num_steps:
                    snapshot(step * dt)
        out = {"iterations": iters}
        if rank == 0:
            out.update(values=np.stack(snaps), times=np.array(times))
        return out

    if stream is not None:
        with torch.cuda.stream(stream):
            out = run()
            eng.close()
    else:
        out = run()
    return out


def solve_elasticity_slabs(grid, el, n_cells, rtol: float = 1e-8, max_iter: int = 400000, group=None, transport: str = "torch"):
    """3-D static elasticity (config 5) over the ranks of the default process group, one z-slab per GPU: single-reduction Jacobi-PCG, halo
    exchange of r (three component planes per neighbour) inside k1, ONE 32-byte all-reduce per iteration; every rank returns the iteration
    count, rank 0 additionally the gathered displacement [N,3] and the projected von Mises field [N] (post-processing on its GPU).
    transport: "torch" (torch.distributed on `group`) or "library" (libpdegpu's own RCCL communicator, see solve_heat_slabs)."""
    import numpy as np
    if transport not in ("torch", "library"):
        raise ValueError(f"transport must be torch or library, got {transport!r}")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    local = int(L.os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0
    if world > 1 and dist.get_backend(group) == "gloo":       # rehearsal: ranks may share a device (see SlabCG._exchange_staged)
        local %= torch.cuda.device_count()
    nx, ny, nz = n_cells
    k0, nk = slab_range(nz + 1, world, rank)
    stream = torch.cuda.Stream(device=local)
    with torch.cuda.stream(stream):
        eng = GpuElasticitySlabEngine(grid, el, k0, nk, local, stream)
        comm = None
        try:
            comm = NativeComm(eng, rank, world, group) if transport == "library" else None
            cg = SlabCG(eng, rank, world, group, comm=comm)
            cg.start(0.0)
            rr0 = eng.rr()
            it, chunk = 0, 16
            rr = rr0
            while rr > rtol * rtol * rr0 and it < max_iter:
                cg.iterate(chunk)
                it += chunk
                rr = eng.rr()                       # the only host synchronisation: once per chunk
                chunk = min(2 * chunk, 256)
            cg.finish()
            u_local = eng.get_u((nx + 1) * (ny + 1)).reshape(nk, (nx + 1) * (ny + 1), 3)
            u_all = gather_slabs(u_local, nz + 1, rank, world, group, local)
            out = {"iterations": it, "rel_residual": (rr / rr0) ** 0.5 if rr0 > 0 else 0.0}
            if rank == 0:
                u = np.ascontiguousarray(u_all.reshape(-1, 3))
                field = np.empty(u.shape[0])
                ctx = eng.ctx
                L.check(ctx, eng.lib.pdegpu_elasticity_postprocess(ctx, grid, el, L.dptr(u), L.dptr(field)))
                out.update(displacement=u, field=field)
        finally:                                    # the communicator goes before the context it was made on, whatever happened
            if comm is not None:
                comm.close()
            eng.close()
    return out
