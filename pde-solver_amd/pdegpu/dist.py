"""Slab-decomposed CG across GPUs of one node: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI) for the only two exchanges the path has —

  * halo: before each operator application every rank sends its first / last owned node plane of r and p
    to the lower / upper neighbour and receives their planes into its ghost planes.  Planes are contiguous
    in the pitched layout ((nx+1 padded) x (ny+1) fp64 = 2.2 MB at 513^2), so there is no pack kernel; the
    four point-to-point transfers of one exchange are issued as one batch (one ncclGroup) and use the two
    xGMI links to the slab neighbours;
  * dot products: one 8-byte all-reduce after K1 (p.q) and one after K2 (r.r).  They are latency-bound, the
    result stays on the device, alpha / beta are formed inside the kernels: the host never synchronises in
    the loop.

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


class GpuSlabEngine:
    """pdegpu_session on one GPU; device memory stays owned by the library, torch only sees plane views."""

    def __init__(self, grid, heat, k0: int, nk: int, device: int, stream: Optional[torch.cuda.Stream] = None):
        self.lib = L.load_library()
        self.device = device
        self.stream = stream
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

    def residual(self):
        self._ck(self.lib.pdegpu_session_residual(self.s))

    def k1(self):
        self._ck(self.lib.pdegpu_session_k1(self.s))

    def k2(self):
        self._ck(self.lib.pdegpu_session_k2(self.s))

    def flush(self):
        self._ck(self.lib.pdegpu_session_flush(self.s))

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

    def iterate_native(self, iters: int, time_kernels: bool, stats=None):
        self._ck(self.lib.pdegpu_session_iterate(self.s, iters, 1 if time_kernels else 0, stats))

    def rr(self) -> float:
        v = C.c_double()
        self._ck(self.lib.pdegpu_session_rr(self.s, C.byref(v)))
        return v.value

    def sync(self):
        self._ck(self.lib.pdegpu_session_sync(self.s))


class SlabCG:
    """CG over a 1-D slab decomposition.  `engine` owns planes [k0, k0+nk) plus one ghost plane per side."""

    def __init__(self, engine, rank: int, world: int, group=None):
        self.e, self.rank, self.world, self.group = engine, rank, world, group
        self.iterations = 0

    # ---- the two exchanges -------------------------------------------------------------------------
    def exchange_halos(self, vecs=(VEC_R, VEC_P)):
        if self.world == 1:
            return
        ops: List[dist.P2POp] = []
        lo, hi = self.rank - 1, self.rank + 1
        for v in vecs:
            if lo >= 0:
                ops.append(dist.P2POp(dist.isend, self.e.plane_tensor(v, 0), lo, self.group, tag=2 * v))
                ops.append(dist.P2POp(dist.irecv, self.e.plane_tensor(v, -1), lo, self.group, tag=2 * v + 1))
            if hi < self.world:
                ops.append(dist.P2POp(dist.isend, self.e.plane_tensor(v, self.e.nk - 1), hi, self.group, tag=2 * v + 1))
                ops.append(dist.P2POp(dist.irecv, self.e.plane_tensor(v, self.e.nk), hi, self.group, tag=2 * v))
        for w in dist.batch_isend_irecv(ops):
            w.wait()          # stream-ordered for NCCL (no host block); completes the copy for gloo

    def allreduce(self, which: int):
        """local partial -> global scalar (sum over ranks); which: 0 = p.q, 1 = r.r"""
        self.e.local_to_global(which)
        if self.world > 1:
            slot = SC_PQ if which == 0 else SC_RR
            dist.all_reduce(self.e.scalars()[slot:slot + 1], op=dist.ReduceOp.SUM, group=self.group)

    # ---- CG ---------------------------------------------------------------------------------------
    def start(self, x_fill: float = 0.0):
        """x = Dirichlet values / fill (ghost planes included: they are functions of the index), b = f m,
        r = P(b - A x), rr all-reduced"""
        self.e.begin(x_fill)
        self.e.residual()
        self.allreduce(1)
        self.iterations = 0

    def iterate(self, iters: int):
        # K1 needs r_n and p_{n-1} on the ghost planes; engines whose K1 maintains p there itself only exchange r
        vecs = (VEC_R, VEC_P) if getattr(self.e, "needs_p_halo", lambda: True)() else (VEC_R,)
        for _ in range(iters):
            self.exchange_halos(vecs)
            self.e.k1()
            self.allreduce(0)
            self.e.k2()
            self.allreduce(1)
        self.iterations += iters

    def finish(self):
        self.e.flush()
