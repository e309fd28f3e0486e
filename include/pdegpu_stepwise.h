/* pdegpu_stepwise.h — the DRIVER / REHEARSAL API of libpdegpu.so: the single steps of a CG / PCG iteration on a slab.
 *
 * include/pdegpu.h holds what a binding of the reference's solve path needs: the tool-level solves (fenics_mcp_server.py:204-762, :1470-1892), the
 * slab sessions as whole solves / chunks of iterations (pdegpu_session_iterate, pdegpu_session_iterate_slab with the library's own RCCL communicator).
 * THIS header exposes the pieces those loops are made of, for hosts that drive an iteration themselves:
 *   - pdegpu/dist.py (SlabCG): the same loop issued from Python with torch.distributed as transport — the configuration bench.py measures at N > 1;
 *   - rehearsals of the multi-rank protocol on one GPU (tests/test_gpu_multirank.py, test_gpu_native_comm.py, tools/slab_surrogate.py): several
 *     slabs in one process, sends to self, staged planes;
 *   - diagnostics (kernel placement probe).
 * Nothing here is needed to replace the reference's `solve(a == L, u, bcs)`; every function keeps the conventions of pdegpu.h (0 / negative status,
 * nothing thrown, work enqueued on the context stream without host synchronisation unless stated).  The reference has no counterpart: its only
 * MPI-adjacent line is fenics_mcp_server.py:141. */
#ifndef PDEGPU_STEPWISE_H
#define PDEGPU_STEPWISE_H
#include "pdegpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- heat session (pdegpu_session): one CG iteration = k1 (x += a p; p = r + beta p; q = A p; p.q) + k2 (r -= alpha q; r.r) ---------------- */
/* device pointers / geometry the host needs for halo exchange: plane `which` of vector `vec`
 * (0:x 1:r 2:p(current) 3:q), which = -1 lower ghost, 0 first owned, nk-1 last owned, nk upper ghost.
 * plane_doubles = number of fp64 in one contiguous plane. */
void*   pdegpu_session_plane_ptr(pdegpu_session* s, int vec, int64_t which);
int64_t pdegpu_session_plane_doubles(const pdegpu_session* s);
/* 1 if the ghost planes of p must be exchanged together with r before every k1 (class-table kernels and the 72-B
 * variant); 0 when k1 maintains p on the ghost planes itself (fused z-marching kernels, cg_variant 1): then only
 * r is exchanged. */
int     pdegpu_session_needs_p_halo(const pdegpu_session* s);
/* pointer to the device scalars {pq_local, rr_local, pq_global, rr_global, rr_global_prev, ...}.  residual / k1 /
 * k2 leave this slab's dot product in BOTH the *_local and the *_global slot; the multi-GPU host all-reduces
 * the *_global slot in place (8 bytes, stays on the device); single GPU: nothing to do. */
double* pdegpu_session_scalars(pdegpu_session* s);
int  pdegpu_session_k1(pdegpu_session* s);                  /* x+=a*p; p=r+beta*p; q=A p; pq_local */
int  pdegpu_session_k2(pdegpu_session* s);                  /* r-=alpha*q; rr_local */
/* k1 in two steps, so that the halo exchange overlaps the bulk of it: part 0 = the owned planes whose stencil touches no ghost
 * plane (call it right after posting the exchange), part 1 = the one or two planes next to the ghosts (after the exchange has
 * completed) + the sum of all partials -> pq_local.  can_split_k1 = 1 for the fused z-marching kernels; other sessions do
 * nothing in part 0 and the whole k1 in part 1, so a driver may always call part 0 then part 1. */
int  pdegpu_session_can_split_k1(const pdegpu_session* s);
int  pdegpu_session_k1_part(pdegpu_session* s, int part);
/* K2 in two steps, so that the NEXT iteration's exchange is posted as soon as its planes are final: part 0 = the one or two planes next to the
 * ghosts (one launch) — after it the first / last owned plane of r belong to the new residual: post the exchange —, part 1 = the planes in between
 * + k2's reduce (classic form).  The transfer then hides behind K2's bulk and the next K1's bulk (pdegpu_session_iterate_slab and pdegpu.dist.SlabCG do
 * this inside a chunk of iterations).  can_split_k2 = 1 for marching sessions whose K2 is the stencil form, on a slab with a neighbour, thin enough for the
 * exchange to need more than K1's bulk to hide behind (K1 bulk estimate < 4 plane transfers: ranks of 4 and 8 on 512^3; the split costs a launch); other
 * sessions run all of K2 in part 0 and nothing in part 1, so a driver may always call 0, post, 1. */
int  pdegpu_session_can_split_k2(const pdegpu_session* s);
int  pdegpu_session_k2_part(pdegpu_session* s, int part);
/* re-copies this slab's local dot product into the global slot (undoes an all-reduce in tests) */
int  pdegpu_session_reduce_local(pdegpu_session* s, int which /*0 pq, 1 rr*/, int copy_to_global);
/* Diagnostic: average milliseconds of `reps` launches of the fused CG kernel K1 with its four vectors (x, r, p_old, p_new) taken from the session's
 * six vectors {x, r, p0, p1, q, b} in the order roles4 gives (distinct indices 0..5).  K1's time depends on where its vectors sit in physical
 * memory (profiles/r02_k1_bimodal_between_processes.log); single-slab marching sessions only; the session must be begun again afterwards. */
int         pdegpu_session_debug_time_k1(pdegpu_session* s, const int32_t* roles4, int reps, double* avg_ms_out);

/* ---- elasticity session (pdegpu_el_session) ------------------------------------------------------------------------------------------------ */
int     pdegpu_el_session_k1(pdegpu_el_session* s);         /* x += a p; p = D^-1 r + beta p; q = A p; local p.q */
int     pdegpu_el_session_k2(pdegpu_el_session* s);         /* r -= a q; local r.z (z = D^-1 r), r.r */
double* pdegpu_el_session_scalars(pdegpu_el_session* s);    /* device {p.q, r.z, r.r, r.z_prev, ...} */
int64_t pdegpu_el_session_plane_doubles(const pdegpu_el_session* s);
void*   pdegpu_el_session_plane_ptr(pdegpu_el_session* s, int vec, int comp, int64_t which);
/* k1 in two steps so that the halo exchange overlaps the part of it that reads no ghost plane (same protocol as pdegpu_session_k1_part):
 * part 0 = p-update of the owned planes + the marching apply on the planes away from the ghosts (call it right after posting the exchange),
 * part 1 = p-update of the ghost planes + the apply on the one or two edge planes (one launch) + the p.q sum.  can_split_k1 = 1 for 3-D
 * marching sessions on slabs with a neighbour whose vectors hold at least el_split_min_mb (32) MiB; other sessions do nothing in part 0 and the whole k1 in part 1. */
int     pdegpu_el_session_can_split_k1(const pdegpu_el_session* s);
int     pdegpu_el_session_k1_part(pdegpu_el_session* s, int part);
/* single-reduction form: the update pass of the owned planes, before the exchange of r (a no-op in the classic form) */
int     pdegpu_el_session_k1_pre(pdegpu_el_session* s);

/* ---- building blocks of the library-issued exchange (csrc/comm.hip; protocol: the comment above pdegpu_comm_create in pdegpu.h) ------------ */
/* one-GPU rehearsal (tools/slab_surrogate.py): on != 0 makes both slab neighbours this rank itself — real messages of the real size, meaningless iterates */
int  pdegpu_comm_set_self_loop(pdegpu_comm* c, int on);
int  pdegpu_comm_group_begin(pdegpu_comm* c);               /* comm stream waits for the context stream; ncclGroupStart */
int  pdegpu_comm_group_end(pdegpu_comm* c);                 /* ncclGroupEnd; marks the end of the exchange for exchange_end */
int  pdegpu_session_exchange_post(pdegpu_session* s, pdegpu_comm* c, int what, int x_planes, int lo_rank, int hi_rank);
int  pdegpu_session_exchange_begin(pdegpu_session* s, pdegpu_comm* c, int x_planes);
int  pdegpu_session_exchange_end(pdegpu_session* s, pdegpu_comm* c);
int  pdegpu_el_session_exchange_post(pdegpu_el_session* s, pdegpu_comm* c, int what, int lo_rank, int hi_rank);
int  pdegpu_el_session_exchange_begin(pdegpu_el_session* s, pdegpu_comm* c);
int  pdegpu_el_session_exchange_end(pdegpu_el_session* s, pdegpu_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* PDEGPU_STEPWISE_H */
