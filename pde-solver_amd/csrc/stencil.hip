// stencil.hip — matrix-free operator application for the heat path (gfx950).
//
// Replaces DOLFIN's assembled sparse matrix + `solve(a == L, u, bcs)` inner products
// (fenics_mcp_server.py:265, 311, 397, 440, 661, 709): the operator A = cM*M + cK*K is never stored;
// it is applied as a constant-coefficient node stencil (7-point stiffness / 15-point consistent mass for
// the reference's P1 simplex split, 27-point for Q1) over full-grid fp64 vectors in HBM.
//
// Two implementations of the same three operations (PLAIN apply, RESIDUAL, fused CG "K1"):
//   * general: one thread per node, coupling weights looked up per boundary class from the 27x27 class
//     table in LDS (any dimension, any Dirichlet/natural face mix, P1 or Q1);
//   * fast: 3-D (2-D through a flat view), P1 split (7- / 15-point) or Q1 (27-point), interior nodes.  One wavefront owns a
//     128-node x RY-row column (RY = 2 by default) and marches along z keeping three planes in registers, so every
//     input plane is read once per wave; x-neighbours come from DPP wave shifts (tile edges: LDS hand-off between the
//     workgroup's waves in the CG kernels), y/z neighbours from registers; loads and stores are 16 B per lane
//     (global_load/store_dwordx4), fully coalesced 1 KiB per wave-instruction.  Workgroups are mapped so that one XCD
//     sweeps one z-chunk of the whole x-y plane: the y-halo rows shared by neighbouring workgroups are then served by
//     that XCD's L2.  Grids with natural faces add a second launch of the class-table kernel over the boundary nodes.
//
// CG K1 (both implementations) fuses, per iteration n:  x += alpha_{n-1} p_{n-1};  p_n = r_n + beta_{n-1}
// p_{n-1};  q_n = A p_n;  partial(p_n . q_n)  — p_n is recomputed on the fly at the stencil neighbours
// from r_n and p_{n-1}, which removes the separate p-update pass (72 B/DoF per CG iteration in total).
// CG K2 (fast path, cg_variant 1): r -= alpha (A p) with the SAME marching stencil applied to the stored p_n, so
// K1 does not have to store q at all: K1 = read r, p, x / write p, x (40 B/DoF), K2 = read r, p / write r
// (24 B/DoF) -> 64 B/DoF per iteration.  K1 also stores p_n on the slab's ghost planes (it has r_n and p_{n-1}
// there), so in the multi-GPU loop only r needs a halo exchange.
#include "common.h"
#include "dot_finish.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace {

// ---------------------------------------------------------------------------------------------
// reductions: per-lane partials -> wavefront (__shfl_down over 64 lanes) -> LDS across the 4 waves
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;   // valid in lane 0
}

__device__ __forceinline__ double block_sum_256(double v, double* sm) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);   // fixed order: bitwise reproducible
}

// ---------------------------------------------------------------------------------------------
// general path
// ---------------------------------------------------------------------------------------------
template <int MODE, bool ELL>
__global__ __launch_bounds__(256) void k_general(StencilArgs a) {
    __shared__ double sTab[ELL ? 1 : 27 * 27];
    __shared__ double sRed[4];
    if (!ELL) {
        for (int e = threadIdx.x; e < 27 * 27; e += 256) sTab[e] = a.tabA[e];
        __syncthreads();
    }

    const GridDev& g = a.g;
    const bool bnd = a.bnd_n > 0;
    const long long total = bnd ? a.bnd_n : (long long)g.nkl * g.PP;
    double alpha_prev = 0.0, beta = 0.0, res_scale = 1.0;
    if (MODE == APPLY_CG_K1) cg_scalars_in(a.di, a.scal, sRed, &alpha_prev, &beta);
    if (MODE == APPLY_CG_K2) res_scale = cg_alpha_in(a.di, a.scal, sRed);      // r -= alpha A p on the boundary nodes of a hybrid session
    // x_in_k2 sessions (hybrid: this is the boundary launch next to the marching one, which records the same two scalars): K1 leaves x alone, the
    // K2 of every other iteration updates it for two iterations at once (see k_fast, dot_finish.h)
    const bool k2x = MODE == APPLY_CG_K2 && a.x_in_k2 && a.xmode == 2;
    double cx_p = 0.0, cx_r = 0.0, cx_o = 0.0;
    if (MODE == APPLY_CG_K1 && a.x_in_k2 && a.xd && blockIdx.x == 0 && threadIdx.x == 0) a.xd[1] = beta;
    if (MODE == APPLY_CG_K2 && a.x_in_k2 && !k2x && a.xd && blockIdx.x == 0 && threadIdx.x == 0) a.xd[0] = res_scale;
    if (k2x) xk2_coefficients(a.xd, a.xmode, res_scale, &cx_p, &cx_r, &cx_o);

    double acc = 0.0, acc2 = 0.0;                              // acc2 (K1): (A p).(A p) of the single-reduction form
    for (long long e0 = (long long)blockIdx.x * 256 + threadIdx.x; e0 < total; e0 += (long long)gridDim.x * 256) {
        long long idx = e0, row_idx = 0;                      // row_idx: j + NY * lk (the non-boundary enumeration; weighted operators)
        int i, j, lk;
        if (bnd) {
            // boundary node e0 of the slab: the full planes z = 0 / NZ-1 first, then the ring of every other plane
            // 32-bit arithmetic (64-bit integer division costs ~150 instructions on this ISA; the host only uses this mode
            // while bnd_n < 2^31)
            unsigned e = (unsigned)e0;
            const unsigned plane_nodes = (unsigned)g.NX * (unsigned)g.NY;
            if (a.bnd_full_lo && e < plane_nodes) {
                lk = -g.k0;
                j = (int)(e / (unsigned)g.NX);
                i = (int)(e % (unsigned)g.NX);
            } else {
                e -= a.bnd_full_lo ? plane_nodes : 0u;
                if (a.bnd_full_hi && e < plane_nodes) {
                    lk = g.NZ - 1 - g.k0;
                    j = (int)(e / (unsigned)g.NX);
                    i = (int)(e % (unsigned)g.NX);
                } else {
                    e -= a.bnd_full_hi ? plane_nodes : 0u;
                    lk = a.bnd_lkb + (int)(e / (unsigned)a.bnd_ring);
                    const int t = (int)(e % (unsigned)a.bnd_ring);
                    if (t < g.NX) { j = 0; i = t; }
                    else if (t < 2 * g.NX) { j = g.NY - 1; i = t - g.NX; }
                    else if (a.bnd_xcols == 1) { j = 1 + (t - 2 * g.NX); i = g.NX - 1; }      // only the column x = NX-1 is left to this pass
                    else { j = 1 + (t - 2 * g.NX) / 2; i = ((t - 2 * g.NX) & 1) ? g.NX - 1 : 0; }
                }
            }
            idx = (long long)lk * g.PP + (long long)j * g.PX + i;
        } else if (total <= 0x7fffffffLL) {
            // 32-bit index arithmetic (two 64-bit divisions with remainders cost ~600 instructions per node on this ISA — more than the rest of the loop body)
            const unsigned e = (unsigned)idx;
            const unsigned row = e / (unsigned)g.PX;
            i = (int)(e - row * (unsigned)g.PX);
            lk = (int)(row / (unsigned)g.NY);
            j = (int)(row - (unsigned)lk * (unsigned)g.NY);
            row_idx = (long long)row;
        } else {
            i = (int)(idx % g.PX);
            const long long row = idx / g.PX;
            j = (int)(row % g.NY);
            lk = (int)(row / g.NY);
            row_idx = row;
        }
        const int kg = g.k0 + lk;
        const bool real = i < g.NX;
        double dval = 0.0;
        const bool dir = real && a.mask_dirichlet && pg_dirichlet(g, i, j, kg, &dval);

        double pn = 0.0;
        if (MODE == APPLY_CG_K1) {
            const double po = a.p_old[idx];
            pn = a.in[idx] + beta * po;
            // hybrid: the marching kernel owns p / x of every node of its rows (interior j and k), x-faces included
            const bool theirs = bnd && j >= 1 && j <= g.NY - 2 && kg >= 1 && kg <= g.NZ - 2 && i < a.bnd_xcover;
            if (!theirs) {
                a.p_new[idx] = pn;
                if (!a.x_in_k2 && alpha_prev != 0.0) a.x[idx] += alpha_prev * po;
            }
        }
        // Neighbour loads are unconditional and separated from their use: every neighbour address of every index of the slab lies
        // inside the allocation (ghost planes, row pads, guards: all finite) and meets a weight that is exactly 0 when the neighbour
        // does not exist.  A load inside `if (w != 0)` is waited for inside that branch — 15 to 27 dependent memory round trips
        // per node; this way they are all in flight together.  a.off_mask = offsets that carry a non-zero weight in ANY class
        // (kernel-uniform: scalar branches, no waits).
        double s = 0.0;
        if (ELL) {
            double wq[15], vi[15], vp[15];
            // where the weight does not depend on one coordinate the coefficient rows repeat along it: three representative planes (mode 1: 1 / nz of
            // the 120 B per node, cache-resident) or three entries per grid row (mode 2: one address per wave, a broadcast) stand for all of them
            long long eidx = idx;
            if (a.ell_mode == 1) {
                eidx = (long long)(kg == 0 ? 0 : (kg == g.NZ - 1 ? 2 : 1)) * g.PP + (idx - (long long)lk * g.PP);
            } else if (a.ell_mode == 2) {
                eidx = 3 * row_idx + (i == 0 ? 0 : (i >= g.NX - 1 ? 2 : 1));
            }
#pragma unroll
            for (int q = 0; q < 15; ++q) {
                constexpr int kOff[15] = {13, 12, 14, 10, 16, 4, 22, 9, 17, 3, 23, 1, 25, 0, 26};
                const int o = kOff[q];
                const int dx = o % 3 - 1, dy = (o / 3) % 3 - 1, dz = o / 9 - 1;
                const long long nb = idx + (long long)dz * g.PP + (long long)dy * g.PX + dx;
                wq[q] = a.ell[q * a.ell_stride + eidx];
                vi[q] = a.in[nb];
                vp[q] = (MODE == APPLY_CG_K1) ? a.p_old[nb] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 15; ++q) s += wq[q] * ((MODE == APPLY_CG_K1) ? vi[q] + beta * vp[q] : vi[q] + a.in_shift);
            if (!(real && !dir)) s = 0.0;
        } else {
            const int cls = real ? pg_axis_class(i, g.NX) + 3 * (g.dim >= 2 ? pg_axis_class(j, g.NY) : 1) +
                                       9 * (g.dim >= 3 ? pg_axis_class(kg, g.NZ) : 1) : 13;
            const double* w = &sTab[cls * 27];
            double vi[27], vp[27];
#pragma unroll
            for (int o = 0; o < 27; ++o) {
                vi[o] = 0.0;
                vp[o] = 0.0;
                if ((a.off_mask >> o) & 1u) {
                    const int dx = o % 3 - 1, dy = (o / 3) % 3 - 1, dz = o / 9 - 1;
                    const long long nb = idx + (long long)dz * g.PP + (long long)dy * g.PX + dx;
                    vi[o] = a.in[nb];
                    if (MODE == APPLY_CG_K1) vp[o] = a.p_old[nb];
                }
            }
#pragma unroll
            for (int o = 0; o < 27; ++o) {
                // (a weight 0 must give exactly 0 even next to the shifted value of a PLAIN pass)
                const double v = (MODE == APPLY_CG_K1) ? vi[o] + beta * vp[o] : vi[o] + a.in_shift;
                s += w[o] * v;
            }
            if (!(real && !dir)) s = 0.0;
        }
        if (MODE == APPLY_PLAIN) {
            a.out[idx] = s;
            if (real) acc += a.in[idx] * s;
        } else if (MODE == APPLY_RESIDUAL || MODE == APPLY_CG_K2) {
            double rv = 0.0;
            const double r_in = a.rhs ? a.rhs[idx] : a.rhs_const;
            if (real && !dir) rv = r_in - res_scale * s;
            if (k2x) {
                // the nodes whose p / x this launch owns (as in K1): p_n and r_n vanish on Dirichlet nodes and pads, which then keep their x
                const bool theirs = bnd && j >= 1 && j <= g.NY - 2 && kg >= 1 && kg <= g.NZ - 2 && i < a.bnd_xcover;
                if (!theirs && real) {
                    double xv = a.x[idx] + cx_p * a.in[idx] - cx_r * r_in;
                    if (cx_o != 0.0) xv += cx_o * a.p_old[idx];
                    a.x[idx] = xv;
                }
            }
            a.out[idx] = rv;
            acc += rv * rv;
        } else {
            if (a.out) a.out[idx] = s;                        // (64-B variant: q is not stored, K2 applies the operator again)
            acc += pn * s;
            acc2 += s * s;
        }
    }
    const double tot = block_sum_256(acc, sRed);
    dot_finish(tot, a.partials);
    if (MODE == APPLY_CG_K1 && a.partials2) {                 // kernel-uniform
        __syncthreads();
        dot_finish(block_sum_256(acc2, sRed), a.partials2);
    }
}

// ---------------------------------------------------------------------------------------------
// fast path: z-marching, registers + DPP
// ---------------------------------------------------------------------------------------------
struct FastGeom {
    int ntx, nty;          // 128-node x-tiles, RY-row y-tiles covering the interior
    int wxt, wyt;          // waves of a workgroup along x / y (wxt*wyt == 4); linear: unused
    int nxg, nyg, nzc;     // workgroup counts along x, y; z-chunks (linear: nxg = workgroups per z-chunk, nyg = 1)
    int linear;            // 1: the waves of a z-chunk are numbered row by row (w = yt * ntx + xt) and a workgroup takes four consecutive ones
    int lkb, lke;          // local interior plane range [lkb, lke)
    int lkb2, lke2;        // a second, disjoint plane range of the same launch: z-chunks nzc1 .. nzc-1 (the two edge planes of a slab)
    int nzc1;              // z-chunks of the first range
    int zlen;              // planes per z-chunk
    int jbase, jmax;       // first / last free row of the middle axis (1 .. NY-2; 0 .. 0 for the flat 2-D view)
};

// lane l receives the value of lane l-1 (lane 0 receives 0): v_mov_b32_dpp wave_shr:1
__device__ __forceinline__ double dpp_from_lower(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// lane l receives the value of lane l+1 (lane 63 receives 0): v_mov_b32_dpp wave_shl:1
__device__ __forceinline__ double dpp_from_upper(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

typedef double dbl2 __attribute__((ext_vector_type(2)));

template <int NT>
__device__ __forceinline__ void store_pair(double* p, const double2& v) {
    if (NT & 1) {
        dbl2 t = {v.x, v.y};
        __builtin_nontemporal_store(t, reinterpret_cast<dbl2*>(p));
    } else {
        *reinterpret_cast<double2*>(p) = v;
    }
}
template <int NT>
__device__ __forceinline__ double2 load_pair_x(const double* p) {
    if (NT & 2) {
        dbl2 t = __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(p));
        return make_double2(t.x, t.y);
    }
    return *reinterpret_cast<const double2*>(p);
}

// EX = 1: the tile-edge x-neighbours are handed between the x-adjacent wavefronts of the workgroup through 512 B of
// LDS (one barrier per z-step; the waves march in lockstep anyway) instead of 2-lane global loads; columns outside
// the free range of a CG vector (Dirichlet / beyond the grid) need no load at all.
// PF = 1 (bare apply / residual / stencil K2 only): the loads of plane lk+2 are issued before the stencil of plane lk, so a wave
// waits for a plane one whole z-step after requesting it (these modes hold one vector, the registers are there).
// XU (x_in_k2 sessions: the solution vector is updated by K2, every other iteration): K1 — 1 = x += alpha p here (the classic place), 0 = K1 never
// touches x (24 instead of 40 B/DoF) and records its beta for K2; K2 — 1 = this launch also does x += cx_p p - cx_r r_in (+ cx_o p_other): the
// pending update of the iteration before (its direction rebuilt from p_n = r_n + beta_n p_{n-1}) together with this iteration's; 0 = no x
// (a x_in_k2 session's K2 of the iterations in between records its alpha instead).
template <int NPT, int RY, int MODE, int NT, int EX, int PF, int XU>
__global__ __launch_bounds__(256, 2) void k_fast(StencilArgs a, FastGeom fg) {
    __shared__ double sRed[4];
    __shared__ double sEdge[EX ? 2 : 1][4][RY + 2][2];
    const GridDev& g = a.g;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr bool K1 = (MODE == APPLY_CG_K1);
    constexpr bool K2 = (MODE == APPLY_CG_K2);
    static_assert(!(PF && MODE == APPLY_CG_K1), "the deeper prefetch is for the one-vector modes");
    constexpr bool K1X = K1 && XU != 0;        // K1 with the x update
    constexpr bool K2X = K2 && XU != 0;        // K2 with the x update
    constexpr bool RES = (MODE == APPLY_RESIDUAL) || K2;      // out = rhs - scale * (A in)
    constexpr bool W15 = (NPT == 15) || (NPT == 27);     // every held row of every plane is needed (x-diagonals)
    constexpr bool W27 = (NPT == 27);                     // Q1: all 26 neighbours, weights symmetric under every sign flip
    constexpr int RH = RY + 2;                 // rows held per plane (own rows + one halo row each side)

    // XCD-aware workgroup -> tile map: blocks b and b+8 share an XCD (round-robin dispatch, speed only), so
    // XCD x gets the contiguous work range [x*G/8, (x+1)*G/8): neighbouring y-tiles of one z-chunk.
    const int G = gridDim.x;
    int wid = blockIdx.x;
    if ((G & 7) == 0) wid = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int xg = wid % fg.nxg, yg = (wid / fg.nxg) % fg.nyg, zc = wid / (fg.nxg * fg.nyg);
    int xt = xg * fg.wxt + (wv % fg.wxt), yt = yg * fg.wyt + (wv / fg.wxt);
    if (fg.linear) {                                         // four consecutive tiles of the row-major tile order, whatever the tile count per row
        const int w = xg * 4 + wv;
        xt = w % fg.ntx;
        yt = w / fg.ntx;
    }
    const int i0 = xt * 128 + 2 * lane;
    const int j0 = fg.jbase + yt * RY;
    const bool range2 = zc >= fg.nzc1;
    const int lk0 = range2 ? fg.lkb2 + (zc - fg.nzc1) * fg.zlen : fg.lkb + zc * fg.zlen;
    const int lk1 = min(lk0 + fg.zlen, range2 ? fg.lke2 : fg.lke);
    const bool active = (xt < fg.ntx) && (yt < fg.nty) && (lk0 < lk1);
    const int wave_x = fg.linear ? wv : wv % fg.wxt;         // position of this wave among the workgroup's x-adjacent tiles
    int estep = 0;                                            // LDS edge-exchange counter (same in every wave of the workgroup)

    double acc = 0.0, acc2 = 0.0;                              // acc2 (K1): (A p).(A p) of the single-reduction form
    // CG scalars: every thread of every workgroup takes part (the prologue sum of the previous launch's partials has barriers)
    double alpha_prev = 0.0, beta = 0.0, res_scale = 1.0;
    if (K1) cg_scalars_in(a.di, a.scal, sRed, &alpha_prev, &beta);
    if (K2) res_scale = cg_alpha_in(a.di, a.scal, sRed);      // alpha_n = rr_n / (p_n . A p_n)
    // x_in_k2 sessions (a.xd != nullptr): xd[0] = alpha of the K2 that left its x update out, xd[1] = beta of the latest K1 (slots no workgroup
    // of the recording launch reads)
    double cx_p = 0.0, cx_r = 0.0, cx_o = 0.0;                // K2X: x += cx_p p_n - cx_r r_n (+ cx_o p_{n-1} read from the other p buffer)
    if (K1 && !XU && a.xd && blockIdx.x == 0 && threadIdx.x == 0) a.xd[1] = beta;
    if (K2 && !XU && a.xd && blockIdx.x == 0 && threadIdx.x == 0) a.xd[0] = res_scale;
    if (K2X) xk2_coefficients(a.xd, a.xmode, res_scale, &cx_p, &cx_r, &cx_o);
    // EX: every wave of a workgroup with work runs the loop (inactive ones with all loads / stores predicated off) so
    // that the barriers match; lk0 / lk1 depend on the workgroup only
    if (EX ? (lk0 < lk1) : active) {
        const double w0 = a.w[0], wx = a.w[1], wy = a.w[2], wz = a.w[3];
        const double wxy = a.w[4], wxz = a.w[5], wyz = a.w[6], wxyz = a.w[7];
        const double shift = a.in_shift;

        const bool lane_ok = active && i0 < g.PX;             // pair inside the padded row
        // free_x == 2 (natural-face sessions): the x-face nodes of interior rows are computed HERE.  On the host tables the class of
        // an x-face is the interior stencil with the in-face (dx = 0) terms halved, the inward dx = -+1 terms unchanged and the
        // outward ones absent (K and M, P1 and Q1, any h): q_face = D / 2 + (inward shifted part).  Per-lane factors, all 1 away
        // from the faces; a Dirichlet face keeps its factors at 1 and is masked as before.
        const bool fx = a.free_x == 2;
        const bool lo_free = fx && g.bc_mode == PDEGPU_BC_DIRECTIONAL && !g.bc_present[0];
        const bool hi_free = fx && a.fx_hi && g.bc_mode == PDEGPU_BC_DIRECTIONAL && !g.bc_present[1];
        const int xlo = lo_free ? 0 : 1, xhi = hi_free ? g.NX - 1 : g.NX - 2;
        const bool ma = active && (i0 >= xlo) && (i0 <= xhi);      // node a = i0 is a free node computed by this kernel
        const bool mb = active && (i0 + 1 <= xhi);                 // node b = i0+1
        const bool fa_lo = lo_free && i0 == 0, fa_hi = hi_free && i0 == g.NX - 1, fb_hi = hi_free && i0 + 1 == g.NX - 1;
        const double sa = (fa_lo || fa_hi) ? 0.5 : 1.0, la = fa_lo ? 0.0 : 1.0, ra = fa_hi ? 0.0 : 1.0;
        const double sb = fb_hi ? 0.5 : 1.0, rb = fb_hi ? 0.0 : 1.0;
        const bool edge_lane = (lane == 0) || (lane == 63);
        const long long edge_off = (lane == 0) ? -1 : 2;      // lane 0: node i0-1, lane 63: node i0+2

        // value of one row-plane: K1 -> p_n = r + beta p_{n-1}; PLAIN -> in + shift; RESIDUAL -> in
        auto row_ok = [&](int j) { return (j >= 0) && (j < g.NY); };
        auto load_pair = [&](const double* v, long long off) {
            return *reinterpret_cast<const double2*>(v + off);
        };
        // Loads are UNCONDITIONAL and their values are NOT masked: a lane / row outside the grid reads offset 0 of the vector (always
        // mapped, finite) and whatever it reads only ever reaches outputs that are masked or never stored — every row a computed
        // output uses lies inside the grid, pad lanes read the zero pads, and the lanes beyond the padded row only feed the masked
        // node PX-1.  A load inside an exec-masked branch is followed by `s_waitcnt vmcnt(0)` as soon as its value is used in
        // that branch (p = r + beta p, in + shift), and a `cond ? f(load) : 0` select is turned back into such a branch by the
        // compiler: every row of a plane then costs its own memory round trip.  This way all loads of a z-step are in flight
        // together and the first wait sits in front of the stencil.
        auto value_row = [&](int lk, int j, double2* po_out) {
            const bool ok = row_ok(j) && lane_ok;
            const long long off = ok ? (long long)lk * g.PP + (long long)j * g.PX + i0 : 0;
            double2 val;
            if (K1) {
                const double2 rr = load_pair(a.in, off), po = load_pair(a.p_old, off);
                val.x = rr.x + beta * po.x;
                val.y = rr.y + beta * po.y;
                if (po_out) *po_out = po;
            } else {
                // (non-temporal loads here were measured slower, profiles/r02_exp14_nontemporal_input_loads.log: the y-neighbours' halo rows are L2 hits)
                const double2 v = load_pair(a.in, off);
                val.x = v.x + shift;
                val.y = v.y + shift;
            }
            return val;
        };
        auto edge_value = [&](int lk, int j) {
            const bool ok = row_ok(j) && edge_lane;
            const long long off = ok ? (long long)lk * g.PP + (long long)j * g.PX + i0 + edge_off : 0;
            if (K1) return a.in[off] + beta * a.p_old[off];
            return a.in[off] + shift;
        };
        // edge values of one plane: E[r] = value at node i0-1 (lane 0) / i0+2 (lane 63) of row r
        // EX: tile-edge neighbours that belong to ANOTHER workgroup (or to no tile: the x-face node NX-1 of a natural-face grid at 2^k + 1
        // nodes per row) still come from global memory.  Those loads are issued with the plane's other loads (edge_global, unconditional
        // like them, wave-uniform branch) and only selected here: inside `if (lane == 63)` their first use was a `s_waitcnt vmcnt(0)` in
        // every z-step that also drained the prefetch of the next plane (natural-face K2: 0.75 ms against 0.55 ms on a Dirichlet grid).
        const int il_t = xt * 128 - 1, ir_t = xt * 128 + 128;
        const bool cgvec_t = (K1 || K2) && !a.free_x;
        // (linear: wave wv - 1 / wv + 1 of the workgroup holds the tile to the left / right unless this tile starts / ends its row)
        const bool has_l_t = fg.linear ? (wv > 0 && xt > 0) : wave_x > 0;
        const bool has_r_t = fg.linear ? (wv < 3 && xt + 1 < fg.ntx) : (wave_x < fg.wxt - 1) && (xt + 1 < fg.ntx);
        const bool gl_t = !has_l_t && il_t >= 0 && (il_t >= 1 || !cgvec_t) && active;
        const bool gr_t = !has_r_t && ir_t <= g.NX - 1 && (ir_t <= g.NX - 2 || !cgvec_t) && active;
        const bool need_ge = EX && __builtin_amdgcn_readfirstlane((int)(gl_t || gr_t)) != 0;      // wave-uniform
        // (the empty asm with a memory clobber pins the loads here: left alone the compiler sinks them to their only use — the edge lanes'
        // branch of plane_edges, behind the wait for the whole plane — and waits for them right there; the arithmetic p = r + beta p /
        // in + shift happens at the use for the same reason.  A volatile load is no alternative: it is followed by `s_waitcnt vmcnt(0)`.)
        auto edge_global = [&](int lk, double* EG, double* EGp, bool all_rows) {
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                EG[r] = 0.0;
                EGp[r] = 0.0;
            }
            if (need_ge) {
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    const bool own = (r >= 1) && (r <= RY);
                    if (all_rows || own) {
                        const bool want = (lane == 0 && gl_t) || (lane == 63 && gr_t);
                        const int j = j0 - 1 + r;
                        const long long off = (want && row_ok(j)) ? (long long)lk * g.PP + (long long)j * g.PX + i0 + edge_off : 0;
                        EG[r] = a.in[off];
                        if (K1) EGp[r] = a.p_old[off];
                        // (a lane that does not want the value reads offset 0 and never uses it; a row outside the grid: offset 0, masked below)
                    }
                }
                asm volatile("" ::: "memory");
            }
        };
        auto plane_edges = [&](int lk, const double2* rows, double* E, bool all_rows, const double* EG, const double* EGp) {
            if (!EX) {
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    const bool own = (r >= 1) && (r <= RY);
                    E[r] = (all_rows || own) ? edge_value(lk, j0 - 1 + r) : 0.0;
                }
                return;
            }
            const int buf = estep & 1;
            ++estep;
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                if (lane == 0) sEdge[buf][wv][r][0] = rows[r].x;
                if (lane == 63) sEdge[buf][wv][r][1] = rows[r].y;
            }
            __syncthreads();
            // x-neighbour wave inside this workgroup (and covering real nodes: an inactive right neighbour holds zeros); a global value
            // (EG, see edge_global) only where the neighbour tile belongs to another workgroup AND the node can be non-zero: CG vectors
            // (K1: r, p; K2: p) vanish on the Dirichlet columns, PLAIN / RESIDUAL inputs carry boundary values
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                const bool own = (r >= 1) && (r <= RY);
                double e = 0.0;
                if (all_rows || own) {
                    const int j = j0 - 1 + r;
                    if (lane == 0) {
                        if (has_l_t) e = sEdge[buf][wv - 1][r][1];
                        else if (gl_t && row_ok(j)) e = K1 ? EG[r] + beta * EGp[r] : EG[r] + shift;
                    } else if (lane == 63) {
                        if (has_r_t) e = sEdge[buf][wv + 1][r][0];
                        else if (gr_t && row_ok(j)) e = K1 ? EG[r] + beta * EGp[r] : EG[r] + shift;
                    }
                }
                E[r] = e;
            }
        };

        // K1: x += alpha p_{n-1}, store p_n for an owned row-plane.  The x row is requested together with the plane's r / p rows
        // (own_xload, unconditional like them) and consumed here, after the stencil.
        auto own_xload = [&](int lk, int j) {
            const bool ok = j <= fg.jmax && lane_ok;
            const long long off = ok ? (long long)lk * g.PP + (long long)j * g.PX + i0 : 0;
            return load_pair_x<NT>(a.x + off);
        };
        auto own_update = [&](int lk, int j, const double2& pn, const double2& po, double2 xv) {
            if (K1) {
                if (j <= fg.jmax && lane_ok) {
                    const long long off = (long long)lk * g.PP + (long long)j * g.PX + i0;
                    store_pair<NT>(a.p_new + off, pn);
                    if (K1X && alpha_prev != 0.0) {
                        xv.x += alpha_prev * po.x;
                        xv.y += alpha_prev * po.y;
                        store_pair<NT>(a.x + off, xv);
                    }
                }
            }
        };

        // registers: P = plane lk-1, C = plane lk, N = plane lk+1; index 0..RH-1 <-> rows j0-1 .. j0+RY
        double2 P[RH], C[RH], N[RH];
        double EP[RH], EC[RH], EN[RH];      // edge values (lane 0: node i0-1, lane 63: node i0+2)
#pragma unroll
        for (int r = 0; r < RH; ++r) {
            const bool own = (r >= 1) && (r <= RY);
            const int j = j0 - 1 + r;
            // plane lk0-1: 7-point needs own rows only; 15-point rows j-1..j (r = 0..RY); 27-point all rows
            if (W27 || (W15 ? (r <= RY) : own)) P[r] = value_row(lk0 - 1, j, nullptr);
            else P[r] = make_double2(0.0, 0.0);
            EP[r] = 0.0;
            double2 po = make_double2(0.0, 0.0);
            C[r] = value_row(lk0, j, &po);
            if (K1 && own) own_update(lk0, j, C[r], po, K1X ? own_xload(lk0, j) : make_double2(0.0, 0.0));
        }
        double EG[RH], EGp[RH];
        if (W15) {
            edge_global(lk0 - 1, EG, EGp, true);
            plane_edges(lk0 - 1, P, EP, true, EG, EGp);
        }
        edge_global(lk0, EG, EGp, W15);
        plane_edges(lk0, C, EC, W15, EG, EGp);
        if (PF) {
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                N[r] = value_row(lk0 + 1, j0 - 1 + r, nullptr);
                EN[r] = 0.0;
            }
            if (!EX) plane_edges(lk0 + 1, N, EN, W15, EG, EGp);
        }

        // K1 keeps p_n valid on the ghost planes of the slab (same formula, same inputs as the neighbour rank)
        auto ghost_store = [&](int lk, const double2* rows) {
            if (K1) {
#pragma unroll
                for (int r = 1; r <= RY; ++r) {
                    const int j = j0 - 1 + r;
                    if (j <= fg.jmax && lane_ok)
                        *reinterpret_cast<double2*>(a.p_new + (long long)lk * g.PP + (long long)j * g.PX + i0) = rows[r];
                }
            }
        };
        if (K1 && lk0 == 0 && g.k0 > 0) ghost_store(-1, P);              // (the chunk that starts the slab)

        for (int lk = lk0; lk < lk1; ++lk) {
            // ---- issue the loads of plane lk+1 (PF: of plane lk+2; lk1 + 1 would leave the allocation and is never used) ----
            double2 PO[RH];
            double2 XV[RY];
            double2 N2[PF ? RH : 1];
            double EN2[PF ? RH : 1];
            if (PF) {
                const int lkn = min(lk + 2, lk1);
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    N2[r] = value_row(lkn, j0 - 1 + r, nullptr);
                    EN2[r] = 0.0;
                }
                if (!EX) plane_edges(lkn, N2, EN2, W15, EG, EGp);
            } else {
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    const int j = j0 - 1 + r;
                    PO[r] = make_double2(0.0, 0.0);
                    N[r] = value_row(lk + 1, j, &PO[r]);
                }
                if (K1X && lk + 1 < lk1) {
#pragma unroll
                    for (int r = 0; r < RY; ++r) XV[r] = own_xload(lk + 1, j0 + r);
                }
                if (K1 && !XU) {
#pragma unroll
                    for (int r = 0; r < RY; ++r) XV[r] = make_double2(0.0, 0.0);
                }
                if (!EX) plane_edges(lk + 1, N, EN, W15, EG, EGp);      // global edge loads: issue them with the plane's loads
                else edge_global(lk + 1, EG, EGp, W15);                 // (EX: the few that no wave of the workgroup can hand over)
            }
            double2 RHS[RY];
            double2 XK[K2X ? RY : 1], PO2[K2X ? RY : 1];      // K2X: the x rows of plane lk (and, after a restart only, the other buffer's p rows)
            if (RES) {
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    const int j = j0 + r;
                    RHS[r] = make_double2(sa * a.rhs_const, sb * a.rhs_const);      // (a constant load is halved on an x-face)
                    if (a.rhs) {                                   // kernel-uniform; the load itself is unconditional (see value_row)
                        const bool ok = j <= fg.jmax && lane_ok;
                        RHS[r] = load_pair_x<NT>(a.rhs + (ok ? (long long)lk * g.PP + (long long)j * g.PX + i0 : 0));      // read once: streaming hint
                    }
                    if (K2X) {
                        const bool ok = j <= fg.jmax && lane_ok;
                        const long long off = ok ? (long long)lk * g.PP + (long long)j * g.PX + i0 : 0;
                        XK[r] = load_pair_x<NT>(a.x + off);
                        PO2[r] = make_double2(0.0, 0.0);
                        if (cx_o != 0.0) PO2[r] = load_pair(a.p_old, off);      // kernel-uniform, rare
                    }
                }
            }

            if (EX) {
                if (PF) edge_global(lk + 1, EG, EGp, W15);          // (with the deeper prefetch the plane was requested a step ago: not these)
                plane_edges(lk + 1, N, EN, W15, EG, EGp);      // LDS hand-off of the new plane's tile-edge values
            }
            // ---- stencil of plane lk -----------------------------------------------------------
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int c = r + 1;
                const int j = j0 + r;
                if (j <= fg.jmax) {                                   // wave-uniform
                    double lC = dpp_from_lower(C[c].y), rC = dpp_from_upper(C[c].x);
                    lC = (lane == 0) ? EC[c] : lC;
                    rC = (lane == 63) ? EC[c] : rC;
                    double qa = w0 * C[c].x + wx * (lC + C[c].y) + wy * (C[c - 1].x + C[c + 1].x) + wz * (P[c].x + N[c].x);
                    double qb = w0 * C[c].y + wx * (C[c].x + rC) + wy * (C[c - 1].y + C[c + 1].y) + wz * (P[c].y + N[c].y);
                    if (fx && !W15) {                                 // 7-point with x-face rows (kernel-uniform branch)
                        qa = sa * (w0 * C[c].x + wy * (C[c - 1].x + C[c + 1].x) + wz * (P[c].x + N[c].x)) + wx * (la * lC + ra * C[c].y);
                        qb = sb * (w0 * C[c].y + wy * (C[c - 1].y + C[c + 1].y) + wz * (P[c].y + N[c].y)) + wx * (C[c].x + rb * rC);
                    }
                    if (W27) {
                        // Q1: q = w0 C + wy Sy + wz Sz + wyz Syz + l(T) + r(T),  T = wx C + wxy Sy + wxz Sz + wxyz Syz with the row
                        // sums Sy = C[j-1] + C[j+1], Sz = P + N, Syz = P[j-1] + P[j+1] + N[j-1] + N[j+1]: two x-shifts per node
                        const double2 Sy = make_double2(C[c - 1].x + C[c + 1].x, C[c - 1].y + C[c + 1].y);
                        const double2 Sz = make_double2(P[c].x + N[c].x, P[c].y + N[c].y);
                        const double2 Syz = make_double2((P[c - 1].x + P[c + 1].x) + (N[c - 1].x + N[c + 1].x),
                                                         (P[c - 1].y + P[c + 1].y) + (N[c - 1].y + N[c + 1].y));
                        const double2 T = make_double2(wx * C[c].x + wxy * Sy.x + wxz * Sz.x + wxyz * Syz.x,
                                                       wx * C[c].y + wxy * Sy.y + wxz * Sz.y + wxyz * Syz.y);
                        const double Te = wx * EC[c] + wxy * (EC[c - 1] + EC[c + 1]) + wxz * (EP[c] + EN[c]) +
                                          wxyz * ((EP[c - 1] + EP[c + 1]) + (EN[c - 1] + EN[c + 1]));
                        double lT = dpp_from_lower(T.y), rT = dpp_from_upper(T.x);
                        lT = (lane == 0) ? Te : lT;
                        rT = (lane == 63) ? Te : rT;
                        qa = w0 * C[c].x + wy * Sy.x + wz * Sz.x + wyz * Syz.x + lT + T.y;
                        qb = w0 * C[c].y + wy * Sy.y + wz * Sz.y + wyz * Syz.y + T.x + rT;
                        if (fx) {
                            qa = sa * (w0 * C[c].x + wy * Sy.x + wz * Sz.x + wyz * Syz.x) + la * lT + ra * T.y;
                            qb = sb * (w0 * C[c].y + wy * Sy.y + wz * Sz.y + wyz * Syz.y) + T.x + rb * rT;
                        }
                    } else if (W15) {
                        // same-sign diagonals of the Kuhn split: +-(1,1,0), +-(1,0,1), +-(0,1,1), +-(1,1,1).  Everything whose
                        // source sits at x+1 is accumulated in U, at x-1 in L, and the x-shift is applied to the two sums:
                        // 2 DPP pairs per node instead of 8 (the x-term of the 7-point part included)
                        const double2 U = make_double2(wx * C[c].x + wxy * C[c + 1].x + wxz * N[c].x + wxyz * N[c + 1].x,
                                                       wx * C[c].y + wxy * C[c + 1].y + wxz * N[c].y + wxyz * N[c + 1].y);
                        const double2 Lw = make_double2(wx * C[c].x + wxy * C[c - 1].x + wxz * P[c].x + wxyz * P[c - 1].x,
                                                        wx * C[c].y + wxy * C[c - 1].y + wxz * P[c].y + wxyz * P[c - 1].y);
                        const double Ue = wx * EC[c] + wxy * EC[c + 1] + wxz * EN[c] + wxyz * EN[c + 1];     // lane 63: node i0+2
                        const double Le = wx * EC[c] + wxy * EC[c - 1] + wxz * EP[c] + wxyz * EP[c - 1];     // lane 0: node i0-1
                        double lL = dpp_from_lower(Lw.y), rU = dpp_from_upper(U.x);
                        lL = (lane == 0) ? Le : lL;
                        rU = (lane == 63) ? Ue : rU;
                        qa = w0 * C[c].x + wy * (C[c - 1].x + C[c + 1].x) + wz * (P[c].x + N[c].x) + wyz * (N[c + 1].x + P[c - 1].x) + U.y + lL;
                        qb = w0 * C[c].y + wy * (C[c - 1].y + C[c + 1].y) + wz * (P[c].y + N[c].y) + wyz * (N[c + 1].y + P[c - 1].y) + rU + Lw.x;
                        if (fx) {
                            qa = sa * (w0 * C[c].x + wy * (C[c - 1].x + C[c + 1].x) + wz * (P[c].x + N[c].x) + wyz * (N[c + 1].x + P[c - 1].x)) +
                                 ra * U.y + la * lL;
                            qb = sb * (w0 * C[c].y + wy * (C[c - 1].y + C[c + 1].y) + wz * (P[c].y + N[c].y) + wyz * (N[c + 1].y + P[c - 1].y)) +
                                 rb * rU + Lw.x;
                        }
                    }
                    double2 o;
                    if (RES) {
                        o.x = ma ? RHS[r].x - res_scale * qa : 0.0;
                        o.y = mb ? RHS[r].y - res_scale * qb : 0.0;
                        acc += o.x * o.x + o.y * o.y;
                    } else {
                        o.x = ma ? qa : 0.0;
                        o.y = mb ? qb : 0.0;
                        acc += (C[c].x - (K1 ? 0.0 : shift)) * o.x + (C[c].y - (K1 ? 0.0 : shift)) * o.y;
                        if (K1) acc2 += o.x * o.x + o.y * o.y;
                    }
                    if (lane_ok && (!K1 || a.out)) store_pair<NT>(a.out + (long long)lk * g.PP + (long long)j * g.PX + i0, o);
                    if (K2X && lane_ok) {
                        // x += alpha_n p_n (+ the update the K2 before left out): p_n = C, r_n = RHS — both vanish on Dirichlet nodes and pads
                        double2 xv = XK[r];
                        xv.x += cx_p * C[c].x - cx_r * RHS[r].x + cx_o * PO2[r].x;
                        xv.y += cx_p * C[c].y - cx_r * RHS[r].y + cx_o * PO2[r].y;
                        store_pair<NT>(a.x + (long long)lk * g.PP + (long long)j * g.PX + i0, xv);
                    }
                }
            }

            // ---- K1: owned rows of plane lk+1: x update + p_n store ------------------------------
            if (K1 && lk + 1 < lk1) {
#pragma unroll
                for (int r = 1; r <= RY; ++r) own_update(lk + 1, j0 - 1 + r, N[r], PO[r], XV[r - 1]);
            }
            // ---- rotate planes -----------------------------------------------------------------
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                P[r] = C[r];
                EP[r] = EC[r];
                C[r] = N[r];
                EC[r] = EN[r];
                if (PF) {
                    N[r] = N2[r];
                    if (!EX) EN[r] = EN2[r];
                }
            }
        }
        // after the last rotation C holds plane lk1: the upper ghost plane when this is the slab's last chunk
        if (K1 && lk1 == g.nkl && g.k0 + g.nkl < g.NZ) ghost_store(lk1, C);
    }
    const double tot = block_sum_256(acc, sRed);
    dot_finish(tot, a.partials);
    if (K1 && a.partials2) {                                  // kernel-uniform
        __syncthreads();
        dot_finish(block_sum_256(acc2, sRed), a.partials2);
    }
}


// ---------------------------------------------------------------------------------------------
// 2-D grids: the two CG kernels without marching
// ---------------------------------------------------------------------------------------------
// A 2-D grid has no third axis to march along: through the flat view every row is a "plane", so a wave's z-chunk is a chain of
// dependent one-row load rounds (config 2: K1 = 17 us for 1 M nodes, all latency).  Here a wavefront owns a 128-node x RY-row
// tile, requests its RY + 2 rows (r and p_{n-1}, or the stored p_n) in ONE round, forms p at every held row and applies the
// in-plane stencil w0, wx, wy, wd1 (offsets +-(1,1), the P1 diagonal), wd2 (offsets +-(1,-1); Q1: wd2 = wd1) with the same DPP
// x-shifts as the marching kernel.  The four waves of a workgroup are stacked in y, so their shared halo rows meet in one L1.
// Every boundary node is Dirichlet here (the 2-D tools have one boundary value): CG vectors vanish on the boundary columns.
template <int RY, int MODE>
__global__ __launch_bounds__(256) void k_plane2d(StencilArgs a, int ntx, int nty) {
    __shared__ double sRed[4];
    constexpr bool K1 = (MODE == APPLY_CG_K1);
    constexpr int RH = RY + 2;
    const GridDev& g = a.g;
    double alpha_prev = 0.0, beta = 0.0, res_scale = 1.0;
    if (K1) cg_scalars_in(a.di, a.scal, sRed, &alpha_prev, &beta);
    else res_scale = cg_alpha_in(a.di, a.scal, sRed);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int xt = (int)blockIdx.x % ntx, yt = ((int)blockIdx.x / ntx) * 4 + wv;
    const bool active = yt < nty;
    const int i0 = xt * 128 + 2 * lane;
    const int j0 = 1 + yt * RY;
    const bool lane_ok = active && i0 < g.PX;
    const bool ma = active && i0 >= 1 && i0 <= g.NX - 2;
    const bool mb = active && i0 + 1 <= g.NX - 2;
    const double w0 = a.w[0], wx = a.w[1], wy = a.w[2], wd1 = a.w[3], wd2 = a.w[4];
    // tile-edge neighbours: lane 0 needs node i0-1, lane 63 node i0+2 (CG vectors are zero in memory on the Dirichlet columns and
    // the pads; what lies beyond only reaches masked outputs)
    const bool elane = (lane == 0 || lane == 63) && active;
    const long long eoff = (lane == 0) ? -1 : 2;

    double2 C[RH], PO[RH];
    double E[RH];
#pragma unroll
    for (int r = 0; r < RH; ++r) {
        const int j = j0 - 1 + r;
        C[r] = make_double2(0.0, 0.0);
        PO[r] = make_double2(0.0, 0.0);
        E[r] = 0.0;
        {
            // unconditional loads, unmasked values (see k_fast::value_row): rows / lanes outside the grid read offset 0 and only
            // reach outputs that are masked or not stored; all rows of the tile are requested in ONE round
            const bool rok = j <= g.NY - 1;
            const long long off = (rok && lane_ok) ? (long long)j * g.PX + i0 : 0;
            const long long offe = (rok && elane) ? (long long)j * g.PX + i0 + eoff : 0;
            const double2 v = *reinterpret_cast<const double2*>(a.in + off);
            if (K1) {
                PO[r] = *reinterpret_cast<const double2*>(a.p_old + off);
                C[r] = make_double2(v.x + beta * PO[r].x, v.y + beta * PO[r].y);
                E[r] = a.in[offe] + beta * a.p_old[offe];
            } else {
                C[r] = v;
                E[r] = a.in[offe];
            }
        }
    }
    double acc = 0.0, acc2 = 0.0;
#pragma unroll
    for (int r = 1; r <= RY; ++r) {
        const int j = j0 - 1 + r;
        if (j <= g.NY - 2) {                                  // wave-uniform
            const long long off = (long long)j * g.PX + i0;
            double lC = dpp_from_lower(C[r].y), rC = dpp_from_upper(C[r].x);
            double lCm = dpp_from_lower(C[r - 1].y), rCm = dpp_from_upper(C[r - 1].x);
            double lCp = dpp_from_lower(C[r + 1].y), rCp = dpp_from_upper(C[r + 1].x);
            if (lane == 0) { lC = E[r]; lCm = E[r - 1]; lCp = E[r + 1]; }
            if (lane == 63) { rC = E[r]; rCm = E[r - 1]; rCp = E[r + 1]; }
            const double qa = w0 * C[r].x + wx * (lC + C[r].y) + wy * (C[r - 1].x + C[r + 1].x) + wd1 * (C[r + 1].y + lCm) +
                              wd2 * (C[r - 1].y + lCp);
            const double qb = w0 * C[r].y + wx * (C[r].x + rC) + wy * (C[r - 1].y + C[r + 1].y) + wd1 * (rCp + C[r - 1].x) +
                              wd2 * (rCm + C[r + 1].x);
            if (K1) {
                if (lane_ok) {
                    *reinterpret_cast<double2*>(a.p_new + off) = C[r];
                    if (alpha_prev != 0.0) {
                        double2 xv = *reinterpret_cast<double2*>(a.x + off);
                        xv.x += alpha_prev * PO[r].x;
                        xv.y += alpha_prev * PO[r].y;
                        *reinterpret_cast<double2*>(a.x + off) = xv;
                    }
                }
                const double2 o = make_double2(ma ? qa : 0.0, mb ? qb : 0.0);
                acc += C[r].x * o.x + C[r].y * o.y;
                acc2 += o.x * o.x + o.y * o.y;
                if (lane_ok && a.out) *reinterpret_cast<double2*>(a.out + off) = o;
            } else {
                const double2 rv = *reinterpret_cast<const double2*>(a.rhs + (lane_ok ? off : 0));
                const double2 o = make_double2(ma ? rv.x - res_scale * qa : 0.0, mb ? rv.y - res_scale * qb : 0.0);
                acc += o.x * o.x + o.y * o.y;
                if (lane_ok) *reinterpret_cast<double2*>(a.out + off) = o;
            }
        }
    }
    const double tot = block_sum_256(acc, sRed);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    if (K1 && a.partials2) {
        __syncthreads();
        dot_finish(block_sum_256(acc2, sRed), a.partials2);
    }
}


// ---------------------------------------------------------------------------------------------
// small 3-D grids: the two CG kernels without marching
// ---------------------------------------------------------------------------------------------
// On a grid whose vectors fit in L2 the marching kernel has nothing to save (every re-read is an L2 hit) and pays for its
// serial plane steps with memory latency: with chunks of one or two planes a 64^3 K1 is three dependent load rounds.  This
// form gives a wavefront ONE plane of a 128-node x RY-row tile: the RY + 2 rows of the three planes it needs are requested in a
// single round, p is formed at every held position and the same 7- / 15- / 27-point formulas are applied.  Uniform Dirichlet
// boundary, one slab (no ghost planes to maintain); the four waves of a workgroup are stacked in y.
// XM = 1 (grids beyond L2): XCD-aware order.  Workgroups b and b + 8 share an XCD (round-robin dispatch, speed only): XCD x owns the
// y-range [x, x+1) * nyg/8 of every plane and takes its tiles plane by plane in dispatch order, so the three planes a tile reads are
// the ones its own XCD's L2 has just streamed (0.27 MB per plane and vector at 513^2 / 8) — every input plane leaves HBM once although
// no workgroup lives longer than one tile.  Streaming stores are non-temporal.  PLAIN: out = A (in + shift), inputs carry the
// boundary values (tile-edge loads reach the Dirichlet columns).
template <int NPT, int RY, int MODE, int XM>
__global__ __launch_bounds__(256) void k_tile3d(StencilArgs a, int ntx, int nty) {
    __shared__ double sRed[4];
    constexpr bool K1 = (MODE == APPLY_CG_K1);
    constexpr bool K2 = (MODE == APPLY_CG_K2);
    constexpr bool PL = (MODE == APPLY_PLAIN);
    constexpr bool W27 = (NPT == 27), WD = (NPT != 7);        // WD: diagonal neighbours -> every held row of every plane
    constexpr int RH = RY + 2;
    const GridDev& g = a.g;
    double alpha_prev = 0.0, beta = 0.0, res_scale = 1.0;
    if (K1) cg_scalars_in(a.di, a.scal, sRed, &alpha_prev, &beta);
    if (K2) res_scale = cg_alpha_in(a.di, a.scal, sRed);
    const double shift = PL ? a.in_shift : 0.0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nyg = (nty + 3) / 4;
    int xt, yg, lk;
    if (XM) {
        const int xcd = (int)blockIdx.x & 7, t = (int)(blockIdx.x >> 3), ny8 = (nyg + 7) / 8;
        xt = t % ntx;
        yg = xcd * ny8 + (t / ntx) % ny8;
        lk = 1 + t / (ntx * ny8);
        if (yg >= nyg) yg = 0x3fffffff;                        // surplus workgroup of the last XCD range: inactive
    } else {
        xt = (int)blockIdx.x % ntx;
        yg = ((int)blockIdx.x / ntx) % nyg;
        lk = 1 + (int)blockIdx.x / (ntx * nyg);               // interior planes 1 .. NZ-2
    }
    const int yt = (yg < 0x10000000) ? yg * 4 + wv : 0x3fffffff;
    const bool active = yt < nty;
    const int i0 = xt * 128 + 2 * lane;
    const int j0 = active ? 1 + yt * RY : 1;
    const bool lane_ok = active && i0 < g.PX;
    const bool ma = active && i0 >= 1 && i0 <= g.NX - 2;
    const bool mb = active && i0 + 1 <= g.NX - 2;
    const double w0 = a.w[0], wx = a.w[1], wy = a.w[2], wz = a.w[3];
    const double wxy = a.w[4], wxz = a.w[5], wyz = a.w[6], wxyz = a.w[7];
    // tile-edge neighbours (lane 0: node i0-1, lane 63: node i0+2): CG vectors are zero in memory on the Dirichlet columns and the
    // pads, a PLAIN input carries the boundary values there; what lies beyond only reaches masked outputs
    const bool elane = (lane == 0 || lane == 63) && active;
    const long long eoff = (lane == 0) ? -1 : 2;

    double2 P[RH], C[RH], N[RH], PO[RH];
    double EP[RH], EC[RH], EN[RH];
    auto value = [&](long long off, double2* po) {
        const double2 v = *reinterpret_cast<const double2*>(a.in + off);
        if (K1) {
            const double2 o = *reinterpret_cast<const double2*>(a.p_old + off);
            if (po) *po = o;
            return make_double2(v.x + beta * o.x, v.y + beta * o.y);
        }
        return make_double2(v.x + shift, v.y + shift);
    };
    auto edge = [&](long long off) { return K1 ? a.in[off + eoff] + beta * a.p_old[off + eoff] : a.in[off + eoff] + shift; };
    auto put = [&](double* p, const double2& v) {
        if (XM) {
            dbl2 t = {v.x, v.y};
            __builtin_nontemporal_store(t, reinterpret_cast<dbl2*>(p));
        } else {
            *reinterpret_cast<double2*>(p) = v;
        }
    };
#pragma unroll
    for (int r = 0; r < RH; ++r) {
        const int j = j0 - 1 + r;
        const bool own = r >= 1 && r <= RY;
        P[r] = C[r] = N[r] = PO[r] = make_double2(0.0, 0.0);
        EP[r] = EC[r] = EN[r] = 0.0;
        {
            // unconditional loads, unmasked values (see k_fast::value_row): everything a tile needs is requested in ONE round.  The
            // fall-back address is plane 1, row 0 (its planes 0 and 2 exist for every tile session)
            const bool rok = j <= g.NY - 1;
            const long long off = (rok && lane_ok) ? (long long)lk * g.PP + (long long)j * g.PX + i0 : g.PP;
            const long long offe = (rok && elane) ? (long long)lk * g.PP + (long long)j * g.PX + i0 : g.PP + 1;
            C[r] = value(off, &PO[r]);
            if (WD || own) {
                P[r] = value(off - g.PP, nullptr);
                N[r] = value(off + g.PP, nullptr);
                EC[r] = edge(offe);
            }
            if (WD) {
                EP[r] = edge(offe - g.PP);
                EN[r] = edge(offe + g.PP);
            }
        }
    }
    double acc = 0.0, acc2 = 0.0;
#pragma unroll
    for (int c = 1; c <= RY; ++c) {
        const int j = j0 - 1 + c;
        if (j <= g.NY - 2) {                                  // wave-uniform
            const long long off = (long long)lk * g.PP + (long long)j * g.PX + i0;
            double qa, qb;
            if (W27) {
                const double2 Sy = make_double2(C[c - 1].x + C[c + 1].x, C[c - 1].y + C[c + 1].y);
                const double2 Sz = make_double2(P[c].x + N[c].x, P[c].y + N[c].y);
                const double2 Syz = make_double2((P[c - 1].x + P[c + 1].x) + (N[c - 1].x + N[c + 1].x),
                                                 (P[c - 1].y + P[c + 1].y) + (N[c - 1].y + N[c + 1].y));
                const double2 T = make_double2(wx * C[c].x + wxy * Sy.x + wxz * Sz.x + wxyz * Syz.x,
                                               wx * C[c].y + wxy * Sy.y + wxz * Sz.y + wxyz * Syz.y);
                const double Te = wx * EC[c] + wxy * (EC[c - 1] + EC[c + 1]) + wxz * (EP[c] + EN[c]) +
                                  wxyz * ((EP[c - 1] + EP[c + 1]) + (EN[c - 1] + EN[c + 1]));
                double lT = dpp_from_lower(T.y), rT = dpp_from_upper(T.x);
                lT = (lane == 0) ? Te : lT;
                rT = (lane == 63) ? Te : rT;
                qa = w0 * C[c].x + wy * Sy.x + wz * Sz.x + wyz * Syz.x + lT + T.y;
                qb = w0 * C[c].y + wy * Sy.y + wz * Sz.y + wyz * Syz.y + T.x + rT;
            } else if (WD) {
                const double2 U = make_double2(wx * C[c].x + wxy * C[c + 1].x + wxz * N[c].x + wxyz * N[c + 1].x,
                                               wx * C[c].y + wxy * C[c + 1].y + wxz * N[c].y + wxyz * N[c + 1].y);
                const double2 Lw = make_double2(wx * C[c].x + wxy * C[c - 1].x + wxz * P[c].x + wxyz * P[c - 1].x,
                                                wx * C[c].y + wxy * C[c - 1].y + wxz * P[c].y + wxyz * P[c - 1].y);
                const double Ue = wx * EC[c] + wxy * EC[c + 1] + wxz * EN[c] + wxyz * EN[c + 1];
                const double Le = wx * EC[c] + wxy * EC[c - 1] + wxz * EP[c] + wxyz * EP[c - 1];
                double lL = dpp_from_lower(Lw.y), rU = dpp_from_upper(U.x);
                lL = (lane == 0) ? Le : lL;
                rU = (lane == 63) ? Ue : rU;
                qa = w0 * C[c].x + wy * (C[c - 1].x + C[c + 1].x) + wz * (P[c].x + N[c].x) + wyz * (N[c + 1].x + P[c - 1].x) + U.y + lL;
                qb = w0 * C[c].y + wy * (C[c - 1].y + C[c + 1].y) + wz * (P[c].y + N[c].y) + wyz * (N[c + 1].y + P[c - 1].y) + rU + Lw.x;
            } else {
                double lC = dpp_from_lower(C[c].y), rC = dpp_from_upper(C[c].x);
                lC = (lane == 0) ? EC[c] : lC;
                rC = (lane == 63) ? EC[c] : rC;
                qa = w0 * C[c].x + wx * (lC + C[c].y) + wy * (C[c - 1].x + C[c + 1].x) + wz * (P[c].x + N[c].x);
                qb = w0 * C[c].y + wx * (C[c].x + rC) + wy * (C[c - 1].y + C[c + 1].y) + wz * (P[c].y + N[c].y);
            }
            if (K1) {
                if (lane_ok) {
                    put(a.p_new + off, C[c]);
                    if (alpha_prev != 0.0) {
                        double2 xv = *reinterpret_cast<double2*>(a.x + off);
                        xv.x += alpha_prev * PO[c].x;
                        xv.y += alpha_prev * PO[c].y;
                        put(a.x + off, xv);
                    }
                }
                const double2 o = make_double2(ma ? qa : 0.0, mb ? qb : 0.0);
                acc += C[c].x * o.x + C[c].y * o.y;
                acc2 += o.x * o.x + o.y * o.y;
                if (lane_ok && a.out) put(a.out + off, o);
            } else if (PL) {
                const double2 o = make_double2(ma ? qa : 0.0, mb ? qb : 0.0);
                acc += (C[c].x - shift) * o.x + (C[c].y - shift) * o.y;
                if (lane_ok) put(a.out + off, o);
            } else {
                const double2 rv = *reinterpret_cast<const double2*>(a.rhs + (lane_ok ? off : 0));
                const double2 o = make_double2(ma ? rv.x - res_scale * qa : 0.0, mb ? rv.y - res_scale * qb : 0.0);
                acc += o.x * o.x + o.y * o.y;
                if (lane_ok) put(a.out + off, o);
            }
        }
    }
    const double tot = block_sum_256(acc, sRed);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    if (K1 && a.partials2) {
        __syncthreads();
        dot_finish(block_sum_256(acc2, sRed), a.partials2);
    }
}

FastGeom fast_geom(const pdegpu_ctx* ctx, const GridDev& g, int RY, int wg_per_cu, int mode, int lk_lo = 0, int lk_hi = 0, int slack8 = 0, int resident_cap = 0,
                   int lk_lo2 = 0, int lk_hi2 = 0) {
    FastGeom fg{};
    fg.ntx = (g.NX - 1 + 127) / 128;
    fg.jbase = g.flat2d ? 0 : 1;
    fg.jmax = g.flat2d ? 0 : g.NY - 2;
    fg.nty = (fg.jmax - fg.jbase + 1 + RY - 1) / RY;
    // waves of a workgroup along x: as many (4, 2, 1) as divide the tile count — a workgroup column with idle waves wastes their slots for the
    // whole march (577 nodes per row = 5 tiles: 4 + 1 left 3 of 8 waves idle, 52 instead of ~85 GDoF/s); the price of fewer x-adjacent waves
    // is more tile-edge values from global memory instead of the LDS hand-off
    fg.wxt = (fg.ntx % 4 == 0) ? 4 : ((fg.ntx % 2 == 0) ? 2 : 1);
    fg.wyt = 4 / fg.wxt;
    fg.nxg = (fg.ntx + fg.wxt - 1) / fg.wxt;
    fg.nyg = (fg.nty + fg.wyt - 1) / fg.wyt;
    // fast_linear (default): a workgroup takes four CONSECUTIVE tiles of the row-major tile order instead of a wxt x wyt block, so three of
    // every four tile boundaries inside a row are handed over through LDS at ANY tile count per row (5 or 7 tiles: none before, 6: three of
    // five); for 1, 2 and 4k tiles per row the two orders are the same workgroups
    fg.linear = ctx->fast_linear != 0;
    if (fg.linear) {
        fg.nxg = (int)(((long long)fg.ntx * fg.nty + 3) / 4);
        fg.nyg = 1;
    }
    fg.lkb = g.k0 >= 1 ? 0 : 1 - g.k0;
    fg.lke = (g.NZ - 1 - g.k0) < g.nkl ? (g.NZ - 1 - g.k0) : g.nkl;
    fg.lkb2 = fg.lke2 = 0;
    if (lk_hi2 > lk_lo2) {                     // second plane range of the same launch, clipped to the free planes like the first
        fg.lkb2 = std::max(fg.lkb, lk_lo2);
        fg.lke2 = std::min(fg.lke, lk_hi2);
        if (fg.lke2 < fg.lkb2) fg.lke2 = fg.lkb2;
    }
    if (lk_hi > lk_lo) {                       // plane-range launch (K1 split of the multi-GPU loop)
        if (lk_lo > fg.lkb) fg.lkb = lk_lo;
        if (lk_hi < fg.lke) fg.lke = lk_hi;
    }
    int planes = fg.lke - fg.lkb;
    if (planes < 0) planes = 0;
    const int xy = fg.nxg * fg.nyg;
    // one resident round: (workgroups that fit per CU) x CUs, at least 8 planes per chunk
    int target = wg_per_cu * ctx->num_cu;
    // slack8 (HBM-resident grids, launch_fast): the chunk count is set directly by the fitted efficiency model further down instead of by
    // the rounds x planes model of the cache-resident grids
    int direct_nzc = 0;
    // chunks of >= 8 planes keep the redundant two-plane prologue below 25 % when there is parallelism to spare; a grid too
    // small to fill the device is latency-bound on the serial march instead (every plane step is a dependent memory round
    // trip), so it gets shorter chunks (down to fast_minlen planes) and up to fast_small_mult resident rounds of workgroups
    int minlen = (int)((long long)planes * xy / (target > 0 ? target : 1));
    if (minlen < 8) {
        target *= ctx->fast_small_mult > 0 ? ctx->fast_small_mult : 1;
        minlen = (int)((long long)planes * xy / (target > 0 ? target : 1));
    }
    if (minlen > 8) minlen = 8;
    if (minlen < ctx->fast_minlen) minlen = ctx->fast_minlen;
    if (minlen < 1) minlen = 1;
    int maxc = planes / minlen;
    if (maxc < 1) maxc = 1;
    // number of z-chunks: minimise (resident rounds) x (planes marched per chunk, two-plane prologue included).  One round
    // when the x-y tiling alone does not fill the device; when it nearly does (1024^3: 1024 of 1280 slots) more, shorter
    // chunks that fill whole rounds beat one partly filled round.
    int nzc = 1;
    long long best = -1;
    // (a single chunk that overflows what is resident at once — 1024^3: 1024 workgroups on 768 slots, 897 nodes per row: 784 — leaves a thin
    // second round marching the whole column: there the rounds x planes model below decides, with the resident capacity as its round size)
    const bool overflow = resident_cap > 0 && xy > resident_cap;
    if (overflow) target = resident_cap;
    if (slack8 && xy > 0) {
        // Chunk count of an HBM-resident grid, fitted to the z-chunk-count sweeps of 19 cube sizes (profiles/r02_exp16_chunk_count_sweep.log):
        //  * one resident round of W = xy * c workgroups runs best with W just below 2 per CU (512^3: 2 x 256 = 512 -> K1 0.78 of the HBM
        //    peak; W = 768 — three per CU, all resident — 0.64: more concurrent streams, not more bandwidth); below that it degrades like
        //    W / 512 (K2 faster: 448 -> 0.82), and the stencil-K2 (fewer streams per workgroup) is as good with W just below 3 per CU;
        //  * otherwise MANY chunks (W >> 512) that the dispatcher hands out as slots free up reach ~0.92 of the perfect fit, whatever the
        //    exact count — where the registers allow exactly 2 workgroups per CU (bare apply with 4 rows and the deeper prefetch) the
        //    rounds are real and W just below a multiple of 512 is best;
        //  * what the old rule (as few chunks as give every CU 2 workgroups, then 1/8 slack) got wrong is everything in between: 768^3 ran
        //    ONE chunk of 576 workgroups (0.56 of the peak; now 12 chunks, 0.72), 700^3 525 (0.49 -> 0.68), 448^3 3 x 224 (0.58 -> 0.70).
        // Every chunk marches a two-plane prologue (useful = planes / (c * (zlen + 2))).
        const double S1 = 2.0 * ctx->num_cu;
        const bool k2 = (mode == APPLY_CG_K2);
        const bool hard_rounds = resident_cap > 0 && resident_cap <= 2 * ctx->num_cu;
        const long long pcap = stencil_max_partials(const_cast<pdegpu_ctx*>(ctx));
        double best_eff = -1.0;
        for (int c = 1; c <= maxc; ++c) {
            const int zl = (planes + c - 1) / c;
            if ((planes + zl - 1) / zl != c) continue;           // (this chunk length gives fewer chunks: already scored)
            const double W = (double)xy * c;
            if ((long long)xy * c > pcap && c > 1) break;
            const double useful = (double)planes / ((double)c * (zl + 2));
            double q;
            if (W <= S1) q = k2 ? std::pow(W / S1, 1.5) : W / S1;
            else if (k2 && W <= 1.5 * S1 && (resident_cap == 0 || W <= resident_cap)) q = 0.985 * W / (1.5 * S1);
            else if (hard_rounds) q = 0.97 * W / (std::ceil(W / S1) * S1);
            else q = 0.92 * W / (W + 0.5 * S1);                 // (tail: about half a round of workgroups short of a full device)
            if (W > S1 && c > 1 && zl < 24 && planes >= 48) continue;      // (chunks this short pay more than the two planes: ramp-up per workgroup)
            const double eff = q * useful;
            if (eff > best_eff * 1.005) {                        // (ties: fewer, longer chunks)
                best_eff = eff;
                direct_nzc = c;
            }
        }
        if (direct_nzc < 1) direct_nzc = 1;
    }
    for (int c = 1; c <= maxc && direct_nzc == 0; ++c) {
        const long long rounds = ((long long)xy * c + target - 1) / (target > 0 ? target : 1);
        const long long cost = rounds * ((planes + c - 1) / c + 2);
        if (best < 0 || cost < best) {
            best = cost;
            nzc = c;
        }
    }
    if (direct_nzc > 0) nzc = direct_nzc;
    if (ctx->fast_zlen > 0 && planes > 0) nzc = (planes + ctx->fast_zlen - 1) / ctx->fast_zlen;      // forced chunk length (tuning)
    {   // one partial per workgroup: the grid must fit the session's partial buffer
        const long long cap = stencil_max_partials(const_cast<pdegpu_ctx*>(ctx));
        if ((long long)xy * nzc > cap) nzc = (int)std::max<long long>(1, cap / (xy > 0 ? xy : 1));
    }
    fg.zlen = planes > 0 ? (planes + nzc - 1) / nzc : 1;
    fg.nzc = planes > 0 ? (planes + fg.zlen - 1) / fg.zlen : 0;
    fg.nzc1 = fg.nzc;
    if (fg.lke2 > fg.lkb2) {                   // the second range is cut with the first one's chunk length (edge planes: one plane each)
        if (planes <= 0) fg.zlen = fg.lke2 - fg.lkb2;
        fg.nzc += (fg.lke2 - fg.lkb2 + fg.zlen - 1) / fg.zlen;
    }
    return fg;
}

struct ModeCfg { int ry, wg, ex, pf; };
// rows per thread, resident workgroups per CU, LDS edge hand-off and deeper prefetch of one marching-kernel mode
ModeCfg mode_cfg(const pdegpu_ctx* ctx, int mode) {
    ModeCfg c;
    c.ry = ctx->fast_ry > 0 ? ctx->fast_ry : ((mode == APPLY_PLAIN || mode == APPLY_RESIDUAL) ? 4 : 2);
    c.wg = (mode == APPLY_CG_K2 && ctx->fast_wg_per_cu_k2 >= 1) ? ctx->fast_wg_per_cu_k2 : ctx->fast_wg_per_cu;
    c.ex = ctx->fast_ex && (mode == APPLY_CG_K1 || mode == APPLY_CG_K2);
    // deeper prefetch: on by default in the bare apply / residual pass (fast_pf = -1; 512^3 SpMV 0.382 -> 0.376 ms, 400^3 0.208 -> 0.195, 640^3
    // 0.955 -> 0.910), 1 = also in the stencil-K2, 0 = nowhere
    c.pf = ctx->fast_pf < 0 ? (mode == APPLY_PLAIN || mode == APPLY_RESIDUAL) : (ctx->fast_pf != 0 && mode != APPLY_CG_K1);
    const int v = ctx->fast_cfg[mode & 3];
    if (v >= 0) {
        c.ry = v & 15;
        c.wg = (v >> 4) & 15;
        c.ex = (v >> 8) & 1;
        c.pf = ((v >> 9) & 1) && mode != APPLY_CG_K1;
    }
    return c;
}

template <int NPT, int RY, int NT, int EX, int PF>
int launch_fast(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li) {
    static int occ[4] = {0, 0, 0, 0};   // resident workgroups per CU of each instantiation
    if (occ[mode] == 0) {
        int nb = 0;
        hipError_t e = hipSuccess;
        switch (mode) {
            case APPLY_PLAIN: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fast<NPT, RY, APPLY_PLAIN, NT, EX, PF, 0>, 256, 0); break;
            case APPLY_RESIDUAL: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fast<NPT, RY, APPLY_RESIDUAL, NT, EX, PF, 0>, 256, 0); break;
            case APPLY_CG_K2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fast<NPT, RY, APPLY_CG_K2, NT, EX, PF, 1>, 256, 0); break;
            default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fast<NPT, RY, APPLY_CG_K1, NT, EX, 0, 1>, 256, 0); break;
        }
        if (e != hipSuccess || nb < 1) nb = 2;
        if (nb > 8) nb = 8;
        occ[mode] = nb;
    }
    int wg_per_cu = occ[mode];
    const ModeCfg mc = mode_cfg(ctx, mode);
    const int res_occ = occ[mode];
    // HBM-resident vectors (the same threshold as the non-temporal hints): two workgroups per CU, i.e. as few and as long z-chunks
    // as fill the device; fast_geom adds 1/8 of slack so that e.g. 3 x 192 tiles count as one round
    const bool hbm = (long long)a.g.nkl * a.g.PP * 8 > (64LL << 20);
    int slack8 = 0;
    if (mc.wg >= 1 && mc.wg <= 16) wg_per_cu = mc.wg;
    else if (hbm) { wg_per_cu = 2; slack8 = 1; }
    // (a K1 that leaves x alone streams what the stencil-K2 streams — two vectors read, one written — but K2's chunk policy does not suit it better
    // than its own: measured over 16 cube sizes, round 4)
    FastGeom fg = fast_geom(ctx, a.g, RY, wg_per_cu, mode, a.lk_lo, a.lk_hi, slack8, res_occ * ctx->num_cu, a.lk_lo2, a.lk_hi2);
    // a multiple of 8 workgroups, so that the XCD-aware tile order of the kernel always applies (the surplus ones find zc >= nzc and idle;
    // 701 nodes per row: 525 workgroups -> 528)
    const int G = (fg.nxg * fg.nyg * fg.nzc + 7) / 8 * 8;
    if (li) { li->grid = G; li->block = 256; }
    {   // PDEGPU_DEBUG_GEOM=1: the geometry of every new (mode, chunking) on stderr (tools/)
        static const bool dbg = getenv("PDEGPU_DEBUG_GEOM") != nullptr;
        static long long last[4] = {-1, -1, -1, -1};
        const long long key = ((long long)fg.nxg * fg.nyg) * 100000 + fg.nzc * 1000LL + a.g.NX % 1000;
        if (dbg && last[mode & 3] != key) {
            last[mode & 3] = key;
            fprintf(stderr, "[pdegpu geom] mode %d npt %d ry %d ex %d pf %d: %d x %d x %d nodes, xy %d x %d workgroups (%d x %d waves), %d chunks of %d planes, occ %d -> grid %d\n",
                    mode, NPT, RY, EX, PF, a.g.NX, a.g.NY, a.g.nkl, fg.nxg, fg.nyg, fg.wxt, fg.wyt, fg.nzc, fg.zlen, occ[mode], G);
        }
    }
    if (G == 0 || (li && li->dry)) return PDEGPU_OK;
    const bool k2x = a.x_in_k2 && a.xmode == 2, k1x = !a.x_in_k2;
    switch (mode) {
        case APPLY_PLAIN: hipLaunchKernelGGL((k_fast<NPT, RY, APPLY_PLAIN, NT, EX, PF, 0>), dim3(G), dim3(256), 0, ctx->stream, a, fg); break;
        case APPLY_RESIDUAL: hipLaunchKernelGGL((k_fast<NPT, RY, APPLY_RESIDUAL, NT, EX, PF, 0>), dim3(G), dim3(256), 0, ctx->stream, a, fg); break;
        case APPLY_CG_K2:
            if (k2x) hipLaunchKernelGGL((k_fast<NPT, RY, APPLY_CG_K2, NT, EX, PF, 1>), dim3(G), dim3(256), 0, ctx->stream, a, fg);
            else hipLaunchKernelGGL((k_fast<NPT, RY, APPLY_CG_K2, NT, EX, PF, 0>), dim3(G), dim3(256), 0, ctx->stream, a, fg);
            break;
        default:
            if (k1x) hipLaunchKernelGGL((k_fast<NPT, RY, APPLY_CG_K1, NT, EX, 0, 1>), dim3(G), dim3(256), 0, ctx->stream, a, fg);
            else hipLaunchKernelGGL((k_fast<NPT, RY, APPLY_CG_K1, NT, EX, 0, 0>), dim3(G), dim3(256), 0, ctx->stream, a, fg);
            break;
    }
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

template <int NPT, int RY>
int launch_fast_nt(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li) {
    int nt = ctx->fast_nt;
    if (nt < 0) nt = ((long long)a.g.nkl * a.g.PP * 8 > (64LL << 20)) ? 3 : 0;
    const ModeCfg mc = mode_cfg(ctx, mode);
    // deeper prefetch: one-vector modes of every operator, RY <= 4 (512^3 bare apply, one box, interleaved, 4 rows per thread: 15 points 0.433-0.449 ->
    // 0.403-0.407 ms; 27 points (Q1) 0.535-0.560 -> 0.414-0.418 ms; no spill: 117-119 / 149-173 VGPRs)
    const bool pf = mc.pf && RY <= 4;
    if (mc.ex) {
        if constexpr (RY <= 4) {
            if (pf) return (nt & 3) == 3 ? launch_fast<NPT, RY, 3, 1, 1>(ctx, a, mode, li) : launch_fast<NPT, RY, 0, 1, 1>(ctx, a, mode, li);
        }
        switch (nt & 3) {
            case 3: return launch_fast<NPT, RY, 3, 1, 0>(ctx, a, mode, li);
            default: return launch_fast<NPT, RY, 0, 1, 0>(ctx, a, mode, li);
        }
    }
    if constexpr (RY <= 4) {
        if (pf) return (nt & 3) == 3 ? launch_fast<NPT, RY, 3, 0, 1>(ctx, a, mode, li) : launch_fast<NPT, RY, 0, 0, 1>(ctx, a, mode, li);
    }
    switch (nt & 3) {
        case 1: return launch_fast<NPT, RY, 1, 0, 0>(ctx, a, mode, li);
        case 2: return launch_fast<NPT, RY, 2, 0, 0>(ctx, a, mode, li);
        case 3: return launch_fast<NPT, RY, 3, 0, 0>(ctx, a, mode, li);
        default: return launch_fast<NPT, RY, 0, 0, 0>(ctx, a, mode, li);
    }
}

}  // namespace

// 3-D grids: CG_K1 / CG_K2 / PLAIN as one-round tiles (k_tile3d); a.g is the full single-slab grid with uniform Dirichlet boundary.
// Grids beyond L2 (more than 4 M nodes) take the XCD-aware order with non-temporal stores.
static bool tile3d_xm(const GridDev& g) { return (long long)g.NX * g.NY * g.NZ > 4000000LL; }
template <int NPT>
static int launch_tile3d(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li) {
    constexpr int RY = 2;
    const int ntx = (a.g.NX - 1 + 127) / 128;
    const int nty = (a.g.NY - 2 + RY - 1) / RY;
    const long long G = stencil_tile3d_grid(a.g);
    if (li) { li->grid = (int)G; li->block = 256; }
    if (G <= 0) return PDEGPU_OK;
    const dim3 grid((unsigned)G), blk(256);
    if (tile3d_xm(a.g)) {
        if (mode == APPLY_CG_K1) hipLaunchKernelGGL((k_tile3d<NPT, RY, APPLY_CG_K1, 1>), grid, blk, 0, ctx->stream, a, ntx, nty);
        else if (mode == APPLY_CG_K2) hipLaunchKernelGGL((k_tile3d<NPT, RY, APPLY_CG_K2, 1>), grid, blk, 0, ctx->stream, a, ntx, nty);
        else hipLaunchKernelGGL((k_tile3d<NPT, RY, APPLY_PLAIN, 1>), grid, blk, 0, ctx->stream, a, ntx, nty);
    } else {
        if (mode == APPLY_CG_K1) hipLaunchKernelGGL((k_tile3d<NPT, RY, APPLY_CG_K1, 0>), grid, blk, 0, ctx->stream, a, ntx, nty);
        else if (mode == APPLY_CG_K2) hipLaunchKernelGGL((k_tile3d<NPT, RY, APPLY_CG_K2, 0>), grid, blk, 0, ctx->stream, a, ntx, nty);
        else hipLaunchKernelGGL((k_tile3d<NPT, RY, APPLY_PLAIN, 0>), grid, blk, 0, ctx->stream, a, ntx, nty);
    }
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}
int stencil_tile3d(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li) {
    if (mode != APPLY_CG_K1 && mode != APPLY_CG_K2 && mode != APPLY_PLAIN) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "tile form: K1, K2 or PLAIN");
    if (a.npt == 27) return launch_tile3d<27>(ctx, a, mode, li);
    if (a.npt == 7) return launch_tile3d<7>(ctx, a, mode, li);
    return launch_tile3d<15>(ctx, a, mode, li);
}
// workgroups (= partials) the tile form launches
long long stencil_tile3d_grid(const GridDev& g) {
    const int ntx = (g.NX - 1 + 127) / 128, nty = (g.NY - 2 + 1) / 2, nyg = (nty + 3) / 4;
    if (tile3d_xm(g)) return 8LL * ntx * ((nyg + 7) / 8) * (g.NZ - 2);
    return (long long)ntx * nyg * (g.NZ - 2);
}


// the two CG kernels of a 2-D grid without marching (k_plane2d); a.g is the 2-D grid itself, a.w = {w0, wx, wy, wd1, wd2}
template <int RY>
static int launch_plane2d(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li) {
    const int ntx = (a.g.NX - 1 + 127) / 128;
    const int nty = (a.g.NY - 2 + RY - 1) / RY;
    const int G = ntx * ((nty + 3) / 4);
    if (li) { li->grid = G; li->block = 256; }
    if (G == 0) return PDEGPU_OK;
    if (mode == APPLY_CG_K1) hipLaunchKernelGGL((k_plane2d<RY, APPLY_CG_K1>), dim3(G), dim3(256), 0, ctx->stream, a, ntx, nty);
    else hipLaunchKernelGGL((k_plane2d<RY, APPLY_CG_K2>), dim3(G), dim3(256), 0, ctx->stream, a, ntx, nty);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}
int stencil_plane2d(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li) {
    if (ctx->plane2d_ry == 2) return launch_plane2d<2>(ctx, a, mode, li);
    if (ctx->plane2d_ry == 8) return launch_plane2d<8>(ctx, a, mode, li);
    return launch_plane2d<4>(ctx, a, mode, li);
}

int stencil_general(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li) {
    const long long total = a.bnd_n > 0 ? a.bnd_n : (long long)a.g.nkl * a.g.PP;
    long long nb = (total + 255) / 256;
    const long long cap = (long long)ctx->grid_mult * ctx->num_cu;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    if (li) { li->grid = (int)nb; li->block = 256; }
    if (a.ell) {
        switch (mode) {
            case APPLY_PLAIN: hipLaunchKernelGGL((k_general<APPLY_PLAIN, true>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, a); break;
            case APPLY_RESIDUAL: hipLaunchKernelGGL((k_general<APPLY_RESIDUAL, true>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, a); break;
            default: hipLaunchKernelGGL((k_general<APPLY_CG_K1, true>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, a); break;
        }
    } else {
        switch (mode) {
            case APPLY_PLAIN: hipLaunchKernelGGL((k_general<APPLY_PLAIN, false>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, a); break;
            case APPLY_RESIDUAL: hipLaunchKernelGGL((k_general<APPLY_RESIDUAL, false>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, a); break;
            case APPLY_CG_K2: hipLaunchKernelGGL((k_general<APPLY_CG_K2, false>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, a); break;
            default: hipLaunchKernelGGL((k_general<APPLY_CG_K1, false>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, a); break;
        }
    }
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

// fast path: 3-D (or 2-D through the flat view), P1 split, every boundary node Dirichlet (uniform mode), at least one
// interior node per axis
bool stencil_fast_supported(const GridDev& g, int element_mode) {
    if (element_mode != PDEGPU_ELEM_P1SPLIT && element_mode != PDEGPU_ELEM_Q1) return false;
    if (g.dim == 3) return g.NX >= 3 && g.NY >= 3 && g.NZ >= 3;       // natural faces: hybrid (stencil_all_dirichlet false)
    if (g.bc_mode != PDEGPU_BC_UNIFORM || !g.bc_present[0]) return false;
    if (g.dim == 2) return g.NX >= 3 && g.NY >= 3;
    return false;
}
// every boundary node constrained: the marching kernels alone cover the free nodes
bool stencil_all_dirichlet(const GridDev& g) { return g.bc_mode == PDEGPU_BC_UNIFORM && g.bc_present[0]; }

// The 2-D P1 operator (5-point stiffness, 7-point mass with the +-(1,1) diagonal of the "right" triangle split) is the
// 3-D 15-point kernel restricted to ONE row of the middle axis: (x, y) -> (x, z), weights w0, wx, wz = w_y2d,
// wxz = w_xy2d.  Memory is unchanged: a 2-D row is one "plane" of pitch PX.
GridDev stencil_flat2d_view(const GridDev& g) {
    GridDev v = g;
    v.dim = 3;
    v.NZ = g.NY;
    v.NY = 1;
    v.PP = g.PX;
    v.nkl = g.NY;
    v.k0 = 0;
    v.flat2d = 1;
    return v;
}

// rows per thread of a marching launch (7- / 15-point operator; Q1 is fixed at 2)
static int fast_rows(const pdegpu_ctx* ctx, const GridDev& g, int npt, int mode, int x_in_k2 = 0) {
    const bool hbm = (long long)g.nkl * g.PP * 8 > (64LL << 20);
    int ry = mode_cfg(ctx, mode).ry;
    // a K1 that leaves the solution vector alone (x_in_k2 sessions) holds two vectors instead of three per row: with 4 rows per thread (182 VGPRs; 212 with
    // 15 points; half the halo rows per owned row) it is faster at every cube from 256 to 1024 cells but 512, where it is the same (7 points: K1
    // 0.59-0.69 -> 0.62-0.72 of the HBM peak, 448^3 +3 %, 640^3 +4 % per iteration in one process; 15 points at 512^3: 0.613 -> 0.555 ms)
    if (mode == APPLY_CG_K1 && x_in_k2 && ctx->fast_ry == 0 && ctx->fast_cfg[mode & 3] < 0 && (npt == 7 || npt == 15) && g.dim == 3 && !g.flat2d && hbm) ry = 4;
    // stencil-K2 with 4 rows per thread: only where that makes ONE resident round of (almost) exactly two workgroups per CU and 2 rows do not
    // (1024^3: 8 tiles x 256 row tiles / 4 = 512 workgroups, K2 0.757 of the HBM peak against 0.644 with many chunks of 2-row tiles; at
    // 832^3 / 896^3 / 960^3, where neither fits, 2 rows win by 2-10 % under the fitted chunk policy: profiles/r02_ab_k2_rows.log)
    if (mode == APPLY_CG_K2 && ctx->fast_ry == 0 && ctx->fast_cfg[mode & 3] < 0 && npt == 7 && g.dim == 3 && !g.flat2d && hbm) {
        const long long ntx = (g.NX - 1 + 127) / 128, rows = g.NY - 2;
        const double S1 = 2.0 * ctx->num_cu;
        auto fits = [&](int r) {
            const long long xy = (ntx * ((rows + r - 1) / r) + 3) / 4;
            for (int c = 1; c <= 8; ++c)
                if ((double)xy * c <= S1 && (double)xy * c >= 0.97 * S1) return true;
            return false;
        };
        if (!fits(2) && fits(4)) ry = 4;
    }
    return ry;
}

int stencil_fast(pdegpu_ctx* ctx, const StencilArgs& a_in, int mode, LaunchInfo* li) {
    StencilArgs a = a_in;
    if (a.npt == 27) return launch_fast_nt<27, 2>(ctx, a, mode, li);        // Q1: rows per thread fixed at 2
    const int ry = fast_rows(ctx, a.g, a.npt, mode, a.x_in_k2);
    if (a.npt == 7) {
        if (ry == 1) return launch_fast_nt<7, 1>(ctx, a, mode, li);
        if (ry == 2) return launch_fast_nt<7, 2>(ctx, a, mode, li);
        if (ry == 8) return launch_fast_nt<7, 8>(ctx, a, mode, li);
        return launch_fast_nt<7, 4>(ctx, a, mode, li);
    }
    if (ry == 1) return launch_fast_nt<15, 1>(ctx, a, mode, li);
    if (ry == 2) return launch_fast_nt<15, 2>(ctx, a, mode, li);
    return launch_fast_nt<15, 4>(ctx, a, mode, li);
}

int stencil_max_partials(pdegpu_ctx* ctx) { return 64 * ctx->num_cu + 4096; }


// Host-only (no GPU): the launch geometry the 7-point marching kernels of a 3-D single-slab grid would get on a device with num_cu compute
// units, with the default tunables — what tests/ pin so that a change of the chunk policy cannot silently move the headline geometry.
extern "C" int pdegpu_marching_geometry(int num_cu, int nx_nodes, int ny_nodes, int nz_nodes, int mode, int resident_wg_per_cu, int32_t* out4) {
    if (!out4 || num_cu < 1 || nx_nodes < 3 || ny_nodes < 3 || nz_nodes < 3 || mode < 0 || mode > 3 || resident_wg_per_cu < 1) return PDEGPU_EINVAL;
    pdegpu_ctx c;
    c.num_cu = num_cu;
    GridDev g{};
    g.dim = 3;
    g.NX = nx_nodes;
    g.NY = ny_nodes;
    g.NZ = nz_nodes;
    g.PX = (nx_nodes + 15) / 16 * 16;
    g.PP = (long long)g.PX * ny_nodes;
    g.nkl = nz_nodes;
    g.k0 = 0;
    g.bc_mode = PDEGPU_BC_UNIFORM;
    g.bc_present[0] = 1;
    const int ry = fast_rows(&c, g, 7, mode, c.cg_xdefer != 0);      // (the default configuration: K1 leaves the solution vector to K2)
    const bool hbm = (long long)g.nkl * g.PP * 8 > (64LL << 20);
    const ModeCfg mc = mode_cfg(&c, mode);
    int wg = resident_wg_per_cu, slack8 = 0;
    if (mc.wg >= 1 && mc.wg <= 16) wg = mc.wg;
    else if (hbm) { wg = 2; slack8 = 1; }
    const FastGeom fg = fast_geom(&c, g, ry, wg, mode, 0, 0, slack8, resident_wg_per_cu * num_cu);
    out4[0] = ry;
    out4[1] = fg.nxg * fg.nyg;
    out4[2] = fg.nzc;
    out4[3] = fg.zlen;
    return PDEGPU_OK;
}
