// dot_finish.h — how the CG kernels hand dot products from one launch to the next (gfx950).
//
// Every dot-producing kernel stores ONE partial per workgroup (dot_finish).  The consumer is either a one-workgroup reduce launch
// (k_reduce: the multi-GPU loop, whose all-reduce needs the scalar in memory) or — single-slab loops — the prologue of the next
// kernel, where every workgroup sums all partials in the same fixed order (dot_in_sum) and so holds the bitwise identical scalar.
// (Round 1 also finished the sum inside the producing launch with a last-workgroup ticket; it was measured slower — every workgroup's
// agent-scope release writes back its XCD's dirty L2 lines in a streaming kernel — and was removed in round 3.)
#pragma once
#include "common.h"

__device__ __forceinline__ double df_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Prologue completion of the PREVIOUS launch's dot product (DotIn, common.h): every workgroup of the consuming kernel
// sums all partials in the order k_reduce uses (per-thread stride 256, 64-lane shuffle tree, (s0+s1)+(s2+s3)), so every
// workgroup holds the bitwise identical scalar and no reduce launch is needed; the partials (a few KB) come from L2.
// All 256 threads must call; sRed: >= 4 doubles of LDS.
__device__ __forceinline__ double dot_in_sum(const double* __restrict__ part, int n, double* sRed) {
    double acc = 0.0;
    for (int e = threadIdx.x; e < n; e += 256) acc += part[e];
    acc = df_wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = acc;
    __syncthreads();
    return (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
}
// K1 side: rr_n (summed here or read from its slot), rr_{n-1}, pq_{n-1} -> alpha_{n-1}, beta_{n-1}
__device__ __forceinline__ void cg_scalars_in(const DotIn& di, double* scal, double* sRed, double* alpha_prev, double* beta) {
    if (di.sr) {
        const double pq = scal[SC_PQ], rr_n = scal[SC_RR], qq = scal[SC_QQ];
        const double a = (pq > 0.0 && rr_n > 0.0) ? rr_n / pq : 0.0;             // alpha_{n-1}; 0 on the first iteration (pq = 0)
        const double rr_next = a * a * qq - rr_n;                                 // |r_n|^2 predicted from the block of iteration n-1
        *alpha_prev = di.skip_x ? 0.0 : a;
        *beta = (a > 0.0 && rr_next > 0.0) ? rr_next / rr_n : 0.0;                // a lost prediction restarts with steepest descent
        return;
    }
    double rr;
    if (di.part) {
        rr = dot_in_sum(di.part, di.n, sRed);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            scal[di.s_rr] = rr;
            scal[SC_RR_LOCAL] = rr;
        }
    } else {
        rr = scal[di.s_rr];
    }
    const double pq = scal[SC_PQ], rr_prev = scal[di.s_rr_old];
    // alpha_{n-1} = rr_{n-1} / (p_{n-1}.q_{n-1}),  beta_{n-1} = rr_n / rr_{n-1};  both 0 on the first iteration
    *alpha_prev = (pq > 0.0 && rr_prev > 0.0) ? rr_prev / pq : 0.0;
    *beta = (rr_prev > 0.0) ? rr / rr_prev : 0.0;
}
// K2 side: alpha_n = rr_n / (p_n.q_n), p_n.q_n summed here or read from SC_PQ
__device__ __forceinline__ double cg_alpha_in(const DotIn& di, double* scal, double* sRed) {
    if (di.sr) {
        const double pq_n = scal[SC_PQ];
        return (pq_n > 0.0) ? scal[SC_RR] / pq_n : 0.0;
    }
    double pq;
    if (di.part) {
        pq = dot_in_sum(di.part, di.n, sRed);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            scal[SC_PQ] = pq;
            scal[SC_PQ_LOCAL] = pq;
        }
    } else {
        pq = scal[SC_PQ];
    }
    const double rr = scal[di.s_rr];
    return (pq > 0.0) ? rr / pq : 0.0;
}

// x_in_k2 sessions (StencilArgs::x_in_k2): the K2 that carries the x update does  x += cx_p p_n - cx_r r_n + cx_o p_{n-1}  — this iteration's
// alpha_n p_n and, when the K2 before left its update out (xmode 2, its alpha in xd[0]), that one too, with p_{n-1} = (p_n - r_n) / beta_n rebuilt
// from K1's p_n = r_n + beta_n p_{n-1} (beta_n in xd[1]); a restart or a tiny beta cannot rebuild it: cx_o then takes p_{n-1} from the other p buffer
__device__ __forceinline__ void xk2_coefficients(const double* xd, int xmode, double alpha, double* cx_p, double* cx_r, double* cx_o) {
    *cx_p = alpha;
    *cx_r = 0.0;
    *cx_o = 0.0;
    const double ap = xd[0], bn = xd[1];
    if (xmode == 2 && ap != 0.0) {
        if (bn > 1e-2) {
            *cx_r = ap / bn;
            *cx_p = alpha + *cx_r;
        } else {
            *cx_o = ap;
        }
    }
}

// block_total is valid in thread 0: one partial per workgroup.
__device__ __forceinline__ void dot_finish(double block_total, double* partials) {
    if (threadIdx.x == 0) partials[blockIdx.x] = block_total;
}
