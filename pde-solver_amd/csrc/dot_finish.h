// dot_finish.h — in-launch completion of a grid-wide dot product (gfx950).
//
// Every workgroup stores its partial sum, publishes it with an agent-scope release and takes a ticket from a
// device-scope counter; the workgroup that draws the last ticket acquires (invalidates its CU's L1) and sums
// ALL partials in a fixed order — the result is bitwise independent of which workgroup happens to be last —
// then updates the CG scalars and re-arms the counter.  This replaces a separate one-workgroup reduce launch
// after every dot-producing kernel (2 launches per CG iteration).
// Protocol per /opt/skills/guides/cdna_hip_programming.md §6 Guideline 16: producer = plain store ->
// __threadfence() (buffer_wbl2 sc1 + buffer_inv sc1) -> explicit s_waitcnt vmcnt(0) (the compiler may drop the
// wait after the write-back) -> atomic ticket; consumer = ticket -> agent acquire fence -> barrier -> plain loads.
// No dispatch-order / placement assumption: the last ticket can only be drawn after every workgroup has arrived.
#pragma once
#include "common.h"

struct DotSink {
    double* partials;       // [gridDim.x]
    unsigned* counter;      // device word, zero between launches
    double* scal;           // CG scalar block (SC_*)
    int which;              // 0: p.q -> SC_PQ_LOCAL / SC_PQ;  1: r.r -> SC_RR_PREV <- SC_RR, SC_RR_LOCAL / SC_RR;  < 0: no scalar update
};

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

// block_total is valid in thread 0.  sRed: >= 4 doubles of LDS.  All 256 threads must call.
__device__ __forceinline__ void dot_finish(double block_total, const DotSink& d, double* sRed) {
    __shared__ int sLast;
    if (threadIdx.x == 0) {
        d.partials[blockIdx.x] = block_total;
        int last = 0;
        if (d.which >= 0) {
            __threadfence();                                         // agent-scope release of the partial
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned ticket = atomicAdd(d.counter, 1u);
            last = (ticket == gridDim.x - 1);
        }
        sLast = last;
    }
    __syncthreads();
    if (sLast) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // this CU's L1 may not hold other CUs' partials
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        double acc = 0.0;
        const int n = (int)gridDim.x;
        for (int e = threadIdx.x; e < n; e += 256) acc += d.partials[e];     // same order as the former k_reduce
        acc = df_wave_sum(acc);
        if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double tot = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
            if (d.which == 0) {
                d.scal[SC_PQ_LOCAL] = tot;
                d.scal[SC_PQ] = tot;
            } else {
                d.scal[SC_RR_PREV] = d.scal[SC_RR];
                d.scal[SC_RR_LOCAL] = tot;
                d.scal[SC_RR] = tot;
            }
            *d.counter = 0u;                                         // re-armed for the next launch (kernel boundary orders it)
        }
    }
}
