// vecops.hip — flat vector kernels of the CG iteration and slab bookkeeping (gfx950).
//
// K2 of the fused CG:  alpha_n = rr_n / (p_n.q_n);  r_{n+1} = r_n - alpha_n q_n;  partial(r.r) — a pure
// streaming pass (read r, q; write r: 24 B/DoF) with 16-B-per-lane accesses; the dot product is reduced
// per lane -> __shfl_down across the 64-lane wavefront -> LDS across waves -> one partial per workgroup,
// summed in fixed order by k_reduce (bitwise reproducible, no atomics, alpha/beta never leave the device).
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include "dot_finish.h"
#include <cstdarg>

int pdegpu_set_err(pdegpu_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double block_sum_256(double v, double* sm) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void k_set_bc(GridDev g, double* __restrict__ x, double fill,
                                                 const double* __restrict__ nodal, int only_bc) {
    const long long total = (long long)g.nkl * g.PP;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int i = (int)(idx % g.PX);
        const long long row = idx / g.PX;
        const int j = (int)(row % g.NY);
        const int lk = (int)(row / g.NY);
        if (i >= g.NX) { if (!only_bc) x[idx] = 0.0; continue; }
        double v;
        if (pg_dirichlet(g, i, j, g.k0 + lk, &v)) x[idx] = v;
        else if (!only_bc) x[idx] = nodal ? nodal[((long long)lk * g.NY + j) * g.NX + i] : fill;
    }
}

// r -= alpha q, partial r.r      (n2 = number of double2 elements)
// NT: non-temporal loads and stores (HBM-resident vectors: tools/membw.hip, in-place axpy with 2 workgroups per CU 5.6 -> 6.3 TB/s).  A
// compile-time choice: behind a runtime switch the compiler merges the two forms and drops the hint.
typedef double vk_d2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(256) void k_k2(double2* __restrict__ r, const double2* __restrict__ q, long long n2,
                                             double* scal, double* __restrict__ partials, DotIn di) {
    __shared__ double sRed[4];
    const double alpha = cg_alpha_in(di, scal, sRed);
    double acc = 0.0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n2; e += (long long)gridDim.x * 256) {
        double2 rv, qv;
        if constexpr (NT) {
            const vk_d2 a = __builtin_nontemporal_load(reinterpret_cast<const vk_d2*>(r + e));
            const vk_d2 b = __builtin_nontemporal_load(reinterpret_cast<const vk_d2*>(q + e));
            rv = make_double2(a.x, a.y);
            qv = make_double2(b.x, b.y);
        } else {
            rv = r[e];
            qv = q[e];
        }
        rv.x -= alpha * qv.x;
        rv.y -= alpha * qv.y;
        if constexpr (NT) {
            const vk_d2 o = {rv.x, rv.y};
            __builtin_nontemporal_store(o, reinterpret_cast<vk_d2*>(r + e));
        } else {
            r[e] = rv;
        }
        acc += rv.x * rv.x + rv.y * rv.y;
    }
    const double tot = block_sum_256(acc, sRed);
    __syncthreads();
    dot_finish(tot, partials);
}

// streaming ceilings of the device for 16-B-per-lane accesses: ONE element per thread, as many workgroups as elements / 256 (no
// grid-stride loop).  tools/membw.hip, measured on this pool: this shape copies at 6.3-6.7 TB/s, a grid-stride loop of 8
// workgroups per CU over the same vectors at 4.8-5.3 (in a loop every load waits, through the in-order vmcnt counter, for the
// store in front of it; short-lived workgroups retire with their stores in flight and are replaced in dispatch order)
__global__ __launch_bounds__(256) void k_membench(int which, double2* __restrict__ r, const double2* __restrict__ q,
                                                   double2* __restrict__ b, long long n2, double c, double* __restrict__ sink) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n2) return;
    if (which == 0) b[e] = q[e];
    else if (which == 1) { const double2 v = r[e]; if (v.x + v.y == 1.2345e300) sink[0] = v.x; }      // keeps the load alive
    else if (which == 2) b[e] = make_double2(c, c);
    else if (which == 3) { const double2 rv = r[e], qv = q[e]; b[e] = make_double2(rv.x + c * qv.x, rv.y + c * qv.y); }
    else { double2 rv = r[e]; const double2 qv = q[e]; rv.x += c * qv.x; rv.y += c * qv.y; r[e] = rv; }
}

__global__ __launch_bounds__(256) void k_norm2(const double2* __restrict__ a, long long n2, double* __restrict__ partials) {
    __shared__ double sRed[4];
    double acc = 0.0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n2; e += (long long)gridDim.x * 256) {
        const double2 v = a[e];
        acc += v.x * v.x + v.y * v.y;
    }
    const double tot = block_sum_256(acc, sRed);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// x += alpha_prev p   (the update K1 would apply at the start of the next iteration)
__global__ __launch_bounds__(256) void k_flush(double2* __restrict__ x, const double2* __restrict__ p, long long n2,
                                                const double* __restrict__ scal, int sr) {
    const double pq = scal[SC_PQ], rr_prev = scal[sr ? SC_RR : SC_RR_PREV];       // (single-reduction form: the block holds r_n.r_n, p_n.Ap_n)
    double alpha = (pq > 0.0 && rr_prev > 0.0) ? rr_prev / pq : 0.0;
    if (sr == 2) alpha = scal[0];                 // x_in_k2 sessions: scal = {alpha the latest K2 recorded, ...}
    if (alpha == 0.0) return;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n2; e += (long long)gridDim.x * 256) {
        double2 xv = x[e];
        const double2 pv = p[e];
        xv.x += alpha * pv.x;
        xv.y += alpha * pv.y;
        x[e] = xv;
    }
}

// initial guess of the next backward-Euler solve: x <- 2 x - xprev (linear extrapolation in time), xprev <- x.
// Dirichlet nodes hold the same value in both, pads and ghost planes are zero in both: they stay what they are.
__global__ __launch_bounds__(256) void k_extrapolate(double2* __restrict__ x, double2* __restrict__ xprev, long long n2, int first) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n2; e += (long long)gridDim.x * 256) {
        const double2 xv = x[e];
        if (!first) {
            const double2 pv = xprev[e];
            x[e] = make_double2(2.0 * xv.x - pv.x, 2.0 * xv.y - pv.y);
        }
        xprev[e] = xv;
    }
}

// fixed-order sum of the per-workgroup partials of the previous kernel
__global__ __launch_bounds__(256) void k_reduce(const double* __restrict__ partials, int n, double* __restrict__ scal,
                                                 int which, int copy_to_global) {
    __shared__ double sRed[4];
    double acc = 0.0;
    for (int e = threadIdx.x; e < n; e += 256) acc += partials[e];
    const double tot = block_sum_256(acc, sRed);
    if (threadIdx.x == 0) {
        if (which == 0) {
            scal[SC_PQ_LOCAL] = tot;
            if (copy_to_global) scal[SC_PQ] = tot;
        } else {
            if (which == 1) scal[SC_RR_PREV] = scal[SC_RR];
            scal[SC_RR_LOCAL] = tot;
            if (copy_to_global) scal[SC_RR] = tot;
        }
    }
}

// single-reduction form: three fixed-order sums in one launch (K1's p.Ap and Ap.Ap partials, K2's r.r partials of the iteration before)
__global__ __launch_bounds__(256) void k_reduce_sr(const double* __restrict__ pq_part, const double* __restrict__ qq_part, int n1,
                                                    const double* __restrict__ rr_part, int n3, double* __restrict__ scal) {
    __shared__ double sRed[4];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int e = threadIdx.x; e < n1; e += 256) {
        a += pq_part[e];
        b += qq_part[e];
    }
    for (int e = threadIdx.x; e < n3; e += 256) c += rr_part[e];
    const double ta = block_sum_256(a, sRed);
    __syncthreads();
    const double tb = block_sum_256(b, sRed);
    __syncthreads();
    const double tc = block_sum_256(c, sRed);
    if (threadIdx.x == 0) {
        scal[SC_PQ] = ta;
        scal[SC_QQ] = tb;
        scal[SC_RR] = n3 >= 0 ? tc : scal[SC_RR_LOCAL];      // first iteration: THIS slab's r_0.r_0 of the residual pass (the message is summed over ranks)
    }
}
__global__ __launch_bounds__(256) void k_reduce_slot(const double* __restrict__ partials, int n, double* __restrict__ scal, int slot) {
    __shared__ double sRed[4];
    double acc = 0.0;
    for (int e = threadIdx.x; e < n; e += 256) acc += partials[e];
    const double tot = block_sum_256(acc, sRed);
    if (threadIdx.x == 0) scal[slot] = tot;
}

__global__ __launch_bounds__(256) void k_pack(GridDev g, const double* __restrict__ v, double* __restrict__ dense, int to_dense) {
    const long long total = (long long)g.nkl * g.NY * g.NX;
    for (long long d = (long long)blockIdx.x * 256 + threadIdx.x; d < total; d += (long long)gridDim.x * 256) {
        const int i = (int)(d % g.NX);
        const long long row = d / g.NX;
        const long long idx = row * g.PX + i;       // row = lk*NY + j and PP = PX*NY
        if (to_dense) dense[d] = v[idx];
        else const_cast<double*>(v)[idx] = dense[d];
    }
}

inline int flat_grid(const pdegpu_ctx* ctx, long long n) {
    long long nb = (n + 255) / 256;
    const long long cap = (long long)ctx->grid_mult * ctx->num_cu;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

}  // namespace

int vec_alloc(pdegpu_ctx* ctx, const GridDev& g, DevVec* v) {
    v->total = vec_total(g);
    v->base = nullptr;
    PG_HIP(ctx, hipMalloc(&v->base, v->total * sizeof(double)));
    PG_HIP(ctx, hipMemsetAsync(v->base, 0, v->total * sizeof(double), ctx->stream));
    v->v = v->base + vec_front(g);
    return PDEGPU_OK;
}

void vec_free(DevVec* v) {
    if (v->base) (void)hipFree(v->base);
    v->base = v->v = nullptr;
    v->total = 0;
}

int vec_zero(pdegpu_ctx* ctx, const GridDev& g, DevVec* v) {
    (void)g;
    PG_HIP(ctx, hipMemsetAsync(v->base, 0, v->total * sizeof(double), ctx->stream));
    return PDEGPU_OK;
}

int vec_set_bc(pdegpu_ctx* ctx, const GridDev& g, double* x, double fill, const double* nodal_dense, int only_bc) {
    const long long total = (long long)g.nkl * g.PP;
    hipLaunchKernelGGL(k_set_bc, dim3(flat_grid(ctx, total)), dim3(256), 0, ctx->stream, g, x, fill, nodal_dense, only_bc);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

int vec_k2(pdegpu_ctx* ctx, const GridDev& g, double* r, const double* q, double* scal, double* partials, LaunchInfo* li,
           const DotIn& di) {
    const long long n2 = (long long)g.nkl * g.PP / 2;
    long long nbl = (n2 + 255) / 256;
    if (nbl > (long long)ctx->k2_grid_mult * ctx->num_cu) nbl = (long long)ctx->k2_grid_mult * ctx->num_cu;
    const int nb = nbl < 1 ? 1 : (int)nbl;
    if (li) { li->grid = nb; li->block = 256; }
    const bool nt = ctx->fast_nt < 0 ? n2 * 16 > (64LL << 20) : (ctx->fast_nt & 3) == 3;      // the marching kernels' policy
    if (nt) hipLaunchKernelGGL(k_k2<true>, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<double2*>(r),
                               reinterpret_cast<const double2*>(q), n2, scal, partials, di);
    else hipLaunchKernelGGL(k_k2<false>, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<double2*>(r),
                            reinterpret_cast<const double2*>(q), n2, scal, partials, di);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

int vec_norm2(pdegpu_ctx* ctx, const GridDev& g, const double* a, double* partials, LaunchInfo* li) {
    const long long n2 = (long long)g.nkl * g.PP / 2;
    const int nb = flat_grid(ctx, n2);
    if (li) { li->grid = nb; li->block = 256; }
    hipLaunchKernelGGL(k_norm2, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<const double2*>(a), n2, partials);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

int vec_axpy_flush(pdegpu_ctx* ctx, const GridDev& g, double* x, const double* p, const double* scal, int sr) {
    const long long n2 = (long long)g.nkl * g.PP / 2;
    hipLaunchKernelGGL(k_flush, dim3(flat_grid(ctx, n2)), dim3(256), 0, ctx->stream, reinterpret_cast<double2*>(x),
                       reinterpret_cast<const double2*>(p), n2, scal, sr);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

int vec_extrapolate(pdegpu_ctx* ctx, const GridDev& g, double* x, double* xprev, int first) {
    const long long n2 = (long long)g.nkl * g.PP / 2;
    hipLaunchKernelGGL(k_extrapolate, dim3(flat_grid(ctx, n2)), dim3(256), 0, ctx->stream, reinterpret_cast<double2*>(x),
                       reinterpret_cast<double2*>(xprev), n2, first);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

int vec_reduce(pdegpu_ctx* ctx, const double* partials, int n, double* scal, int which, int copy_to_global) {
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, ctx->stream, partials, n, scal, which, copy_to_global);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

int vec_reduce_sr(pdegpu_ctx* ctx, const double* pq_part, const double* qq_part, int n1, const double* rr_part, int n3, double* scal) {
    hipLaunchKernelGGL(k_reduce_sr, dim3(1), dim3(256), 0, ctx->stream, pq_part, qq_part, n1, rr_part, n3, scal);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}
int vec_reduce_slot(pdegpu_ctx* ctx, const double* partials, int n, double* scal, int slot) {
    hipLaunchKernelGGL(k_reduce_slot, dim3(1), dim3(256), 0, ctx->stream, partials, n, scal, slot);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

int vec_copy(pdegpu_ctx* ctx, const GridDev& g, double* dst, const double* src) {
    PG_HIP(ctx, hipMemcpyAsync(dst, src, (size_t)g.nkl * g.PP * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return PDEGPU_OK;
}

int vec_pack(pdegpu_ctx* ctx, const GridDev& g, const double* v, double* dense) {
    const long long total = (long long)g.nkl * g.NY * g.NX;
    hipLaunchKernelGGL(k_pack, dim3(flat_grid(ctx, total)), dim3(256), 0, ctx->stream, g, v, dense, 1);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

int vec_unpack(pdegpu_ctx* ctx, const GridDev& g, const double* dense, double* v) {
    const long long total = (long long)g.nkl * g.NY * g.NX;
    hipLaunchKernelGGL(k_pack, dim3(flat_grid(ctx, total)), dim3(256), 0, ctx->stream, g, v, const_cast<double*>(dense), 0);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

int vec_membench(pdegpu_ctx* ctx, const GridDev& g, int which, double* r, const double* q, double* b, double* sink) {
    const long long n2 = (long long)g.nkl * g.PP / 2;
    const long long nb = (n2 + 255) / 256;
    if (nb > 0x7fffffffLL) return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "membench: slab too large for a one-shot grid");
    hipLaunchKernelGGL(k_membench, dim3((unsigned)nb), dim3(256), 0, ctx->stream, which, reinterpret_cast<double2*>(r),
                       reinterpret_cast<const double2*>(q), reinterpret_cast<double2*>(b), n2, 0.0, sink);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}
