// weighted.hip — per-node assembled 15-point operators for the reference's curvilinear heat tools
// (_solve_heat_{1d,2d}_{cylindrical,spherical}_raw, _solve_heat_3d_spherical_raw, fenics_mcp_server.py:769-1464).
//
// Those forms multiply mass and stiffness by one coordinate weight w (r, r^2 or r^2 sin(theta)), entered as an
// Expression of degree 1 / 2 = its P1 / P2 interpolant per simplex, integrated exactly.  The stencil is no
// longer translation invariant, so one kernel integrates, for every node, the weighted element matrices of its
// incident simplices (node-centric gather: deterministic, no atomics) into 15 coupling weights stored
// offset-major in HBM (coalesced), plus the Jacobi diagonal.  The class-table kernels then read their weights
// from that array (k_general<..., ELL>), and CG becomes Jacobi-PCG because r^2 sin(theta) spans orders of
// magnitude across the grid.
#include "common.h"
#include "dot_finish.h"
#include "tables_math.h"

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

struct WeightGeom { double origin[3], h[3], lengths[3], core_radius, core_ratio; };

__global__ __launch_bounds__(256) void k_weighted_assemble(GridDev g, WeightGeom wg, int kind, double cM, double cK,
                                                            double* __restrict__ ellA, double* __restrict__ ellM,
                                                            double* __restrict__ dinv, long long stride) {
    const long long total = (long long)g.nkl * g.PP;
    const int nn[3] = {g.NX, g.NY, g.NZ};
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int i = (int)(idx % g.PX);
        const long long row = idx / g.PX;
        const int j = (int)(row % g.NY);
        const int k = g.k0 + (int)(row / g.NY);
        double rowK[PG_NOFF15], rowM[PG_NOFF15], ld = 0.0;
        if (i < g.NX) {
            weighted_row(g.dim, nn, wg.origin, wg.h, wg.lengths, kind, wg.core_radius, wg.core_ratio, i, j, k, rowK, rowM, &ld);
        } else {
            for (int q = 0; q < PG_NOFF15; ++q) rowK[q] = rowM[q] = 0.0;
        }
        for (int q = 0; q < PG_NOFF15; ++q) {
            ellA[q * stride + idx] = cM * rowM[q] + cK * rowK[q];
            ellM[q * stride + idx] = rowM[q];
        }
        const double d = cM * rowM[0] + cK * rowK[0];
        dinv[idx] = d > 0.0 ? 1.0 / d : 0.0;
    }
}

// Compact forms (StencilArgs.ell_mode; 3-D, a slab that owns every plane): only representative nodes are integrated.  mode 1: entry e = pc * PP + j * PX + i
// for the plane classes pc = 0 | 1 | 2 = slowest index 0 | interior (plane 1 stands for all) | NZ-1; mode 2: e = 3 * (j + NY * k) + ic for the column classes
// ic = x index 0 | interior (column 1) | NX-1.  A class without a node (the interior one of a two-plane / two-column grid) keeps the zeros of the memset.
__global__ __launch_bounds__(256) void k_weighted_assemble_compact(GridDev g, WeightGeom wg, int kind, double cM, double cK, double* __restrict__ ellA,
                                                                    double* __restrict__ ellM, long long stride, int ell_mode) {
    const int nn[3] = {g.NX, g.NY, g.NZ};
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < stride; e += (long long)gridDim.x * 256) {
        int i, j, k, cls;
        if (ell_mode == 1) {
            cls = (int)(e / g.PP);
            const int rem = (int)(e - (long long)cls * g.PP);
            j = rem / g.PX;
            i = rem - j * g.PX;
            k = cls == 0 ? 0 : (cls == 2 ? g.NZ - 1 : 1);
            if (i >= g.NX || j >= g.NY || (cls == 1 && g.NZ < 3)) continue;
        } else {
            const long long row = e / 3;
            cls = (int)(e - 3 * row);
            k = (int)(row / g.NY);
            j = (int)(row - (long long)k * g.NY);
            i = cls == 0 ? 0 : (cls == 2 ? g.NX - 1 : 1);
            if (cls == 1 && g.NX < 3) continue;
        }
        double rowK[PG_NOFF15], rowM[PG_NOFF15], ld = 0.0;
        weighted_row(g.dim, nn, wg.origin, wg.h, wg.lengths, kind, wg.core_radius, wg.core_ratio, i, j, k, rowK, rowM, &ld);
        for (int q = 0; q < PG_NOFF15; ++q) {
            ellA[q * stride + e] = cM * rowM[q] + cK * rowK[q];
            ellM[q * stride + e] = rowM[q];
        }
    }
}
// Jacobi diagonal of every node from the compact rows (q = 0 is the centre entry); row pads get 0 as in the one-row-per-node form
__global__ __launch_bounds__(256) void k_weighted_dinv_compact(GridDev g, const double* __restrict__ ellA, double* __restrict__ dinv, int ell_mode) {
    const long long total = (long long)g.nkl * g.PP;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long row = idx / g.PX;
        const int i = (int)(idx - row * g.PX);
        const int k = (int)(row / g.NY);
        long long e;
        if (ell_mode == 1) e = (long long)(k == 0 ? 0 : (k == g.NZ - 1 ? 2 : 1)) * g.PP + (idx - (long long)k * g.PP);
        else e = 3 * row + (i == 0 ? 0 : (i >= g.NX - 1 ? 2 : 1));
        const double d = i < g.NX ? ellA[e] : 0.0;
        dinv[idx] = d > 0.0 ? 1.0 / d : 0.0;
    }
}

// r -= alpha q;  z = D^-1 r;  partial r.z    (use_alpha == 0: z and the sum only)
__global__ __launch_bounds__(256) void k_k2_pc(double* __restrict__ r, const double* __restrict__ q, double* __restrict__ z,
                                                const double* __restrict__ dinv, long long n, double* scal,
                                                double* __restrict__ partials, int use_alpha, DotIn di) {
    __shared__ double sRed[4];
    double alpha = 0.0;
    if (use_alpha) alpha = cg_alpha_in(di, scal, sRed);      // (lazy loop: K1's p.q partials are summed here, the r.z slot is di.s_rr)
    double acc = 0.0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        double rv = r[e];
        if (use_alpha) {
            rv -= alpha * q[e];
            r[e] = rv;
        }
        const double zv = dinv[e] * rv;
        z[e] = zv;
        acc += rv * zv;
    }
    const double tot = block_sum_256(acc, sRed);
    __syncthreads();
    dot_finish(tot, partials);
}

inline int flat_grid(const pdegpu_ctx* ctx, long long n) {
    long long nb = (n + 255) / 256;
    const long long cap = (long long)ctx->grid_mult * ctx->num_cu;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

}  // namespace

int weighted_assemble(pdegpu_ctx* ctx, const GridDev& g, const pdegpu_weight* w, const double* h, const double* lengths, double cM,
                      double cK, double* ellA, double* ellM, double* dinv, long long stride, int ell_mode) {
    WeightGeom wg{};
    for (int r = 0; r < 3; ++r) { wg.origin[r] = w->origin[r]; wg.h[r] = h[r]; wg.lengths[r] = lengths[r]; }
    wg.core_radius = w->core_radius;
    wg.core_ratio = w->core_ratio;
    const int kind = w->kind;
    const long long total = (long long)g.nkl * g.PP;
    if (ell_mode) {
        hipLaunchKernelGGL(k_weighted_assemble_compact, dim3(flat_grid(ctx, stride)), dim3(256), 0, ctx->stream, g, wg, kind, cM, cK, ellA, ellM, stride,
                           ell_mode);
        PG_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_weighted_dinv_compact, dim3(flat_grid(ctx, total)), dim3(256), 0, ctx->stream, g, ellA, dinv, ell_mode);
        PG_HIP(ctx, hipGetLastError());
        return PDEGPU_OK;
    }
    hipLaunchKernelGGL(k_weighted_assemble, dim3(flat_grid(ctx, total)), dim3(256), 0, ctx->stream, g, wg, kind, cM, cK, ellA, ellM,
                       dinv, stride);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

int vec_k2_pc(pdegpu_ctx* ctx, const GridDev& g, double* r, const double* q, double* z, const double* dinv, double* scal,
              double* partials, int use_alpha, LaunchInfo* li, const DotIn& di) {
    const long long n = (long long)g.nkl * g.PP;
    const int nb = flat_grid(ctx, n);
    if (li) { li->grid = nb; li->block = 256; }
    hipLaunchKernelGGL(k_k2_pc, dim3(nb), dim3(256), 0, ctx->stream, r, q, z, dinv, n, scal, partials, use_alpha, di);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

// host twin of k_weighted_assemble (CPU test-suite): rowsK / rowsM [N][15] in node order (x fastest), load [N]
extern "C" int pdegpu_weighted_rows_host(const pdegpu_grid* grid, const pdegpu_weight* weight, double* rowsK, double* rowsM,
                                         double* load) {
    if (!grid || !weight || !rowsK || !rowsM || !load || grid->dim < 1 || grid->dim > 3) return PDEGPU_EINVAL;
    const int kind = weight->kind;
    const double* origin = weight->origin;
    if (kind < PDEGPU_WEIGHT_ONE || kind > PDEGPU_WEIGHT_RCYL) return PDEGPU_EINVAL;
    if (kind == PDEGPU_WEIGHT_R2SIN && grid->dim < 2) return PDEGPU_EINVAL;
    if ((kind == PDEGPU_WEIGHT_RCYL || kind == PDEGPU_WEIGHT_ONE) && grid->dim < 3) return PDEGPU_EINVAL;
    double h[3] = {1, 1, 1}, lengths[3] = {0, 0, 0};
    int nn[3] = {1, 1, 1};
    for (int r = 0; r < grid->dim; ++r) {
        if (grid->n[r] < 1 || !(grid->L[r] > 0)) return PDEGPU_EINVAL;
        h[r] = grid->L[r] / grid->n[r];
        lengths[r] = grid->L[r];
        nn[r] = grid->n[r] + 1;
    }
    long long idx = 0;
    for (int k = 0; k < nn[2]; ++k)
        for (int j = 0; j < nn[1]; ++j)
            for (int i = 0; i < nn[0]; ++i, ++idx)
                weighted_row(grid->dim, nn, origin, h, lengths, kind, weight->core_radius, weight->core_ratio, i, j, k, rowsK + idx * 15, rowsM + idx * 15,
                             load + idx);
    return PDEGPU_OK;
}
