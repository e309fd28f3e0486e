// tables.hip — element-local integration of the macro-element stiffness / mass / load into LDS and the
// fold into per-boundary-class node stencils.
//
// Replaces, for the uniform box meshes of the reference, what UFL -> FFC -> DOLFIN assemble() produce
// from the forms `u*v*dx`, `inner(grad(u),grad(v))*dx`, `f*v*dx` (fenics_mcp_server.py:261-262,
// 304-305, 393-394, 433-434, 657-658, 702-703): every grid cell of a uniform box has the SAME 2^d x 2^d
// macro-element matrix (the cell split into 1 interval / 2 triangles / 6 Kuhn tets for the reference's
// P1 space, or one multilinear Q1 element), so a global matrix is never formed.  One workgroup integrates
// the macro element into LDS, then folds it into the coupling weight of every (boundary class, neighbour
// offset) pair: class = (cx, cy, cz) in {low face, interior, high face}^3 selects which of the up to 2^d
// incident cells exist, which is what truncates the stencil on natural (non-Dirichlet) faces.
#include "common.h"
#include "tables_math.h"

namespace {

__global__ __launch_bounds__(256) void k_build_tables(int dim, int element_mode, double hx, double hy, double hz,
                                                        double* __restrict__ tabK, double* __restrict__ tabM,
                                                        double* __restrict__ load) {
    __shared__ double sK[8][8], sM[8][8];       // the macro-element matrices live in LDS
    __shared__ double sTabM[27 * 27];
    const double h[3] = {hx, hy, hz};
    const int t = threadIdx.x;
    const int nv = 1 << dim;
    if (t < 64) {
        const int a = t >> 3, b = t & 7;
        double k = 0, m = 0;
        if (a < nv && b < nv) macro_entry(dim, element_mode, h, a, b, &k, &m);
        sK[a][b] = k;
        sM[a][b] = m;
    }
    __syncthreads();
    for (int e = t; e < 27 * 27; e += blockDim.x) {
        double wk, wm;
        fold_entry(dim, sK, sM, e / 27, e % 27, &wk, &wm);
        tabK[e] = wk;
        tabM[e] = wm;
        sTabM[e] = wm;
    }
    __syncthreads();
    if (t < 27) {
        double s = 0;
        for (int o = 0; o < 27; ++o) s += sTabM[t * 27 + o];   // m_i = int phi_i = row sum of the mass stencil
        load[t] = s;
    }
}

__global__ void k_combine(const double* __restrict__ tabK, const double* __restrict__ tabM, double cM, double cK,
                          double* __restrict__ out) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 27 * 27) out[e] = cM * tabM[e] + cK * tabK[e];
}

}  // namespace

int tables_build(pdegpu_ctx* ctx, const pdegpu_grid* grid, HeatTables* t) {
    const size_t nb = sizeof(double) * 27 * 27;
    PG_HIP(ctx, hipMalloc(&t->d_tabK, nb));
    PG_HIP(ctx, hipMalloc(&t->d_tabM, nb));
    PG_HIP(ctx, hipMalloc(&t->d_tabA, nb));
    PG_HIP(ctx, hipMalloc(&t->d_load, sizeof(double) * 27));
    double h[3] = {1, 1, 1};
    for (int r = 0; r < grid->dim; ++r) h[r] = grid->L[r] / grid->n[r];
    hipLaunchKernelGGL(k_build_tables, dim3(1), dim3(256), 0, ctx->stream, grid->dim, grid->element_mode, h[0], h[1], h[2],
                       t->d_tabK, t->d_tabM, t->d_load);
    PG_HIP(ctx, hipGetLastError());
    PG_HIP(ctx, hipMemcpyAsync(t->h_tabK, t->d_tabK, nb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(t->h_tabM, t->d_tabM, nb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(t->h_load, t->d_load, sizeof(double) * 27, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PDEGPU_OK;
}

int tables_combine(pdegpu_ctx* ctx, HeatTables* t, double cM, double cK, double* d_out, double* h_out) {
    hipLaunchKernelGGL(k_combine, dim3(3), dim3(256), 0, ctx->stream, t->d_tabK, t->d_tabM, cM, cK, d_out);
    PG_HIP(ctx, hipGetLastError());
    if (h_out)
        for (int e = 0; e < 27 * 27; ++e) h_out[e] = cM * t->h_tabM[e] + cK * t->h_tabK[e];
    return PDEGPU_OK;
}

void tables_free(HeatTables* t) {
    if (t->d_tabK) (void)hipFree(t->d_tabK);
    if (t->d_tabM) (void)hipFree(t->d_tabM);
    if (t->d_tabA) (void)hipFree(t->d_tabA);
    if (t->d_load) (void)hipFree(t->d_load);
    t->d_tabK = t->d_tabM = t->d_tabA = t->d_load = nullptr;
}

// host twin of k_build_tables: same math, no GPU (CPU test-suite checks it against the oracle's assembly)
extern "C" int pdegpu_heat_tables_host(const pdegpu_grid* grid, double* tabK, double* tabM, double* load) {
    if (!grid || grid->dim < 1 || grid->dim > 3) return PDEGPU_EINVAL;
    double h[3] = {1, 1, 1};
    for (int r = 0; r < grid->dim; ++r) {
        if (grid->n[r] < 1 || !(grid->L[r] > 0)) return PDEGPU_EINVAL;
        h[r] = grid->L[r] / grid->n[r];
    }
    double sK[8][8] = {}, sM[8][8] = {};
    const int nv = 1 << grid->dim;
    for (int a = 0; a < nv; ++a)
        for (int b = 0; b < nv; ++b) macro_entry(grid->dim, grid->element_mode, h, a, b, &sK[a][b], &sM[a][b]);
    for (int e = 0; e < 27 * 27; ++e) fold_entry(grid->dim, sK, sM, e / 27, e % 27, &tabK[e], &tabM[e]);
    for (int c = 0; c < 27; ++c) {
        double s = 0;
        for (int o = 0; o < 27; ++o) s += tabM[c * 27 + o];
        load[c] = s;
    }
    return PDEGPU_OK;
}
