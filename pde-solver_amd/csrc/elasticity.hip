// elasticity.hip — static linear elasticity on the structured simplex split (gfx950).
//
// Replaces VectorFunctionSpace assembly + clamp at x = 0 + `solve(a == L, u_sol, bc)` + per-cell von Mises +
// `project(eq_expr, Vs)` of _solve_elasticity_{1,2,3}d_static (fenics_mcp_server.py:1516-1564, 1648-1718,
// 1803-1866).  The operator is applied matrix-free as a d x d-block node stencil over the 15 neighbours the
// Kuhn split couples; the blocks are folded on the device from the 2^d-vertex macro-element matrix per
// boundary class (all faces except x = 0 are traction-free, so boundary nodes use truncated class stencils).
// Solver: Jacobi-preconditioned CG per iteration = flat p-update (p = z + beta p, deferred x += alpha p) -> block apply
// q = A p with the p.q partials (3-D: z-marching k_el_fast + boundary workgroups; 2-D: thread per node) -> K2 (r -= alpha q,
// z = D^-1 r, r.z and r.r); the scalars are summed in the consuming kernel's prologue (no reduce launches in the one-GPU loop).
// Displacement components are stored as separate pitched arrays so every neighbour read is coalesced.
#include "common.h"
#include "tables_math.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

// device scalars of the elasticity PCG; RZ and RR are adjacent so the multi-GPU host all-reduces them as one 16-byte message
enum { ESC_PQ = 0, ESC_RZ = 1, ESC_RR = 2, ESC_RZ_PREV = 3, ESC_COUNT = 8 };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double block_sum_256(double v, double* sm) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void k_el_tables(int dim, double hx, double hy, double hz, double lam, double mu,
                                                    double* __restrict__ tab /*[27][15][9]*/, double* __restrict__ dinv /*[27][3]*/,
                                                    unsigned* __restrict__ mask /*[27] nonzero blocks*/) {
    const double h[3] = {hx, hy, hz};
    for (int e = threadIdx.x; e < 27 * PG_NOFF15; e += blockDim.x) {
        double blk[9];
        fold_elasticity_block(dim, h, lam, mu, e / PG_NOFF15, e % PG_NOFF15, blk);
        for (int t = 0; t < 9; ++t) tab[e * 9 + t] = blk[t];
        if (e % PG_NOFF15 == 0)
            for (int a = 0; a < 3; ++a) dinv[(e / PG_NOFF15) * 3 + a] = blk[a * 3 + a] > 0.0 ? 1.0 / blk[a * 3 + a] : 0.0;
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x < 27) {
        unsigned m = 0;
        for (int q = 0; q < PG_NOFF15; ++q) {
            bool nz = false;
            for (int t = 0; t < 9; ++t) nz |= tab[(threadIdx.x * PG_NOFF15 + q) * 9 + t] != 0.0;
            if (nz) m |= 1u << q;
        }
        mask[threadIdx.x] = m;
    }
}

struct ElArgs {
    GridDev g;
    long long stride;          // component stride (doubles)
    const double* tab;         // [27][15][9]
    const double* dinv;        // [27][3]
    const double* in;          // PLAIN/RESIDUAL input; K1: z
    const double* rhs;         // RESIDUAL
    double* out;               // PLAIN/RESIDUAL output; K1: q
    const double* p_old;
    double* p_new;
    double* x;
    const double* scal;
    double* partials;
    double* partials2;
    int mask_dirichlet;
    // lazy dot products (same scheme as DotIn of the heat path): non-null = per-workgroup partials of the previous launch,
    // summed in this kernel's prologue.  p-update: r.z (din) and r.r (din2) of K2;  K2: p.q (din) of the block apply.
    const double* din = nullptr;
    const double* din2 = nullptr;
    int din_n = 0;
    int s_rz = 1, s_rz_old = 3;     // ESC_RZ / ESC_RZ_PREV: slots of r.z of this and of the previous iteration
};

// fixed-order sum of n partials by all 256 threads (identical in every workgroup; the order k_el_reduce uses)
__device__ __forceinline__ double el_sum_in(const double* __restrict__ part, int n, double* sRed) {
    double acc = 0.0;
    for (int e = threadIdx.x; e < n; e += 256) acc += part[e];
    return block_sum_256(acc, sRed);
}

// block weights of the interior boundary class, passed BY VALUE: kernel arguments live in the kernarg segment and are
// fetched with scalar loads, so interior wavefronts read their 15 x 9 weights from SGPRs (a pointer inside ElArgs
// would be a vector load — the compiler cannot prove it unclobbered)
struct ElInterior { double w[PG_NOFF15 * 9]; };

template <int D, int MODE>
__global__ __launch_bounds__(256) void k_el_apply(ElArgs a, ElInterior wi) {
    __shared__ double sTab[27 * PG_NOFF15 * 9];
    __shared__ unsigned sMask[27];
    __shared__ double sRed[4];
    for (int e = threadIdx.x; e < 27 * PG_NOFF15 * 9; e += 256) sTab[e] = a.tab[e];
    __syncthreads();
    if (threadIdx.x < 27) {
        unsigned m = 0;
        for (int q = 0; q < PG_NOFF15; ++q) {
            bool nz = false;
            for (int t = 0; t < 9; ++t) nz |= sTab[(threadIdx.x * PG_NOFF15 + q) * 9 + t] != 0.0;
            if (nz) m |= 1u << q;
        }
        sMask[threadIdx.x] = m;
    }
    __syncthreads();

    const GridDev& g = a.g;
    const long long total = (long long)g.nkl * g.PP;
    double alpha_prev = 0.0, beta = 0.0;
    if (MODE == APPLY_CG_K1) {
        const double pq = a.scal[ESC_PQ], rz = a.scal[ESC_RZ], rz_prev = a.scal[ESC_RZ_PREV];
        alpha_prev = (pq > 0.0 && rz_prev > 0.0) ? rz_prev / pq : 0.0;
        beta = (rz_prev > 0.0) ? rz / rz_prev : 0.0;
    }
    double acc = 0.0;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int i = (int)(idx % g.PX);
        const long long row = idx / g.PX;
        const int j = (int)(row % g.NY);
        const int lk = (int)(row / g.NY);
        const int kg = g.k0 + lk;
        const bool real = i < g.NX;
        double dval;
        const bool dir = real && a.mask_dirichlet && pg_dirichlet(g, i, j, kg, &dval);
        double pn[D], s[D];
#pragma unroll
        for (int c = 0; c < D; ++c) { pn[c] = 0.0; s[c] = 0.0; }
        if (MODE == APPLY_CG_K1) {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                const double po = a.p_old[c * a.stride + idx];
                pn[c] = a.in[c * a.stride + idx] + beta * po;
                a.p_new[c * a.stride + idx] = pn[c];
                if (alpha_prev != 0.0) a.x[c * a.stride + idx] += alpha_prev * po;
            }
        }
        const bool active = real && !dir;
        const int cls = pg_axis_class(i < g.NX ? i : 1, g.NX) + 3 * (g.dim >= 2 ? pg_axis_class(j, g.NY) : 1) +
                        9 * (g.dim >= 3 ? pg_axis_class(kg, g.NZ) : 1);
        // wave-uniform boundary class (every interior wave): the 15 x 9 weights come through scalar loads into SGPRs
        // instead of 135 LDS reads per lane
        const unsigned long long act = __ballot(active);
        int cls0 = 13;
        bool uniform = false;
        if (act) {
            cls0 = __builtin_amdgcn_readlane(cls, (int)__builtin_ctzll(act));
            uniform = __ballot(active && cls != cls0) == 0ull;
        }
        auto gather = [&](const double* __restrict__ wtab, unsigned m) {
#pragma unroll
            for (int q = 0; q < PG_NOFF15; ++q) {
                if (!((m >> q) & 1u)) continue;
                const int o = pg_off15(q);
                const int dx = o % 3 - 1, dy = (o / 3) % 3 - 1, dz = o / 9 - 1;
                const long long nb = idx + (long long)dz * g.PP + (long long)dy * g.PX + dx;
                const double* blk = wtab + q * 9;
                double v[D];
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    v[c] = a.in[c * a.stride + nb];
                    if (MODE == APPLY_CG_K1) v[c] += beta * a.p_old[c * a.stride + nb];
                }
#pragma unroll
                for (int r = 0; r < D; ++r)
#pragma unroll
                    for (int c = 0; c < D; ++c) s[r] += blk[r * 3 + c] * v[c];
            }
        };
        if (uniform && cls0 == 13) {
            if (active) gather(wi.w, sMask[13]);                                 // interior class: weights from SGPRs
        } else if (active) {
            gather(&sTab[cls * PG_NOFF15 * 9], sMask[cls]);
        }
#pragma unroll
        for (int c = 0; c < D; ++c) {
            if (MODE == APPLY_PLAIN) {
                a.out[c * a.stride + idx] = s[c];
                if (real) acc += a.in[c * a.stride + idx] * s[c];
            } else if (MODE == APPLY_RESIDUAL) {
                double rv = 0.0;
                if (real && !dir) rv = a.rhs[c * a.stride + idx] - s[c];
                a.out[c * a.stride + idx] = rv;
                acc += rv * rv;
            } else {
                a.out[c * a.stride + idx] = s[c];
                acc += pn[c] * s[c];
            }
        }
    }
    const double tot = block_sum_256(acc, sRed);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
}


// ---------------------------------------------------------------------------------------------------------------
// 3-D fast apply: z-marching block stencil for the interior class + the boundary nodes in the same launch
// ---------------------------------------------------------------------------------------------------------------
// On interior nodes of the uniform Kuhn split the 15 blocks are symmetric, B(-o) = B(o), and the four diagonal
// offsets have a zero diagonal (el_fast_weights checks this on the host table before the path is enabled), so
//   y_i = C u_i + sum_{7 offsets o} B_o (u_{i+o} + u_{i-o})
// needs 36 distinct weights (72 SGPRs, fetched from the kernarg segment) and 60 FMAs instead of 135.  One wavefront
// owns a 62-node x RY-row column (lanes 0 and 63 only carry the x-halo, so no edge exchange exists), marches along z
// with three planes of RY+2 rows x 3 components in registers, and takes its x-neighbours from DPP wave shifts: 9 (RY=1)
// or 6 (RY=2) coalesced loads per node instead of 45.  Boundary nodes (traction-free faces, the clamped face, 7 % of
// config 5) are enumerated analytically and handled by extra workgroups of the same launch with the class table in LDS.
struct ElW36 {
    double c[6];        // centre block: xx yy zz xy xz yz
    double ax[3][6];    // offsets (1,0,0) (0,1,0) (0,0,1)
    double dg[4][3];    // offsets (1,1,0) (1,0,1) (0,1,1) (1,1,1): xy xz yz (zero diagonal)
    // x-face nodes (interior in y and z) inside the marching kernel.  On the host tables the class of the face x = NX-1 is: offsets
    // with dx = +1 absent, dx = -1 as in the interior, and for the in-face offsets o (dx = 0)  B_face(o) = B(o) / 2 + A(o)  with
    // A(-o) = -A(o) antisymmetric and non-zero in its first row / column only, A(o) = [[0, a, b], [-a, 0, 0], [-b, 0, 0]] (the Kuhn
    // split is not mirror symmetric in x); the face x = 0 has -A.  xf[t] = {a, b} of o = (0,1,0), (0,0,1), (0,1,1); the centre
    // block is exactly halved.  el_fast_weights checks all of this before xface is set.
    double xf[3][2];
    int xface;          // 1: the marching part computes the x-face rows, the boundary pass only the y / z faces
};
struct ElFastGeom {
    int ntx, nty, nzc, zlen;   // 62-node x-tiles, RY-row y-tiles, z-chunks of zlen planes
    int nyg;                   // groups of 4 y-tiles (one workgroup each) of the y-hand-off variant
    int lkb, lke;              // local interior plane range [lkb, lke)
    int nb_fast;               // workgroups of the marching part (4 waves each); they FOLLOW the nb_bnd boundary workgroups
    int nb_bnd;                // boundary workgroups (launched first so that their short tail overlaps the marching part)
    int full_lo, full_hi;      // 1: this slab owns the global plane 0 / NZ-1 (all of its nodes are boundary nodes)
    int ring;                  // boundary nodes of an interior plane: 2 NX + 2 (NY - 2)
    long long nbnd;            // boundary nodes of the slab
    const unsigned* mask;      // [27] nonzero-block masks of the class table
};

__device__ __forceinline__ double el_dpp_lower(double v) {     // lane l <- lane l-1 (v_mov_b32_dpp wave_shr:1)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double el_dpp_upper(double v) {     // lane l <- lane l+1 (v_mov_b32_dpp wave_shl:1)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void el_sym6(const double* b, const double* v, double* y) {
    y[0] += b[0] * v[0] + b[3] * v[1] + b[4] * v[2];
    y[1] += b[3] * v[0] + b[1] * v[1] + b[5] * v[2];
    y[2] += b[4] * v[0] + b[5] * v[1] + b[2] * v[2];
}
__device__ __forceinline__ void el_off3(const double* b, const double* v, double* y) {
    y[0] += b[0] * v[1] + b[1] * v[2];
    y[1] += b[0] * v[0] + b[2] * v[2];
    y[2] += b[1] * v[0] + b[2] * v[1];
}

template <int RY, bool YX>
__global__ __launch_bounds__(256) void k_el_fast(ElArgs a, ElW36 w, ElFastGeom fg) {
    __shared__ double sTab[27 * PG_NOFF15 * 9];
    __shared__ unsigned sMask[27];
    __shared__ double sRed[8];
    constexpr int RH = RY + 2;
    const GridDev& g = a.g;
    double acc = 0.0;
    if ((int)blockIdx.x < fg.nb_bnd) {
        // ---- boundary nodes: one thread per node, truncated class stencils from LDS -----------------------------
        for (int e = threadIdx.x; e < 27 * PG_NOFF15 * 9; e += 256) sTab[e] = a.tab[e];
        if (threadIdx.x < 27) sMask[threadIdx.x] = fg.mask[threadIdx.x];
        __syncthreads();
        for (long long e0 = (long long)blockIdx.x * 256 + threadIdx.x; e0 < fg.nbnd; e0 += (long long)fg.nb_bnd * 256) {
            unsigned e = (unsigned)e0;                        // 32-bit divisions (nbnd < 2^31 is checked on the host)
            int i, j, lk;
            const unsigned plane_nodes = (unsigned)g.NX * (unsigned)g.NY;
            if (fg.full_lo && e < plane_nodes) {
                lk = -g.k0;                                   // global plane 0
                j = (int)(e / (unsigned)g.NX);
                i = (int)(e % (unsigned)g.NX);
            } else {
                e -= fg.full_lo ? plane_nodes : 0u;
                if (fg.full_hi && e < plane_nodes) {
                    lk = g.NZ - 1 - g.k0;
                    j = (int)(e / (unsigned)g.NX);
                    i = (int)(e % (unsigned)g.NX);
                } else {
                    e -= fg.full_hi ? plane_nodes : 0u;
                    lk = fg.lkb + (int)(e / (unsigned)fg.ring);
                    const int t = (int)(e % (unsigned)fg.ring);
                    if (t < g.NX) { j = 0; i = t; }
                    else if (t < 2 * g.NX) { j = g.NY - 1; i = t - g.NX; }
                    else { j = 1 + (t - 2 * g.NX) / 2; i = ((t - 2 * g.NX) & 1) ? g.NX - 1 : 0; }      // (ring = 2 NX with xface: never here)
                }
            }
            const int kg = g.k0 + lk;
            const long long idx = (long long)lk * g.PP + (long long)j * g.PX + i;
            double dval;
            const bool dir = a.mask_dirichlet && pg_dirichlet(g, i, j, kg, &dval);
            double pc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) pc[c] = a.in[c * a.stride + idx];
            // All 15 neighbours are loaded unconditionally (45 loads in flight instead of 15 dependent round trips): an
            // offset that leaves the grid still lands inside the allocation (ghost planes / guards / row pads, all zero
            // or finite neighbour data) and meets a class weight that is exactly 0.
            double s[3] = {0.0, 0.0, 0.0};
            double nv[PG_NOFF15][3];
#pragma unroll
            for (int q = 0; q < PG_NOFF15; ++q) {
                const int o = pg_off15(q);
                const int dx = o % 3 - 1, dy = (o / 3) % 3 - 1, dz = o / 9 - 1;
                const long long nb = idx + (long long)dz * g.PP + (long long)dy * g.PX + dx;
#pragma unroll
                for (int c = 0; c < 3; ++c) nv[q][c] = a.in[c * a.stride + nb];
            }
            if (!dir) {
                const int cls = pg_axis_class(i, g.NX) + 3 * pg_axis_class(j, g.NY) + 9 * pg_axis_class(kg, g.NZ);
                const double* wtab = &sTab[cls * PG_NOFF15 * 9];
#pragma unroll
                for (int q = 0; q < PG_NOFF15; ++q) {
                    const double* blk = wtab + q * 9;
#pragma unroll
                    for (int r = 0; r < 3; ++r) s[r] += blk[r * 3] * nv[q][0] + blk[r * 3 + 1] * nv[q][1] + blk[r * 3 + 2] * nv[q][2];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a.out[c * a.stride + idx] = s[c];
                acc += pc[c] * s[c];
            }
        }
    } else {
        // ---- interior nodes: z-marching ------------------------------------------------------------------------
        const int lane = threadIdx.x & 63;
        // XCD-aware order: workgroups b and b+8 share an XCD (round-robin dispatch), so XCD x is given the contiguous range
        // [x*G/8, (x+1)*G/8) of tiles: y-neighbouring columns, whose halo rows overlap, then meet in one L2 (speed only)
        int bm = (int)blockIdx.x - fg.nb_bnd;
        if ((fg.nb_fast & 7) == 0) bm = (bm & 7) * (fg.nb_fast >> 3) + (bm >> 3);
        const int wv = threadIdx.x >> 6;
        int xt, yt, zc;
        if (YX) {
            // the four waves of a workgroup are y-neighbours (same x-tile, same z-chunk): they hand their y-halo rows to
            // each other through LDS, so a workgroup loads 4 RY + 2 rows per plane instead of 4 (RY + 2)
            xt = bm % fg.ntx;
            yt = ((bm / fg.ntx) % fg.nyg) * 4 + wv;
            zc = bm / (fg.ntx * fg.nyg);
        } else {
            const int wid = bm * 4 + wv;
            xt = wid % fg.ntx;
            yt = (wid / fg.ntx) % fg.nty;
            zc = wid / (fg.ntx * fg.nty);
        }
        const bool ghost_wave = YX && yt >= fg.nty;           // beyond the last row tile: takes part in the barriers only
        if (ghost_wave) yt = fg.nty - 1;                      // (loads stay inside the grid; nothing is stored)
        const int i = xt * 62 + lane;                         // lanes 0 and 63: x-halo only (lane 0 of the first tile: the face x = 0)
        const int j0 = 1 + yt * RY;                           // first owned row; held rows j0-1 .. j0+RY
        const int lk0 = fg.lkb + zc * fg.zlen;
        const int lk1 = min(lk0 + fg.zlen, fg.lke);
        if (zc < fg.nzc && lk0 < lk1) {
            // Loads are unconditional (branch-free, all of a plane in flight at once): every address a wave can form lies
            // inside the allocation — rows up to NY (= first row of the next plane), columns up to NX+60 < PX+64, planes
            // -1 .. nkl, covered by the ghost planes and the guard of kGuard + one row on both sides — and a value from
            // outside the grid only ever reaches outputs that are not stored (an interior node has all 15 neighbours inside).
            // x-faces: lane 0 of the first tile is the node i = 0 (clamped: its row is 0 when Dirichlet rows are masked), the node
            // i = NX-1 (traction-free) sits on one of the lanes 1 .. 62 of the last tile
            const bool face_lo = w.xface && i == 0, face_hi = w.xface && i == g.NX - 1;
            const bool wave_has_face = w.xface && (xt == 0 || (xt + 1) * 62 >= g.NX - 1);      // wave-uniform
            const double fsc = (face_lo || face_hi) ? 0.5 : 1.0, fsg = face_hi ? 1.0 : (face_lo ? -1.0 : 0.0);
            const bool out_x = (lane >= 1 && lane <= 62 && i <= g.NX - (w.xface ? 1 : 2)) || face_lo;
            bool row_out[RH];
#pragma unroll
            for (int r = 0; r < RH; ++r) row_out[r] = out_x && !ghost_wave && r >= 1 && r <= RY && (j0 - 1 + r) <= g.NY - 2;
            // lanes beyond the padded row (the last 62-node tile overshoots: 558 columns for a 513-node row) read the row's last pad
            // entry instead of the next row's data: same cache line as real data, no traffic of their own (9 % of the loads at
            // NX = 513); what they hold never reaches a stored output
            const long long col = (long long)(j0 - 1) * g.PX + (i < g.PX ? i : g.PX - 1);
            double Pm[RH][3], P0[RH][3], Pp[RH][3];
            auto load_plane = [&](int lk, double (*P)[3]) {
                const long long base = (long long)lk * g.PP + col;
#pragma unroll
                for (int r = 0; r < RH; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) P[r][c] = a.in[c * a.stride + base + (long long)r * g.PX];
            };
            // one more plane in flight: plane lk+2 is requested before plane lk is computed
            double Pn[RH][3];
            // YX: a marching plane is loaded without the halo rows its y-neighbours in the workgroup own (the lowest wave
            // still loads row j0-1, the highest real wave row j0+RY); they arrive through LDS one step later, when the
            // plane has become Pp and its loads have landed.  The LDS block is the class table's (boundary workgroups only).
            const bool lo_global = !YX || wv == 0;
            const bool hi_global = !YX || wv == 3 || (yt + 1 >= fg.nty);
            auto load_own = [&](int lk, double (*P)[3]) {
                const long long base = (long long)lk * g.PP + col;
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    const bool need = (r >= 1 && r <= RY) || (r == 0 ? lo_global : hi_global);      // wave-uniform
#pragma unroll
                    for (int c = 0; c < 3; ++c) P[r][c] = need ? a.in[c * a.stride + base + (long long)r * g.PX] : 0.0;
                }
            };
            double* sY = sTab;                                // [2][4 waves][first | last own row][3][64]
            int ybuf = 0;
            auto y_exchange = [&](double (*P)[3]) {
                double* mine = sY + ((ybuf * 4 + wv) * 2) * 192;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    mine[c * 64 + lane] = P[1][c];
                    mine[192 + c * 64 + lane] = P[RY][c];
                }
                __syncthreads();
                if (!lo_global) {
                    const double* below = sY + ((ybuf * 4 + wv - 1) * 2 + 1) * 192;          // last own row of the wave below
#pragma unroll
                    for (int c = 0; c < 3; ++c) P[0][c] = below[c * 64 + lane];
                }
                if (!hi_global) {
                    const double* above = sY + ((ybuf * 4 + wv + 1) * 2) * 192;              // first own row of the wave above
#pragma unroll
                    for (int c = 0; c < 3; ++c) P[RH - 1][c] = above[c * 64 + lane];
                }
                ybuf ^= 1;
            };
            load_plane(lk0 - 1, Pm);
            load_plane(lk0, P0);
            load_plane(lk0 + 1, Pp);                          // the address is inside the allocation even when lk0 + 1 == lk1 == nkl
            for (int lk = lk0; lk < lk1; ++lk) {
                // lk + 2 can reach nkl + 1 (one plane past the upper ghost plane) only when it is not needed: clamp it
                if (YX) {
                    load_own(min(lk + 2, lk1), Pn);
                    if (lk > lk0) y_exchange(Pp);             // Pp was requested one step ago; every wave of the workgroup passes here
                } else {
                    load_plane(min(lk + 2, lk1), Pn);
                }
                const long long obase = (long long)lk * g.PP + col;
#pragma unroll
                for (int r = 1; r <= RY; ++r) {
                    // y = [unshifted terms] + up(U) + lo(L): the x-shifts act on two accumulated 3-vectors, not on 24 inputs
                    double y[3] = {0.0, 0.0, 0.0}, U[3] = {0.0, 0.0, 0.0}, Lw[3] = {0.0, 0.0, 0.0}, v[3];
                    el_sym6(w.c, P0[r], y);
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c] = P0[r + 1][c] + P0[r - 1][c];
                    el_sym6(w.ax[1], v, y);
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c] = Pp[r][c] + Pm[r][c];
                    el_sym6(w.ax[2], v, y);
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c] = Pp[r + 1][c] + Pm[r - 1][c];
                    el_off3(w.dg[2], v, y);
                    el_sym6(w.ax[0], P0[r], U);               // the (+-1,0,0) term is the same 3-vector for both shifts
                    Lw[0] = U[0]; Lw[1] = U[1]; Lw[2] = U[2];
                    el_off3(w.dg[0], P0[r + 1], U);           // source (i+1, j+1, k)
                    el_off3(w.dg[1], Pp[r], U);               //        (i+1, j, k+1)
                    el_off3(w.dg[3], Pp[r + 1], U);           //        (i+1, j+1, k+1)
                    el_off3(w.dg[0], P0[r - 1], Lw);          //        (i-1, j-1, k)
                    el_off3(w.dg[1], Pm[r], Lw);              //        (i-1, j, k-1)
                    el_off3(w.dg[3], Pm[r - 1], Lw);          //        (i-1, j-1, k-1)
                    if (wave_has_face) {
                        // x-face lanes: in-face part halved + the antisymmetric remainder, no coupling across the face; the
                        // clamped face gives 0 when Dirichlet rows are masked
                        double d1[3], d2[3], d3[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            d1[c] = P0[r + 1][c] - P0[r - 1][c];
                            d2[c] = Pp[r][c] - Pm[r][c];
                            d3[c] = Pp[r + 1][c] - Pm[r - 1][c];
                        }
                        const double cx = w.xf[0][0] * d1[1] + w.xf[0][1] * d1[2] + w.xf[1][0] * d2[1] + w.xf[1][1] * d2[2] +
                                          w.xf[2][0] * d3[1] + w.xf[2][1] * d3[2];
                        const double cy = -(w.xf[0][0] * d1[0] + w.xf[1][0] * d2[0] + w.xf[2][0] * d3[0]);
                        const double cz = -(w.xf[0][1] * d1[0] + w.xf[1][1] * d2[0] + w.xf[2][1] * d3[0]);
                        const double corr[3] = {cx, cy, cz};
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const double up = face_hi ? 0.0 : el_dpp_upper(U[c]), lo = face_lo ? 0.0 : el_dpp_lower(Lw[c]);
                            y[c] = fsc * y[c] + (up + lo) + fsg * corr[c];     // interior lanes: bitwise the plain y += up + lo
                            if (face_lo && a.mask_dirichlet) y[c] = 0.0;
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; ++c) y[c] += el_dpp_upper(U[c]) + el_dpp_lower(Lw[c]);
                    }
                    if (row_out[r]) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            a.out[c * a.stride + obase + (long long)r * g.PX] = y[c];
                            acc += P0[r][c] * y[c];
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < RH; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        Pm[r][c] = P0[r][c];
                        P0[r][c] = Pp[r][c];
                        Pp[r][c] = Pn[r][c];
                    }
            }
        }
    }
    const double tot = block_sum_256(acc, sRed);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
}

// r -= alpha q;  partial r.z (z = D^-1 r, not stored) and r.r     (use_alpha == 0: alpha = 0, the sums only)
template <int D>
__global__ __launch_bounds__(256) void k_el_k2(ElArgs a, const double* __restrict__ q, double* __restrict__ r, int use_alpha) {
    __shared__ double sRed[4];
    __shared__ double sDinv[27 * 3];
    if (threadIdx.x < 81) sDinv[threadIdx.x] = a.dinv[threadIdx.x];
    __syncthreads();
    const GridDev& g = a.g;
    const long long total = (long long)g.nkl * g.PP;
    double alpha = 0.0;
    if (use_alpha) {
        double pq;
        if (a.din) {
            pq = el_sum_in(a.din, a.din_n, sRed);
            if (blockIdx.x == 0 && threadIdx.x == 0) const_cast<double*>(a.scal)[ESC_PQ] = pq;
        } else {
            pq = a.scal[ESC_PQ];
        }
        const double rz = a.scal[a.s_rz];
        alpha = pq > 0.0 ? rz / pq : 0.0;
    }
    double acc_rz = 0.0, acc_rr = 0.0;
    // 16 B per lane: element pairs (i, i+1) of a row (PX is even, so a pair never straddles rows); (i, j, lk) of the pair advance
    // incrementally by the grid stride — one division per thread instead of two per element
    const long long total2 = total / 2;
    const int PX2 = g.PX / 2;
    const long long step = (long long)gridDim.x * 256;
    const int di = (int)(step % PX2);
    const long long drow = step / PX2;
    const int dj = (int)(drow % g.NY), dk = (int)(drow / g.NY);
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    int i2 = (int)(e % PX2), j = (int)((e / PX2) % g.NY), lk = (int)((e / PX2) / g.NY);
    for (; e < total2; e += step) {
        const long long idx = 2 * e;
        const int kg = g.k0 + lk;
        const int i = 2 * i2;
        const int cyz = 3 * (g.dim >= 2 ? pg_axis_class(j, g.NY) : 1) + 9 * (g.dim >= 3 ? pg_axis_class(kg, g.NZ) : 1);
        const int cls0 = pg_axis_class(i < g.NX ? i : 1, g.NX) + cyz;
        const int cls1 = pg_axis_class(i + 1 < g.NX ? i + 1 : 1, g.NX) + cyz;
        i2 += di;
        j += dj;
        lk += dk;
        if (i2 >= PX2) { i2 -= PX2; ++j; }
        if (j >= g.NY) { j -= g.NY; ++lk; }
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double2 rv = *reinterpret_cast<double2*>(r + c * a.stride + idx);
            if (use_alpha) {
                const double2 qv = *reinterpret_cast<const double2*>(q + c * a.stride + idx);
                rv.x -= alpha * qv.x;
                rv.y -= alpha * qv.y;
                *reinterpret_cast<double2*>(r + c * a.stride + idx) = rv;
            }
            // z = D^-1 r is not stored: the p-update forms it again from r (one vector pass less per iteration)
            const double2 zv = make_double2(sDinv[cls0 * 3 + c] * rv.x, sDinv[cls1 * 3 + c] * rv.y);
            acc_rz += rv.x * zv.x + rv.y * zv.y;
            acc_rr += rv.x * rv.x + rv.y * rv.y;
        }
    }
    const double t1 = block_sum_256(acc_rz, sRed);
    const double t2 = block_sum_256(acc_rr, sRed);
    if (threadIdx.x == 0) {
        a.partials[blockIdx.x] = t1;
        a.partials2[blockIdx.x] = t2;
    }
}

// p_n = z_n + beta p_{n-1};  x += alpha_{n-1} p_{n-1}   over the owned planes AND the ghost planes inside the domain
// (so that the block apply that follows finds p_n on the ghosts: z and p_{n-1} were exchanged before)
template <int D>
__global__ __launch_bounds__(256) void k_el_pupdate(ElArgs a, long long first, long long total) {
    __shared__ double sRed2[8];
    __shared__ double sDinv[27 * 3];
    if (threadIdx.x < 81) sDinv[threadIdx.x] = a.dinv[threadIdx.x];
    __syncthreads();
    double rz;
    if (a.din) {
        // both sums in one pass: all loads issued before the first barrier (each is an L2 miss right after a kernel boundary)
        double a1 = 0.0, a2 = 0.0;
        for (int e = threadIdx.x; e < a.din_n; e += 256) {
            a1 += a.din[e];
            a2 += a.din2[e];
        }
        a1 = wave_sum(a1);
        a2 = wave_sum(a2);
        if ((threadIdx.x & 63) == 0) {
            sRed2[threadIdx.x >> 6] = a1;
            sRed2[4 + (threadIdx.x >> 6)] = a2;
        }
        __syncthreads();
        rz = (sRed2[0] + sRed2[1]) + (sRed2[2] + sRed2[3]);
        const double rr = (sRed2[4] + sRed2[5]) + (sRed2[6] + sRed2[7]);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            double* sc = const_cast<double*>(a.scal);
            sc[a.s_rz] = rz;          // a slot no workgroup of this launch reads
            sc[ESC_RR] = rr;
        }
    } else {
        rz = a.scal[a.s_rz];
    }
    const double pq = a.scal[ESC_PQ], rz_prev = a.scal[a.s_rz_old];
    const double alpha_prev = (pq > 0.0 && rz_prev > 0.0) ? rz_prev / pq : 0.0;
    const double beta = (rz_prev > 0.0) ? rz / rz_prev : 0.0;
    // 16 B per lane: first, total and the component stride are multiples of 16 doubles (PX is), the origins 128-B aligned.
    // a.in = r; z = D^-1 r is formed here from the class of the node (Jacobi diagonal of its truncated stencil), so K2 does not
    // have to store z.  (i, j, lk) of a pair advance incrementally by the grid stride; lk starts at -1 when a lower ghost plane exists.
    const GridDev& g = a.g;
    const long long total2 = total / 2;
    const int PX2 = g.PX / 2;
    const long long step = (long long)gridDim.x * 256;
    const int di = (int)(step % PX2);
    const long long drow = step / PX2;
    const int dj = (int)(drow % g.NY), dk = (int)(drow / g.NY);
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    int i2 = (int)(e % PX2), j = (int)((e / PX2) % g.NY), lk = (int)((e / PX2) / g.NY) + (int)(first / g.PP);
    for (; e < total2; e += step) {
        const long long idx = first + 2 * e;
        const int kg = g.k0 + lk;
        const int i = 2 * i2;
        const int cyz = 3 * (g.dim >= 2 ? pg_axis_class(j, g.NY) : 1) + 9 * (g.dim >= 3 ? pg_axis_class(kg, g.NZ) : 1);
        const int cls0 = pg_axis_class(i < g.NX ? i : 1, g.NX) + cyz;
        const int cls1 = pg_axis_class(i + 1 < g.NX ? i + 1 : 1, g.NX) + cyz;
        i2 += di;
        j += dj;
        lk += dk;
        if (i2 >= PX2) { i2 -= PX2; ++j; }
        if (j >= g.NY) { j -= g.NY; ++lk; }
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const double2 po = *reinterpret_cast<const double2*>(a.p_old + c * a.stride + idx);
            const double2 rv = *reinterpret_cast<const double2*>(a.in + c * a.stride + idx);
            const double2 zv = make_double2(sDinv[cls0 * 3 + c] * rv.x, sDinv[cls1 * 3 + c] * rv.y);
            *reinterpret_cast<double2*>(a.p_new + c * a.stride + idx) = make_double2(zv.x + beta * po.x, zv.y + beta * po.y);
            if (alpha_prev != 0.0) {
                double2 xv = *reinterpret_cast<double2*>(a.x + c * a.stride + idx);
                xv.x += alpha_prev * po.x;
                xv.y += alpha_prev * po.y;
                *reinterpret_cast<double2*>(a.x + c * a.stride + idx) = xv;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_el_reduce(const double* __restrict__ p1, const double* __restrict__ p2, int n,
                                                    double* __restrict__ scal, int which) {
    __shared__ double sRed[4];
    double a1 = 0.0, a2 = 0.0;
    for (int e = threadIdx.x; e < n; e += 256) {
        a1 += p1[e];
        if (p2) a2 += p2[e];
    }
    const double t1 = block_sum_256(a1, sRed);
    const double t2 = block_sum_256(a2, sRed);
    if (threadIdx.x == 0) {
        if (which == 0) scal[ESC_PQ] = t1;
        else {
            if (which == 1) scal[ESC_RZ_PREV] = scal[ESC_RZ];      // which == 2: ESC_RZ_PREV already holds the previous value
            scal[ESC_RZ] = t1;
            scal[ESC_RR] = t2;
        }
    }
}

template <int D>
__global__ __launch_bounds__(256) void k_el_flush(ElArgs a, const double* __restrict__ p) {
    const double pq = a.scal[ESC_PQ], rz_prev = a.scal[ESC_RZ_PREV];
    const double alpha = (pq > 0.0 && rz_prev > 0.0) ? rz_prev / pq : 0.0;
    if (alpha == 0.0) return;
    const long long total = (long long)a.g.nkl * a.g.PP;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256)
#pragma unroll
        for (int c = 0; c < D; ++c) a.x[c * a.stride + idx] += alpha * p[c * a.stride + idx];
}

// b_c(node) = body_c * m(node), m = int phi  (`dot(b, v)*dx`, :1678, :1828)
__global__ __launch_bounds__(256) void k_el_load(GridDev g, long long stride, int D, const double* __restrict__ load27,
                                                  double b0, double b1, double b2, double* __restrict__ b) {
    const long long total = (long long)g.nkl * g.PP;
    const double body[3] = {b0, b1, b2};
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int i = (int)(idx % g.PX);
        const long long row = idx / g.PX;
        const int j = (int)(row % g.NY);
        const int kg = g.k0 + (int)(row / g.NY);
        double m = 0.0;
        if (i < g.NX)
            m = load27[pg_axis_class(i, g.NX) + 3 * (g.dim >= 2 ? pg_axis_class(j, g.NY) : 1) +
                       9 * (g.dim >= 3 ? pg_axis_class(kg, g.NZ) : 1)];
        for (int c = 0; c < D; ++c) b[c * stride + idx] = body[c] * m;
    }
}

// per-cell, per-simplex von Mises equivalent stress / strain (:1690-1711, :1840-1859): eps = sym(grad u) is
// constant on each simplex; the deviator uses tr/3 with the d x d identity in 2-D as well (:1699-1700).
struct VmGeom {
    double g[6][4][3];     // gradients of the barycentric basis of each simplex type
    double vol[6];
    int verts[6][4];
    int ns;
};

template <int D>
__global__ __launch_bounds__(256) void k_vm_cells(GridDev g, long long stride, const double* __restrict__ u, VmGeom vg,
                                                   double lam, double mu, int quantity, double* __restrict__ vm /*[ns][ncell]*/) {
    const int cx = g.NX - 1, cy = D >= 2 ? g.NY - 1 : 1, cz = D >= 3 ? g.NZ - 1 : 1;
    const long long ncell = (long long)cx * cy * cz;
    for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < ncell; c += (long long)gridDim.x * 256) {
        const int i = (int)(c % cx), j = (int)((c / cx) % cy), k = (int)(c / ((long long)cx * cy));
        const long long base = (long long)k * g.PP + (long long)j * g.PX + i;
        double uu[8][D];
        const int nv = 1 << D;
        for (int l = 0; l < nv; ++l) {
            const long long off = base + (l & 1) + (long long)((l >> 1) & 1) * g.PX + (long long)((l >> 2) & 1) * g.PP;
            for (int a = 0; a < D; ++a) uu[l][a] = u[a * stride + off];
        }
        for (int s = 0; s < vg.ns; ++s) {
            double G[D][D];                       // grad u
            for (int a = 0; a < D; ++a)
                for (int r = 0; r < D; ++r) {
                    double t = 0.0;
                    for (int p = 0; p <= D; ++p) t += uu[vg.verts[s][p]][a] * vg.g[s][p][r];
                    G[a][r] = t;
                }
            double e[D][D], tr = 0.0;
            for (int a = 0; a < D; ++a) {
                for (int r = 0; r < D; ++r) e[a][r] = 0.5 * (G[a][r] + G[r][a]);
                tr += e[a][a];
            }
            double val;
            if (quantity == PDEGPU_QUANTITY_STRAIN) {
                double ss = 0.0;
                for (int a = 0; a < D; ++a)
                    for (int r = 0; r < D; ++r) {
                        const double dv = e[a][r] - (a == r ? tr / 3.0 : 0.0);
                        ss += dv * dv;
                    }
                val = sqrt(2.0 / 3.0 * ss);
            } else {
                double sg[D][D], trs = 0.0;
                for (int a = 0; a < D; ++a) {
                    for (int r = 0; r < D; ++r) sg[a][r] = 2.0 * mu * e[a][r] + (a == r ? lam * tr : 0.0);
                    trs += sg[a][a];
                }
                double ss = 0.0;
                for (int a = 0; a < D; ++a)
                    for (int r = 0; r < D; ++r) {
                        const double dv = sg[a][r] - (a == r ? trs / 3.0 : 0.0);
                        ss += dv * dv;
                    }
                val = sqrt(3.0 / 2.0 * ss);
            }
            vm[(long long)s * ncell + c] = val;
        }
    }
}

// right-hand side of project(eq_expr, Vs): r_i = sum_{T contains i} val_T |T| / (d+1)   (:1714, :1862) —
// node-centric gather over the incident cells (deterministic, no atomics)
template <int D>
__global__ __launch_bounds__(256) void k_vm_gather(GridDev g, VmGeom vg, const double* __restrict__ vm, double* __restrict__ rhs) {
    const int cx = g.NX - 1, cy = D >= 2 ? g.NY - 1 : 1, cz = D >= 3 ? g.NZ - 1 : 1;
    const long long ncell = (long long)cx * cy * cz;
    const long long total = (long long)g.nkl * g.PP;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int i = (int)(idx % g.PX);
        const long long row = idx / g.PX;
        const int j = (int)(row % g.NY);
        const int k = (int)(row / g.NY);
        double acc = 0.0;
        if (i < g.NX) {
            for (int sbits = 0; sbits < (1 << D); ++sbits) {       // incident cell (i - sx, j - sy, k - sz)
                const int sx = sbits & 1, sy = (sbits >> 1) & 1, sz = (sbits >> 2) & 1;
                const int ci = i - sx, cj = j - sy, ck = k - sz;
                if (ci < 0 || ci >= cx || cj < 0 || cj >= cy || ck < 0 || ck >= cz) continue;
                const long long c = ((long long)ck * cy + cj) * cx + ci;
                for (int s = 0; s < vg.ns; ++s) {
                    bool has = false;
                    for (int p = 0; p <= D; ++p) has |= (vg.verts[s][p] == sbits);
                    if (has) acc += vm[(long long)s * ncell + c] * (vg.vol[s] / (D + 1));
                }
            }
        }
        rhs[idx] = acc;
    }
}


// ---------------------------------------------------------------------------------------------
// element-by-element apply on the matrix cores: Y_e[24 x 16 elements] = K_e[24 x 24] U_e[24 x 16] with
// v_mfma_f64_16x16x4_f64 (2 M-tiles x 6 K-steps per 16 elements), gather from / scatter-add to the node arrays.
// The only dense contraction on the path (north_star).  Measured against the vector block stencil in
// fp64 MFMA issues at the fp64 VALU rate on MI355X and the 24 -> 32 row padding plus the
// scatter atomics make it the slower form, so the solver keeps the stencil; this kernel stays selectable
// (use_mfma = 1) and parity-tested.
// ---------------------------------------------------------------------------------------------
typedef double dbl4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_el_apply_mfma(GridDev g, long long stride, const double* __restrict__ Ke /*[24][24]*/,
                                                        const double* __restrict__ u, double* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int e = lane & 15, kg = lane >> 4;
    const int cx = g.NX - 1, cy = g.NY - 1, cz = g.NZ - 1;
    const int nbx = (cx + 15) / 16;
    const long long nbatch = (long long)nbx * cy * cz;
    // A operands: K_e tile entries, constant for every element (lane: row = l&15, k = l>>4)
    double A[2][6];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const int row = mt * 16 + e;
            A[mt][ks] = row < 24 ? Ke[row * 24 + ks * 4 + kg] : 0.0;
        }
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long b = wave; b < nbatch; b += nwaves) {
        const int bx = (int)(b % nbx), j = (int)((b / nbx) % cy), k = (int)(b / ((long long)nbx * cy));
        const int i = bx * 16 + e;                     // this lane's element (column of the MFMA)
        const bool live = i < cx;
        const long long base = (long long)k * g.PP + (long long)j * g.PX + (live ? i : 0);
        dbl4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const int dof = ks * 4 + kg, v = dof / 3, c = dof % 3;          // B[k = l>>4][col = l&15]
            const long long off = base + (v & 1) + (long long)((v >> 1) & 1) * g.PX + (long long)((v >> 2) & 1) * g.PP;
            const double bv = live ? u[c * stride + off] : 0.0;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[0][ks], bv, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[1][ks], bv, acc1, 0, 0, 0);
        }
        // D: col = lane & 15 (element), row = (lane >> 4) + 4 * reg   (f64 layout)
        if (live) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = kg + 4 * reg, v = row / 3, c = row % 3;
                const long long off = base + (v & 1) + (long long)((v >> 1) & 1) * g.PX + (long long)((v >> 2) & 1) * g.PP;
                atomicAdd(&y[c * stride + off], acc0[reg]);
            }
#pragma unroll
            for (int reg = 0; reg < 2; ++reg) {
                const int row = 16 + kg + 4 * reg, v = row / 3, c = row % 3;
                const long long off = base + (v & 1) + (long long)((v >> 1) & 1) * g.PX + (long long)((v >> 2) & 1) * g.PP;
                atomicAdd(&y[c * stride + off], acc1[reg]);
            }
        }
    }
}

// halo planes of the slab loop as ONE contiguous message per neighbour: [side][r0 r1 r2 p0 p1 p2][PP].  pack: first / last owned
// plane -> send buffer; unpack: receive buffer -> ghost planes.  PP is a multiple of 16 doubles, origins are 128-B aligned.
__global__ __launch_bounds__(256) void k_el_halo(double* __restrict__ r, double* __restrict__ p, long long stride, long long PP, int nkl,
                                                  int lo, int hi, double* __restrict__ buf, int unpack) {
    const long long per2 = PP / 2;                       // double2 per plane
    const long long total = 12 * per2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int slot = (int)(e / per2);                // side * 6 + vector * 3 + component
        const long long w = e % per2;
        const int side = slot / 6, vc = slot % 6;
        if ((side == 0 && !lo) || (side == 1 && !hi)) continue;
        double* v = (vc < 3 ? r : p) + (long long)(vc % 3) * stride;
        const long long plane = unpack ? (side == 0 ? -1 : nkl) : (side == 0 ? 0 : nkl - 1);
        double2* g2 = reinterpret_cast<double2*>(v + plane * PP) + w;
        double2* b2 = reinterpret_cast<double2*>(buf) + e;
        if (unpack) *g2 = *b2;
        else *b2 = *g2;
    }
}

inline int flat_grid(const pdegpu_ctx* ctx, long long n) {
    long long nb = (n + 255) / 256;
    const long long cap = 8LL * ctx->num_cu;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

struct EventPair {                 // RAII so that early returns do not leak events
    hipEvent_t a = nullptr, b = nullptr;
    bool ok = false;
    EventPair() { ok = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess; }
    ~EventPair() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
    float elapsed() {
        float ms = 0;
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
        return ms;
    }
};

struct ElSession {
    pdegpu_ctx* ctx = nullptr;
    GridDev g{};
    int D = 3;
    double lam = 0, mu = 0;
    long long stride = 0;
    double *tab = nullptr, *dinv = nullptr, *scal = nullptr, *part = nullptr, *part2 = nullptr, *part3 = nullptr;   // part3: p.q partials of the apply
    double *x = nullptr, *r = nullptr, *p[2] = {nullptr, nullptr}, *q = nullptr, *b = nullptr;   // origins
    std::vector<double*> bases;
    int pcur = 0;
    int nblk = 1;
    HeatTables ht{};
    double* h_pin = nullptr;
    ElInterior interior{};
    // 3-D fast apply (k_el_fast): interior class by z-marching, boundary nodes by extra workgroups of the same launch
    bool fast = false;
    int fast_ry = 1;
    ElW36 w36{};
    ElFastGeom fgeom{};
    unsigned* d_mask = nullptr;
    int last_apply_blocks = 1;      // workgroups (= partials) of the most recent apply launch
    double *halo_send = nullptr, *halo_recv = nullptr;      // [2 sides][6 planes][PP] each (slab loop, allocated on first use)

    double* alloc_vec() {
        double* base = nullptr;
        const size_t tot = (size_t)D * stride;
        if (hipMalloc(&base, tot * sizeof(double)) != hipSuccess) return nullptr;
        (void)hipMemsetAsync(base, 0, tot * sizeof(double), ctx->stream);
        bases.push_back(base);
        return base + vec_front(g);
    }
    ~ElSession() {
        for (double* b0 : bases) (void)hipFree(b0);
        if (tab) (void)hipFree(tab);
        if (dinv) (void)hipFree(dinv);
        if (scal) (void)hipFree(scal);
        if (part) (void)hipFree(part);
        if (part2) (void)hipFree(part2);
        if (part3) (void)hipFree(part3);
        if (d_mask) (void)hipFree(d_mask);
        if (halo_send) (void)hipFree(halo_send);
        if (halo_recv) (void)hipFree(halo_recv);
        if (h_pin) (void)hipHostFree(h_pin);
        tables_free(&ht);
    }
    ElArgs args() const {
        ElArgs a{};
        a.g = g;
        a.stride = stride;
        a.tab = tab;
        a.dinv = dinv;
        a.scal = scal;
        a.partials = part;
        a.partials2 = part2;
        a.mask_dirichlet = 1;
        return a;
    }
};

template <int D>
int el_launch_apply(ElSession* s, ElArgs a, int mode) {
    const int nb = s->nblk;
    switch (mode) {
        case APPLY_PLAIN: hipLaunchKernelGGL((k_el_apply<D, APPLY_PLAIN>), dim3(nb), dim3(256), 0, s->ctx->stream, a, s->interior); break;
        case APPLY_RESIDUAL: hipLaunchKernelGGL((k_el_apply<D, APPLY_RESIDUAL>), dim3(nb), dim3(256), 0, s->ctx->stream, a, s->interior); break;
        default: hipLaunchKernelGGL((k_el_apply<D, APPLY_CG_K1>), dim3(nb), dim3(256), 0, s->ctx->stream, a, s->interior); break;
    }
    PG_HIP(s->ctx, hipGetLastError());
    return PDEGPU_OK;
}
// interior weights in the 36-value form of k_el_fast; false if the host table does not have the structure it relies on
bool el_fast_weights(const double* w15 /*[15][9]*/, ElW36* out) {
    auto qof = [](int dx, int dy, int dz) {
        const int o = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
        for (int q = 0; q < PG_NOFF15; ++q)
            if (pg_off15(q) == o) return q;
        return -1;
    };
    double scale = 0.0;
    for (int t = 0; t < 9; ++t) scale = std::max(scale, std::fabs(w15[qof(0, 0, 0) * 9 + t]));
    const double tol = 1e-13 * scale;
    auto sym6 = [&](const double* b, double* o6) {
        if (std::fabs(b[1] - b[3]) > tol || std::fabs(b[2] - b[6]) > tol || std::fabs(b[5] - b[7]) > tol) return false;
        o6[0] = b[0]; o6[1] = b[4]; o6[2] = b[8]; o6[3] = b[1]; o6[4] = b[2]; o6[5] = b[5];
        return true;
    };
    auto same = [&](const double* b1, const double* b2) {
        for (int t = 0; t < 9; ++t)
            if (std::fabs(b1[t] - b2[t]) > tol) return false;
        return true;
    };
    if (!sym6(w15 + qof(0, 0, 0) * 9, out->c)) return false;
    const int axes[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r) {
        const double* bp = w15 + qof(axes[r][0], axes[r][1], axes[r][2]) * 9;
        const double* bm = w15 + qof(-axes[r][0], -axes[r][1], -axes[r][2]) * 9;
        if (!same(bp, bm) || !sym6(bp, out->ax[r])) return false;
    }
    const int dgs[4][3] = {{1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
    for (int r = 0; r < 4; ++r) {
        const double* bp = w15 + qof(dgs[r][0], dgs[r][1], dgs[r][2]) * 9;
        const double* bm = w15 + qof(-dgs[r][0], -dgs[r][1], -dgs[r][2]) * 9;
        double o6[6];
        if (!same(bp, bm) || !sym6(bp, o6)) return false;
        if (std::fabs(o6[0]) > tol || std::fabs(o6[1]) > tol || std::fabs(o6[2]) > tol) return false;
        out->dg[r][0] = o6[3]; out->dg[r][1] = o6[4]; out->dg[r][2] = o6[5];
    }
    return true;
}

// x-face structure of the class table (see ElW36::xf): fills xf and returns true when the two x-face classes (interior in y, z)
// have exactly that form
bool el_xface_weights(const double* tab /*[27][15][9]*/, ElW36* out) {
    auto qof = [](int dx, int dy, int dz) {
        const int o = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
        for (int q = 0; q < PG_NOFF15; ++q)
            if (pg_off15(q) == o) return q;
        return -1;
    };
    const double* ti = tab + 13 * PG_NOFF15 * 9;
    double scale = 0.0;
    for (int t = 0; t < 9; ++t) scale = std::max(scale, std::fabs(ti[qof(0, 0, 0) * 9 + t]));
    const double tol = 1e-12 * scale;
    const int inface[3][3] = {{0, 1, 0}, {0, 0, 1}, {0, 1, 1}};
    for (int face = 0; face < 2; ++face) {                    // 0: x = 0 (class 12, sign -1), 1: x = NX-1 (class 14, sign +1)
        const double* tf = tab + (face ? 14 : 12) * PG_NOFF15 * 9;
        const double sg = face ? 1.0 : -1.0;
        for (int q = 0; q < PG_NOFF15; ++q) {
            const int o = pg_off15(q), dx = o % 3 - 1;
            for (int t = 0; t < 9; ++t) {
                const double bf = tf[q * 9 + t], bi = ti[q * 9 + t];
                if (dx == (face ? 1 : -1) && std::fabs(bf) > tol) return false;             // outward: absent
                if (dx == (face ? -1 : 1) && std::fabs(bf - bi) > tol) return false;        // inward: as in the interior
            }
        }
        for (int t = 0; t < 9; ++t)
            if (std::fabs(tf[qof(0, 0, 0) * 9 + t] - 0.5 * ti[qof(0, 0, 0) * 9 + t]) > tol) return false;
        for (int k = 0; k < 3; ++k)
            for (int sgn = -1; sgn <= 1; sgn += 2) {
                const int q = qof(0, sgn * inface[k][1], sgn * inface[k][2]);
                double A[9];
                for (int t = 0; t < 9; ++t) A[t] = tf[q * 9 + t] - 0.5 * ti[q * 9 + t];
                const double a = sg * sgn * A[1], b = sg * sgn * A[2];
                if (face == 1 && sgn == 1) { out->xf[k][0] = a; out->xf[k][1] = b; }
                const double expect[9] = {0, sg * sgn * out->xf[k][0], sg * sgn * out->xf[k][1], -sg * sgn * out->xf[k][0], 0, 0, -sg * sgn * out->xf[k][1], 0, 0};
                if (!(face == 1 && sgn == 1))
                    for (int t = 0; t < 9; ++t)
                        if (std::fabs(A[t] - expect[t]) > tol) return false;
                if (std::fabs(A[0]) > tol || std::fabs(A[4]) > tol || std::fabs(A[5]) > tol || std::fabs(A[7]) > tol || std::fabs(A[8]) > tol ||
                    std::fabs(A[1] + A[3]) > tol || std::fabs(A[2] + A[6]) > tol) return false;
            }
    }
    return true;
}

void el_fast_geometry(ElSession* s) {
    const GridDev& g = s->g;
    ElFastGeom& fg = s->fgeom;
    const int RY = s->fast_ry;
    fg.full_lo = g.k0 == 0 ? 1 : 0;
    fg.full_hi = g.k0 + g.nkl == g.NZ ? 1 : 0;
    fg.lkb = fg.full_lo;
    fg.lke = g.nkl - fg.full_hi;
    const int planes = std::max(0, fg.lke - fg.lkb);
    const bool xface = s->w36.xface != 0;
    fg.ntx = (g.NX - (xface ? 1 : 2) + 61) / 62;
    fg.nty = (g.NY - 2 + RY - 1) / RY;
    fg.nyg = (fg.nty + 3) / 4;
    const bool yx = s->ctx->el_fast_yx != 0;
    const long long cols = yx ? (long long)fg.ntx * fg.nyg * 4 : (long long)fg.ntx * fg.nty;      // wavefront columns launched
    long long nzc = (s->ctx->el_fast_waves + cols - 1) / cols;
    nzc = std::max(1LL, std::min<long long>(nzc, std::max(1, planes)));
    fg.zlen = std::max(1, (int)((planes + nzc - 1) / nzc));
    fg.nzc = planes > 0 ? (planes + fg.zlen - 1) / fg.zlen : 0;
    fg.nb_fast = (int)((cols * fg.nzc + 3) / 4);
    fg.nb_fast = (fg.nb_fast + 7) / 8 * 8;          // multiple of 8 for the XCD-aware tile order (surplus waves find zc >= nzc and idle)
    fg.ring = 2 * g.NX + (xface ? 0 : 2 * (g.NY - 2));        // boundary nodes of an interior plane left to the class-table pass
    fg.nbnd = (long long)(fg.full_lo + fg.full_hi) * g.NX * g.NY + (long long)planes * fg.ring;
    fg.nb_bnd = (int)std::min<long long>((fg.nbnd + 255) / 256, 2LL * s->ctx->num_cu);      // grid-stride: the 29 KB table is loaded per workgroup
    fg.mask = s->d_mask;
}

int el_apply(ElSession* s, const ElArgs& a, int mode) {
    if (mode == APPLY_PLAIN && s->fast && s->ctx->el_fast) {
        const ElFastGeom& fg = s->fgeom;
        const int nb = fg.nb_fast + fg.nb_bnd;
        const bool yx = s->ctx->el_fast_yx != 0;
        if (s->fast_ry == 2) {
            if (yx) hipLaunchKernelGGL((k_el_fast<2, true>), dim3(nb), dim3(256), 0, s->ctx->stream, a, s->w36, fg);
            else hipLaunchKernelGGL((k_el_fast<2, false>), dim3(nb), dim3(256), 0, s->ctx->stream, a, s->w36, fg);
        } else {
            if (yx) hipLaunchKernelGGL((k_el_fast<1, true>), dim3(nb), dim3(256), 0, s->ctx->stream, a, s->w36, fg);
            else hipLaunchKernelGGL((k_el_fast<1, false>), dim3(nb), dim3(256), 0, s->ctx->stream, a, s->w36, fg);
        }
        PG_HIP(s->ctx, hipGetLastError());
        s->last_apply_blocks = nb;
        return PDEGPU_OK;
    }
    s->last_apply_blocks = s->nblk;
    return s->D == 3 ? el_launch_apply<3>(s, a, mode) : el_launch_apply<2>(s, a, mode);
}
// lazy != nullptr: no reduce launches — K2's prologue sums the apply's p.q partials, the next p-update's prologue sums
// K2's r.z / r.r partials; *lazy carries the state of one chunk (which r.z slot is current, whether partials are pending)
struct ElLazy {
    int cur = ESC_RZ;
    bool pend = false;
};
int el_k2(ElSession* s, int use_alpha, ElLazy* lazy = nullptr) {
    ElArgs a = s->args();
    if (lazy) {
        a.din = s->part3;
        a.din_n = s->last_apply_blocks;
        a.s_rz = lazy->cur;
    }
    if (s->D == 3) hipLaunchKernelGGL(k_el_k2<3>, dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, s->q, s->r, use_alpha);
    else hipLaunchKernelGGL(k_el_k2<2>, dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, s->q, s->r, use_alpha);
    PG_HIP(s->ctx, hipGetLastError());
    if (lazy) {
        lazy->pend = true;
        return PDEGPU_OK;
    }
    hipLaunchKernelGGL(k_el_reduce, dim3(1), dim3(256), 0, s->ctx->stream, s->part, s->part2, s->nblk, s->scal, 1);
    PG_HIP(s->ctx, hipGetLastError());
    return PDEGPU_OK;
}
// end of a lazy chunk: the pending r.z / r.r partials -> canonical slots (ESC_RZ current, ESC_RZ_PREV previous)
int el_lazy_finish(ElSession* s, ElLazy* lazy) {
    if (!lazy->pend) return PDEGPU_OK;
    hipLaunchKernelGGL(k_el_reduce, dim3(1), dim3(256), 0, s->ctx->stream, s->part, s->part2, s->nblk, s->scal, lazy->cur == ESC_RZ ? 1 : 2);
    PG_HIP(s->ctx, hipGetLastError());
    lazy->pend = false;
    lazy->cur = ESC_RZ;
    return PDEGPU_OK;
}
// K1 = flat p-update (+ deferred x-update) followed by the plain block apply q = P A p with the p.q partials.
// (The heat path recomputes p at the stencil neighbours to save an HBM pass; here everything is cache-resident and
// the 15 x 3 x 2 neighbour reads of the fused form cost more than the extra launch: measured 110 -> ~60 us.)
int el_k1(ElSession* s, ElLazy* lazy = nullptr) {
    ElArgs a = s->args();
    a.in = s->r;                 // the p-update forms z = D^-1 r itself
    a.p_old = s->p[s->pcur];
    a.p_new = s->p[1 - s->pcur];
    a.x = s->x;
    if (lazy) {
        if (lazy->pend) {
            a.din = s->part;
            a.din2 = s->part2;
            a.din_n = s->nblk;
            a.s_rz_old = lazy->cur;
            a.s_rz = lazy->cur == ESC_RZ ? ESC_RZ_PREV : ESC_RZ;
            lazy->cur = a.s_rz;
            lazy->pend = false;
        } else {
            a.s_rz = lazy->cur;
            a.s_rz_old = lazy->cur == ESC_RZ ? ESC_RZ_PREV : ESC_RZ;
        }
    }
    const int lo = s->g.k0 > 0 ? 1 : 0, hi = (s->g.k0 + s->g.nkl < s->g.NZ) ? 1 : 0;
    const long long first = -(long long)lo * s->g.PP, total = (long long)(s->g.nkl + lo + hi) * s->g.PP;
    if (s->D == 3) hipLaunchKernelGGL(k_el_pupdate<3>, dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, first, total);
    else hipLaunchKernelGGL(k_el_pupdate<2>, dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, first, total);
    PG_HIP(s->ctx, hipGetLastError());
    s->pcur ^= 1;
    ElArgs b = s->args();
    b.in = s->p[s->pcur];
    b.out = s->q;
    b.partials = s->part3;
    PG_TRY(el_apply(s, b, APPLY_PLAIN));
    if (lazy) return PDEGPU_OK;
    hipLaunchKernelGGL(k_el_reduce, dim3(1), dim3(256), 0, s->ctx->stream, s->part3, (const double*)nullptr, s->last_apply_blocks, s->scal, 0);
    PG_HIP(s->ctx, hipGetLastError());
    return PDEGPU_OK;
}
int el_scalars(ElSession* s, double* out8) {
    PG_HIP(s->ctx, hipMemcpyAsync(s->h_pin, s->scal, ESC_COUNT * sizeof(double), hipMemcpyDeviceToHost, s->ctx->stream));
    PG_HIP(s->ctx, hipStreamSynchronize(s->ctx->stream));
    memcpy(out8, s->h_pin, ESC_COUNT * sizeof(double));
    return PDEGPU_OK;
}

int el_create(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el, ElSession* s, int64_t k0 = 0, int64_t nk = -1) {
    s->ctx = ctx;
    s->D = grid->dim;
    pdegpu_heat bc{};
    bc.bc_mode = PDEGPU_BC_DIRECTIONAL;      // clamp x = 0 (:1681-1684, :1831-1834); everything else natural
    bc.bc_present[0] = 1;
    bc.bc_value[0] = 0.0;
    const int64_t nslow = grid->dim == 3 ? (int64_t)grid->n[2] + 1 : 1;
    if (nk < 0) nk = nslow;
    if (k0 < 0 || nk < 1 || k0 + nk > nslow) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "slab outside the node planes");
    s->g = make_grid_dev(grid, &bc, k0, nk);
    s->stride = (long long)vec_total(s->g);
    s->mu = el->E / (2.0 * (1.0 + el->nu));
    if (grid->dim == 2 && el->plane_stress) s->lam = el->E * el->nu / (1.0 - el->nu * el->nu);         // :1660-1661
    else s->lam = el->E * el->nu / ((1.0 + el->nu) * (1.0 - 2.0 * el->nu));                           // :1664-1665, :1813-1814
    {
        long long nb = ((long long)s->g.nkl * s->g.PP + 255) / 256;
        const long long cap = (long long)ctx->el_grid_mult * ctx->num_cu;
        s->nblk = (int)(nb > cap ? cap : (nb < 1 ? 1 : nb));
    }
    PG_HIP(ctx, hipMalloc(&s->tab, 27 * PG_NOFF15 * 9 * sizeof(double)));
    PG_HIP(ctx, hipMalloc(&s->dinv, 81 * sizeof(double)));
    PG_HIP(ctx, hipMalloc(&s->scal, ESC_COUNT * sizeof(double)));
    PG_HIP(ctx, hipMalloc(&s->d_mask, 32 * sizeof(unsigned)));
    PG_HIP(ctx, hipHostMalloc(&s->h_pin, ESC_COUNT * sizeof(double)));
    PG_HIP(ctx, hipMemsetAsync(s->scal, 0, ESC_COUNT * sizeof(double), ctx->stream));
    double h[3] = {1, 1, 1};
    for (int r = 0; r < grid->dim; ++r) h[r] = grid->L[r] / grid->n[r];
    hipLaunchKernelGGL(k_el_tables, dim3(1), dim3(256), 0, ctx->stream, grid->dim, h[0], h[1], h[2], s->lam, s->mu, s->tab, s->dinv, s->d_mask);
    PG_HIP(ctx, hipGetLastError());
    for (int q = 0; q < PG_NOFF15; ++q) fold_elasticity_block(grid->dim, h, s->lam, s->mu, 13, q, s->interior.w + q * 9);
    size_t npart = (size_t)s->nblk;
    s->fast_ry = ctx->el_fast_ry == 2 ? 2 : 1;
    s->fast = grid->dim == 3 && s->g.NX >= 3 && s->g.NY >= 3 && s->g.NZ >= 3 && el_fast_weights(s->interior.w, &s->w36);
    s->w36.xface = 0;
    if (s->fast && ctx->el_fast_xface) {
        // x-face rows inside the marching kernel: only when the host table has the structure the kernel relies on
        std::vector<double> htab((size_t)27 * PG_NOFF15 * 9);
        for (int e = 0; e < 27 * PG_NOFF15; ++e) fold_elasticity_block(grid->dim, h, s->lam, s->mu, e / PG_NOFF15, e % PG_NOFF15, htab.data() + (size_t)e * 9);
        s->w36.xface = el_xface_weights(htab.data(), &s->w36) ? 1 : 0;
    }
    if (s->fast) {
        el_fast_geometry(s);
        if (s->fgeom.nbnd > 0x7fffffffLL) s->fast = false;
        npart = std::max(npart, (size_t)s->fgeom.nb_fast + (size_t)((s->fgeom.nbnd + 255) / 256));
    }
    PG_HIP(ctx, hipMalloc(&s->part, npart * sizeof(double)));
    PG_HIP(ctx, hipMalloc(&s->part2, npart * sizeof(double)));
    PG_HIP(ctx, hipMalloc(&s->part3, npart * sizeof(double)));
    PG_TRY(tables_build(ctx, grid, &s->ht));
    double** vecs[] = {&s->x, &s->r, &s->p[0], &s->p[1], &s->q, &s->b};
    for (double** v : vecs) {
        *v = s->alloc_vec();
        if (!*v) return pdegpu_set_err(ctx, PDEGPU_EHIP, "hipMalloc of an elasticity vector failed");
    }
    return PDEGPU_OK;
}

// b = body * m ; x = 0 ; r = P(b - A 0) = P b ; z = D^-1 r ; local rho and r.r  (no halo needed: x = 0)
int el_begin(ElSession* s, const pdegpu_elasticity* el) {
    pdegpu_ctx* ctx = s->ctx;
    const size_t bytes = (size_t)s->D * s->stride * sizeof(double);
    for (double* v : {s->x, s->p[0], s->p[1], s->q})
        PG_HIP(ctx, hipMemsetAsync(v - vec_front(s->g), 0, bytes, ctx->stream));
    PG_HIP(ctx, hipMemsetAsync(s->scal, 0, ESC_COUNT * sizeof(double), ctx->stream));
    s->pcur = 0;
    hipLaunchKernelGGL(k_el_load, dim3(s->nblk), dim3(256), 0, ctx->stream, s->g, s->stride, s->D, s->ht.d_load, el->body[0], el->body[1],
                       s->D == 3 ? el->body[2] : 0.0, s->b);
    PG_HIP(ctx, hipGetLastError());
    ElArgs a = s->args();
    a.in = s->x;
    a.rhs = s->b;
    a.out = s->r;
    PG_TRY(el_apply(s, a, APPLY_RESIDUAL));
    return el_k2(s, 0);
}

int el_flush(ElSession* s) {
    ElArgs a = s->args();
    a.x = s->x;
    if (s->D == 3) hipLaunchKernelGGL(k_el_flush<3>, dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, s->p[s->pcur]);
    else hipLaunchKernelGGL(k_el_flush<2>, dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, s->p[s->pcur]);
    PG_HIP(s->ctx, hipGetLastError());
    PG_HIP(s->ctx, hipMemsetAsync(s->scal + ESC_PQ, 0, sizeof(double), s->ctx->stream));   // applied: next K1 skips it
    return PDEGPU_OK;
}

VmGeom make_vm_geom(const pdegpu_grid* grid) {
    VmGeom vg{};
    double h[3] = {1, 1, 1};
    for (int r = 0; r < grid->dim; ++r) h[r] = grid->L[r] / grid->n[r];
    vg.ns = pg_num_simplices(grid->dim);
    for (int s = 0; s < vg.ns; ++s) {
        pg_simplex_verts(grid->dim, s, vg.verts[s]);
        simplex_grads(grid->dim, vg.verts[s], h, vg.g[s], &vg.vol[s]);
    }
    return vg;
}

int copy_out_component(ElSession* s, const double* comp, double* dense_dev, double* host, size_t N) {
    PG_TRY(vec_pack(s->ctx, s->g, comp, dense_dev));
    PG_HIP(s->ctx, hipMemcpyAsync(host, dense_dev, N * sizeof(double), hipMemcpyDeviceToHost, s->ctx->stream));
    PG_HIP(s->ctx, hipStreamSynchronize(s->ctx->stream));
    return PDEGPU_OK;
}

// von Mises per simplex -> projection right-hand side -> consistent-mass solve; optional displacement download.
// Needs the whole grid on this device (single slab).
int el_postprocess(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el, ElSession* sp, double* field_out,
                   double* disp_out, pdegpu_stats* st2_out) {
    ElSession& s = *sp;
    const int D = s.D;
    const size_t N = (size_t)pdegpu_grid_num_nodes(grid);
    const int cx = s.g.NX - 1, cy = s.g.NY - 1, cz = D == 3 ? s.g.NZ - 1 : 1;
    const long long ncell = (long long)cx * cy * cz;
    VmGeom vg = make_vm_geom(grid);
    double *d_vm = nullptr, *d_dense = nullptr;
    if (hipMalloc(&d_vm, (size_t)vg.ns * ncell * sizeof(double)) != hipSuccess ||
        hipMalloc(&d_dense, N * sizeof(double)) != hipSuccess) {
        if (d_vm) (void)hipFree(d_vm);
        return pdegpu_set_err(ctx, PDEGPU_EHIP, "hipMalloc of the von Mises work arrays failed");
    }
    if (D == 3) {
        hipLaunchKernelGGL(k_vm_cells<3>, dim3(flat_grid(ctx, ncell)), dim3(256), 0, ctx->stream, s.g, s.stride, s.x, vg, s.lam, s.mu,
                           el->quantity, d_vm);
        hipLaunchKernelGGL(k_vm_gather<3>, dim3(s.nblk), dim3(256), 0, ctx->stream, s.g, vg, d_vm, s.q);
    } else {
        hipLaunchKernelGGL(k_vm_cells<2>, dim3(flat_grid(ctx, ncell)), dim3(256), 0, ctx->stream, s.g, s.stride, s.x, vg, s.lam, s.mu,
                           el->quantity, d_vm);
        hipLaunchKernelGGL(k_vm_gather<2>, dim3(s.nblk), dim3(256), 0, ctx->stream, s.g, vg, d_vm, s.q);
    }
    int rc2 = hipGetLastError() == hipSuccess ? PDEGPU_OK : pdegpu_set_err(ctx, PDEGPU_EHIP, "von Mises kernels failed to launch");
    std::vector<double> rhs(N);
    if (rc2 == PDEGPU_OK) rc2 = copy_out_component(&s, s.q, d_dense, rhs.data(), N);
    if (rc2 == PDEGPU_OK && disp_out) {
        std::vector<double> comp(N);
        for (int c = 0; c < D && rc2 == PDEGPU_OK; ++c) {
            rc2 = copy_out_component(&s, s.x + (long long)c * s.stride, d_dense, comp.data(), N);
            for (size_t i = 0; i < N; ++i) disp_out[i * D + c] = comp[i];
        }
    }
    (void)hipFree(d_vm);
    (void)hipFree(d_dense);
    pdegpu_stats st2{};
    if (rc2 == PDEGPU_OK) rc2 = pdegpu_mass_solve(ctx, grid, rhs.data(), field_out, 1e-13, 100000, &st2);
    if (st2_out) *st2_out = st2;
    return rc2;
}

// 1-D axial bar (:1516-1564): EA u'' = -b, u(0) = 0; eps = project(u'), sigma = project(E u')
int bar_1d(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el, double* field_out, double* disp_out,
           pdegpu_stats* stats) {
    const int nx = grid->n[0];
    const size_t N = (size_t)nx + 1;
    pdegpu_heat h{};
    h.kappa = el->E * el->area;                 // EA (:1523)
    h.steady = 1;
    h.source = el->body[0];
    h.bc_mode = PDEGPU_BC_DIRECTIONAL;
    h.bc_present[0] = 1;
    h.bc_value[0] = 0.0;
    h.rtol = el->rtol > 0 ? el->rtol : 1e-12;
    h.max_iter = el->max_iter > 0 ? el->max_iter : 200000;
    std::vector<double> u(N), t(1), rhs(N, 0.0);
    pdegpu_stats st{};
    if (!(h.kappa > 0.0)) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "E*area must be > 0");
    PG_TRY(pdegpu_heat_solve(ctx, grid, &h, nullptr, u.data(), t.data(), &st));
    const double hx = grid->L[0] / nx;
    const double scale = el->quantity == PDEGPU_QUANTITY_STRAIN ? 1.0 : el->E;
    for (int c = 0; c < nx; ++c) {              // r_i = sum_T val_T |T| / 2
        const double val = scale * (u[c + 1] - u[c]) / hx;
        rhs[c] += val * hx * 0.5;
        rhs[c + 1] += val * hx * 0.5;
    }
    pdegpu_stats st2{};
    PG_TRY(pdegpu_mass_solve(ctx, grid, rhs.data(), field_out, 1e-13, 100000, &st2));
    if (disp_out) memcpy(disp_out, u.data(), N * sizeof(double));
    if (stats) {
        *stats = st;
        stats->cg_iterations += st2.cg_iterations;
        stats->linear_solves += st2.linear_solves;
    }
    return PDEGPU_OK;
}

}  // namespace

extern "C" int pdegpu_elasticity_solve(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el, double* field_out,
                                       double* disp_out, pdegpu_stats* stats) {
    if (!ctx) return PDEGPU_EINVAL;
    if (!el || !field_out) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "null argument");
    PG_TRY(validate_grid(ctx, grid));
    if (grid->element_mode != PDEGPU_ELEM_P1SPLIT) return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "elasticity uses the P1 split only");
    if (!(el->E > 0.0)) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "E must be > 0");
    if (el->quantity != PDEGPU_QUANTITY_STRESS && el->quantity != PDEGPU_QUANTITY_STRAIN)
        return pdegpu_set_err(ctx, PDEGPU_EINVAL, "quantity must be stress or strain");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    if (grid->dim == 1) return bar_1d(ctx, grid, el, field_out, disp_out, stats);
    if (!(el->nu > -1.0 && el->nu < 0.5)) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "nu must lie in (-1, 0.5)");

    ElSession s;
    PG_TRY(el_create(ctx, grid, el, &s));
    const int D = s.D;
    const size_t N = (size_t)pdegpu_grid_num_nodes(grid);
    pdegpu_stats st{};
    st.converged = 1;
    st.n_nodes = (int64_t)N;
    st.n_unknowns = (int64_t)D * (int64_t)(N - N / (grid->n[0] + 1));
    EventPair whole, cgt;
    if (!whole.ok || !cgt.ok) return pdegpu_set_err(ctx, PDEGPU_EHIP, "hipEventCreate failed");
    PG_HIP(ctx, hipEventRecord(whole.a, ctx->stream));

    PG_TRY(el_begin(&s, el));
    double sc[ESC_COUNT];
    PG_TRY(el_scalars(&s, sc));
    const double bnorm2 = sc[ESC_RR];
    const double rtol = el->rtol > 0 ? el->rtol : 1e-10;
    const int max_iter = el->max_iter > 0 ? el->max_iter : 200000;
    const double target = rtol * rtol * bnorm2;
    double rr = bnorm2;
    int it = 0, chunk = 16;
    PG_HIP(ctx, hipEventRecord(cgt.a, ctx->stream));
    while (rr > target && it < max_iter && std::isfinite(rr)) {
        const int n = std::min(chunk, max_iter - it);
        ElLazy lazy;
        ElLazy* lz = ctx->lazy_dot ? &lazy : nullptr;
        for (int i = 0; i < n; ++i) {
            PG_TRY(el_k1(&s, lz));
            PG_TRY(el_k2(&s, 1, lz));
        }
        if (lz) PG_TRY(el_lazy_finish(&s, lz));
        it += n;
        const double rr_before = rr;
        PG_TRY(el_scalars(&s, sc));
        rr = sc[ESC_RR];
        chunk = next_poll_chunk(rr_before, rr, target, n, chunk, 256);
    }
    PG_TRY(el_flush(&s));
    PG_HIP(ctx, hipEventRecord(cgt.b, ctx->stream));
    const float cms = cgt.elapsed();
    st.cg_iterations = it;
    st.cg_ms = cms;
    st.linear_solves = 1;
    st.final_rel_residual = bnorm2 > 0 ? std::sqrt(rr / bnorm2) : 0.0;
    int rc = PDEGPU_OK;
    if (!std::isfinite(rr)) rc = pdegpu_set_err(ctx, PDEGPU_ENOCONV, "elasticity PCG diverged after %d iterations", it);
    else if (rr > target) rc = pdegpu_set_err(ctx, PDEGPU_ENOCONV, "elasticity PCG did not reach rtol=%g in %d iterations (rel=%g)", rtol, it,
                                              st.final_rel_residual);
    if (rc != PDEGPU_OK) st.converged = 0;

    pdegpu_stats st2{};
    const int rc2 = el_postprocess(ctx, grid, el, &s, field_out, disp_out, &st2);
    st.linear_solves += st2.linear_solves;
    PG_HIP(ctx, hipEventRecord(whole.b, ctx->stream));
    st.solve_ms = whole.elapsed();
    if (stats) *stats = st;
    return rc2 != PDEGPU_OK ? rc2 : rc;
}

namespace {
// y = K u through the matrix cores (3-D): zero y, one launch, scatter with fp64 atomics
int el_apply_mfma(ElSession* s, const double* u, double* y, double* d_Ke) {
    pdegpu_ctx* ctx = s->ctx;
    PG_HIP(ctx, hipMemsetAsync(y - vec_front(s->g), 0, (size_t)3 * s->stride * sizeof(double), ctx->stream));
    const long long nbatch = (long long)((s->g.NX - 1 + 15) / 16) * (s->g.NY - 1) * (s->g.NZ - 1);
    long long nb = (nbatch + 3) / 4;
    if (nb > 8LL * ctx->num_cu) nb = 8LL * ctx->num_cu;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_el_apply_mfma, dim3((unsigned)nb), dim3(256), 0, ctx->stream, s->g, s->stride, d_Ke, u, y);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}
int upload_macro_element(ElSession* s, const pdegpu_grid* grid, double** d_Ke) {
    double h[3] = {1, 1, 1};
    for (int r = 0; r < grid->dim; ++r) h[r] = grid->L[r] / grid->n[r];
    double Ke[24 * 24];
    for (int la = 0; la < 8; ++la)
        for (int lb = 0; lb < 8; ++lb) {
            double blk[9];
            macro_elasticity_block(3, h, s->lam, s->mu, la, lb, blk);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) Ke[(la * 3 + a) * 24 + lb * 3 + b] = blk[a * 3 + b];
        }
    PG_HIP(s->ctx, hipMalloc(d_Ke, sizeof Ke));
    PG_HIP(s->ctx, hipMemcpyAsync(*d_Ke, Ke, sizeof Ke, hipMemcpyHostToDevice, s->ctx->stream));
    PG_HIP(s->ctx, hipStreamSynchronize(s->ctx->stream));
    return PDEGPU_OK;
}
}  // namespace

// y = K u on the full grid (no clamp), u and y host arrays [N][dim] — for parity tests
extern "C" int pdegpu_elasticity_apply(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el, const double* u,
                                       double* y, int use_mfma) {
    if (!ctx) return PDEGPU_EINVAL;
    if (!el || !u || !y) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "null argument");
    PG_TRY(validate_grid(ctx, grid));
    if (grid->dim < 2) return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "block operator exists for dim 2 and 3");
    if (use_mfma && grid->dim != 3) return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "the MFMA element-by-element apply is 3-D only");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    ElSession s;
    PG_TRY(el_create(ctx, grid, el, &s));
    const int D = s.D;
    const size_t N = (size_t)pdegpu_grid_num_nodes(grid);
    double* d_dense = nullptr;
    PG_HIP(ctx, hipMalloc(&d_dense, N * sizeof(double)));
    std::vector<double> comp(N);
    int rc = PDEGPU_OK;
    for (int c = 0; c < D && rc == PDEGPU_OK; ++c) {
        for (size_t i = 0; i < N; ++i) comp[i] = u[i * D + c];
        if (hipMemcpyAsync(d_dense, comp.data(), N * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "upload failed");
        if (rc == PDEGPU_OK) rc = vec_unpack(ctx, s.g, d_dense, s.x + (long long)c * s.stride);
        if (rc == PDEGPU_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "sync failed");
    }
    if (rc == PDEGPU_OK && use_mfma) {
        double* d_Ke = nullptr;
        rc = upload_macro_element(&s, grid, &d_Ke);
        if (rc == PDEGPU_OK) rc = el_apply_mfma(&s, s.x, s.q, d_Ke);
        if (rc == PDEGPU_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "sync failed");
        if (d_Ke) (void)hipFree(d_Ke);
    } else if (rc == PDEGPU_OK) {
        ElArgs a = s.args();
        a.in = s.x;
        a.out = s.q;
        a.mask_dirichlet = 0;
        rc = el_apply(&s, a, APPLY_PLAIN);
    }
    for (int c = 0; c < D && rc == PDEGPU_OK; ++c) {
        rc = copy_out_component(&s, s.q + (long long)c * s.stride, d_dense, comp.data(), N);
        for (size_t i = 0; i < N; ++i) y[i * D + c] = comp[i];
    }
    (void)hipFree(d_dense);
    return rc;
}

// host twin of k_el_tables for the CPU test-suite: tab [27][15][9]
extern "C" int pdegpu_elasticity_tables_host(const pdegpu_grid* grid, double lam, double mu, double* tab) {
    if (!grid || !tab || grid->dim < 2 || grid->dim > 3) return PDEGPU_EINVAL;
    double h[3] = {1, 1, 1};
    for (int r = 0; r < grid->dim; ++r) {
        if (grid->n[r] < 1 || !(grid->L[r] > 0)) return PDEGPU_EINVAL;
        h[r] = grid->L[r] / grid->n[r];
    }
    for (int e = 0; e < 27 * PG_NOFF15; ++e) fold_elasticity_block(grid->dim, h, lam, mu, e / PG_NOFF15, e % PG_NOFF15, tab + e * 9);
    return PDEGPU_OK;
}

// roofline harness for the elasticity operator: average device time of `reps` full-grid applications y = K u
// (u = the load vector pattern) with the vector block stencil (use_mfma = 0) or the MFMA element-by-element form
extern "C" int pdegpu_elasticity_time_apply(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el, int use_mfma,
                                            int reps, double* avg_ms_out) {
    if (!ctx) return PDEGPU_EINVAL;
    if (!el || !avg_ms_out || reps < 1) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "bad argument");
    PG_TRY(validate_grid(ctx, grid));
    if (grid->dim != 3) return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "3-D only");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    ElSession s;
    PG_TRY(el_create(ctx, grid, el, &s));
    hipLaunchKernelGGL(k_el_load, dim3(s.nblk), dim3(256), 0, ctx->stream, s.g, s.stride, 3, s.ht.d_load, 1.0, -2.0, 3.0, s.x);
    PG_HIP(ctx, hipGetLastError());
    double* d_Ke = nullptr;
    if (use_mfma) PG_TRY(upload_macro_element(&s, grid, &d_Ke));
    ElArgs a = s.args();
    a.in = s.x;
    a.out = s.q;
    a.mask_dirichlet = 0;
    hipEvent_t e0, e1;
    PG_HIP(ctx, hipEventCreate(&e0));
    PG_HIP(ctx, hipEventCreate(&e1));
    int rc = PDEGPU_OK;
    for (int i = -1; i < reps && rc == PDEGPU_OK; ++i) {
        if (i == 0) (void)hipEventRecord(e0, ctx->stream);
        rc = use_mfma ? el_apply_mfma(&s, s.x, s.q, d_Ke) : el_apply(&s, a, APPLY_PLAIN);
    }
    (void)hipEventRecord(e1, ctx->stream);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (d_Ke) (void)hipFree(d_Ke);
    *avg_ms_out = ms / reps;
    return rc;
}

// ---------------------------------------------------------------------------------------------
// slab-resident elasticity PCG session: the unit the multi-GPU driver steps (pdegpu/dist.py), one slab of node
// planes [k0, k0+nk) per GPU plus one ghost plane per side.  Same protocol as the heat session.
// ---------------------------------------------------------------------------------------------
struct pdegpu_el_session {
    ElSession s;
    pdegpu_elasticity el{};
    pdegpu_grid grid{};
};

extern "C" int pdegpu_el_session_create(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el, int64_t k0, int64_t nk,
                                        pdegpu_el_session** out) {
    if (!ctx) return PDEGPU_EINVAL;
    if (!el || !out) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "null argument");
    *out = nullptr;
    PG_TRY(validate_grid(ctx, grid));
    if (grid->dim != 3) return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "slab sessions exist for 3-D elasticity");
    if (!(el->E > 0.0) || !(el->nu > -1.0 && el->nu < 0.5)) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "need E > 0 and nu in (-1, 0.5)");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    pdegpu_el_session* h = new pdegpu_el_session();
    h->el = *el;
    h->grid = *grid;
    const int rc = el_create(ctx, grid, el, &h->s, k0, nk);
    if (rc != PDEGPU_OK) {
        delete h;
        return rc;
    }
    *out = h;
    return PDEGPU_OK;
}

extern "C" void pdegpu_el_session_destroy(pdegpu_el_session* h) {
    if (!h) return;
    (void)hipSetDevice(h->s.ctx->device);
    (void)hipStreamSynchronize(h->s.ctx->stream);
    delete h;
}

extern "C" int pdegpu_el_session_begin(pdegpu_el_session* h) { return h ? el_begin(&h->s, &h->el) : PDEGPU_EINVAL; }
extern "C" int pdegpu_el_session_k1(pdegpu_el_session* h) { return h ? el_k1(&h->s) : PDEGPU_EINVAL; }
extern "C" int pdegpu_el_session_k2(pdegpu_el_session* h) { return h ? el_k2(&h->s, 1) : PDEGPU_EINVAL; }
extern "C" int pdegpu_el_session_flush(pdegpu_el_session* h) { return h ? el_flush(&h->s) : PDEGPU_EINVAL; }
extern "C" double* pdegpu_el_session_scalars(pdegpu_el_session* h) { return h ? h->s.scal : nullptr; }
extern "C" int64_t pdegpu_el_session_plane_doubles(const pdegpu_el_session* h) { return h ? (int64_t)h->s.g.PP : -1; }

// vec: 0 x, 1 r, 2 p (current), 3 q, 4 z;  comp in [0, 3);  which in [-1, nk]
extern "C" void* pdegpu_el_session_plane_ptr(pdegpu_el_session* h, int vec, int comp, int64_t which) {
    if (!h || comp < 0 || comp >= h->s.D || which < -1 || which > h->s.g.nkl) return nullptr;
    double* base = nullptr;
    switch (vec) {
        case 0: base = h->s.x; break;
        case 1: base = h->s.r; break;
        case 2: base = h->s.p[h->s.pcur]; break;
        case 3: base = h->s.q; break;
        default: return nullptr;
    }
    return base + (long long)comp * h->s.stride + which * h->s.g.PP;
}

// one contiguous halo message per neighbour: buffers [2 sides][r0 r1 r2 p0 p1 p2][plane_doubles]; side 0 = lower neighbour.
// which = 0 send, 1 receive; side_doubles = 6 * plane_doubles
extern "C" void* pdegpu_el_session_halo_buffer(pdegpu_el_session* h, int which, int side) {
    if (!h || (which != 0 && which != 1) || (side != 0 && side != 1)) return nullptr;
    ElSession& s = h->s;
    const size_t bytes = (size_t)12 * s.g.PP * sizeof(double);
    if (!s.halo_send) {
        (void)hipSetDevice(s.ctx->device);
        if (hipMalloc(&s.halo_send, bytes) != hipSuccess || hipMalloc(&s.halo_recv, bytes) != hipSuccess) return nullptr;
        (void)hipMemsetAsync(s.halo_send, 0, bytes, s.ctx->stream);
        (void)hipMemsetAsync(s.halo_recv, 0, bytes, s.ctx->stream);
    }
    return (which == 0 ? s.halo_send : s.halo_recv) + (size_t)side * 6 * s.g.PP;
}
// unpack = 0: first / last owned plane of r and p (current) -> send buffers; 1: receive buffers -> ghost planes.  Sides
// without a slab neighbour are skipped.
extern "C" int pdegpu_el_session_halo_pack(pdegpu_el_session* h, int unpack) {
    if (!h) return PDEGPU_EINVAL;
    ElSession& s = h->s;
    if (!pdegpu_el_session_halo_buffer(h, 0, 0)) return pdegpu_set_err(s.ctx, PDEGPU_EHIP, "hipMalloc of the halo buffers failed");
    const int lo = s.g.k0 > 0 ? 1 : 0, hi = (s.g.k0 + s.g.nkl < s.g.NZ) ? 1 : 0;
    if (!lo && !hi) return PDEGPU_OK;
    const long long total = 12 * (s.g.PP / 2);
    const int nb = (int)std::min<long long>((total + 255) / 256, 2LL * s.ctx->num_cu);
    hipLaunchKernelGGL(k_el_halo, dim3(nb), dim3(256), 0, s.ctx->stream, s.r, s.p[s.pcur], s.stride, s.g.PP, s.g.nkl, lo, hi,
                       unpack ? s.halo_recv : s.halo_send, unpack);
    PG_HIP(s.ctx, hipGetLastError());
    return PDEGPU_OK;
}

// synchronises; out4 = {p.q, r.z, r.r, r.z of the previous iteration} (global values once the host has all-reduced them)
extern "C" int pdegpu_el_session_scalars_host(pdegpu_el_session* h, double* out4) {
    if (!h || !out4) return PDEGPU_EINVAL;
    double sc[ESC_COUNT];
    PG_TRY(el_scalars(&h->s, sc));
    for (int i = 0; i < 4; ++i) out4[i] = sc[i];
    return PDEGPU_OK;
}

// displacement of the owned planes, host [nk * (ny+1) * (nx+1)][3]
extern "C" int pdegpu_el_session_get_u(pdegpu_el_session* h, double* u_out) {
    if (!h || !u_out) return PDEGPU_EINVAL;
    ElSession& s = h->s;
    const size_t n = (size_t)s.g.nkl * s.g.NY * s.g.NX;
    double* d_dense = nullptr;
    PG_HIP(s.ctx, hipMalloc(&d_dense, n * sizeof(double)));
    std::vector<double> comp(n);
    int rc = PDEGPU_OK;
    for (int c = 0; c < s.D && rc == PDEGPU_OK; ++c) {
        rc = copy_out_component(&s, s.x + (long long)c * s.stride, d_dense, comp.data(), n);
        for (size_t i = 0; i < n; ++i) u_out[i * s.D + c] = comp[i];
    }
    (void)hipFree(d_dense);
    return rc;
}

// von Mises + project() from a given displacement field (host [N][dim]) on one GPU — the post-processing of
// _solve_elasticity_{2,3}d_static (:1690-1718, :1840-1866) after a distributed solve has been gathered
extern "C" int pdegpu_elasticity_postprocess(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el, const double* disp,
                                             double* field_out) {
    if (!ctx) return PDEGPU_EINVAL;
    if (!el || !disp || !field_out) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "null argument");
    PG_TRY(validate_grid(ctx, grid));
    if (grid->dim < 2) return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "dim 2 or 3");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    ElSession s;
    PG_TRY(el_create(ctx, grid, el, &s));
    const size_t N = (size_t)pdegpu_grid_num_nodes(grid);
    double* d_dense = nullptr;
    PG_HIP(ctx, hipMalloc(&d_dense, N * sizeof(double)));
    std::vector<double> comp(N);
    int rc = PDEGPU_OK;
    for (int c = 0; c < s.D && rc == PDEGPU_OK; ++c) {
        for (size_t i = 0; i < N; ++i) comp[i] = disp[i * s.D + c];
        if (hipMemcpyAsync(d_dense, comp.data(), N * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "upload failed");
        if (rc == PDEGPU_OK) rc = vec_unpack(ctx, s.g, d_dense, s.x + (long long)c * s.stride);
        if (rc == PDEGPU_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "sync failed");
    }
    (void)hipFree(d_dense);
    if (rc == PDEGPU_OK) rc = el_postprocess(ctx, grid, el, &s, field_out, nullptr, nullptr);
    return rc;
}
