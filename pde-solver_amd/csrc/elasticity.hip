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
#include <type_traits>

namespace {

// device scalars of the elasticity PCG; RZ and RR are adjacent so the multi-GPU host all-reduces them as one 16-byte message
// Single-reduction form (el_cg_variant 1): the block {ESC_PQ, ESC_RZ, ESC_RR, ESC_QMQ} = {p.Ap, r.z, r.r, Ap.D^-1 Ap} is ONE 32-byte message per iteration
// (ESC_QMQ takes the slot the classic form uses for the previous r.z).
enum { ESC_PQ = 0, ESC_RZ = 1, ESC_RR = 2, ESC_RZ_PREV = 3, ESC_QMQ = 3,
       ESC_XD_FLAG = 4, ESC_XD_ALPHA = 5, ESC_XD_BETA = 6,      // deferred x update of the single-reduction form (el_xdefer, see k_el_update)
       ESC_COUNT = 8 };

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
    // single-reduction form: the apply also sums (A p) . D^-1 (A p) into partials_qmq (same indexing as partials); the fused update pass finds the
    // apply's two partial arrays in din3 / din4 (din_n2 entries each) next to its own r.z / r.r partials of the iteration before (din / din2)
    double* partials_qmq = nullptr;
    const double* din3 = nullptr;
    const double* din4 = nullptr;
    int din_n2 = 0;
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
    double acc = 0.0, acc2 = 0.0;
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
                if (real) {
                    acc += a.in[c * a.stride + idx] * s[c];
                    if (a.partials_qmq) acc2 += a.dinv[cls * 3 + c] * s[c] * s[c];      // (kernel-uniform branch)
                }
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
    if (a.partials_qmq) {
        const double t2 = block_sum_256(acc2, sRed);
        if (threadIdx.x == 0) a.partials_qmq[blockIdx.x] = t2;
    }
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
    // Face nodes inside the marching kernel.  On the host tables the class of a face with normal axis N and sign s (+1: high) is: offsets
    // pointing out of the grid absent, inward ones as in the interior, and for the in-face offsets o  B_face(+-o) = B(o) / 2 +- s A(g(o))
    // with an antisymmetric remainder that couples the normal component with the tangential ones, (A(g) v)_N = g . v, (A(g) v)_t = -g_t v_N
    // (the Kuhn split is not mirror symmetric); the centre block is exactly halved.  The 27 numbers g (3 axes x 3 in-face offsets e_p, e_q,
    // e_p + e_q) travel in ElFastGeom::gf (device memory -> LDS).  el_face_weights checks all of this before a bit is set.
    int faces;          // the marching part computes the nodes of: bit 0 the x-faces, bit 1 the y-faces, bit 2 the z-faces (each interior in the
                        // other two directions; a bit implies the lower ones) — with all three the class-table pass keeps the edges and corners
};
// The 36 interior weights are 18 numbers: on the uniform Kuhn split every off-diagonal entry of the 15 blocks is one of a = B(1,0,0)_xy,
// b = B(1,0,0)_xz, c = B(1,1,0)_yz times a signed power of two (xy: centre -2a, (1,0,0) a, (0,1,0) a, (0,0,1) -a/2, (1,1,0) -a, (1,0,1) a/2,
// (0,1,1) a/2, (1,1,1) -a/2; xz and yz alike), and the axis blocks have two distinct diagonal entries (normal, tangential).  The kernel takes
// these 18 (36 SGPRs instead of 72 — with 72 the compiler kept every other scalar of the marching loop in VGPR lanes, 40 v_readlane per plane)
// and rebuilds the 36 with signs as source modifiers; el_pack_weights checks the relations on the host table before the path is enabled.
struct ElW18 {
    double dc[3];             // centre block diagonal: xx yy zz
    double dp[2], dq[2], dr[2];      // (1,0,0), (0,1,0), (0,0,1): {normal-normal entry, the two tangential ones}
    double a[3], b[3], c[3];  // {a/2, a, 2a}, {b/2, b, 2b}, {c, 2c, 4c}
    int faces;                // see ElW36::faces
};
__device__ __forceinline__ ElW36 el_expand_weights(const ElW18& k) {
    ElW36 w;
    const double c6[6] = {k.dc[0], k.dc[1], k.dc[2], -k.a[2], -k.b[2], -k.c[2]};
    const double x6[6] = {k.dp[0], k.dp[1], k.dp[1], k.a[1], k.b[1], -k.c[0]};
    const double y6[6] = {k.dq[1], k.dq[0], k.dq[1], k.a[1], -k.b[0], k.c[1]};
    const double z6[6] = {k.dr[1], k.dr[1], k.dr[0], -k.a[0], k.b[1], k.c[1]};
    const double d3[4][3] = {{-k.a[1], k.b[0], k.c[0]}, {k.a[0], -k.b[1], k.c[0]}, {k.a[0], k.b[0], -k.c[1]}, {-k.a[0], -k.b[0], -k.c[0]}};
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        w.c[t] = c6[t];
        w.ax[0][t] = x6[t];
        w.ax[1][t] = y6[t];
        w.ax[2][t] = z6[t];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t) w.dg[r][t] = d3[r][t];
    w.faces = k.faces;
    return w;
}
// nodes a wave owns per row (lanes 1 .. 62; lanes 0 and 63 carry the x-halo).  48-node tiles on 128-byte boundaries were measured twice and are
// gone: slower at every size, HBM-resident grids included (profiles/r03_elasticity_apply_large_grids.log)
constexpr int kElXt = 62;
struct ElFastGeom {
    int ntx, nty, nzc, zlen;   // 62-node x-tiles, RY-row y-tiles, z-chunks of zlen planes
    int nlong, z0;             // the z-chunks z0 .. z0+nlong-1 of the first range have zlen planes, the others zlen - 1 (z0 = 0, nlong = nzc1: all of one
                               // length; z0 = 1 keeps the chunk that also computes the plane 0 short, like the last one with the plane NZ-1)
    int nyg;                   // groups of 4 y-tiles (one workgroup each) of the y-hand-off variant
    int lkb, lke;              // local interior plane range [lkb, lke)
    int lkb2, lke2, nzc1;      // plane-range launches of the slab loop: a second, disjoint range [lkb2, lke2) = the z-chunks nzc1 .. nzc-1 (the two edge
                               // planes of a slab in ONE launch); nzc1 = nzc and an empty range otherwise
    int nb_fast;               // workgroups of the marching part (4 waves each); they FOLLOW the nb_bnd boundary workgroups
    int nb_bnd;                // boundary workgroups
    int bnd_gran;              // nodes a boundary workgroup takes per pass (256 | 64)
    int bnd_last;              // 1: they take the LAST block indices
    int full_lo, full_hi;      // 1: this slab owns the global plane 0 / NZ-1 (all of its nodes are boundary nodes)
    int ring;                  // boundary nodes of an interior plane: 2 NX + 2 (NY - 2)
    long long nbnd;            // boundary nodes of the slab
    const unsigned* mask;      // [27] nonzero-block masks of the class table
    const double* gf;          // [3][3][3] face remainder vectors g (see ElW36::faces)
    unsigned long long* trace; // tuning aid (PDEGPU_EL_TRACE_FILE, pdegpu_elasticity_time_apply): per workgroup {start, end} of the 100 MHz clock + XCD / CU ids
};

__device__ __forceinline__ double el_dpp_lower(double v) {     // lane l <- lane l-1 (v_mov_b32_dpp wave_shr:1)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);      // bound_ctrl: lane 0 receives 0, no `old` operand to set up
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double el_dpp_upper(double v) {     // lane l <- lane l+1 (v_mov_b32_dpp wave_shl:1)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// block products as explicit FMA chains (y += b0 v0 + b3 v1 + b4 v2 written as a sum is contracted to mul + 2 fma + add: a third more
// issue slots in a kernel whose marching loop is VALU-co-bound)
__device__ __forceinline__ void el_sym6(const double* b, const double* v, double* y) {
    y[0] = fma(b[4], v[2], fma(b[3], v[1], fma(b[0], v[0], y[0])));
    y[1] = fma(b[5], v[2], fma(b[1], v[1], fma(b[3], v[0], y[1])));
    y[2] = fma(b[2], v[2], fma(b[5], v[1], fma(b[4], v[0], y[2])));
}
__device__ __forceinline__ void el_off3(const double* b, const double* v, double* y) {
    y[0] = fma(b[1], v[2], fma(b[0], v[1], y[0]));
    y[1] = fma(b[2], v[2], fma(b[0], v[0], y[1]));
    y[2] = fma(b[2], v[1], fma(b[1], v[0], y[2]));
}

// y += sg A(g) v with the face remainder A(g) v = e_N (g . v) - g v_N  (N = face normal, g_N = 0, sg = +-1)
template <int N>
__device__ __forceinline__ void el_rem(const double* g, double sg, const double* v, double* y) {
    constexpr int T1 = (N + 1) % 3, T2 = (N + 2) % 3;
    const double g1 = sg * g[T1], g2 = sg * g[T2];
    y[N] = fma(g2, v[T2], fma(g1, v[T1], y[N]));
    y[T1] = fma(-g1, v[N], y[T1]);
    y[T2] = fma(-g2, v[N], y[T2]);
}

// One node of a y- or z-face (interior in the two other directions), in the marching kernel's terms.  On the host tables the class of a
// face with normal axis N and sign s (+1: high face) is: centre block halved; for the in-face offsets +-o  B(o) / 2 +- s A(g(o)) with
// the antisymmetric remainder above (the Kuhn split is not mirror symmetric); inward offsets as in the interior, outward ones absent
// (el_face_weights checks exactly this before the path is enabled).  c = the node, (e1p, e1m) = its neighbours along the second
// in-face axis (+, -) — they are also the sources of the in-face x-diagonal —, n0 / n1 = the inward neighbour and the inward
// (0,1,1) diagonal neighbour.  G[k] = g of the in-face offsets e_x, e_second, e_x + e_second.  y: unshifted part, U / Lw: the parts the
// lanes i-1 / i+1 pick up through the x-shifts, exactly as for interior rows.
template <int N, bool HIGH>
__device__ __forceinline__ void el_face_node(const ElW36& w, const double (*G)[3], const double* c, const double* e1p, const double* e1m,
                                             const double* n0, const double* n1, double* y, double* U, double* Lw) {
    constexpr double s = HIGH ? 1.0 : -1.0;
    const double* Bin = N == 1 ? w.ax[2] : w.ax[1];       // second in-face axis: z for a y-face, y for a z-face
    const double* Din = N == 1 ? w.dg[1] : w.dg[0];       // in-face x-diagonal: (1,0,1) | (1,1,0)
    const double* Bnor = N == 1 ? w.ax[1] : w.ax[2];      // normal axis
    const double* Dnor = N == 1 ? w.dg[0] : w.dg[1];      // x-diagonal with a normal component: (1,1,0) | (1,0,1)
    // (every temporary lives only as long as its one use: the four plane buffers leave about 60 VGPRs at three waves per SIMD)
    double t[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        t[e] = 0.5 * c[e];
        y[e] = 0.0;
        U[e] = 0.0;
    }
    el_sym6(w.c, t, y);
    el_sym6(w.ax[0], t, U);
    Lw[0] = U[0]; Lw[1] = U[1]; Lw[2] = U[2];
    el_rem<N>(G[0], s, c, U);
    el_rem<N>(G[0], -s, c, Lw);
#pragma unroll
    for (int e = 0; e < 3; ++e) t[e] = 0.5 * (e1p[e] + e1m[e]);
    el_sym6(Bin, t, y);
#pragma unroll
    for (int e = 0; e < 3; ++e) t[e] = e1p[e] - e1m[e];
    el_rem<N>(G[1], s, t, y);
#pragma unroll
    for (int e = 0; e < 3; ++e) t[e] = 0.5 * e1p[e];
    el_off3(Din, t, U);
    el_rem<N>(G[2], s, e1p, U);
#pragma unroll
    for (int e = 0; e < 3; ++e) t[e] = 0.5 * e1m[e];
    el_off3(Din, t, Lw);
    el_rem<N>(G[2], -s, e1m, Lw);
    el_sym6(Bnor, n0, y);
    el_off3(w.dg[2], n1, y);
    el_off3(Dnor, n0, HIGH ? Lw : U);
    el_off3(w.dg[3], n1, HIGH ? Lw : U);
}

// Boundary workgroups of the marching launches: the nodes the marching part does not compute (all boundary nodes, or with faces folded in the
// edges and corners of the box), one thread per node, truncated class stencils from LDS.  Returns this thread's share of p . q.
__device__ __forceinline__ double el_boundary_role(const ElArgs& a, int faces, const ElFastGeom& fg, int bidx, double* sTab, unsigned* sMask, double* acc2_out) {
    const GridDev& g = a.g;
    double acc = 0.0, acc2 = 0.0;
    // ---- boundary nodes: one thread per node, truncated class stencils from LDS -----------------------------
    for (int e = threadIdx.x; e < 27 * PG_NOFF15 * 9; e += 256) sTab[e] = a.tab[e];
    if (threadIdx.x < 27) sMask[threadIdx.x] = fg.mask[threadIdx.x];
    __syncthreads();
    // (fg.bnd_gran nodes per workgroup and pass: 256, or 64 — one wavefront's worth — where the nodes are few: every thread's 48 loads go to 48 different
    // cache lines, and 256 such threads on ONE CU made the last boundary workgroup the longest-lived of config 5's launch, 18 us)
    for (long long e0 = (long long)bidx * fg.bnd_gran + threadIdx.x; threadIdx.x < fg.bnd_gran && e0 < fg.nbnd; e0 += (long long)fg.nb_bnd * fg.bnd_gran) {
        unsigned e = (unsigned)e0;                        // 32-bit divisions (nbnd < 2^31 is checked on the host)
        int i, j, lk;
        const unsigned plane_nodes = (unsigned)g.NX * (unsigned)g.NY;
        if (faces & 2) {
            // y-faces folded: the planes 0 / NZ-1 this slab owns (their perimeter only when the z-faces are folded as well), then the
            // four corners of every other plane
            const unsigned ringp = 2u * (unsigned)g.NX + 2u * (unsigned)(g.NY - 2);
            const unsigned per = (faces & 4) ? ringp : plane_nodes;
            const unsigned nlo = fg.full_lo ? per : 0u, nhi = fg.full_hi ? per : 0u;
            if (e < nlo + nhi) {
                lk = e < nlo ? -g.k0 : g.NZ - 1 - g.k0;
                const int t = (int)(e < nlo ? e : e - nlo);
                if (!(faces & 4)) { j = t / g.NX; i = t % g.NX; }
                else if (t < g.NX) { j = 0; i = t; }
                else if (t < 2 * g.NX) { j = g.NY - 1; i = t - g.NX; }
                else { j = 1 + (t - 2 * g.NX) / 2; i = ((t - 2 * g.NX) & 1) ? g.NX - 1 : 0; }
            } else {
                e -= nlo + nhi;
                const int pl = (int)(e >> 2), n1 = fg.lke - fg.lkb;
                lk = pl < n1 ? fg.lkb + pl : fg.lkb2 + (pl - n1);
                j = (e & 2u) ? g.NY - 1 : 0;
                i = (e & 1u) ? g.NX - 1 : 0;
            }
        } else if (fg.full_lo && e < plane_nodes) {
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
                const int pl = (int)(e / (unsigned)fg.ring), n1 = fg.lke - fg.lkb;
                lk = pl < n1 ? fg.lkb + pl : fg.lkb2 + (pl - n1);
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
        const int cls = pg_axis_class(i, g.NX) + 3 * pg_axis_class(j, g.NY) + 9 * pg_axis_class(kg, g.NZ);
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
            if (a.partials_qmq) acc2 += a.dinv[cls * 3 + c] * s[c] * s[c];      // (kernel-uniform branch; 81 doubles, cache-resident)
        }
    }
    *acc2_out = acc2;
    return acc;
}

// (min. 3 workgroups per CU for RY <= 2: the code fits 168 VGPRs without a spill, but left alone the scheduler trades the third wave per SIMD away)
// NTS: non-temporal stores of the output (a compile-time choice: behind a runtime branch the compiler merges the two stores and drops the hint)
template <int RY, bool YX, bool ZF, bool NTS>
__global__ __launch_bounds__(256, (RY <= 2) ? 3 : 2) void k_el_fast(ElArgs a, ElW18 wk, ElFastGeom fg) {
    const ElW36 w = el_expand_weights(wk);
    unsigned long long t_start = 0, t_pro = 0, t_planes = 0, t_loop = 0;      // (trace only: phase stamps of a marching workgroup)
    if (fg.trace) t_start = wall_clock64();
    __shared__ double sTab[27 * PG_NOFF15 * 9];
    __shared__ unsigned sMask[27];
    __shared__ double sRed[8];
    constexpr int RH = RY + 2;
    const GridDev& g = a.g;
    double acc = 0.0, acc2 = 0.0;                             // acc2: (A p) . D^-1 (A p) of the single-reduction PCG (a.partials_qmq)
    const bool qmq = a.partials_qmq != nullptr;              // kernel-uniform
    // workgroup roles: bnd_last = 0: boundary workgroups take the first nb_bnd block indices; 1: the last ones (they then start as the
    // marching workgroups retire, filling the tail of the launch)
    const int bidx = fg.bnd_last ? ((int)blockIdx.x >= fg.nb_fast ? (int)blockIdx.x - fg.nb_fast : -1) : ((int)blockIdx.x < fg.nb_bnd ? (int)blockIdx.x : -1);
    if (bidx >= 0) {
        acc = el_boundary_role(a, w.faces, fg, bidx, sTab, sMask, &acc2);
    } else {
        // ---- interior nodes: z-marching ------------------------------------------------------------------------
        const int lane = threadIdx.x & 63;
        // XCD-aware order: workgroups b and b+8 share an XCD (round-robin dispatch), so XCD x is given the contiguous range
        // [x*G/8, (x+1)*G/8) of tiles: y-neighbouring columns, whose halo rows overlap, then meet in one L2 (speed only)
        int bm = fg.bnd_last ? (int)blockIdx.x : (int)blockIdx.x - fg.nb_bnd;
        if ((fg.nb_fast & 7) == 0) bm = (bm & 7) * (fg.nb_fast >> 3) + (bm >> 3);
        if (w.faces || qmq) {                                 // every wave of the workgroup passes here
            if (w.faces && threadIdx.x < 27) sTab[3200 + threadIdx.x] = fg.gf[threadIdx.x];
            if (qmq && threadIdx.x < 81) sTab[3240 + threadIdx.x] = a.dinv[threadIdx.x];      // Jacobi diagonals of the 27 boundary classes
            __syncthreads();
        }
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform, and known to the compiler as such:
        int xt, yt, zc;                                                       // the row / halo conditions below stay scalar branches
        if (YX) {
            // the four waves of a workgroup are y-neighbours (same x-tile, same z-chunk): they hand their y-halo rows to
            // each other through LDS, so a workgroup loads 4 RY + 2 rows per plane instead of 4 (RY + 2)
            xt = bm % fg.ntx;
            yt = ((bm / fg.ntx) % fg.nyg) * 4 + wv;
            zc = fg.nzc - 1 - bm / (fg.ntx * fg.nyg);        // the chunk with the extra plane NZ-1 first; surplus workgroups get zc < 0
        } else {
            const int wid = bm * 4 + wv;
            xt = wid % fg.ntx;
            yt = (wid / fg.ntx) % fg.nty;
            zc = fg.nzc - 1 - wid / (fg.ntx * fg.nty);
        }
        const bool ghost_wave = YX && yt >= fg.nty;           // beyond the last row tile: takes part in the barriers only
        if (ghost_wave) yt = fg.nty - 1;                      // (loads stay inside the grid; nothing is stored)
        const int i = xt * kElXt + lane;                         // lanes 0 and 63: x-halo only (lane 0 of the first tile: the face x = 0)
        const int j0 = 1 + yt * RY;                           // first owned row; held rows j0-1 .. j0+RY
        const bool range2 = zc >= fg.nzc1;
        const int lk0 = range2 ? fg.lkb2 + (zc - fg.nzc1) * fg.zlen : fg.lkb + zc * (fg.zlen - 1) + min(max(zc - fg.z0, 0), fg.nlong);
        const int lk1 = min(lk0 + ((range2 || (zc >= fg.z0 && zc < fg.z0 + fg.nlong)) ? fg.zlen : fg.zlen - 1), range2 ? fg.lke2 : fg.lke);
        if (zc >= 0 && lk0 < lk1) {
            // Loads are unconditional (branch-free, all of a plane in flight at once): every address a wave can form lies
            // inside the allocation — rows up to NY (= first row of the next plane), columns up to NX+60 < PX+64, planes
            // -1 .. nkl, covered by the ghost planes and the guard of kGuard + one row on both sides — and a value from
            // outside the grid only ever reaches outputs that are not stored (an interior node has all 15 neighbours inside).
            // x-faces: lane 0 of the first tile is the node i = 0 (clamped: its row is 0 when Dirichlet rows are masked), the node
            // i = NX-1 (traction-free) sits on one of the lanes 1 .. 62 of the last tile
            const bool face_lo = (w.faces & 1) && i == 0, face_hi = (w.faces & 1) && i == g.NX - 1;
            const bool wave_has_face = (w.faces & 1) && (xt == 0 || (xt + 1) * kElXt >= g.NX - 1);      // wave-uniform
            const double kup = face_hi ? 0.0 : 1.0, klo = face_lo ? 0.0 : 1.0;      // no coupling across the face (the lanes beyond hold finite pad values)
            const bool out_x = (lane >= 1 && lane <= kElXt && i <= g.NX - ((w.faces & 1) ? 1 : 2)) || face_lo;
            const int cls_x = i == 0 ? 0 : (i == g.NX - 1 ? 2 : 1);      // boundary class of the lane's column along x
            double S2[3] = {0.0, 0.0, 0.0};                              // sum of y_c^2 over the lane's interior-row outputs (single-reduction PCG)
            const int rows_out = ghost_wave ? 0 : min(RY, g.NY - 1 - j0);      // owned rows inside the grid (wave-uniform)
            // lanes beyond the padded row (the last 62-node tile overshoots: 558 columns for a 513-node row) read the row's last pad
            // entry instead of the next row's data: same cache line as real data, no traffic of their own (9 % of the loads at
            // NX = 513); what they hold never reaches a stored output
            const long long col = (long long)(j0 - 1) * g.PX + (i < g.PX ? i : g.PX - 1);
            // four plane buffers whose roles (k-1, k, k+1, the plane in flight) rotate by NAME: the z-loop is unrolled four times, so no
            // register is ever copied (the copying form spent 96 of its ~450 VALU instructions per plane on the rotation)
            double PA[RH][3], PB[RH][3], PC[RH][3], PD[RH][3];
            auto load_plane = [&](int lk, double (*P)[3]) __attribute__((always_inline)) {
                const long long base = (long long)lk * g.PP + col;
#pragma unroll
                for (int r = 0; r < RH; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) P[r][c] = a.in[c * a.stride + base + (long long)r * g.PX];
            };
            // YX: a marching plane is loaded without the halo rows its y-neighbours in the workgroup own (the lowest wave
            // still loads row j0-1, the highest real wave row j0+RY); they arrive through LDS one step later, when the
            // plane has become the k+1 plane and its loads have landed.  The LDS block is the class table's (boundary workgroups only).
            const bool lo_global = !YX || wv == 0;
            const bool hi_global = !YX || wv == 3 || (yt + 1 >= fg.nty);
            auto load_own = [&](int lk, double (*P)[3]) __attribute__((always_inline)) {
                const long long base = (long long)lk * g.PP + col;
#pragma unroll
                for (int r = 1; r <= RY; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const double* src = a.in + c * a.stride + base + (long long)r * g.PX;
                        P[r][c] = *src;
                    }
                if (lo_global) {                              // wave-uniform
#pragma unroll
                    for (int c = 0; c < 3; ++c) P[0][c] = a.in[c * a.stride + base];
                }
                if (hi_global) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) P[RH - 1][c] = a.in[c * a.stride + base + (long long)(RH - 1) * g.PX];
                }
            };
            double* sY = sTab;                                // [2][4 waves][first | last own row][3][64]
            // face remainder vectors of axis ax, beyond the hand-off block.  The offset passes through an empty asm so that the reads stay where
            // they are used: hoisted out of the z-loop the 27 numbers cost 54 VGPRs and a third of the occupancy
            auto face_g = [&](int ax) __attribute__((always_inline)) {
                int off = 3200 + 9 * ax;
                asm volatile("" : "+v"(off));
                return reinterpret_cast<const double (*)[3]>(sTab + off);
            };
            auto y_exchange = [&](int ybuf, double (*P)[3]) __attribute__((always_inline)) {
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
            };
            // y- and z-face nodes (faces bit 1): which of them this wave computes
            const bool yz = (w.faces & 2) != 0, zf = ZF && (w.faces & 4) != 0;
            const bool ylo_wave = yz && !ghost_wave && yt == 0;                       // row 0 = the halo row below the first row tile
            const int r_hi = (yz && !ghost_wave && yt == fg.nty - 1) ? g.NY - j0 : 0;       // held-row index of the row NY-1 (2 .. RY+1)
            // (the chunk that starts at local plane 1 of a slab owning the global plane 0 / that ends below the global plane NZ-1)
            const bool zlo_chunk = zf && fg.full_lo && lk0 == 1, zhi_chunk = zf && fg.full_hi && lk1 == g.nkl - 1;
            const bool out_f = lane >= 1 && lane <= kElXt && i >= 1 && i <= g.NX - 2;      // face nodes proper (the x-ends are edges of the box)
            // D^-1 of one boundary class from LDS (read where it is used: hoisted out of the z-loop the values would cost registers the
            // marching loop does not have)
            auto dinv_of = [&](int cls) __attribute__((always_inline)) {
                int off = 3240 + 3 * cls;
                asm volatile("" : "+v"(off));
                return sTab + off;
            };
            // cls_face: boundary class of the face nodes this call finishes (x interior: 1 + 3 cy + 9 cz)
            auto face_finish = [&](long long o, const double* c0, double* y, const double* U, const double* Lw, int cls_face) __attribute__((always_inline)) {
#pragma unroll
                for (int c = 0; c < 3; ++c) y[c] += el_dpp_upper(U[c]) + el_dpp_lower(Lw[c]);
                if (out_f) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        a.out[c * a.stride + o] = y[c];
                        acc += c0[c] * y[c];
                    }
                    if (qmq) {
                        const double* dv = dinv_of(cls_face);
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc2 += dv[c] * y[c] * y[c];
                    }
                }
            };
            // the plane 0 / NZ-1 of the grid (owned rows of this tile; the perimeter of the plane stays with the class-table pass): Pin = the
            // inward neighbour plane
            auto zface_lo = [&](int lk, double (*P0)[3], double (*Pin)[3]) __attribute__((always_inline)) {
                const long long obase = (long long)lk * g.PP + col;
#pragma unroll
                for (int r = 1; r <= RY; ++r) {
                    double y[3], U[3], Lw[3];
                    el_face_node<2, false>(w, face_g(2), P0[r], P0[r + 1], P0[r - 1], Pin[r], Pin[r + 1], y, U, Lw);
                    if (r <= rows_out) face_finish(obase + (long long)r * g.PX, P0[r], y, U, Lw, 1 + 3 + 0);
                }
            };
            auto zface_hi = [&](int lk, double (*P0)[3], double (*Pin)[3]) __attribute__((always_inline)) {
                const long long obase = (long long)lk * g.PP + col;
#pragma unroll
                for (int r = 1; r <= RY; ++r) {
                    double y[3], U[3], Lw[3];
                    el_face_node<2, true>(w, face_g(2), P0[r], P0[r + 1], P0[r - 1], Pin[r], Pin[r - 1], y, U, Lw);
                    if (r <= rows_out) face_finish(obase + (long long)r * g.PX, P0[r], y, U, Lw, 1 + 3 + 18);
                }
            };
            auto step = [&](int lk, double (*Pm)[3], double (*P0)[3], double (*Pp)[3], double (*Pn)[3]) __attribute__((always_inline)) {
                // one more plane in flight: plane lk+2 is requested before plane lk is computed; lk + 2 can reach nkl + 1 (one plane
                // past the upper ghost plane) only when it is not needed: clamp it
                if (YX) {
                    load_own(min(lk + 2, lk1), Pn);
                    if (lk > lk0) y_exchange((lk - lk0) & 1, Pp);      // Pp was requested one step ago; every wave of the workgroup passes here
                } else {
                    load_plane(min(lk + 2, lk1), Pn);
                }
                const long long obase = (long long)lk * g.PP + col;
#pragma unroll
                for (int r = 1; r <= RY; ++r) {
                    // y = [unshifted terms] + up(U) + lo(L): the x-shifts act on two accumulated 3-vectors, not on 24 inputs
                    double y[3] = {0.0, 0.0, 0.0}, U[3] = {0.0, 0.0, 0.0}, Lw[3], v[3];
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
                        // clamped face gives 0 when Dirichlet rows are masked.  The six remainder weights come from LDS (uniform
                        // address: a broadcast read) — as kernel arguments they pushed the interior weights out of the SGPR file
                        // (one difference vector at a time: this branch is where the kernel's register pressure peaks)
                        const double (*G0)[3] = face_g(0);
                        double corr[3] = {0.0, 0.0, 0.0};
                        {
                            const double dv[3] = {P0[r + 1][0] - P0[r - 1][0], P0[r + 1][1] - P0[r - 1][1], P0[r + 1][2] - P0[r - 1][2]};
                            el_rem<0>(G0[0], 1.0, dv, corr);
                        }
                        {
                            const double dv[3] = {Pp[r][0] - Pm[r][0], Pp[r][1] - Pm[r][1], Pp[r][2] - Pm[r][2]};
                            el_rem<0>(G0[1], 1.0, dv, corr);
                        }
                        {
                            const double dv[3] = {Pp[r + 1][0] - Pm[r - 1][0], Pp[r + 1][1] - Pm[r - 1][1], Pp[r + 1][2] - Pm[r - 1][2]};
                            el_rem<0>(G0[2], 1.0, dv, corr);
                        }
                        const double fsc = 0.5 * (kup + klo), fsg = klo - kup;      // face lanes: 1/2, +-1 (high / low face); interior lanes: 1, 0
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            // the shifts run with every lane active and their results are masked by factors: written as
                            // `face_hi ? 0.0 : shift(U)` the compiler moved the DPP under exec = !face_hi, and the lane next to the face
                            // then read a disabled lane (= 0) instead of the face node's contribution
                            const double up = el_dpp_upper(U[c]), lo = el_dpp_lower(Lw[c]);
                            y[c] = fsc * y[c] + (kup * up + klo * lo) + fsg * corr[c];     // interior lanes: bitwise the plain y += up + lo
                            if (face_lo && a.mask_dirichlet) y[c] = 0.0;
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; ++c) y[c] += el_dpp_upper(U[c]) + el_dpp_lower(Lw[c]);
                    }
                    if (r <= rows_out && out_x) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            double* dst = a.out + c * a.stride + obase + (long long)r * g.PX;
                            if constexpr (NTS) __builtin_nontemporal_store(y[c], dst);      // the output is not read again by this launch
                            else *dst = y[c];
                            acc += P0[r][c] * y[c];
                        }
                        if (qmq) {                            // rows and planes interior: the class is that of the lane's x-position for the whole
#pragma unroll                                                // march, so the lane sums y_c^2 per component and applies D^-1 once, after the loop
                            for (int c = 0; c < 3; ++c) S2[c] = fma(y[c], y[c], S2[c]);
                        }
                    }
                }
                // y-face rows of this plane (wave-uniform branches): row 0 by the first row tile, row NY-1 by the last one
                if (ylo_wave) {
                    double y[3], U[3], Lw[3];
                    el_face_node<1, false>(w, face_g(1), P0[0], Pp[0], Pm[0], P0[1], Pp[1], y, U, Lw);
                    face_finish(obase, P0[0], y, U, Lw, 1 + 0 + 9);
                }
                if (r_hi > 0) {
#pragma unroll
                    for (int r = 2; r < RH; ++r)
                        if (r == r_hi) {
                            double y[3], U[3], Lw[3];
                            el_face_node<1, true>(w, face_g(1), P0[r], Pp[r], Pm[r], P0[r - 1], Pm[r - 1], y, U, Lw);
                            face_finish(obase + (long long)r * g.PX, P0[r], y, U, Lw, 1 + 6 + 9);
                        }
                }
            };
            if (fg.trace) t_pro = wall_clock64();
            load_plane(lk0 - 1, PA);
            load_plane(lk0, PB);
            load_plane(lk0 + 1, PC);                          // the address is inside the allocation even when lk0 + 1 == lk1 == nkl
            if (fg.trace) {                                   // (trace only: when the three planes of the first step have landed)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                t_planes = wall_clock64();
            }
            if constexpr (ZF)
                if (zlo_chunk) zface_lo(lk0 - 1, PA, PB);
            for (int lk = lk0; lk < lk1; lk += 4) {
                step(lk, PA, PB, PC, PD);
                if (lk + 1 >= lk1) break;
                step(lk + 1, PB, PC, PD, PA);
                if (lk + 2 >= lk1) break;
                step(lk + 2, PC, PD, PA, PB);
                if (lk + 3 >= lk1) break;
                step(lk + 3, PD, PA, PB, PC);
            }
            if (fg.trace) t_loop = wall_clock64();
            if (qmq) {
                const double* dv = dinv_of(cls_x + 3 + 9);
#pragma unroll
                for (int c = 0; c < 3; ++c) acc2 = fma(dv[c], S2[c], acc2);
            }
            if constexpr (ZF) if (zhi_chunk) {
                // the plane NZ-1 and its inward neighbour are read once more (cache hits) rather than taken from the rotating buffers: which
                // two they are depends on the chunk length, and four exit blocks with four buffer pairs made the compiler carry copies of
                // two planes through the loop (+70 VGPRs, a third of the occupancy).  This chunk is dispatched first (see zc above).
                int lkf = lk1;
                asm volatile("" : "+s"(lkf));                 // (keeps these loads below the loop)
                load_plane(lkf, PA);
                load_plane(lkf - 1, PB);
                zface_hi(lkf, PA, PB);
            }
        }
    }
    unsigned long long t_red = 0;
    if (fg.trace) t_red = wall_clock64();
    const double tot = block_sum_256(acc, sRed);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
    if (qmq) {
        const double t2 = block_sum_256(acc2, sRed);
        if (threadIdx.x == 0) a.partials_qmq[blockIdx.x] = t2;
    }
    if (fg.trace && threadIdx.x == 0) {
        unsigned long long* tr = fg.trace + 8 * blockIdx.x;
        tr[0] = t_start;
        tr[1] = wall_clock64();
        // HW_REG_XCC_ID (id 20) bits [3:0]; HW_REG_HW_ID (id 4): CU id bits [11:8], SH [12], SE [15:13]
        tr[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((15 << 11) | 4);
        tr[3] = t_pro;         // entry .. here: kernel arguments, weights, table fill and its barrier, index arithmetic
        tr[4] = t_planes;      // .. the three planes of the first step have landed
        tr[5] = t_loop;        // .. the z-loop is done
        tr[6] = t_red;         // .. (the plane NZ-1 of the last chunk) the block sums begin
        tr[7] = 0;
    }
}

// 16-byte accesses of the flat PCG kernels; NT = non-temporal (compile-time: HBM-resident grids only — on config 5 each kernel reads what the
// previous one left in L2 / MALL and the hint costs 3-5 us per iteration)
typedef double el_dbl2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 el_ld2(const double* p) {
    if constexpr (NT) {
        const el_dbl2 t = __builtin_nontemporal_load(reinterpret_cast<const el_dbl2*>(p));
        return make_double2(t.x, t.y);
    } else {
        return *reinterpret_cast<const double2*>(p);
    }
}
template <bool NT>
__device__ __forceinline__ void el_st2(double* p, const double2& v) {
    if constexpr (NT) {
        const el_dbl2 t = {v.x, v.y};
        __builtin_nontemporal_store(t, reinterpret_cast<el_dbl2*>(p));
    } else {
        *reinterpret_cast<double2*>(p) = v;
    }
}

// r -= alpha q;  partial r.z (z = D^-1 r, not stored) and r.r     (use_alpha == 0: alpha = 0, the sums only)
template <int D, bool NT>
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
            double2 rv = el_ld2<NT>(r + c * a.stride + idx);
            if (use_alpha) {
                const double2 qv = el_ld2<NT>(q + c * a.stride + idx);
                rv.x -= alpha * qv.x;
                rv.y -= alpha * qv.y;
                el_st2<NT>(r + c * a.stride + idx, rv);
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
// Two index ranges in one launch: workgroups [0, nb1) take [first, first + total), the rest [first2, first2 + total2) — the two ghost planes of a
// slab, updated after the halo exchange has completed (nb1 = gridDim.x and an empty second range otherwise).
template <int D, bool NT>
__global__ __launch_bounds__(256) void k_el_pupdate(ElArgs a, long long first, long long total, long long first2, long long total2, int nb1) {
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
    const bool second = (int)blockIdx.x >= nb1;               // workgroup-uniform
    if (second) {
        first = first2;
        total = total2;
    }
    const int bid = second ? (int)blockIdx.x - nb1 : (int)blockIdx.x, nbl = second ? (int)gridDim.x - nb1 : nb1;
    const long long pairs = total / 2;
    const int PX2 = g.PX / 2;
    const long long step = (long long)nbl * 256;
    const int di = (int)(step % PX2);
    const long long drow = step / PX2;
    const int dj = (int)(drow % g.NY), dk = (int)(drow / g.NY);
    long long e = (long long)bid * 256 + threadIdx.x;
    // (first is a multiple of PP, possibly negative: the lower ghost plane)
    int i2 = (int)(e % PX2), j = (int)((e / PX2) % g.NY), lk = (int)((e / PX2) / g.NY) + (int)(first / g.PP);
    for (; e < pairs; e += step) {
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
            const double2 po = el_ld2<NT>(a.p_old + c * a.stride + idx);
            const double2 rv = el_ld2<NT>(a.in + c * a.stride + idx);
            const double2 zv = make_double2(sDinv[cls0 * 3 + c] * rv.x, sDinv[cls1 * 3 + c] * rv.y);
            // (p_n is read by the block apply that follows: a plain store even with NT)
            *reinterpret_cast<double2*>(a.p_new + c * a.stride + idx) = make_double2(zv.x + beta * po.x, zv.y + beta * po.y);
            if (alpha_prev != 0.0) {
                double2 xv = el_ld2<NT>(a.x + c * a.stride + idx);
                xv.x += alpha_prev * po.x;
                xv.y += alpha_prev * po.y;
                el_st2<NT>(a.x + c * a.stride + idx, xv);
            }
        }
    }
}

// Single-reduction PCG, the fused update pass (K2 + p-update in one sweep):
//   alpha = r.z / p.Ap,  beta = (alpha^2 (Ap . D^-1 Ap) - r.z) / r.z   from the block of the iteration before ({p.Ap, r.z, r.r, Ap.D^-1 Ap}: read from the
//   scalars, or — single-slab loops — summed here from the partials of the apply (din3 / din4) and of the previous update pass (din / din2));
//   r -= alpha q;  z = D^-1 r;  x += alpha p_old;  p_new = z + beta p_old;  partials of r.z and r.r.
// The recurrence |r_new|_D^2 = alpha^2 |Ap|_D^2 - r.z (exact arithmetic: q.z = p.Ap) predicts the next r.z for beta only; the true r.z summed here joins
// the next block and forms the next alpha.  7 vector passes (read r, q, p_old, x; write r, p_new, x) where K2 + p-update make 8, one launch less, and
// ONE scalar reduction per iteration.  ghost != 0: only p_new = D^-1 r + beta p_old on up to two index ranges (the ghost planes of a slab, whose r has
// just arrived from the neighbour), no sums.
// xmode (el_xdefer): the x update of every other pass is left out and made up for by the next one — 6 instead of 7 vector passes on average.
//   1: this pass may SKIP x += alpha p_old (it does when its beta is not small) and records {skipped, alpha, beta} in the scalars;
//   2: the pass right after a mode-1 pass: if that one skipped, x += alpha p_old + alpha' p_old' with the previous pass's p_old' = (p_old - z) / beta'
//      (p_old = z + beta' p_old' was formed there; z = D^-1 r of the r this pass reads, before its own update) — nothing more is read for it;
//   0: the plain pass.  The host never ends a chunk on a mode-1 pass, so nothing is pending when x or the scalars are looked at.
template <int D, bool NT>
__global__ __launch_bounds__(256) void k_el_update(ElArgs a, const double* __restrict__ q, double* __restrict__ r, int ghost, long long first,
                                                    long long total, long long first2, long long total2, int nb1, int xmode) {
    __shared__ double sRed4[16];
    __shared__ double sDinv[27 * 3];
    if (threadIdx.x < 81) sDinv[threadIdx.x] = a.dinv[threadIdx.x];
    // the first node pair of this thread (config 5: one or two pairs per thread) is requested BEFORE the scalars of the iteration are summed: p_old, r and q
    // do not depend on them (x does: the deferred form may skip it), so three of the pass's four read streams are in flight during the prologue
    const bool second = (int)blockIdx.x >= nb1;               // workgroup-uniform
    if (second) {
        first = first2;
        total = total2;
    }
    const int bid = second ? (int)blockIdx.x - nb1 : (int)blockIdx.x, nbl = second ? (int)gridDim.x - nb1 : nb1;
    const long long pairs = total / 2;
    long long e = (long long)bid * 256 + threadIdx.x;
    double2 ppo[D], prv[D], pqv[D];
    if (e < pairs) {
        const long long idx = first + 2 * e;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            ppo[c] = el_ld2<NT>(a.p_old + c * a.stride + idx);
            prv[c] = el_ld2<NT>(r + c * a.stride + idx);
            pqv[c] = ghost ? make_double2(0.0, 0.0) : el_ld2<NT>(q + c * a.stride + idx);
        }
    }
    __syncthreads();
    double pq, qmq, rz;
    if (a.din) {
        double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;       // all loads issued before the first barrier
        for (int e = threadIdx.x; e < a.din_n; e += 256) {
            s1 += a.din[e];
            s2 += a.din2[e];
        }
        for (int e = threadIdx.x; e < a.din_n2; e += 256) {
            s3 += a.din3[e];
            s4 += a.din4[e];
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        s3 = wave_sum(s3);
        s4 = wave_sum(s4);
        if ((threadIdx.x & 63) == 0) {
            const int w = threadIdx.x >> 6;
            sRed4[w] = s1;
            sRed4[4 + w] = s2;
            sRed4[8 + w] = s3;
            sRed4[12 + w] = s4;
        }
        __syncthreads();
        rz = (sRed4[0] + sRed4[1]) + (sRed4[2] + sRed4[3]);
        const double rr = (sRed4[4] + sRed4[5]) + (sRed4[6] + sRed4[7]);
        pq = (sRed4[8] + sRed4[9]) + (sRed4[10] + sRed4[11]);
        qmq = (sRed4[12] + sRed4[13]) + (sRed4[14] + sRed4[15]);
        if (blockIdx.x == 0 && threadIdx.x == 0) {            // for the host's polls; no workgroup of this launch reads the scalars
            double* sc = const_cast<double*>(a.scal);
            sc[ESC_PQ] = pq;
            sc[ESC_RZ] = rz;
            sc[ESC_RR] = rr;
            sc[ESC_QMQ] = qmq;
        }
    } else {
        pq = a.scal[ESC_PQ];
        rz = a.scal[ESC_RZ];
        qmq = a.scal[ESC_QMQ];
    }
    const double alpha = (pq > 0.0 && rz > 0.0) ? rz / pq : 0.0;                  // 0 on the first iteration (p.Ap = 0)
    const double rz_next = alpha * alpha * qmq - rz;
    const double beta = (alpha > 0.0 && rz_next > 0.0) ? rz_next / rz : 0.0;      // a lost prediction restarts with the preconditioned residual
    // deferred x update (kernel-uniform): x += cx_p p_old - cx_z D^-1 r_in
    const bool skip_x = xmode == 1 && alpha != 0.0 && beta > 1e-2;
    const bool dbl = xmode == 2 && a.scal[ESC_XD_FLAG] != 0.0;
    double cx_p = alpha, cx_z = 0.0;
    if (dbl) {
        cx_z = a.scal[ESC_XD_ALPHA] / a.scal[ESC_XD_BETA];
        cx_p = alpha + cx_z;
    }
    if (xmode == 1 && blockIdx.x == 0 && threadIdx.x == 0) {      // (slots no workgroup of this launch reads)
        double* sc = const_cast<double*>(a.scal);
        sc[ESC_XD_FLAG] = skip_x ? 1.0 : 0.0;
        sc[ESC_XD_ALPHA] = alpha;
        sc[ESC_XD_BETA] = beta;
    }
    const GridDev& g = a.g;
    const int PX2 = g.PX / 2;
    const long long step = (long long)nbl * 256;
    const int di = (int)(step % PX2);
    const long long drow = step / PX2;
    const int dj = (int)(drow % g.NY), dk = (int)(drow / g.NY);
    bool pre = true;                                          // the first pair's p_old / r / q are already on their way
    int i2 = (int)(e % PX2), j = (int)((e / PX2) % g.NY), lk = (int)((e / PX2) / g.NY) + (int)(first / g.PP);
    double acc_rz = 0.0, acc_rr = 0.0;
    for (; e < pairs; e += step) {
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
        // every load of the node pair (4 vectors x D components) is issued before the first store: in program order a load behind a store waits for it
        // (stores count in the in-order vmcnt, and the vectors may alias as far as the compiler knows), which made the D components 2 D dependent round trips
        const bool upd = !ghost && alpha != 0.0;              // kernel-uniform
        const bool xdo = !ghost && ((upd && !skip_x) || dbl);
        double2 po[D], rv[D], qv[D], xv[D];
        if (pre) {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                po[c] = ppo[c];
                rv[c] = prv[c];
                qv[c] = upd ? pqv[c] : make_double2(0.0, 0.0);
                xv[c] = make_double2(0.0, 0.0);
                if (xdo) xv[c] = el_ld2<NT>(a.x + c * a.stride + idx);
            }
            pre = false;
        } else {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                po[c] = el_ld2<NT>(a.p_old + c * a.stride + idx);
                rv[c] = el_ld2<NT>(r + c * a.stride + idx);
                qv[c] = make_double2(0.0, 0.0);
                xv[c] = make_double2(0.0, 0.0);
                if (upd) qv[c] = el_ld2<NT>(q + c * a.stride + idx);
                if (xdo) xv[c] = el_ld2<NT>(a.x + c * a.stride + idx);
            }
        }
#pragma unroll
        for (int c = 0; c < D; ++c) {
            if (xdo) {
                // (cx_z = 0 unless this pass makes up for the one before: then D^-1 r of the r just read is the z that pass formed p_old from)
                xv[c].x += cx_p * po[c].x - cx_z * (sDinv[cls0 * 3 + c] * rv[c].x);
                xv[c].y += cx_p * po[c].y - cx_z * (sDinv[cls1 * 3 + c] * rv[c].y);
                el_st2<NT>(a.x + c * a.stride + idx, xv[c]);
            }
            if (upd) {
                rv[c].x -= alpha * qv[c].x;
                rv[c].y -= alpha * qv[c].y;
                el_st2<NT>(r + c * a.stride + idx, rv[c]);
            }
            const double2 zv = make_double2(sDinv[cls0 * 3 + c] * rv[c].x, sDinv[cls1 * 3 + c] * rv[c].y);
            acc_rz += rv[c].x * zv.x + rv[c].y * zv.y;
            acc_rr += rv[c].x * rv[c].x + rv[c].y * rv[c].y;
            // (p_n is read by the block apply that follows: a plain store even with NT)
            *reinterpret_cast<double2*>(a.p_new + c * a.stride + idx) = make_double2(zv.x + beta * po[c].x, zv.y + beta * po[c].y);
        }
    }
    if (!ghost) {
        const double t1 = block_sum_256(acc_rz, sRed4);
        const double t2 = block_sum_256(acc_rr, sRed4);
        if (threadIdx.x == 0) {
            a.partials[blockIdx.x] = t1;
            a.partials2[blockIdx.x] = t2;
        }
    }
}

// single-reduction PCG: the block {p.Ap, r.z, r.r, Ap.D^-1 Ap} from the apply's partials (nA each) and the update pass's (nU each), one launch
__global__ __launch_bounds__(256) void k_el_reduce4(const double* __restrict__ pq, const double* __restrict__ qmq, int nA, const double* __restrict__ rz,
                                                     const double* __restrict__ rr, int nU, double* __restrict__ scal) {
    __shared__ double sRed[4];
    double a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
    for (int e = threadIdx.x; e < nA; e += 256) {
        a1 += pq[e];
        a2 += qmq[e];
    }
    for (int e = threadIdx.x; e < nU; e += 256) {
        a3 += rz[e];
        a4 += rr[e];
    }
    const double t1 = block_sum_256(a1, sRed), t2 = block_sum_256(a2, sRed), t3 = block_sum_256(a3, sRed), t4 = block_sum_256(a4, sRed);
    if (threadIdx.x == 0) {
        scal[ESC_PQ] = t1;
        scal[ESC_QMQ] = t2;
        scal[ESC_RZ] = t3;
        scal[ESC_RR] = t4;
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

// el_xdefer: the x += alpha p_old an update pass left out (recorded in the scalars), applied on its own — where the loop stops between two passes
template <int D>
__global__ __launch_bounds__(256) void k_el_xflush(ElArgs a, const double* __restrict__ p) {
    if (a.scal[ESC_XD_FLAG] == 0.0) return;
    const double alpha = a.scal[ESC_XD_ALPHA];
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
    ElW18 w18{};               // what k_el_fast is launched with (w36 in 18 numbers + the faces bits)
    ElFastGeom fgeom{};
    bool xd_pending = false;        // el_xdefer: the latest update pass may have left its x += alpha p out (the scalars say whether it did)
    unsigned* d_mask = nullptr;
    int last_apply_blocks = 1;      // workgroups (= partials) of the most recent apply launch
    int k1_parts = 0;               // workgroups (partials) of the parts of a split K1 so far
    size_t npart3 = 0;              // doubles in part3
    // single-reduction PCG (variant 1): part4 = the apply's Ap.D^-1 Ap partials; the update pass writes its r.z / r.r partials alternately to
    // (part, part2) and (partB, partB2) — in a single-slab loop its own prologue still reads the previous pair
    int variant = 0;
    double *part4 = nullptr, *partB = nullptr, *partB2 = nullptr;
    int ucur = 0;                   // which pair the NEXT update pass writes
    bool sr_mem = true;             // the next update pass reads the block from the scalars (first pass of a solve / of a chunk; slab loop: always)

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
        if (part4) (void)hipFree(part4);
        if (partB) (void)hipFree(partB);
        if (partB2) (void)hipFree(partB2);
        if (d_mask) (void)hipFree(d_mask);
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

// the 18-number form of the interior weights (see ElW18); false if the 36 do not have that structure
bool el_pack_weights(const ElW36& w, ElW18* k) {
    double scale = 0.0;
    for (int t = 0; t < 6; ++t) scale = std::max(scale, std::fabs(w.c[t]));
    const double tol = 1e-12 * scale;
    auto eq = [&](double x, double y) { return std::fabs(x - y) <= tol; };
    const double a = w.ax[0][3], b = w.ax[0][4], c = w.dg[0][2];
    bool ok = eq(w.ax[0][1], w.ax[0][2]) && eq(w.ax[1][0], w.ax[1][2]) && eq(w.ax[2][0], w.ax[2][1]);
    ok = ok && eq(w.c[3], -2 * a) && eq(w.c[4], -2 * b) && eq(w.c[5], -4 * c);
    ok = ok && eq(w.ax[0][5], -c) && eq(w.ax[1][3], a) && eq(w.ax[1][4], -0.5 * b) && eq(w.ax[1][5], 2 * c);
    ok = ok && eq(w.ax[2][3], -0.5 * a) && eq(w.ax[2][4], b) && eq(w.ax[2][5], 2 * c);
    ok = ok && eq(w.dg[0][0], -a) && eq(w.dg[0][1], 0.5 * b);
    ok = ok && eq(w.dg[1][0], 0.5 * a) && eq(w.dg[1][1], -b) && eq(w.dg[1][2], c);
    ok = ok && eq(w.dg[2][0], 0.5 * a) && eq(w.dg[2][1], 0.5 * b) && eq(w.dg[2][2], -2 * c);
    ok = ok && eq(w.dg[3][0], -0.5 * a) && eq(w.dg[3][1], -0.5 * b) && eq(w.dg[3][2], -c);
    if (!ok) return false;
    for (int t = 0; t < 3; ++t) k->dc[t] = w.c[t];
    k->dp[0] = w.ax[0][0]; k->dp[1] = w.ax[0][1];
    k->dq[0] = w.ax[1][1]; k->dq[1] = w.ax[1][0];
    k->dr[0] = w.ax[2][2]; k->dr[1] = w.ax[2][0];
    k->a[0] = 0.5 * a; k->a[1] = a; k->a[2] = 2 * a;
    k->b[0] = 0.5 * b; k->b[1] = b; k->b[2] = 2 * b;
    k->c[0] = c; k->c[1] = 2 * c; k->c[2] = 4 * c;
    k->faces = w.faces;
    return true;
}

// Face structure of the class table (see ElW36::faces): fills G[k] = g of the in-face offsets e_p, e_q, e_p + e_q (p < q the two in-face
// axes) and returns true when the two face classes of `axis` (interior in the other two directions) have exactly that form
bool el_face_weights(const double* tab /*[27][15][9]*/, int axis, double (*G)[3]) {
    auto qof = [](const int* d) {
        const int o = (d[0] + 1) + 3 * (d[1] + 1) + 9 * (d[2] + 1);
        for (int q = 0; q < PG_NOFF15; ++q)
            if (pg_off15(q) == o) return q;
        return -1;
    };
    const int pw[3] = {1, 3, 9};
    const int P = (axis + 1) % 3 < (axis + 2) % 3 ? (axis + 1) % 3 : (axis + 2) % 3, Q = 3 - axis - P;
    const double* ti = tab + 13 * PG_NOFF15 * 9;
    const int zero[3] = {0, 0, 0};
    const int qc = qof(zero);
    double scale = 0.0;
    for (int t = 0; t < 9; ++t) scale = std::max(scale, std::fabs(ti[qc * 9 + t]));
    const double tol = 1e-12 * scale;
    for (int k = 0; k < 3; ++k)
        for (int t = 0; t < 3; ++t) G[k][t] = 0.0;
    for (int fi = 0; fi < 2; ++fi) {                          // the high face first (g is read from its +o blocks), then the low face
        const int face = 1 - fi;                              // 0: low face (sign -1), 1: high face (sign +1)
        const double* tf = tab + (13 + (face ? pw[axis] : -pw[axis])) * PG_NOFF15 * 9;
        const double sg = face ? 1.0 : -1.0;
        for (int q = 0; q < PG_NOFF15; ++q) {
            const int o = pg_off15(q);
            const int d[3] = {o % 3 - 1, (o / 3) % 3 - 1, o / 9 - 1};
            for (int t = 0; t < 9; ++t) {
                const double bf = tf[q * 9 + t], bi = ti[q * 9 + t];
                if (d[axis] == (face ? 1 : -1) && std::fabs(bf) > tol) return false;             // outward: absent
                if (d[axis] == (face ? -1 : 1) && std::fabs(bf - bi) > tol) return false;        // inward: as in the interior
            }
        }
        for (int t = 0; t < 9; ++t)
            if (std::fabs(tf[qc * 9 + t] - 0.5 * ti[qc * 9 + t]) > tol) return false;
        for (int k = 0; k < 3; ++k)
            for (int sgn = 1; sgn >= -1; sgn -= 2) {
                int d[3] = {0, 0, 0};
                if (k != 1) d[P] = sgn;
                if (k != 0) d[Q] = sgn;
                const int q = qof(d);
                if (q < 0) return false;
                double A[9];
                for (int t = 0; t < 9; ++t) A[t] = tf[q * 9 + t] - 0.5 * ti[q * 9 + t];
                // expected: sg * sgn * Ahigh(g),  Ahigh[axis][t] = g_t, Ahigh[t][axis] = -g_t
                if (face == 1 && sgn == 1)
                    for (int t = 0; t < 3; ++t) G[k][t] = t == axis ? 0.0 : A[axis * 3 + t];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) {
                        double expect = 0.0;
                        if (r == axis && c != axis) expect = sg * sgn * G[k][c];
                        if (c == axis && r != axis) expect = -sg * sgn * G[k][r];
                        if (std::fabs(A[r * 3 + c] - expect) > tol) return false;
                    }
            }
    }
    return true;
}

bool el_vec_nt(const ElSession* s);

// Launch geometry of the marching apply over the local planes [a, b) (and, in the same launch, [a2, b2): the two edge planes of a slab with
// neighbours); the whole slab is (0, nkl).  A range takes the slab's global end plane (z-face: every node a boundary node) along when it
// contains it.
int el_fast_wg_per_cu(const ElSession* s);
void el_fast_geometry_range(const ElSession* s, int a, int b, int a2, int b2, ElFastGeom* out) {
    const GridDev& g = s->g;
    ElFastGeom& fg = *out;
    const int RY = s->fast_ry;
    const bool two = b2 > a2;
    fg.full_lo = (g.k0 == 0 && a == 0) ? 1 : 0;
    fg.full_hi = (g.k0 + g.nkl == g.NZ && (two ? b2 : b) == g.nkl) ? 1 : 0;
    fg.lkb = a + fg.full_lo;
    fg.lke = b - ((fg.full_hi && !two) ? 1 : 0);
    fg.lkb2 = two ? a2 : 0;
    fg.lke2 = two ? b2 - fg.full_hi : 0;
    const int planes = std::max(0, fg.lke - fg.lkb), planes2 = std::max(0, fg.lke2 - fg.lkb2);
    const bool xface = s->w36.faces != 0;
    fg.ntx = (g.NX - (xface ? 1 : 2) + kElXt - 1) / kElXt;
    fg.nty = (g.NY - 2 + RY - 1) / RY;
    fg.nyg = (fg.nty + 3) / 4;
    const bool yx = s->ctx->el_fast_yx != 0;
    const long long cols = yx ? (long long)fg.ntx * fg.nyg * 4 : (long long)fg.ntx * fg.nty;      // wavefront columns launched
    // wavefront target: 4096 (1.3 resident rounds) on cache-resident grids; with the non-temporal stores of HBM-resident grids shorter chunks
    // win (512x128x128: 8 instead of 16 planes per chunk, 89 -> 85 us; 500x100x100: 58.5 -> 53.7 us)
    const long long waves = s->ctx->el_fast_waves > 0 ? s->ctx->el_fast_waves : (el_vec_nt(s) ? 8192 : 4096);
    long long nzc = (waves + cols - 1) / cols;
    nzc = std::max(1LL, std::min<long long>(nzc, std::max(1, planes)));
    fg.zlen = std::max(1, (int)((planes + nzc - 1) / nzc));
    if (planes == 0 && planes2 > 0) fg.zlen = planes2;
    fg.nzc1 = planes > 0 ? (planes + fg.zlen - 1) / fg.zlen : 0;
    fg.nzc = fg.nzc1 + (planes2 > 0 ? (planes2 + fg.zlen - 1) / fg.zlen : 0);
    fg.nb_fast = (int)((cols * fg.nzc + 3) / 4);
    fg.nb_fast = (fg.nb_fast + 7) / 8 * 8;          // multiple of 8 for the XCD-aware tile order (surplus waves find zc >= nzc and idle)
    fg.ring = 2 * g.NX + (xface ? 0 : 2 * (g.NY - 2));        // boundary nodes of an interior plane left to the class-table pass
    fg.nbnd = (long long)(fg.full_lo + fg.full_hi) * g.NX * g.NY + (long long)(planes + planes2) * fg.ring;
    if (s->w36.faces & 2) {                                   // y-faces folded: 4 corners per interior plane; the planes 0 / NZ-1: perimeter | whole
        fg.ring = 4;
        const long long per = (s->w36.faces & 4) ? 2LL * g.NX + 2LL * (g.NY - 2) : (long long)g.NX * g.NY;
        fg.nbnd = (long long)(fg.full_lo + fg.full_hi) * per + 4LL * (planes + planes2);
    }
    fg.nb_bnd = (int)std::min<long long>((fg.nbnd + 255) / 256, 2LL * s->ctx->num_cu);      // grid-stride: the 29 KB table is loaded per workgroup
    fg.bnd_gran = 256;
    if (s->ctx->el_bnd_gran == 64 && fg.nbnd <= 64LL * s->ctx->num_cu / 4) {                // few boundary nodes (edges and corners only): a wavefront's worth each
        fg.bnd_gran = 64;
        fg.nb_bnd = (int)((fg.nbnd + 63) / 64);
    }
    fg.nlong = fg.nzc1;
    fg.z0 = 0;
    // A launch between one and two resident rounds is refitted to ONE (el_fast_fit): as many z-chunks as fill the slots, of two lengths (zlen and
    // zlen - 1).  What the per-workgroup timeline shows at config 5 (tools/el_trace.py, profiles/r04_el_apply_timelines.log): every workgroup lives
    // 7 us + 1 us per plane whatever the chunk length, all 768 resident ones start inside the first microsecond, and the 78 of the 846 that do not
    // fit wait for a slot until t = 8-11 us — the kernel lasts two workgroup lifetimes for 9 % more work than fits one.
    if (s->ctx->el_fast_fit && yx && planes > 0 && planes2 == 0 && s->ctx->el_fast_waves <= 0) {
        const long long wgc = cols / 4, slots = (long long)el_fast_wg_per_cu(s) * s->ctx->num_cu;
        const long long nb_default = (wgc * fg.nzc1 + 7) / 8 * 8 + fg.nb_bnd;
        if (nb_default > slots && nb_default < 2 * slots) {
            long long nfit = (slots - fg.nb_bnd) / wgc;
            while (nfit >= 1 && (wgc * nfit + 7) / 8 * 8 + fg.nb_bnd > slots) --nfit;
            if (nfit >= 1 && nfit <= planes) {
                fg.zlen = (int)((planes + nfit - 1) / nfit);
                fg.nzc1 = fg.nzc = (int)nfit;
                fg.nlong = fg.zlen > 1 ? (int)(planes - nfit * (fg.zlen - 1)) : (int)nfit;
                // the first and the last chunk also compute the planes 0 / NZ-1 where the z-faces are inside the marching part: the short ones
                fg.z0 = (fg.nlong + 2 <= nfit) ? 1 : 0;
                fg.nb_fast = (int)((wgc * nfit + 7) / 8 * 8);
            }
        }
    }
    fg.bnd_last = s->ctx->el_bnd_last != 0 ? 1 : 0;
    fg.mask = s->d_mask;
    fg.gf = reinterpret_cast<const double*>(s->d_mask + 32);
    fg.trace = nullptr;
}
void el_fast_geometry(ElSession* s) { el_fast_geometry_range(s, 0, s->g.nkl, 0, 0, &s->fgeom); }

// non-temporal hints where the vectors are HBM-resident (one component set beyond 64 MiB, the heat kernels' threshold); el_fast_nt = 0 | 1 forces
bool el_vec_nt(const ElSession* s) {
    return s->ctx->el_fast_nt < 0 ? (long long)s->stride * 3 * 8 > (64LL << 20) : (s->ctx->el_fast_nt & 1) != 0;
}

// workgroups of the marching apply a CU holds at once (the 2-row y-hand-off instantiations; without a device — host-only callers — the 3 of its launch bounds)
int el_fast_wg_per_cu(const ElSession* s) {
    static int occ[2] = {0, 0};
    const int zf = (s->w36.faces & 4) ? 1 : 0;
    if (occ[zf] == 0) {
        int nb = 0;
        const hipError_t e = zf ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_el_fast<2, true, true, false>, 256, 0)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_el_fast<2, true, false, false>, 256, 0);
        if (e != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 3;
        }
        occ[zf] = nb;
    }
    return occ[zf];
}

// fgr != nullptr: a plane-range launch of the marching apply (el_fast_geometry_range)
int el_apply(ElSession* s, const ElArgs& a, int mode, const ElFastGeom* fgr = nullptr) {
    if (mode == APPLY_PLAIN && s->fast && s->ctx->el_fast) {
        const ElFastGeom& fg = fgr ? *fgr : s->fgeom;
        const int nb = fg.nb_fast + fg.nb_bnd;
        const bool yx = s->ctx->el_fast_yx != 0;
        // (ZF = with the code for the planes 0 / NZ-1: it costs 60 VGPRs, so slabs without such a plane and sessions that leave those
        // planes to the class-table pass run the lean instantiation)
        const bool zf = (s->w36.faces & 4) != 0;
        auto launch = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, s->ctx->stream, a, s->w18, fg); };
        // non-temporal stores where the vectors are HBM-resident (beyond 64 MiB per component set, the heat kernels' threshold): on cache-
        // resident grids the next kernel finds q in L2 / MALL and the hint costs time
        const bool nts = el_vec_nt(s);
        const int ry = s->fast_ry >= 2 ? 2 : 1;
        const int key = (ry == 2 ? 8 : 0) | (yx ? 4 : 0) | (zf ? 2 : 0) | (nts ? 1 : 0);
        {
            switch (key) {
#define EL_CASE1(K, RYv, YXv, ZFv, NTv) case K: launch(k_el_fast<RYv, YXv, ZFv, NTv>); break;
                EL_CASE1(0, 1, false, false, false) EL_CASE1(1, 1, false, false, true) EL_CASE1(2, 1, false, true, false) EL_CASE1(3, 1, false, true, true)
                EL_CASE1(4, 1, true, false, false) EL_CASE1(5, 1, true, false, true) EL_CASE1(6, 1, true, true, false) EL_CASE1(7, 1, true, true, true)
                EL_CASE1(8, 2, false, false, false) EL_CASE1(9, 2, false, false, true) EL_CASE1(10, 2, false, true, false) EL_CASE1(11, 2, false, true, true)
                EL_CASE1(12, 2, true, false, false) EL_CASE1(13, 2, true, false, true) EL_CASE1(14, 2, true, true, false) EL_CASE1(15, 2, true, true, true)
#undef EL_CASE1
            }
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
    if (s->D == 3) {
        if (el_vec_nt(s)) hipLaunchKernelGGL((k_el_k2<3, true>), dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, s->q, s->r, use_alpha);
        else hipLaunchKernelGGL((k_el_k2<3, false>), dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, s->q, s->r, use_alpha);
    } else {
        hipLaunchKernelGGL((k_el_k2<2, false>), dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, s->q, s->r, use_alpha);
    }
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
// The p-update covers the ghost planes too: p_n there comes from the exchanged r and the slab's own p_{n-1} of the ghost plane (same formula,
// same global scalars as the neighbour rank), so ONLY r travels in the halo exchange.
// part (slab loop, marching apply): -1 = the whole K1; 0 = everything that reads no ghost plane — the p-update of the owned planes and the apply
// on the planes away from the ghosts — while the r planes travel; 1 = p-update of the ghost planes, apply on the edge planes (ONE launch for
// both), p.q reduce.
bool el_k1_splittable(const ElSession* s) {
    const int lo = s->g.k0 > 0 ? 1 : 0, hi = (s->g.k0 + s->g.nkl < s->g.NZ) ? 1 : 0;
    // (two more launches than the unsplit K1, ~9 us per iteration on the per-rank surrogate — config 5: 26 -> 34 us at 8 planes per rank, 44 -> 53 us
    // at 33: it pays only where the apply on the planes away from the ghosts is long enough to hide a sizeable exchange, i.e. on big slabs)
    const double slab_mb = (double)s->g.nkl * (double)s->g.PP * 3.0 * 8.0 / (1 << 20);
    return s->D == 3 && s->fast && s->ctx->el_fast && (lo || hi) && s->g.nkl >= 4 && slab_mb >= (double)s->ctx->el_split_min_mb;
}
int el_pupdate_launch(ElSession* s, const ElArgs& a, long long first, long long total, long long first2 = 0, long long total2 = 0) {
    int nb1 = s->nblk, nb = s->nblk;
    if (total2 > 0) {                                          // two short ranges (ghost planes): workgroups in proportion, at least one each
        auto want = [&](long long t) { return (int)std::max<long long>(1, std::min<long long>(s->nblk, (t / 2 + 255) / 256)); };
        nb1 = want(total);
        nb = nb1 + want(total2);
    } else if (total < (long long)s->g.nkl * s->g.PP) {
        nb1 = nb = (int)std::max<long long>(1, std::min<long long>(s->nblk, (total / 2 + 255) / 256));
    }
    if (s->D == 3) {
        if (el_vec_nt(s)) hipLaunchKernelGGL((k_el_pupdate<3, true>), dim3(nb), dim3(256), 0, s->ctx->stream, a, first, total, first2, total2, nb1);
        else hipLaunchKernelGGL((k_el_pupdate<3, false>), dim3(nb), dim3(256), 0, s->ctx->stream, a, first, total, first2, total2, nb1);
    } else {
        hipLaunchKernelGGL((k_el_pupdate<2, false>), dim3(nb), dim3(256), 0, s->ctx->stream, a, first, total, first2, total2, nb1);
    }
    PG_HIP(s->ctx, hipGetLastError());
    return PDEGPU_OK;
}
int el_k1(ElSession* s, ElLazy* lazy = nullptr, int part = -1) {
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
    const GridDev& g = s->g;
    const int lo = g.k0 > 0 ? 1 : 0, hi = (g.k0 + g.nkl < g.NZ) ? 1 : 0;
    if (part >= 0 && !el_k1_splittable(s)) {                   // a driver may always call part 0 then part 1
        if (part == 0) return PDEGPU_OK;
        part = -1;
    }
    ElArgs b = s->args();
    b.out = s->q;
    if (part == 0) {
        PG_TRY(el_pupdate_launch(s, a, 0, (long long)g.nkl * g.PP));
        ElFastGeom fgr;
        el_fast_geometry_range(s, lo, g.nkl - hi, 0, 0, &fgr);
        b.in = s->p[1 - s->pcur];
        b.partials = s->part3;
        PG_TRY(el_apply(s, b, APPLY_PLAIN, &fgr));
        s->k1_parts = s->last_apply_blocks;
        return PDEGPU_OK;
    }
    if (part == 1) {
        const long long lo_first = -(long long)g.PP, hi_first = (long long)g.nkl * g.PP;
        if (lo && hi) PG_TRY(el_pupdate_launch(s, a, lo_first, g.PP, hi_first, g.PP));
        else PG_TRY(el_pupdate_launch(s, a, lo ? lo_first : hi_first, g.PP));
        s->pcur ^= 1;
        ElFastGeom fgr;
        if (lo && hi) el_fast_geometry_range(s, 0, 1, g.nkl - 1, g.nkl, &fgr);
        else if (lo) el_fast_geometry_range(s, 0, 1, 0, 0, &fgr);
        else el_fast_geometry_range(s, g.nkl - 1, g.nkl, 0, 0, &fgr);
        if ((size_t)s->k1_parts + fgr.nb_fast + fgr.nb_bnd > s->npart3)
            return pdegpu_set_err(s->ctx, PDEGPU_EUNSUPPORTED, "split K1: %d + %d workgroups exceed the %zu partials of the session", s->k1_parts,
                                  fgr.nb_fast + fgr.nb_bnd, s->npart3);
        b.in = s->p[s->pcur];
        b.partials = s->part3 + s->k1_parts;
        PG_TRY(el_apply(s, b, APPLY_PLAIN, &fgr));
        s->k1_parts += s->last_apply_blocks;
        hipLaunchKernelGGL(k_el_reduce, dim3(1), dim3(256), 0, s->ctx->stream, s->part3, (const double*)nullptr, s->k1_parts, s->scal, 0);
        PG_HIP(s->ctx, hipGetLastError());
        return PDEGPU_OK;
    }
    const long long first = -(long long)lo * g.PP, total = (long long)(g.nkl + lo + hi) * g.PP;
    PG_TRY(el_pupdate_launch(s, a, first, total));
    s->pcur ^= 1;
    b.in = s->p[s->pcur];
    b.partials = s->part3;
    PG_TRY(el_apply(s, b, APPLY_PLAIN));
    if (lazy) return PDEGPU_OK;
    hipLaunchKernelGGL(k_el_reduce, dim3(1), dim3(256), 0, s->ctx->stream, s->part3, (const double*)nullptr, s->last_apply_blocks, s->scal, 0);
    PG_HIP(s->ctx, hipGetLastError());
    return PDEGPU_OK;
}
// ---- single-reduction PCG (variant 1) ---------------------------------------------------------------------------------------------------
// update pass over the owned planes; lazy: sum the previous block from partials in the prologue (single-slab loops) unless the block has just
// been reduced into the scalars (sr_mem)
int el_sr_update(ElSession* s, bool lazy, int xmode = -1) {
    const GridDev& g = s->g;
    // xmode -1 (the stepwise protocol): the passes alternate by themselves — leave x out | make up for it; el_xflush covers a stop in between
    if (xmode < 0) xmode = !s->ctx->el_xdefer ? 0 : (s->xd_pending ? 2 : 1);
    s->xd_pending = xmode == 1;
    ElArgs a = s->args();
    a.p_old = s->p[s->pcur];
    a.p_new = s->p[1 - s->pcur];
    a.x = s->x;
    a.partials = s->ucur ? s->partB : s->part;
    a.partials2 = s->ucur ? s->partB2 : s->part2;
    if (lazy && !s->sr_mem) {
        a.din = s->ucur ? s->part : s->partB;
        a.din2 = s->ucur ? s->part2 : s->partB2;
        a.din_n = s->nblk;
        a.din3 = s->part3;
        a.din4 = s->part4;
        a.din_n2 = s->last_apply_blocks;
    }
    const long long total = (long long)g.nkl * g.PP;
    if (s->D == 3) {
        if (el_vec_nt(s)) hipLaunchKernelGGL((k_el_update<3, true>), dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, s->q, s->r, 0, 0LL, total, 0LL, 0LL, s->nblk, xmode);
        else hipLaunchKernelGGL((k_el_update<3, false>), dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, s->q, s->r, 0, 0LL, total, 0LL, 0LL, s->nblk, xmode);
    } else {
        hipLaunchKernelGGL((k_el_update<2, false>), dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, s->q, s->r, 0, 0LL, total, 0LL, 0LL, s->nblk, xmode);
    }
    PG_HIP(s->ctx, hipGetLastError());
    s->ucur ^= 1;
    s->sr_mem = false;
    return PDEGPU_OK;
}
// p_new = D^-1 r + beta p_old on the ghost planes (r has arrived from the neighbours; the same beta: the scalars still hold the block the update pass used)
int el_sr_ghost_update(ElSession* s) {
    const GridDev& g = s->g;
    const int lo = g.k0 > 0 ? 1 : 0, hi = (g.k0 + g.nkl < g.NZ) ? 1 : 0;
    if (!lo && !hi) return PDEGPU_OK;
    ElArgs a = s->args();
    a.p_old = s->p[s->pcur];
    a.p_new = s->p[1 - s->pcur];
    a.x = s->x;
    const long long lo_first = -(long long)g.PP, hi_first = (long long)g.nkl * g.PP;
    const long long f1 = lo ? lo_first : hi_first, f2 = (lo && hi) ? hi_first : 0, t2 = (lo && hi) ? (long long)g.PP : 0;
    const int nbp = (int)std::max<long long>(1, std::min<long long>(s->nblk, ((long long)g.PP / 2 + 255) / 256));
    const int nb1 = nbp, nb = t2 > 0 ? 2 * nbp : nbp;
    if (s->D == 3) hipLaunchKernelGGL((k_el_update<3, false>), dim3(nb), dim3(256), 0, s->ctx->stream, a, s->q, s->r, 1, f1, (long long)g.PP, f2, t2, nb1, 0);
    else hipLaunchKernelGGL((k_el_update<2, false>), dim3(nb), dim3(256), 0, s->ctx->stream, a, s->q, s->r, 1, f1, (long long)g.PP, f2, t2, nb1, 0);
    PG_HIP(s->ctx, hipGetLastError());
    return PDEGPU_OK;
}
// the block {p.Ap, r.z, r.r, Ap.D^-1 Ap} -> scalars (this slab's sums; the multi-GPU host all-reduces scalars[0..3], 32 bytes)
int el_sr_reduce(ElSession* s, int n_apply) {
    const double* urz = s->ucur ? s->part : s->partB;        // the pair the LAST update pass wrote (ucur has been flipped since)
    const double* urr = s->ucur ? s->part2 : s->partB2;
    hipLaunchKernelGGL(k_el_reduce4, dim3(1), dim3(256), 0, s->ctx->stream, s->part3, s->part4, n_apply, urz, urr, s->nblk, s->scal);
    PG_HIP(s->ctx, hipGetLastError());
    s->sr_mem = true;
    return PDEGPU_OK;
}
// One iteration = update pass + apply.  part: -1 whole (slabs without neighbours; lazy: no reduce launch), 0 = the update pass of the owned planes
// (the host then posts the exchange of r), 1 = the apply on the planes away from the ghosts (overlaps the exchange), 2 = the ghost planes'
// p-update, the remaining apply and the block reduce.
int el_sr_k1(ElSession* s, bool lazy, int part, int xmode = -1) {
    const GridDev& g = s->g;
    const int lo = g.k0 > 0 ? 1 : 0, hi = (g.k0 + g.nkl < g.NZ) ? 1 : 0;
    ElArgs b = s->args();
    b.out = s->q;
    b.partials_qmq = s->part4;
    if (part < 0) {
        if (lo || hi) return pdegpu_set_err(s->ctx, PDEGPU_EINVAL, "single-reduction PCG on a slab with neighbours: the residual update is part of k1, "
                                                                    "so r travels INSIDE it — call k1_pre, exchange r, then k1_part(0) and k1_part(1)");
        PG_TRY(el_sr_update(s, lazy, xmode));
        s->pcur ^= 1;
        b.in = s->p[s->pcur];
        b.partials = s->part3;
        PG_TRY(el_apply(s, b, APPLY_PLAIN));
        if (lazy) return PDEGPU_OK;
        return el_sr_reduce(s, s->last_apply_blocks);
    }
    if (part == 0) return el_sr_update(s, false, xmode);
    const bool split = el_k1_splittable(s);
    if (part == 1) {
        s->k1_parts = 0;
        if (!split) return PDEGPU_OK;
        ElFastGeom fgr;
        el_fast_geometry_range(s, lo, g.nkl - hi, 0, 0, &fgr);
        b.in = s->p[1 - s->pcur];
        b.partials = s->part3;
        PG_TRY(el_apply(s, b, APPLY_PLAIN, &fgr));
        s->k1_parts = s->last_apply_blocks;
        return PDEGPU_OK;
    }
    PG_TRY(el_sr_ghost_update(s));
    s->pcur ^= 1;
    b.in = s->p[s->pcur];
    if (split) {
        ElFastGeom fgr;
        if (lo && hi) el_fast_geometry_range(s, 0, 1, g.nkl - 1, g.nkl, &fgr);
        else if (lo) el_fast_geometry_range(s, 0, 1, 0, 0, &fgr);
        else el_fast_geometry_range(s, g.nkl - 1, g.nkl, 0, 0, &fgr);
        if ((size_t)s->k1_parts + fgr.nb_fast + fgr.nb_bnd > s->npart3)
            return pdegpu_set_err(s->ctx, PDEGPU_EUNSUPPORTED, "split K1: partials of the session exceeded");
        b.partials = s->part3 + s->k1_parts;
        b.partials_qmq = s->part4 + s->k1_parts;
        PG_TRY(el_apply(s, b, APPLY_PLAIN, &fgr));
        s->k1_parts += s->last_apply_blocks;
    } else {
        b.partials = s->part3;
        PG_TRY(el_apply(s, b, APPLY_PLAIN));
        s->k1_parts = s->last_apply_blocks;
    }
    return el_sr_reduce(s, s->k1_parts);
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
    PG_HIP(ctx, hipMalloc(&s->d_mask, 32 * sizeof(unsigned) + 32 * sizeof(double)));      // + the 27 face remainder numbers
    PG_HIP(ctx, hipHostMalloc(&s->h_pin, ESC_COUNT * sizeof(double)));
    PG_HIP(ctx, hipMemsetAsync(s->scal, 0, ESC_COUNT * sizeof(double), ctx->stream));
    double h[3] = {1, 1, 1};
    for (int r = 0; r < grid->dim; ++r) h[r] = grid->L[r] / grid->n[r];
    hipLaunchKernelGGL(k_el_tables, dim3(1), dim3(256), 0, ctx->stream, grid->dim, h[0], h[1], h[2], s->lam, s->mu, s->tab, s->dinv, s->d_mask);
    PG_HIP(ctx, hipGetLastError());
    for (int q = 0; q < PG_NOFF15; ++q) fold_elasticity_block(grid->dim, h, s->lam, s->mu, 13, q, s->interior.w + q * 9);
    size_t npart = (size_t)s->nblk;
    s->fast_ry = ctx->el_fast_ry >= 2 ? 2 : 1;          // (4 rows per wave were measured, never faster, and are gone)
    s->fast = grid->dim == 3 && s->g.NX >= 3 && s->g.NY >= 3 && s->g.NZ >= 3 && el_fast_weights(s->interior.w, &s->w36);
    s->w36.faces = 0;
    if (s->fast && ctx->el_fast_xface) {
        // x-face rows inside the marching kernel: only when the host table has the structure the kernel relies on
        std::vector<double> htab((size_t)27 * PG_NOFF15 * 9);
        for (int e = 0; e < 27 * PG_NOFF15; ++e) fold_elasticity_block(grid->dim, h, s->lam, s->mu, e / PG_NOFF15, e % PG_NOFF15, htab.data() + (size_t)e * 9);
        double gf[3][3][3] = {};
        s->w36.faces = el_face_weights(htab.data(), 0, gf[0]) ? 1 : 0;
        // y- and z-faces as well: they are natural faces here (only x = 0 is clamped); a slab without an interior plane has no
        // marching chunk to attach its end planes to
        const bool yz_natural = s->g.bc_mode == PDEGPU_BC_DIRECTIONAL && !s->g.bc_present[2];
        if (s->w36.faces && ctx->el_fast_yzface && yz_natural && el_face_weights(htab.data(), 1, gf[1])) {
            const bool has_lo = s->g.k0 == 0, has_hi = s->g.k0 + s->g.nkl == s->g.NZ;
            const int planes = s->g.nkl - (has_hi ? 1 : 0) - (has_lo ? 1 : 0);
            s->w36.faces |= 2;
            if ((ctx->el_fast_yzface & 2) && (has_lo || has_hi) && planes >= 1 && el_face_weights(htab.data(), 2, gf[2])) s->w36.faces |= 4;
        }
        PG_HIP(ctx, hipMemcpyAsync(s->d_mask + 32, gf, sizeof(gf), hipMemcpyHostToDevice, ctx->stream));
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));      // gf is a local
    }
    if (s->fast && !el_pack_weights(s->w36, &s->w18)) s->fast = false;
    if (s->fast) {
        el_fast_geometry(s);
        if (s->fgeom.nbnd > 0x7fffffffLL) s->fast = false;
        npart = std::max(npart, (size_t)s->fgeom.nb_fast + (size_t)std::max<long long>((s->fgeom.nbnd + 255) / 256, s->fgeom.nb_bnd));
        // a split K1 (slab with neighbours) appends the partials of its two launches
        const int lo = s->g.k0 > 0 ? 1 : 0, hi = (s->g.k0 + s->g.nkl < s->g.NZ) ? 1 : 0;
        if ((lo || hi) && s->g.nkl >= 4) {
            ElFastGeom f0, f1;
            el_fast_geometry_range(s, lo, s->g.nkl - hi, 0, 0, &f0);
            if (lo && hi) el_fast_geometry_range(s, 0, 1, s->g.nkl - 1, s->g.nkl, &f1);
            else if (lo) el_fast_geometry_range(s, 0, 1, 0, 0, &f1);
            else el_fast_geometry_range(s, s->g.nkl - 1, s->g.nkl, 0, 0, &f1);
            npart = std::max(npart, (size_t)f0.nb_fast + f0.nb_bnd + f1.nb_fast + f1.nb_bnd);
        }
    }
    s->npart3 = npart;
    PG_HIP(ctx, hipMalloc(&s->part, npart * sizeof(double)));
    PG_HIP(ctx, hipMalloc(&s->part2, npart * sizeof(double)));
    PG_HIP(ctx, hipMalloc(&s->part3, npart * sizeof(double)));
    PG_HIP(ctx, hipMalloc(&s->part4, npart * sizeof(double)));
    PG_HIP(ctx, hipMalloc(&s->partB, npart * sizeof(double)));
    PG_HIP(ctx, hipMalloc(&s->partB2, npart * sizeof(double)));
    s->variant = ctx->el_cg_variant != 0 ? 1 : 0;
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
    s->ucur = 0;
    s->sr_mem = true;
    s->xd_pending = false;
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

// single-reduction form with el_xdefer: apply the x update the latest pass left out, if it did (the device flag decides)
int el_xflush(ElSession* s) {
    if (!s->xd_pending) return PDEGPU_OK;
    ElArgs a = s->args();
    a.x = s->x;
    const double* p_old = s->p[1 - s->pcur];                  // (the pass has flipped the buffers: its p_old is the other one now)
    if (s->D == 3) hipLaunchKernelGGL(k_el_xflush<3>, dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, p_old);
    else hipLaunchKernelGGL(k_el_xflush<2>, dim3(s->nblk), dim3(256), 0, s->ctx->stream, a, p_old);
    PG_HIP(s->ctx, hipGetLastError());
    PG_HIP(s->ctx, hipMemsetAsync(s->scal + ESC_XD_FLAG, 0, sizeof(double), s->ctx->stream));
    s->xd_pending = false;
    return PDEGPU_OK;
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
        if (s.variant) {
            // single-reduction PCG: update pass (K2 and p-update fused) + block apply; with lazy_dot the update pass sums the previous block in its
            // prologue and one reduce launch closes the chunk for the poll
            const bool lz = ctx->lazy_dot != 0;
            // el_xdefer: the passes of a chunk go in pairs (may skip the x update | makes up for it); an odd last pass is a plain one, so that
            // nothing is pending when the chunk's poll — or the caller — looks
            for (int i = 0; i < n; ++i) {
                const int xmode = !ctx->el_xdefer ? 0 : ((i & 1) ? 2 : (i + 1 < n ? 1 : 0));
                PG_TRY(el_sr_k1(&s, lz, -1, xmode));
            }
            if (lz) PG_TRY(el_sr_reduce(&s, s.last_apply_blocks));
        } else {
            ElLazy lazy;
            ElLazy* lz = ctx->lazy_dot ? &lazy : nullptr;
            for (int i = 0; i < n; ++i) {
                PG_TRY(el_k1(&s, lz));
                PG_TRY(el_k2(&s, 1, lz));
            }
            if (lz) PG_TRY(el_lazy_finish(&s, lz));
        }
        it += n;
        const double rr_before = rr;
        PG_TRY(el_scalars(&s, sc));
        rr = sc[ESC_RR];
        chunk = next_poll_chunk(rr_before, rr, target, n, chunk, 256);
    }
    if (!s.variant) PG_TRY(el_flush(&s));      // (single-reduction form: x is the iterate whose residual the last poll read; nothing is pending)
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
    if (use_mfma == 1 && grid->dim != 3) return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "the MFMA element-by-element apply is 3-D only");
    if (use_mfma < 0 || use_mfma > 2) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "form: 0 vector block stencil, 1 MFMA, 2 the PCG loop's masked launch");
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
    if (rc == PDEGPU_OK && use_mfma == 1) {
        double* d_Ke = nullptr;
        rc = upload_macro_element(&s, grid, &d_Ke);
        if (rc == PDEGPU_OK) rc = el_apply_mfma(&s, s.x, s.q, d_Ke);
        if (rc == PDEGPU_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "sync failed");
        if (d_Ke) (void)hipFree(d_Ke);
    } else if (rc == PDEGPU_OK) {
        ElArgs a = s.args();
        a.in = s.x;
        a.out = s.q;
        a.mask_dirichlet = use_mfma == 2 ? 1 : 0;      // 2: the launch of the PCG loop — rows of the clamped face give 0
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
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        if (d_Ke) (void)hipFree(d_Ke);
        return pdegpu_set_err(ctx, PDEGPU_EHIP, "hipEventCreate failed");
    }
    int rc = PDEGPU_OK;
    for (int i = -1; i < reps && rc == PDEGPU_OK; ++i) {          // (reps >= 1 is checked above: e0 is always recorded)
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
    // tuning aid: one more launch of the marching apply with per-workgroup timestamps, written as text to $PDEGPU_EL_TRACE_FILE
    // (block, start, end in 10-ns ticks from the earliest start, XCD, HW_ID; tools/el_trace.py)
    const char* tf = getenv("PDEGPU_EL_TRACE_FILE");
    if (rc == PDEGPU_OK && tf && *tf && !use_mfma && s.fast && ctx->el_fast) {
        const ElFastGeom& fgs = s.fgeom;
        const int nb = fgs.nb_fast + fgs.nb_bnd;
        unsigned long long* d_tr = nullptr;
        if (hipMalloc(&d_tr, (size_t)nb * 8 * sizeof(unsigned long long)) == hipSuccess && hipMemsetAsync(d_tr, 0, (size_t)nb * 64, ctx->stream) == hipSuccess) {
            ElFastGeom fgt = fgs;
            fgt.trace = d_tr;
            a.partials_qmq = s.part4;
            rc = el_apply(&s, a, APPLY_PLAIN, &fgt);
            std::vector<unsigned long long> h((size_t)nb * 8);
            if (rc == PDEGPU_OK && hipMemcpyAsync(h.data(), d_tr, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                hipStreamSynchronize(ctx->stream) == hipSuccess) {
                unsigned long long t0 = ~0ULL;
                for (int b = 0; b < nb; ++b) t0 = std::min(t0, h[8 * b]);
                if (FILE* f = fopen(tf, "w")) {
                    fprintf(f, "# nb_fast %d nb_bnd %d bnd_last %d ntx %d nty %d nyg %d nzc %d zlen %d nlong %d z0 %d\n", fgs.nb_fast, fgs.nb_bnd, fgs.bnd_last, fgs.ntx,
                            fgs.nty, fgs.nyg, fgs.nzc, fgs.zlen, fgs.nlong, fgs.z0);
                    auto rel = [&](unsigned long long t) { return t ? t - t0 : 0ULL; };
                    for (int b = 0; b < nb; ++b)
                        fprintf(f, "%d %llu %llu %llu %llu %llu %llu %llu %llu\n", b, h[8 * b] - t0, h[8 * b + 1] - t0, h[8 * b + 2] >> 32, h[8 * b + 2] & 0xffffffffULL,
                                rel(h[8 * b + 3]), rel(h[8 * b + 4]), rel(h[8 * b + 5]), rel(h[8 * b + 6]));
                    fclose(f);
                }
            }
            (void)hipFree(d_tr);
        }
    }
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
extern "C" int pdegpu_el_session_k1(pdegpu_el_session* h) {
    if (!h) return PDEGPU_EINVAL;
    return h->s.variant ? el_sr_k1(&h->s, false, -1) : el_k1(&h->s);
}
extern "C" int pdegpu_el_session_can_split_k1(const pdegpu_el_session* h) { return (h && el_k1_splittable(&h->s)) ? 1 : 0; }
extern "C" int pdegpu_el_session_k1_part(pdegpu_el_session* h, int part) {
    if (!h || (part != 0 && part != 1)) return PDEGPU_EINVAL;
    return h->s.variant ? el_sr_k1(&h->s, false, part + 1) : el_k1(&h->s, nullptr, part);
}
// 0 = classic Jacobi-PCG (scalars[0] after k1, scalars[1..2] after k2); 1 = single-reduction form: the residual update moves INTO k1 (fused with
// the p-update), k2 does nothing, and after k1 / k1_part(1) the block scalars[0..3] = {p.Ap, r.z, r.r, Ap.D^-1 Ap} is the ONE message of the
// iteration.  On a slab with neighbours r changes inside k1, so the protocol becomes: k1_pre (update pass of the owned planes) -> exchange the
// first / last owned plane of r -> k1_part(0) (apply away from the ghosts; may run while the planes travel) -> k1_part(1).  Choose before begin().
extern "C" int pdegpu_el_session_set_cg_variant(pdegpu_el_session* h, int variant) {
    if (!h || (variant != 0 && variant != 1)) return PDEGPU_EINVAL;
    h->s.variant = variant;
    h->s.sr_mem = true;
    return PDEGPU_OK;
}
extern "C" int pdegpu_el_session_cg_variant(const pdegpu_el_session* h) { return h ? h->s.variant : -1; }
extern "C" int pdegpu_el_session_k1_pre(pdegpu_el_session* h) {
    if (!h) return PDEGPU_EINVAL;
    return h->s.variant ? el_sr_k1(&h->s, false, 0) : PDEGPU_OK;
}
extern "C" int pdegpu_el_session_k2(pdegpu_el_session* h) {
    if (!h) return PDEGPU_EINVAL;
    return h->s.variant ? PDEGPU_OK : el_k2(&h->s, 1);
}
extern "C" int pdegpu_el_session_flush(pdegpu_el_session* h) {
    if (!h) return PDEGPU_EINVAL;
    // (single-reduction form: x matches the r that r.r reports — up to the update a pass of the deferred form may have left out)
    return h->s.variant ? el_xflush(&h->s) : el_flush(&h->s);
}
// ---- exchange and scalar reduction through the library's own RCCL communicator (comm.hip): same protocol as the heat session ----
namespace {
// first / last owned plane of r (3 component planes, each contiguous) <-> the neighbours' ghost planes; what: bit 0 sends, bit 1 receives
int el_exchange_post(ElSession* s, pdegpu_comm* c, int what, int lo_rank, int hi_rank) {
    const GridDev& g = s->g;
    const bool lo = g.k0 > 0, hi = g.k0 + g.nkl < g.NZ;
    if (lo_rank < 0) lo_rank = comm_rank(c) - (comm_self_loop(c) ? 0 : 1);
    if (hi_rank < 0) hi_rank = comm_rank(c) + (comm_self_loop(c) ? 0 : 1);
    const size_t n = (size_t)g.PP;
    for (int comp = 0; comp < s->D; ++comp) {
        double* base = s->r + (long long)comp * s->stride;
        if (what & 1) {
            if (lo) PG_TRY(comm_send(c, base, n, lo_rank));
            if (hi) PG_TRY(comm_send(c, base + (long long)(g.nkl - 1) * g.PP, n, hi_rank));
        }
        if (what & 2) {
            if (lo) PG_TRY(comm_recv(c, base - g.PP, n, lo_rank));
            if (hi) PG_TRY(comm_recv(c, base + (long long)g.nkl * g.PP, n, hi_rank));
        }
    }
    return PDEGPU_OK;
}
}  // namespace

extern "C" int pdegpu_el_session_exchange_post(pdegpu_el_session* h, pdegpu_comm* c, int what, int lo_rank, int hi_rank) {
    if (!h || !c || (what & ~3) || !what) return PDEGPU_EINVAL;
    PG_HIP(h->s.ctx, hipSetDevice(h->s.ctx->device));
    return el_exchange_post(&h->s, c, what, lo_rank, hi_rank);
}
extern "C" int pdegpu_el_session_exchange_begin(pdegpu_el_session* h, pdegpu_comm* c) {
    if (!h || !c) return PDEGPU_EINVAL;
    ElSession* s = &h->s;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    const bool any = s->g.k0 > 0 || s->g.k0 + s->g.nkl < s->g.NZ;
    PG_TRY(comm_ready(c, s->ctx->stream));
    if (any) {
        PG_TRY(comm_group_start(c));
        const int rc = el_exchange_post(s, c, 3, -1, -1);
        PG_TRY(comm_group_end(c));
        PG_TRY(rc);
    }
    return comm_done(c);
}
extern "C" int pdegpu_el_session_exchange_end(pdegpu_el_session* h, pdegpu_comm* c) {
    if (!h || !c) return PDEGPU_EINVAL;
    PG_HIP(h->s.ctx, hipSetDevice(h->s.ctx->device));
    return comm_wait(c, h->s.ctx->stream);
}
extern "C" int pdegpu_el_session_allreduce(pdegpu_el_session* h, pdegpu_comm* c, int first, int count) {
    if (!h || !c || first < 0 || count < 1 || first + count > ESC_COUNT) return PDEGPU_EINVAL;
    PG_HIP(h->s.ctx, hipSetDevice(h->s.ctx->device));
    return comm_allreduce(c, h->s.scal + first, (size_t)count, h->s.ctx->stream);
}
// `iters` PCG iterations of this rank's slab, exchange and reduction(s) issued from here (the loop of pdegpu.dist.SlabCG.iterate).  Single-reduction
// form: update pass of the owned planes -> exchange of r posted -> apply away from the ghosts -> wait -> ghost update + edge planes + reduce -> ONE
// 32-byte all-reduce; classic form: exchange around the split apply, 8 bytes after it, 16 after k2.
extern "C" int pdegpu_el_session_iterate_slab(pdegpu_el_session* h, pdegpu_comm* c, int iters) {
    if (!h || !c || iters < 0) return PDEGPU_EINVAL;
    ElSession* s = &h->s;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    const bool multi = comm_world(c) > 1 || comm_self_loop(c);
    for (int it = 0; it < iters; ++it) {
        PG_TRY(pdegpu_el_session_k1_pre(h));
        if (multi) PG_TRY(pdegpu_el_session_exchange_begin(h, c));
        PG_TRY(pdegpu_el_session_k1_part(h, 0));
        if (multi) PG_TRY(pdegpu_el_session_exchange_end(h, c));
        PG_TRY(pdegpu_el_session_k1_part(h, 1));
        PG_TRY(comm_allreduce(c, s->scal, s->variant ? 4 : 1, s->ctx->stream));
        if (!s->variant) {
            PG_TRY(pdegpu_el_session_k2(h));
            PG_TRY(comm_allreduce(c, s->scal + 1, 2, s->ctx->stream));
        }
    }
    return PDEGPU_OK;
}

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
    if (s.variant) PG_TRY(el_xflush(&s));      // (a deferred x update must not be missing from what is read)
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
