// wmarch.hip — z-marching kernels for the weighted 3-D heat operators (gfx950).
//
// The reference's curvilinear and composite tools (_solve_heat_3d_spherical_raw, fenics_mcp_server.py:1333-1464; the BoxMesh fallback of
// geometry_type="cylinder", :523-529, :645; the composite core, :537-572) multiply mass and stiffness by a coordinate weight / a per-cell
// diffusivity, so every node has its own 15 coupling weights (weighted.hip integrates them).  In all of them the weight does NOT depend on
// one grid coordinate, and that decides where a marching wavefront keeps its coefficients:
//   CM = 1  r^2 sin(theta): independent of phi, the slowest index = the marching direction.  A lane loads the 15 weights of its two nodes ONCE
//           per z-chunk (plane class "interior" of the compact rows) and keeps them in registers for the whole march.
//   CM = 2  sqrt(y^2 + z^2), the unit weight, the core factor: independent of x = the lane direction.  The 15 weights of a grid row are
//           wave-uniform: scalar loads (constant address space -> s_load into SGPRs) from a row table [NZ * NY][16], one 128-byte line per row
//           and plane step; entry 15 of the line is the row's Jacobi diagonal^-1.
// Otherwise the kernel is the plain 15-point marching kernel of stencil.hip: a wavefront owns 128 nodes x RY rows, keeps three planes in
// registers, takes x-neighbours by DPP wave shifts (shifted ONCE per plane and row: the shifted value is part of the rotating plane state, because
// with per-node weights the constant-coefficient trick of shifting weighted sums is gone), reads every input plane once.
// Jacobi-PCG with ONE residual vector: the session's r holds z = D^-1 r, the only form K1 needs (p = z + beta p, exactly the plain kernel's
// recurrence, no diagonal in K1); the residual pass stores z = D^-1 (b - A x), K2 updates z -= alpha D^-1 q and sums r.z = sum D z^2 with D^-1 from
// the compact table (a flat pass with registers to spare).  An iteration moves  K1: read z, p, x; write p, x, q (48 B/DoF)  +  K2: read z, q; write
// z (24 B/DoF)  =  72 B/DoF instead of the 88 of the class-table form (k_general<..., ELL> + k_k2_pc, which also costs 45 load instructions per
// node).  Where a vector exceeds 32 MiB (wfast_xdefer) K2 is a mode of the marching kernel instead — the weighted stencil on the stored p again — and
// carries the x update of every other iteration: K1 24 + K2 24 | 40 = 56 B/DoF (see the comment above k_wmarch).
// Sessions: one slab owning every plane, every boundary node Dirichlet (the uniform boundary value of the tools).
#include "common.h"
#include "dot_finish.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace {

#define PG_CONSTANT __attribute__((address_space(4)))

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
// lane l receives the value of lane l-1 / l+1 (v_mov_b32_dpp wave_shr:1 / wave_shl:1; the end lanes receive 0)
__device__ __forceinline__ double dpp_from_lower(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_upper(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// non-temporal 16-byte accesses (a compile-time choice: behind a runtime switch the compiler merges the two forms and drops the hint)
typedef double wm_d2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 wm_ld(const double* p) {
    if constexpr (NT) {
        const wm_d2 v = __builtin_nontemporal_load(reinterpret_cast<const wm_d2*>(p));
        return make_double2(v.x, v.y);
    } else {
        return *reinterpret_cast<const double2*>(p);
    }
}
template <bool NT>
__device__ __forceinline__ void wm_st(double* p, const double2& v) {
    if constexpr (NT) {
        const wm_d2 t = {v.x, v.y};
        __builtin_nontemporal_store(t, reinterpret_cast<wm_d2*>(p));
    } else {
        *reinterpret_cast<double2*>(p) = v;
    }
}

struct WGeom {
    int ntx, nty;        // 128-node x-tiles, RY-row y-tiles of the free rows jlo .. jhi
    int nwg_xy;          // workgroups per z-chunk (four consecutive tiles of the row-major tile order each)
    int nzc, zlen;       // z-chunks of zlen planes over the free planes klo .. khi
    int jlo, jhi, klo, khi;   // rows / planes that carry unknowns: 1 .. N-2 where the y / z faces are Dirichlet, 0 .. N-1 where they are natural
    int xlo, xhi;             // first / last column that carries unknowns: 1 / NX-2, or 0 / NX-1 where that x-face is natural (FX instantiations)
};

// one row of one held plane: the two nodes of the lane and their x-neighbours (l = node i0-1, r = node i0+2)
struct WRow { double l, x, y, r; };

// LQ (CM 1): the last LQ of a node pair's 15 weights live in LDS instead of registers — as a per-lane scratch (a lane reads back what it wrote: no
// barrier, conflict-free 16-byte slots) — which is what lets a wavefront with per-lane weights own TWO rows: 2 x 30 weights in registers leave no
// room for the plane state (> 256 VGPRs); with 8 of 15 in LDS (64 KB per workgroup, two workgroups per CU) it fits.
// FX (CM 2): a natural x-face (T_left or T_right not given, fenics_mcp_server.py:606-623).  Its nodes are lanes of ordinary rows, but their weights are
// those of the column class "face" (truncated cells), not the wave-uniform ones: every lane reads the 128-byte line of ITS face table (lanes without a
// face node: any line, unused) beside the scalar line of the row, the stencil of the two nodes is formed a second time with it, and the face lanes
// keep that second result.  A template flag: sessions without a natural x-face run the instantiation without those loads.
// NT: non-temporal stores of q / p / x (and loads of x, which nothing reads again before the next iteration) on HBM-resident grids, as in k_fast.
// EX: the tile-edge x-neighbours (node i0-1 of lane 0, node i0+2 of lane 63) are handed between the x-adjacent wavefronts of the workgroup through LDS
// (one barrier per plane; the four waves of a workgroup march in lockstep anyway) instead of 2-lane global loads — 2 (RY + 2) of them per plane and
// vector, more load instructions than the wide ones; only a wave whose neighbour tile belongs to another workgroup still loads its edge from memory.
// MODE CG_K2 + XU (x_in_k2 sessions, as in k_fast — tunable cg_xdefer): K2 applies the weighted stencil to the stored p again,
// z -= alpha D^-1 (A p) with the partial sum of D z^2, instead of reading a stored q; every other such launch (XU = 1) also updates x for two iterations
// at once with p_{n-1} = (p_n - z_n) / beta_n.  K1 of such a session (XU = 0) neither touches x nor stores q: 24 + 24 | 40 = 56 instead of 72 B/DoF.
template <int CM, int RY, int MODE, int LQ, bool FX, bool NT, bool EX, int XU>
__global__ __launch_bounds__(256, 2) void k_wmarch(StencilArgs a, WGeom wgm) {
    __shared__ double sEdge[EX ? 2 : 1][4][RY + 2][2];
    static_assert(!FX || CM == 2, "natural x-faces: weights independent of x only");
    __shared__ double sRed[4];
    __shared__ double2 sC[LQ > 0 ? 4 : 1][LQ > 0 ? RY : 1][LQ > 0 ? LQ : 1][LQ > 0 ? 64 : 1];
    constexpr int NQR = 15 - LQ;               // weights of a node pair kept in registers
    constexpr bool K1 = (MODE == APPLY_CG_K1), RES = (MODE == APPLY_RESIDUAL), K2 = (MODE == APPLY_CG_K2);
    constexpr bool K1X = K1 && XU != 0, K2X = K2 && XU != 0;
    constexpr bool DI = RES || K2;             // the pass needs the Jacobi diagonal^-1 (of its own rows)
    constexpr int RH = RY + 2;
    const GridDev& g = a.g;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD-aware order (workgroups b and b + 8 share an XCD): XCD x sweeps the contiguous range [x G/8, (x+1) G/8) — y-neighbouring tiles of one z-chunk
    const int G = gridDim.x;
    int wid = blockIdx.x;
    if ((G & 7) == 0) wid = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int zc = wid / wgm.nwg_xy;
    const int w = (wid - zc * wgm.nwg_xy) * 4 + wv;
    const int yt = w / wgm.ntx, xt = w - yt * wgm.ntx;
    const int i0 = xt * 128 + 2 * lane;
    const int j0 = wgm.jlo + yt * RY;
    const int jmax = wgm.jhi;
    const int lk0 = wgm.klo + zc * wgm.zlen;
    const int lk1 = min(lk0 + wgm.zlen, wgm.khi + 1);
    const bool wg_active = zc < wgm.nzc && lk0 < lk1;                       // workgroup-uniform
    const bool wave_active = wg_active && yt < wgm.nty;                     // wave-uniform
    // EX: every wave of a workgroup with work runs the march (an inactive one with its loads at offset 0 and its stores masked) so that the barriers match
    const bool active = EX ? wg_active : wave_active;
    int estep = 0;                                                          // edge hand-off counter (the same in every wave of the workgroup)

    double alpha_prev = 0.0, beta = 0.0, res_scale = 1.0;
    if (K1) cg_scalars_in(a.di, a.scal, sRed, &alpha_prev, &beta);          // (every thread of every workgroup: the prologue sum has barriers)
    if (K2) res_scale = cg_alpha_in(a.di, a.scal, sRed);
    double cx_p = 0.0, cx_r = 0.0, cx_o = 0.0;
    if (K1 && !XU && a.xd && blockIdx.x == 0 && threadIdx.x == 0) a.xd[1] = beta;
    if (K2 && !XU && a.xd && blockIdx.x == 0 && threadIdx.x == 0) a.xd[0] = res_scale;
    if (K2X) xk2_coefficients(a.xd, a.xmode, res_scale, &cx_p, &cx_r, &cx_o);
    double acc = 0.0;
    if (active) {
        const double shift = a.in_shift;
        const bool lane_ok = wave_active && i0 < g.PX;
        const bool ma = (i0 >= wgm.xlo) && (i0 <= wgm.xhi);                 // node a = i0 is a free node
        const bool mb = (i0 + 1 <= wgm.xhi);                                // node b = i0 + 1
        // FX: which of the lane's two nodes sits on a natural x-face, and the face table of that node (0: x = 0, 1: x = NX-1)
        const bool fa_lo = FX && wgm.xlo == 0 && i0 == 0, fa_hi = FX && wgm.xhi == g.NX - 1 && i0 == g.NX - 1;
        const bool fb_hi = FX && wgm.xhi == g.NX - 1 && i0 + 1 == g.NX - 1;
        const bool face_a = fa_lo || fa_hi, face_b = fb_hi;
        // (lanes without a face node read a line of whichever face table exists and never use it; only one of the two may be allocated)
        const double* ftab = (fa_hi || fb_hi || !a.wface[0]) ? a.wface[1] : a.wface[0];
        // the x-neighbour tile of this wave inside the workgroup (waves are consecutive tiles of the row-major tile order)
        const bool has_l = EX && wv > 0 && xt > 0, has_r = EX && wv < 3 && xt + 1 < wgm.ntx;
        const bool elane = wave_active && ((lane == 0 && i0 >= 1 && !has_l) || (lane == 63 && i0 + 2 < g.PX && !has_r));      // edge from global memory
        const bool need_ge = !EX || __builtin_amdgcn_readfirstlane((int)((xt > 0 && !has_l) || (xt * 128 + 128 < g.PX && !has_r))) != 0;
        const long long eoff = (lane == 0) ? -1 : 2;
        // (natural y / z faces: the rows -1 / NY and the planes -1 / NZ a face row asks for do not exist — they meet weights that are exactly 0: the row
        // reads offset 0 of the vector, the ghost planes are zero)
        auto row_ok = [&](int j) { return j >= 0 && j < g.NY; };
        auto load2 = [&](const double* v, long long off) { return *reinterpret_cast<const double2*>(v + off); };
        const double PG_CONSTANT* tab = (const double PG_CONSTANT*)a.ell;    // CM 2: the row table

        // ---- coefficients -------------------------------------------------------------------------------------------------------------------
        // CM 1: cA / cB = the 15 weights of node a / b of every own row, D = diagonal^-1 of the own rows (residual pass)
        double cA[CM == 1 ? RY : 1][NQR], cB[CM == 1 ? RY : 1][NQR];
        double2 D[(CM == 1 && DI) ? RY : 1];
        if (CM == 1) {
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int j = j0 + r;
                const long long off = (j <= jmax && lane_ok) ? (long long)j * g.PX + i0 : 0;
#pragma unroll
                for (int q = 0; q < 15; ++q) {
                    const double2 c = load2(a.ell + (long long)q * a.ell_stride, off);
                    if (q < NQR) {
                        cA[r][q] = c.x;
                        cB[r][q] = c.y;
                    } else {
                        sC[wv][r][q - NQR][lane] = c;
                    }
                }
            }
            if (DI) {
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    const int j = j0 + r;
                    D[r] = load2(a.wdinv, (j <= jmax && lane_ok) ? (long long)j * g.PX + i0 : 0);
                }
            }
        }

        // ---- one plane of the marched vector into registers ---------------------------------------------------------------------------------
        // Loads are unconditional and unmasked (see k_fast::value_row): a lane / row outside the grid reads offset 0 of the vector and what it reads
        // only reaches outputs that are masked or never stored.  K1: value = p_n = z + beta p_{n-1}; PLAIN: in + shift; RESIDUAL: in.
        auto load_plane = [&](int lk, WRow* rows, double2* po_out) {
            double2 vin[RH], vp[RH];
            double ein[RH], ep[RH];
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                const int j = j0 - 1 + r;
                const bool rok = row_ok(j);
                const long long off = (rok && lane_ok) ? (long long)lk * g.PP + (long long)j * g.PX + i0 : 0;
                const long long offe = (rok && elane) ? (long long)lk * g.PP + (long long)j * g.PX + i0 + eoff : 0;
                vin[r] = load2(a.in, off);
                ein[r] = 0.0;
                ep[r] = 0.0;
                if (need_ge) ein[r] = a.in[offe];                 // (wave-uniform branch)
                if (K1) {
                    vp[r] = load2(a.p_old, off);
                    if (need_ge) ep[r] = a.p_old[offe];
                }
            }
            double2 vals[RH];
            double eg[RH];
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                if (K1) {
                    vals[r].x = vin[r].x + beta * vp[r].x;
                    vals[r].y = vin[r].y + beta * vp[r].y;
                    eg[r] = ein[r] + beta * ep[r];
                    if (po_out) po_out[r] = vp[r];
                } else {
                    vals[r].x = vin[r].x + shift;
                    vals[r].y = vin[r].y + shift;
                    eg[r] = ein[r] + shift;
                }
            }
            const int buf = estep & 1;
            if (EX) {
                ++estep;
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    if (lane == 0) sEdge[buf][wv][r][0] = vals[r].x;
                    if (lane == 63) sEdge[buf][wv][r][1] = vals[r].y;
                }
                __syncthreads();
            }
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                double e = eg[r];
                if (EX) {
                    if (lane == 0 && has_l) e = sEdge[buf][wv - 1][r][1];
                    if (lane == 63 && has_r) e = sEdge[buf][wv + 1][r][0];
                }
                const double lv = dpp_from_lower(vals[r].y), rv = dpp_from_upper(vals[r].x);
                rows[r].x = vals[r].x;
                rows[r].y = vals[r].y;
                rows[r].l = (lane == 0) ? e : lv;
                rows[r].r = (lane == 63) ? e : rv;
            }
        };
        auto own_xload = [&](int lk, int j) {
            const bool ok = j <= jmax && lane_ok;
            return wm_ld<NT>(a.x + (ok ? (long long)lk * g.PP + (long long)j * g.PX + i0 : 0));
        };
        // K1: store p_n and x += alpha_{n-1} p_{n-1} of an owned row (every node of the row pair: Dirichlet nodes and pads hold p = 0, r = 0)
        auto own_update = [&](int lk, int j, const WRow& pn, const double2& po, double2 xv) {
            if (K1 && j <= jmax && lane_ok) {
                const long long off = (long long)lk * g.PP + (long long)j * g.PX + i0;
                wm_st<NT>(a.p_new + off, make_double2(pn.x, pn.y));
                if (K1X && alpha_prev != 0.0) {
                    xv.x += alpha_prev * po.x;
                    xv.y += alpha_prev * po.y;
                    wm_st<NT>(a.x + off, xv);
                }
            }
        };

        WRow P[RH], C[RH], N[RH];
        double2 PO[RH];
        load_plane(lk0 - 1, P, nullptr);
        load_plane(lk0, C, PO);
        if (K1) {
#pragma unroll
            for (int r = 1; r <= RY; ++r) own_update(lk0, j0 - 1 + r, C[r], PO[r], K1X ? own_xload(lk0, j0 - 1 + r) : make_double2(0.0, 0.0));
        }

        // CM 2: the 128-byte line of every own row of a plane (15 weights + D^-1), requested a whole plane step AHEAD: at the end of the step before, behind
        // the stencil that consumed the previous lines.  (Scalar loads and LDS share one counter — lgkmcnt, and scalar loads return out of order, so any
        // wait with one outstanding is a wait for all — hence not at the top of the step in front of the edge hand-off's LDS traffic.  Measured: no
        // difference either way, 572-584 us against 568-595 us at 385^3; the hand-off stays slower for this form on HBM-resident grids, see launch_wmarch.)
        double CS[CM == 2 ? RY : 1][16];
        auto load_lines = [&](int lk) {
            if (CM == 2) {
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    const long long e = ((long long)min(lk, wgm.khi) * g.NY + min(j0 + r, jmax)) * 16;
#pragma unroll
                    for (int q = 0; q < 16; ++q) CS[r][q] = tab[e + q];
                }
            }
        };
        load_lines(lk0);
        for (int lk = lk0; lk < lk1; ++lk) {
            double2 CF[FX ? RY : 1][8];             // FX: the face line of every own row (per-lane address)
            if (FX) {
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    const long long e = ((long long)lk * g.NY + min(j0 + r, jmax)) * 16;
#pragma unroll
                    for (int q = 0; q < 8; ++q) CF[r][q] = load2(ftab, e + 2 * q);
                }
            }
            double2 XV[RY];
#pragma unroll
            for (int r = 0; r < RY; ++r) XV[r] = make_double2(0.0, 0.0);
            if (K1X && lk + 1 < lk1) {
#pragma unroll
                for (int r = 0; r < RY; ++r) XV[r] = own_xload(lk + 1, j0 + r);
            }
            double2 RHS[RY];
            double2 XK[K2X ? RY : 1], PO2[K2X ? RY : 1];      // K2X: the x rows of this plane (and, after a restart only, p of the iteration before)
            if (RES || K2) {
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    const int j = j0 + r;
                    const long long off = (j <= jmax && lane_ok) ? (long long)lk * g.PP + (long long)j * g.PX + i0 : 0;
                    RHS[r] = make_double2(a.rhs_const, a.rhs_const);
                    if (a.rhs) RHS[r] = load2(a.rhs, off);
                    if (K2X) {
                        XK[r] = wm_ld<NT>(a.x + off);
                        PO2[r] = make_double2(0.0, 0.0);
                        if (cx_o != 0.0) PO2[r] = load2(a.p_old, off);      // kernel-uniform, rare
                    }
                }
            }
            load_plane(lk + 1, N, PO);

#pragma unroll
            for (int r = 0; r < RY; ++r) {
                // (no branch around the stencil: a row beyond jmax — the second row of the last tile when RY = 2 — is computed on whatever its registers hold and
                // neither stored nor summed; in one basic block the scalar loads of the weights are scheduled ahead of the waits for the plane)
                const int c = r + 1;
                const int j = j0 + r;
                const bool jok = j <= jmax;                                  // wave-uniform
                double ca[15], cb[15], da = 0.0, db = 0.0;
                if (CM == 1) {
#pragma unroll
                    for (int q = 0; q < NQR; ++q) { ca[q] = cA[r][q]; cb[q] = cB[r][q]; }
#pragma unroll
                    for (int q = NQR; q < 15; ++q) {
                        const double2 c = sC[wv][r][q - NQR][lane];
                        ca[q] = c.x;
                        cb[q] = c.y;
                    }
                    if (DI) { da = D[r].x; db = D[r].y; }
                } else {
#pragma unroll
                    for (int q = 0; q < 15; ++q) ca[q] = cb[q] = CS[r][q];
                    if (DI) da = db = CS[r][15];
                }
                // 15-offset list (tables_math.h pg_off15): 0 centre, 1/2 -+x, 3/4 -+y, 5/6 -+z, 7/8 -+(1,1,0), 9/10 -+(1,0,1), 11/12 -+(0,1,1), 13/14 -+(1,1,1)
                auto stencil_a = [&](const double* w) {
                    return w[0] * C[c].x + w[1] * C[c].l + w[2] * C[c].y + w[3] * C[c - 1].x + w[4] * C[c + 1].x + w[5] * P[c].x + w[6] * N[c].x +
                           w[7] * C[c - 1].l + w[8] * C[c + 1].y + w[9] * P[c].l + w[10] * N[c].y + w[11] * P[c - 1].x + w[12] * N[c + 1].x +
                           w[13] * P[c - 1].l + w[14] * N[c + 1].y;
                };
                auto stencil_b = [&](const double* w) {
                    return w[0] * C[c].y + w[1] * C[c].x + w[2] * C[c].r + w[3] * C[c - 1].y + w[4] * C[c + 1].y + w[5] * P[c].y + w[6] * N[c].y +
                           w[7] * C[c - 1].x + w[8] * C[c + 1].r + w[9] * P[c].x + w[10] * N[c].r + w[11] * P[c - 1].y + w[12] * N[c + 1].y +
                           w[13] * P[c - 1].x + w[14] * N[c + 1].r;
                };
                double qa = stencil_a(ca), qb = stencil_b(cb);
                double diag_a = ca[0], diag_b = cb[0];
                if (FX) {                                                    // the face nodes of this wave take the second result
                    double cf[16];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { cf[2 * q] = CF[r][q].x; cf[2 * q + 1] = CF[r][q].y; }
                    const double fa = stencil_a(cf), fb = stencil_b(cf);
                    qa = face_a ? fa : qa;
                    qb = face_b ? fb : qb;
                    diag_a = face_a ? cf[0] : diag_a;
                    diag_b = face_b ? cf[0] : diag_b;
                    if (DI) {
                        da = face_a ? cf[15] : da;
                        db = face_b ? cf[15] : db;
                    }
                }
                bool fra = ma && jok, frb = mb && jok;                       // the node carries an unknown
                if (FX) {
                    double dv;
                    fra = fra && i0 < g.NX && !pg_dirichlet(g, i0, j, lk, &dv);
                    frb = frb && i0 + 1 < g.NX && !pg_dirichlet(g, i0 + 1, j, lk, &dv);
                }
                double2 o;
                if (RES) {                                                   // z = D^-1 (rhs - A in);  r.z = D z^2
                    o.x = fra ? da * (RHS[r].x - qa) : 0.0;
                    o.y = frb ? db * (RHS[r].y - qb) : 0.0;
                    acc += diag_a * o.x * o.x + diag_b * o.y * o.y;          // (the centre weight IS the diagonal)
                } else if (K2) {                                             // z -= alpha D^-1 (A p);  r.z = D z^2  (in = the stored p_n, rhs = out = z)
                    o.x = fra ? RHS[r].x - res_scale * da * qa : 0.0;
                    o.y = frb ? RHS[r].y - res_scale * db * qb : 0.0;
                    acc += diag_a * o.x * o.x + diag_b * o.y * o.y;
                } else {
                    o.x = fra ? qa : 0.0;
                    o.y = frb ? qb : 0.0;
                    acc += (C[c].x - (K1 ? 0.0 : shift)) * o.x + (C[c].y - (K1 ? 0.0 : shift)) * o.y;
                }
                if (lane_ok && jok && (!K1 || a.out)) wm_st<NT>(a.out + (long long)lk * g.PP + (long long)j * g.PX + i0, o);
                if (K2X && lane_ok && jok) {
                    // x += alpha_n p_n (+ the update the K2 before left out, p_{n-1} = (p_n - z_n) / beta_n): p_n = C, z_n = RHS (0 on Dirichlet nodes and pads)
                    double2 xv = XK[r];
                    xv.x += cx_p * C[c].x - cx_r * RHS[r].x + cx_o * PO2[r].x;
                    xv.y += cx_p * C[c].y - cx_r * RHS[r].y + cx_o * PO2[r].y;
                    wm_st<NT>(a.x + (long long)lk * g.PP + (long long)j * g.PX + i0, xv);
                }
            }
            if (K1 && lk + 1 < lk1) {
#pragma unroll
                for (int r = 1; r <= RY; ++r) own_update(lk + 1, j0 - 1 + r, N[r], PO[r], XV[r - 1]);
            }
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                P[r] = C[r];
                C[r] = N[r];
            }
            load_lines(lk + 1);
        }
    }
    const double tot = block_sum_256(acc, sRed);
    dot_finish(tot, a.partials);
}

// K2 of the weighted PCG on z = D^-1 r: z -= alpha D^-1 q on the free rows, partial r.z = sum D z^2.  A wavefront takes whole grid rows (j, k), so the
// diagonal comes without index arithmetic: CM 1 from the interior plane class of the compact rows (D = centre weight, D^-1: [j * PX + i]), CM 2 as two
// scalars of the row's line (entry 0 = D, entry 15 = D^-1).  Dirichlet nodes and pads of a free row hold z = q = 0 and add 0.
// NT: non-temporal loads / stores of the two streamed vectors (HBM-resident grids: nothing of them is re-read before it has left every cache; a
// compile-time choice — behind a runtime switch the compiler merges the two forms and drops the hint)
typedef double wk_d2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 wk_ld(const double* p) {
    if constexpr (NT) {
        const wk_d2 v = __builtin_nontemporal_load(reinterpret_cast<const wk_d2*>(p));
        return make_double2(v.x, v.y);
    } else {
        return *reinterpret_cast<const double2*>(p);
    }
}
template <bool NT>
__device__ __forceinline__ void wk_st(double* p, const double2& v) {
    if constexpr (NT) {
        const wk_d2 t = {v.x, v.y};
        __builtin_nontemporal_store(t, reinterpret_cast<wk_d2*>(p));
    } else {
        *reinterpret_cast<double2*>(p) = v;
    }
}
template <int CM, bool NT>
__global__ __launch_bounds__(256) void k_wk2(double* __restrict__ r, const double* __restrict__ q, const double* __restrict__ dtab, const double* __restrict__ ditab,
                                              GridDev g, int nrows, int jlo, int ny_free, int klo, const double* __restrict__ flo, const double* __restrict__ fhi,
                                              double* scal, double* __restrict__ partials, DotIn di) {
    __shared__ double sRed[4];
    const double alpha = cg_alpha_in(di, scal, sRed);
    double acc = 0.0;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ny2 = ny_free, px2 = g.PX / 2;
    const double PG_CONSTANT* tab = (const double PG_CONSTANT*)dtab;
    for (int row = blockIdx.x * 4 + wv; row < nrows; row += gridDim.x * 4) {       // wave-uniform
        const int kk = row / ny2, j = row - kk * ny2 + jlo, k = kk + klo;
        const long long base = (long long)k * g.PP + (long long)j * g.PX;
        double ds = 0.0, dis = 0.0, dlo = 0.0, dilo = 0.0, dhi = 0.0, dihi = 0.0;
        if (CM == 2) {
            const long long e = ((long long)k * g.NY + j) * 16;
            ds = tab[e];
            dis = tab[e + 15];
            // natural x-faces (kernel-uniform): the diagonal of the face node of this row from its own table
            if (flo) { dlo = ((const double PG_CONSTANT*)flo)[e]; dilo = ((const double PG_CONSTANT*)flo)[e + 15]; }
            if (fhi) { dhi = ((const double PG_CONSTANT*)fhi)[e]; dihi = ((const double PG_CONSTANT*)fhi)[e + 15]; }
        }
        for (int s0 = 0; s0 < px2; s0 += 128) {
            // two 64-pair segments per round: their loads are in flight together
            const int sa = s0 + lane, sb = s0 + 64 + lane;
            const bool oka = sa < px2, okb = sb < px2;
            const long long offa = base + 2LL * (oka ? sa : 0), offb = base + 2LL * (okb ? sb : 0);
            double2 ra = wk_ld<NT>(r + offa), rb = wk_ld<NT>(r + offb);
            const double2 qa = wk_ld<NT>(q + offa), qb = wk_ld<NT>(q + offb);
            double2 da = make_double2(ds, ds), db = da, ia = make_double2(dis, dis), ib = ia;
            if (CM == 2 && (flo || fhi)) {
                const int ihi = g.NX - 1;
                if (flo && sa == 0) { da.x = dlo; ia.x = dilo; }
                if (fhi) {
                    if (2 * sa == ihi) { da.x = dhi; ia.x = dihi; }
                    if (2 * sa + 1 == ihi) { da.y = dhi; ia.y = dihi; }
                    if (2 * sb == ihi) { db.x = dhi; ib.x = dihi; }
                    if (2 * sb + 1 == ihi) { db.y = dhi; ib.y = dihi; }
                }
            }
            if (CM == 1) {
                const long long ta = (long long)j * g.PX + 2LL * (oka ? sa : 0), tb = (long long)j * g.PX + 2LL * (okb ? sb : 0);
                da = *reinterpret_cast<const double2*>(dtab + ta);
                db = *reinterpret_cast<const double2*>(dtab + tb);
                ia = *reinterpret_cast<const double2*>(ditab + ta);
                ib = *reinterpret_cast<const double2*>(ditab + tb);
            }
            ra.x -= alpha * (ia.x * qa.x);
            ra.y -= alpha * (ia.y * qa.y);
            rb.x -= alpha * (ib.x * qb.x);
            rb.y -= alpha * (ib.y * qb.y);
            if (oka) {
                wk_st<NT>(r + offa, ra);
                acc += da.x * ra.x * ra.x + da.y * ra.y * ra.y;
            }
            if (okb) {
                wk_st<NT>(r + offb, rb);
                acc += db.x * rb.x * rb.x + db.y * rb.y * rb.y;
            }
        }
    }
    const double tot = block_sum_256(acc, sRed);
    __syncthreads();
    dot_finish(tot, partials);
}

// CM 2: the row tables [NZ * NY][16] from the compact rows of the column class "interior" (entry e = 3 * row + 1 of the q-major arrays):
// 15 weights + (table of A) the Jacobi diagonal^-1
__global__ __launch_bounds__(256) void k_wrowtab(const double* __restrict__ ellA, const double* __restrict__ ellM, long long stride, long long rows,
                                                  double* __restrict__ rowA, double* __restrict__ rowM, int cls) {
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < rows * 16; t += (long long)gridDim.x * 256) {
        const long long row = t >> 4;
        const int q = (int)(t & 15);
        const long long e = 3 * row + cls;                   // column class: 0 = x-face 0, 1 = interior, 2 = x-face NX-1
        if (q < 15) {
            rowA[t] = ellA[q * stride + e];
            rowM[t] = ellM[q * stride + e];
        } else {
            const double d = ellA[e];
            rowA[t] = d > 0.0 ? 1.0 / d : 0.0;
            rowM[t] = 0.0;
        }
    }
}

// the rows / planes / columns with unknowns: Dirichlet faces -> the interior range; natural faces (directional conditions, value not given) -> the face too
void wmarch_ranges(const GridDev& g, int* jlo, int* jhi, int* klo, int* khi, int* xlo = nullptr, int* xhi = nullptr) {
    const bool dirn = g.bc_mode == PDEGPU_BC_DIRECTIONAL;
    // (a natural x-face has free nodes on its edges whatever the side faces are — T_side constrains 0 < x < Lx only, :614-617 — so every row and plane
    // is visited then and the kernels mask node by node with the Dirichlet predicate)
    const bool sides_free = dirn && (!g.bc_present[2] || !g.bc_present[0] || !g.bc_present[1]);
    *jlo = sides_free ? 0 : 1;
    *jhi = sides_free ? g.NY - 1 : g.NY - 2;
    *klo = sides_free ? 0 : 1;
    *khi = sides_free ? g.NZ - 1 : g.NZ - 2;
    if (xlo) *xlo = (dirn && !g.bc_present[0]) ? 0 : 1;
    if (xhi) *xhi = (dirn && !g.bc_present[1]) ? g.NX - 1 : g.NX - 2;
}

// one_round_cap > 0 (HBM-resident grids, wfast_one_round): the workgroups the device really holds at once (occupancy x CUs).  If some chunking fills at
// least 1.6 workgroups per CU within that, the one with the most workgroups is taken — ONE resident round; the cost model below counts rounds against
// wg_per_cu x CUs, but the hardware places what fits: 385^3 nodes, wave-uniform weights (occupancy 3), modelled as two rounds of 504 (7 chunks) is in fact
// one of 768 and a tail of 240.  Composite core 384^3: 7 chunks 74.2, 3 chunks (432 workgroups) 77.5, 5 chunks (720) 83.3 GDoF/s; spherical
// (occupancy 2) 7 chunks 78.4, 3 chunks 81.4.
WGeom wmarch_geom(const pdegpu_ctx* ctx, const GridDev& g, int RY, int wg_per_cu, int one_round_cap = 0) {
    WGeom w{};
    wmarch_ranges(g, &w.jlo, &w.jhi, &w.klo, &w.khi, &w.xlo, &w.xhi);
    w.ntx = (w.xhi + 1 + 127) / 128;          // tiles start at node 0 and must reach the last free column
    w.nty = (w.jhi - w.jlo + 1 + RY - 1) / RY;
    w.nwg_xy = (int)(((long long)w.ntx * w.nty + 3) / 4);
    const int planes = w.khi - w.klo + 1;
    // z-chunks: as few resident rounds of as short marches as possible (every chunk marches a two-plane prologue; CM 1 also loads its weights)
    const int target = std::max(1, wg_per_cu * ctx->num_cu);
    const int pcap = stencil_max_partials(const_cast<pdegpu_ctx*>(ctx));
    int best_c = 1;
    long long best = -1;
    const int maxc = std::max(1, planes / 4);
    for (int c = 1; c <= maxc; ++c) {
        if ((long long)w.nwg_xy * c > pcap && c > 1) break;
        const long long rounds = ((long long)w.nwg_xy * c + target - 1) / target;
        const long long cost = rounds * ((planes + c - 1) / c + 3);
        if (best < 0 || cost < best) {
            best = cost;
            best_c = c;
        }
    }
    if (one_round_cap > 0 && w.nwg_xy > 0) {
        const int c1 = (int)std::min<long long>(std::min(maxc, one_round_cap / w.nwg_xy), std::max(1, pcap / w.nwg_xy));
        if (c1 >= 1 && (long long)w.nwg_xy * c1 * 10 >= 16LL * ctx->num_cu) best_c = c1;
    }
    if (ctx->fast_zlen > 0) best_c = std::max(1, (planes + ctx->fast_zlen - 1) / ctx->fast_zlen);
    w.zlen = (planes + best_c - 1) / best_c;
    w.nzc = (planes + w.zlen - 1) / w.zlen;
    return w;
}

template <int CM, int RY, int LQ, bool FX, bool NT, bool EX>
int launch_wmarch_nt(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li) {
    static int occ[4] = {0, 0, 0, 0};      // (of the instantiation that carries x: the two forms of a mode share one geometry)
    if (occ[mode & 3] == 0) {
        int nb = 0;
        hipError_t e;
        switch (mode) {
            case APPLY_PLAIN: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_wmarch<CM, RY, APPLY_PLAIN, LQ, FX, NT, EX, 0>, 256, 0); break;
            case APPLY_RESIDUAL: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_wmarch<CM, RY, APPLY_RESIDUAL, LQ, FX, NT, EX, 0>, 256, 0); break;
            case APPLY_CG_K2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_wmarch<CM, RY, APPLY_CG_K2, LQ, FX, NT, EX, 1>, 256, 0); break;
            default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_wmarch<CM, RY, APPLY_CG_K1, LQ, FX, NT, EX, 1>, 256, 0); break;
        }
        if (e != hipSuccess || nb < 1) nb = 2;
        occ[mode & 3] = std::min(nb, 8);
    }
    // HBM-resident vectors: two workgroups per CU — fewer, longer-lived streams move more bytes per second on this memory system (the plain marching
    // kernels' finding, §4b of docs/history.md; here composite core 320^3 70.2 -> 71.1, 384^3 65.7 -> 73.0, spherical 384^3 61.7 -> 69.8 GDoF/s; 512^3 unchanged)
    const bool hbm = (long long)a.g.nkl * a.g.PP * 8 > (64LL << 20);
    const int wg = ctx->wfast_wg_per_cu > 0 ? ctx->wfast_wg_per_cu : (hbm ? std::min(occ[mode & 3], 2) : occ[mode & 3]);
    const int cap1 = (hbm && ctx->wfast_wg_per_cu <= 0 && ctx->wfast_one_round) ? occ[mode & 3] * ctx->num_cu : 0;
    const WGeom wgm = wmarch_geom(ctx, a.g, RY, wg, cap1);
    const int G = (wgm.nwg_xy * wgm.nzc + 7) / 8 * 8;      // a multiple of 8: the XCD-aware order always applies (surplus workgroups idle)
    if (li) { li->grid = G; li->block = 256; }
    {
        static const bool dbg = getenv("PDEGPU_DEBUG_GEOM") != nullptr;
        static int last[4] = {-1, -1, -1, -1};
        if (dbg && last[mode & 3] != G) {
            last[mode & 3] = G;
            fprintf(stderr, "[pdegpu wmarch] cm %d ry %d mode %d: %d x %d x %d nodes, %d x %d tiles, %d workgroups per chunk, %d chunks of %d planes, occ %d -> grid %d\n", CM, RY,
                    mode, a.g.NX, a.g.NY, a.g.NZ, wgm.ntx, wgm.nty, wgm.nwg_xy, wgm.nzc, wgm.zlen, occ[mode & 3], G);
        }
    }
    if (G == 0 || (li && li->dry)) return PDEGPU_OK;
    if (G > stencil_max_partials(ctx))         // one partial per workgroup (rows of more than ~80 000 x 2 tiles: nothing this device holds)
        return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "weighted marching kernels: %d workgroups exceed the %d partials of a session", G, stencil_max_partials(ctx));
    switch (mode) {
        case APPLY_PLAIN: hipLaunchKernelGGL((k_wmarch<CM, RY, APPLY_PLAIN, LQ, FX, NT, EX, 0>), dim3(G), dim3(256), 0, ctx->stream, a, wgm); break;
        case APPLY_RESIDUAL: hipLaunchKernelGGL((k_wmarch<CM, RY, APPLY_RESIDUAL, LQ, FX, NT, EX, 0>), dim3(G), dim3(256), 0, ctx->stream, a, wgm); break;
        case APPLY_CG_K2:
            if (a.xmode == 2) hipLaunchKernelGGL((k_wmarch<CM, RY, APPLY_CG_K2, LQ, FX, NT, EX, 1>), dim3(G), dim3(256), 0, ctx->stream, a, wgm);
            else hipLaunchKernelGGL((k_wmarch<CM, RY, APPLY_CG_K2, LQ, FX, NT, EX, 0>), dim3(G), dim3(256), 0, ctx->stream, a, wgm);
            break;
        default:
            if (a.x_in_k2) hipLaunchKernelGGL((k_wmarch<CM, RY, APPLY_CG_K1, LQ, FX, NT, EX, 0>), dim3(G), dim3(256), 0, ctx->stream, a, wgm);
            else hipLaunchKernelGGL((k_wmarch<CM, RY, APPLY_CG_K1, LQ, FX, NT, EX, 1>), dim3(G), dim3(256), 0, ctx->stream, a, wgm);
            break;
    }
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

// non-temporal hints where one vector of the session exceeds 64 MiB (wfast_nt: -1 auto, 0 never, 1 always)
template <int CM, int RY, int LQ, bool FX>
int launch_wmarch(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li) {
    int nt = ctx->wfast_nt;
    if (nt < 0) nt = ((long long)a.g.nkl * a.g.PP * 8 > (64LL << 20)) ? 1 : 0;
    // edge hand-off through LDS (wfast_ex: -1 auto, 0 off, 1 on).  Measured by kernel time, alternating processes on one box (K1, 193^3 / 257^3 / 385^3 / 513^3
    // nodes): per-lane weights 88.6 -> 80.5, 171 -> 162, 592 -> 572, 1283 -> 1172 us with it; wave-uniform weights 68.7 -> 66.3 us at 193^3 but 162 -> 165,
    // 568 -> 600, 1236 -> 1265 us on the HBM-resident grids (the barrier takes the four waves' freedom to run ahead of each other, which that form —
    // three waves per SIMD, scalar loads per plane step — uses): on for the former always, for the latter on cache-resident grids only
    int ex = ctx->wfast_ex;
    if (ex < 0) ex = (CM == 1 || (long long)a.g.nkl * a.g.PP * 8 <= (64LL << 20)) ? 1 : 0;
    if (ex) return nt ? launch_wmarch_nt<CM, RY, LQ, FX, true, true>(ctx, a, mode, li) : launch_wmarch_nt<CM, RY, LQ, FX, false, true>(ctx, a, mode, li);
    return nt ? launch_wmarch_nt<CM, RY, LQ, FX, true, false>(ctx, a, mode, li) : launch_wmarch_nt<CM, RY, LQ, FX, false, false>(ctx, a, mode, li);
}

}  // namespace

// a weighted session whose operator applications can march: 3-D, one slab owning every plane, at least one interior node per axis, and either every
// boundary node Dirichlet (the uniform boundary value; T_left, T_right and T_side all given) or — weights that do not depend on x (cm 2), whose row
// tables hold the truncated rows of the face nodes — any of the directional values missing (the cylinder fallback with T_left / T_right, whose T_side
// matches no facet, fenics_mcp_server.py:590-594; T_left alone; ...)
bool wmarch_supported(const GridDev& g, int cm) {
    if (g.dim != 3 || g.k0 != 0 || g.nkl != g.NZ || g.flat2d) return false;
    if (g.NX < 3 || g.NY < 3 || g.NZ < 3) return false;
    if ((long long)g.NZ * g.NY >= 0x7fffffffLL) return false;
    if (g.bc_mode == PDEGPU_BC_UNIFORM) return g.bc_present[0] != 0;
    if (cm == 2) return true;                              // any mix of T_left / T_right / T_side: natural faces come from the row / face tables
    return g.bc_present[0] && g.bc_present[1] && g.bc_present[2];
}

// a.ell / a.ell_stride / a.ell_mode (= CM) / a.wdinv describe the coefficients (see the header comment); mode: PLAIN, RESIDUAL, CG_K1
int wmarch_apply(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li) {
    if (mode != APPLY_PLAIN && mode != APPLY_RESIDUAL && mode != APPLY_CG_K1 && mode != APPLY_CG_K2)
        return pdegpu_set_err(ctx, PDEGPU_EINVAL, "weighted marching: PLAIN, RESIDUAL, K1 or K2");
    const int ry = ctx->wfast_ry;
    if (a.ell_mode == 1) return ctx->wfast_ry1 == 2 ? launch_wmarch<1, 2, 8, false>(ctx, a, mode, li) : launch_wmarch<1, 1, 0, false>(ctx, a, mode, li);
    if (a.wface[0] || a.wface[1]) return ry == 1 ? launch_wmarch<2, 1, 0, true>(ctx, a, mode, li) : launch_wmarch<2, 2, 0, true>(ctx, a, mode, li);
    if (ry == 1) return launch_wmarch<2, 1, 0, false>(ctx, a, mode, li);
    return launch_wmarch<2, 2, 0, false>(ctx, a, mode, li);
}

// CM 1: dtab = the diagonal (compact rows of A, interior plane class, weight 0), ditab = its inverse, both [j * PX + i]; CM 2: dtab = the row table of A
int wmarch_k2(pdegpu_ctx* ctx, const GridDev& g, int cm, double* r, const double* q, const double* dtab, const double* ditab, const double* flo,
              const double* fhi, double* scal, double* partials, LaunchInfo* li, const DotIn& di) {
    int jlo, jhi, klo, khi;
    wmarch_ranges(g, &jlo, &jhi, &klo, &khi);
    const int ny_free = jhi - jlo + 1;
    const int nrows = (khi - klo + 1) * ny_free;
    long long nb = ((long long)nrows + 3) / 4;
    const long long cap = (long long)(ctx->wfast_k2_mult > 0 ? ctx->wfast_k2_mult : 4) * ctx->num_cu;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    if (li) { li->grid = (int)nb; li->block = 256; }
    // z is read again by the next K1 and q was just written by this iteration's: on grids whose vectors stay in the caches the default policy keeps them there
    int nt = ctx->wfast_k2_nt;
    if (nt < 0) nt = ((long long)g.nkl * g.PP * 8 > (64LL << 20)) ? 1 : 0;
    if (cm == 1 && nt) hipLaunchKernelGGL((k_wk2<1, true>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, r, q, dtab, ditab, g, nrows, jlo, ny_free, klo, flo, fhi, scal, partials, di);
    else if (cm == 1) hipLaunchKernelGGL((k_wk2<1, false>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, r, q, dtab, ditab, g, nrows, jlo, ny_free, klo, flo, fhi, scal, partials, di);
    else if (nt) hipLaunchKernelGGL((k_wk2<2, true>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, r, q, dtab, ditab, g, nrows, jlo, ny_free, klo, flo, fhi, scal, partials, di);
    else hipLaunchKernelGGL((k_wk2<2, false>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, r, q, dtab, ditab, g, nrows, jlo, ny_free, klo, flo, fhi, scal, partials, di);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

// cls: the column class the table is made of (1 = interior: the wave-uniform rows; 0 / 2 = the nodes of the x-face 0 / NX-1)
int wmarch_rowtab(pdegpu_ctx* ctx, const GridDev& g, const double* ellA, const double* ellM, long long stride, double* rowA, double* rowM, int cls) {
    const long long rows = (long long)g.NZ * g.NY;
    long long nb = (rows * 16 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_wrowtab, dim3((unsigned)nb), dim3(256), 0, ctx->stream, ellA, ellM, stride, rows, rowA, rowM, cls);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}
