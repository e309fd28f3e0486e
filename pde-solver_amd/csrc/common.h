// common.h — internal declarations shared by the HIP translation units of libpdegpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/pdegpu.h"
#include "../../include/pdegpu_stepwise.h"

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct pdegpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    // tunables (pdegpu_set_tunable / PDEGPU_* environment at context creation)
    int fast_ry = 0;            // rows per thread of the z-marching kernels (1, 2, 4, 8); 0 = per mode: 4 for the one-vector passes
                                // (bare apply, residual), 2 for the CG kernels (round-2 sweeps, 512^3: SpMV 0.43 -> 0.38 ms)
    int fast_nt = -1;           // bit 0: non-temporal stores of q / p / x; bit 1: non-temporal loads of x; -1 = auto
                                // (3 when one slab vector exceeds 64 MiB, i.e. nothing stays cache-resident)
    int fast_ex = 1;            // 1: CG kernels hand tile-edge x-neighbours between the workgroup's waves through LDS
                                // instead of 2-lane global loads (measured: 1.65 -> 1.52 ms per 512^3 CG iteration)
    int fast_pf = -1;           // deeper prefetch (plane k+2 requested before plane k is computed): -1 = bare apply / residual pass only, 1 = every one-vector mode, 0 = off
    // per-mode overrides of the marching kernels, index = ApplyMode (PLAIN, RESIDUAL, CG_K1, CG_K2); -1 = follow the global tunables.
    // value = ry + 16 * wg_per_cu + 256 * ex + 512 * pf  (ry in {1, 2, 4, 8}, wg_per_cu 0 = occupancy API)
    int fast_cfg[4] = {-1, -1, -1, -1};
    int place_tune = 1;         // HBM-resident marching sessions: 1 = at creation K1 is timed with each of the six (still interchangeable) vectors as x and the
                                // fastest becomes x (K1's time depends on the physical placement of its read-modify-write stream), ~20 ms
    int fast_linear = 1;        // marching kernels: 1 = a workgroup takes four consecutive tiles of the row-major tile order (LDS edge hand-off at any tile count per row)
    int fast_zlen = 0;          // > 0: planes per z-chunk of the marching kernels (0 = chosen by fast_geom)
    int fast_minlen = 1;        // shortest z-chunk of the marching kernels on grids too small to fill the device
    int fast_small_mult = 1;    // such grids may launch this many resident rounds of workgroups
    int fast_wg_per_cu_k2 = 0;  // the same for the stencil K2 alone (0 = follow fast_wg_per_cu)
    int fast_wg_per_cu = 0;     // workgroups per CU the z-chunking aims for; 0 = 2 (+ 1/8 slack) when the vectors live in HBM — FEWER,
                                // longer-lived streams move more bytes per second on this memory system (512^3: K1 1.03 -> 0.92 ms,
                                // K2 0.62 -> 0.55 ms against the occupancy limit of 5-8) — and the occupancy API on cache-resident grids
    int grid_mult = 8;          // flat kernels: workgroups per CU
    int wfast_xdefer = -1;      // weighted marching sessions (-1 = where one vector exceeds 32 MiB): 1 = K2 applies the weighted stencil to the stored p again (z -= alpha D^-1 A p: no q) and updates x
                                // for two iterations at once, every other iteration, as the plain marching sessions do (cg_xdefer): 56 instead of 72 B/DoF
    int cg_xdefer = 1;          // marching sessions in the 64 B/DoF form: 1 = the solution vector is updated by K2 (which holds p_n, r_n and alpha_n) instead of
                                // the next K1, and only by every other K2 — the one in between records its alpha, the next forms both updates with
                                // p_{n-1} = (p_n - r_n) / beta_n: K1 24 B/DoF, K2 24 | 40 B/DoF = 56 instead of 64 B/DoF per iteration; 0 = x += alpha p in K1
    int cg_variant = 1;         // fast path: 0 = K1 stores q, flat K2 (72 B/DoF); 1 = K2 recomputes q = A p (64 B/DoF)
    int el_grid_mult = 4;       // elasticity kernels: workgroups per CU (each loads the 29 KB class table into LDS)
    int wfast = 1;              // weighted 3-D operators whose coefficient rows are compact (ell_compact) on an all-Dirichlet box: 1 = z-marching kernels with the
                                // weights in registers / SGPRs (wmarch.hip, 72 B/DoF per PCG iteration); 0 = class-table kernels reading the rows (88 B/DoF)
    int wfast_ry = 2;           // rows per wavefront column of the weighted marching kernels whose weights are wave-uniform (1 | 2; 192^3 composite core 53.7 -> 62.3 GDoF/s
                                // with 2); per-lane weights: always 1 (the 30 weights of a row pair leave no registers for a second row)
    int wfast_ry1 = 2;          // ... and where they are per-lane (r^2 sin(theta)): 2 = two rows, 8 of a node pair's 15 weights in LDS; 1 = one row, all in registers
    int wfast_ex = -1;          // their marching kernels: 1 = tile-edge x-neighbours handed between the workgroup's waves through LDS instead of 2-lane global loads;
                                // -1 = where measured faster (per-lane weights: always; wave-uniform weights: cache-resident grids)
    int wfast_nt = -1;          // their marching kernels: non-temporal stores of q / p / x and loads of x: 1 | 0; -1 = where one vector exceeds 64 MiB
    int wfast_k2_nt = -1;       // K2 of those sessions: non-temporal loads / stores 1 | 0; -1 = where one vector exceeds 64 MiB
    int wfast_k2_mult = 4;      // K2 of those sessions: workgroups per CU (a wavefront per grid row; 192^3 composite core: 57 -> 60 GDoF/s from 2 to 4)
    int wfast_one_round = 1;    // their z-chunking on HBM-resident grids: 1 = the chunking with the most workgroups that are all resident at once, where one exists
    int wfast_wg_per_cu = 0;    // workgroups per CU their z-chunking aims for; 0 = what the occupancy API reports
    int ell_compact = 1;        // weighted operators whose weight does not depend on one coordinate keep the coefficient rows of three representative
                                // planes / columns instead of one row per node (0: always one row per node)
    int el_fast = 1;            // 3-D elasticity: 1 = z-marching block stencil (k_el_fast); 0 = thread-per-node class-table kernel
    int el_fast_yx = 1;         // k_el_fast: 1 = the 4 waves of a workgroup are y-neighbours and hand their y-halo rows through LDS
    int el_fast_xface = 1;      // k_el_fast: 1 = the x-face nodes (interior in y, z) are computed by the marching part (edge lanes carry the
                                // face-class weights); 0 = by the class-table boundary pass (45 uncoalesced loads per node)
    int el_fast_yzface = 3;     // k_el_fast also computes: bit 0 the y-face nodes, bit 1 (with bit 0) the nodes of the planes 0 / NZ-1
    int el_fast_nt = -1;        // k_el_fast: non-temporal stores of the output: 1 | 0; -1 = where the vectors are HBM-resident
    int el_bnd_last = 0;        // k_el_fast: 1 = the boundary workgroups take the last block indices instead of the first
    int el_fast_ry = 2;         // rows per wavefront column of k_el_fast (1 | 2)
    int el_bnd_gran = 64;       // k_el_fast: boundary nodes a boundary workgroup takes per pass where they are few (edges and corners only): 64 | 256
    int el_fast_fit = 1;        // k_el_fast: 1 = a launch that would need between one and two resident rounds of workgroups is cut into as many z-chunks as fill
                                // ONE round, of two lengths (config 5: 19 chunks of 4 / 3 planes = 766 workgroups on 768 slots instead of 21 x 3 = 846)
    int el_xdefer = 1;          // single-reduction elasticity PCG of a whole solve: 1 = x += alpha p is left out of every other update pass and made up for by the
                                // next one from what that pass reads anyway (6 instead of 7 vector passes on average); 0 = x updated in every pass
    int el_cg_variant = 1;      // elasticity PCG: 1 = single-reduction form (K2 and p-update fused into one pass, one scalar reduction per iteration); 0 = classic
    int el_split_min_mb = 32;   // slab loop: K1 is split around the halo exchange where one vector of the slab (3 components) holds at least this many MiB
    int el_fast_waves = 0;      // wavefronts the marching part aims for (sets the z-chunk length); 0 = 4096, 8192 on HBM-resident grids
    int tile3d_max_nodes = 3000000;   // 3-D grids up to this many nodes (vectors resident in L2) use the one-round tile kernels; 0 = never
    int tile3d_plain = 1;       // tile sessions: the bare masked apply uses the tile form too
    int plane2d_ry = 4;         // rows per wavefront tile of k_plane2d (2 | 4 | 8)
    int snapshot_async = 1;     // transient solves: 1 = the snapshot of a step is copied to the host beside the next step (helper thread, second stream)
    int persist2d_cheb = 8;     // persistent 2-D CG, fresh solves: > 1 = Chebyshev polynomial preconditioner with that many terms (bands of <= 4 rows); 0 = plain CG
    int persist2d_cheb_ratio = 100;  // ... built on [lambda_max / ratio, lambda_max]
    int persist2d_sr = 1;       // persistent 2-D CG: 1 = fresh solves use the single-reduction (Chronopoulos-Gear) recurrences: one grid barrier +
                                // one neighbour hand-off per iteration instead of two grid barriers (same iterates in exact arithmetic; measured
                                // the same iteration counts, config 2: 13.3 -> 11.7 us per iteration); 0 = classic recurrences
    int persist2d = 1;          // 2-D grids up to 1024 cells wide and 8 rows per CU: the whole CG loop in ONE cooperative launch with the state
                                // in registers (persist2d.hip); 0 = one launch per kernel
    int persist2d_launch = 0;   // that launch: 0 = hipLaunchCooperativeKernel (the runtime checks co-residency), 1 = ordinary launch of the same grid (profiling runs)
    int plane2d = 1;            // 2-D grids: 1 = the two CG kernels load their tile in one round (k_plane2d); 0 = flat view of the marching kernels
    int hybrid_k2 = 1;          // hybrid sessions: 1 = 64 B/DoF variant as on Dirichlet grids (K1 stores no q, K2 applies the operator again: marching + boundary launch)
    int hybrid_fx = 1;          // hybrid sessions: 1 = the marching kernel computes the x-face rows too (the class-table pass keeps the y / z faces)
    int hybrid_bc = 1;          // 3-D grids with natural faces: 1 = marching kernel for the interior + class-table kernel for the boundary
                                // nodes; 0 = class-table kernel everywhere
    int extrapolate = 1;        // transient solves start CG from 2 u^n - u^{n-1} instead of u^n (fewer iterations, same tolerance)
    int lazy_dot = 1;           // single-slab CG loops: 1 = the next kernel's prologue sums the previous launch's partials (no reduce
                                // launches); 0 = one-workgroup k_reduce launch after every dot-producing kernel
    int direct_1d = 1;          // 1-D grids: 1 = Thomas recurrence inside one launch (tridiag.hip); 0 = the CG path (diagnostic)
    int k2_grid_mult = 2;       // K2: workgroups per CU (few, deep streams beat many shallow ones on mixed R/W)
    std::string err;
    char name[256] = {0};
};

int pdegpu_set_err(pdegpu_ctx* ctx, int code, const char* fmt, ...);

#define PG_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return pdegpu_set_err((ctx), PDEGPU_EHIP, "%s failed: %s (%s:%d)", #call,              \
                                  hipGetErrorString(e__), __FILE__, __LINE__);                     \
    } while (0)

#define PG_TRY(call)                                                                               \
    do {                                                                                           \
        int rc__ = (call);                                                                         \
        if (rc__ != PDEGPU_OK) return rc__;                                                        \
    } while (0)

// ------------------------------------------------------------------------------------------------
// slab-resident pitched grid.  Node (i, j, lk) of a vector v lives at v[lk*PP + j*PX + i], lk = local
// plane index in [-1, nkl] (-1 / nkl = ghost planes).  Rows are padded to PX (multiple of 16 doubles =
// 128 B); pad entries, ghost planes of boundary slabs and guards are zero and stay zero.
// ------------------------------------------------------------------------------------------------
struct GridDev {
    int dim;
    int NX, NY, NZ;        // global node counts (1 on unused axes)
    int PX;                // row pitch in doubles
    int nkl;               // owned planes of this slab (slowest axis)
    int k0;                // global index of local plane 0
    long long PP;          // plane pitch in doubles = PX*NY
    int bc_mode;           // PDEGPU_BC_*
    int bc_present[3];
    double bc_value[3];
    int flat2d;            // fast-kernel view of a 2-D grid: (x, y) -> (x, z) with ONE row in the middle axis, no y-boundary
};

static constexpr int kGuard = 64;   // extra doubles (beyond one row) before/after the ghost planes

struct DevVec {
    double* base = nullptr;    // allocation
    double* v = nullptr;       // origin: node (0,0,lk=0)
    size_t total = 0;          // doubles allocated
};


inline size_t vec_front(const GridDev& g) { return (size_t)g.PX + kGuard + (size_t)g.PP; }
inline size_t vec_total(const GridDev& g) { return 2 * ((size_t)g.PX + kGuard) + (size_t)(g.nkl + 2) * g.PP; }

__host__ __device__ inline bool pg_on_boundary(const GridDev& g, int i, int j, int kg) {
    bool b = (i == 0) | (i == g.NX - 1);
    if (g.dim >= 2) b |= (j == 0) | (j == g.NY - 1);
    if (g.dim >= 3) b |= (kg == 0) | (kg == g.NZ - 1);
    return b;
}

// Dirichlet set + value.  Mirrors the reference predicates (fenics_mcp_server.py:233-241, 373-376,
// 606-628) with exact index tests instead of near().
__host__ __device__ inline bool pg_dirichlet(const GridDev& g, int i, int j, int kg, double* val) {
    if (!pg_on_boundary(g, i, j, kg)) return false;
    if (g.bc_mode == PDEGPU_BC_UNIFORM) {
        if (!g.bc_present[0]) return false;
        *val = g.bc_value[0];
        return true;
    }
    if (i == 0) {
        if (!g.bc_present[0]) return false;
        *val = g.bc_value[0];
        return true;
    }
    if (i == g.NX - 1) {
        if (!g.bc_present[1]) return false;
        *val = g.bc_value[1];
        return true;
    }
    if (!g.bc_present[2]) return false;
    *val = g.bc_value[2];
    return true;
}

// boundary class of an index along one axis: 0 = low face, 1 = interior, 2 = high face
__host__ __device__ inline int pg_axis_class(int idx, int n_nodes) {
    return idx == 0 ? 0 : (idx == n_nodes - 1 ? 2 : 1);
}

// ------------------------------------------------------------------------------------------------
// class tables (tables.hip): integrated on the device from the macro-element matrices
// ------------------------------------------------------------------------------------------------
struct HeatTables {
    double* d_tabK = nullptr;   // [27][27]
    double* d_tabM = nullptr;   // [27][27]
    double* d_load = nullptr;   // [27]
    double* d_tabA = nullptr;   // [27][27] = cM*tabM + cK*tabK (filled by tables_combine)
    double h_tabK[27 * 27], h_tabM[27 * 27], h_load[27];
};
int tables_build(pdegpu_ctx* ctx, const pdegpu_grid* grid, HeatTables* t);
int tables_combine(pdegpu_ctx* ctx, HeatTables* t, double cM, double cK, double* d_out, double* h_out);
void tables_free(HeatTables* t);

// ------------------------------------------------------------------------------------------------
// kernels launched through host wrappers (stencil.hip / vecops.hip)
// ------------------------------------------------------------------------------------------------
enum ApplyMode { APPLY_PLAIN = 0, APPLY_RESIDUAL = 1, APPLY_CG_K1 = 2, APPLY_CG_K2 = 3 };

// device scalar block used by the CG kernels
// Single-reduction form (pdegpu_session_set_cg_variant 1): the block {SC_PQ, SC_RR, SC_QQ} = {p_n.Ap_n, r_n.r_n, Ap_n.Ap_n} is ONE 24-byte
// all-reduce message per iteration; SC_QQ takes the slot the classic form uses for r_{n-1}.r_{n-1}, SC_RR_POLL receives r.r of the
// residual the last K2 left (convergence polls only).
enum { SC_PQ_LOCAL = 0, SC_RR_LOCAL = 1, SC_PQ = 2, SC_RR = 3, SC_RR_PREV = 4, SC_QQ = 4, SC_BNORM = 5 /* reference norm of a solve, host-read only */,
       SC_RR_POLL = 6, SC_COUNT = 8 };

// where a CG kernel finds the dot product of the launch before it.  part == nullptr: already reduced, rr_n in
// scal[s_rr], rr_{n-1} in scal[s_rr_old], p.q in scal[SC_PQ] (the eager protocol of the multi-GPU driver: SC_RR /
// SC_RR_PREV).  part != nullptr: the n per-workgroup partials of the previous launch, summed in the kernel prologue
// (dot_finish.h); workgroup 0 records the sum (K1: rr_n -> scal[s_rr], a slot nobody reads in that launch; K2: -> SC_PQ).
struct DotIn {
    const double* part = nullptr;
    int n = 0;
    int s_rr = SC_RR;
    int s_rr_old = SC_RR_PREV;
    // single-reduction form: the scalars are the all-reduced block {SC_PQ, SC_RR, SC_QQ} of the PREVIOUS iteration (K1) / of this one (K2):
    //   alpha = r.r / p.Ap;   K1: beta = (alpha^2 Ap.Ap - r.r) / r.r  — the recurrence |r_{n+1}|^2 = alpha^2 |Ap|^2 - |r_n|^2 predicts the
    //   next r.r for beta only; the TRUE r.r of K2's residual joins the next message and forms the next alpha (part / n unused)
    int sr = 0;
    int skip_x = 0;           // sr: the pending x += alpha p has been applied by flush
};

struct StencilArgs {
    GridDev g;
    const double* tabA;       // device [27][27] class table (general path)
    unsigned off_mask = 0x7ffffffu;   // general path: bit o set = offset o has a non-zero weight in some class (others are never loaded)
    double w[8];              // interior coefficients: c0, cx, cy, cz, cxy, cxz, cyz, cxyz (fast path)
    const double* in;         // PLAIN/RESIDUAL: input vector;  CG_K1: r
    const double* rhs;        // RESIDUAL: right-hand side vector (nullable -> rhs_const)
    double rhs_const;         // RESIDUAL with rhs == nullptr: constant right-hand side per free node
    double in_shift;          // PLAIN: out = T (in + in_shift)
    double* out;              // PLAIN/RESIDUAL: output;  CG_K1: q
    const double* p_old;      // CG_K1
    double* p_new;            // CG_K1
    double* x;                // CG_K1 (updated in place)
    double* scal;             // device scalars (SC_*): read by K1 / K2
    double* partials;         // per-workgroup partial dot products
    double* partials2 = nullptr;   // CG_K1, single-reduction form: per-workgroup partials of (A p).(A p), same indexing
    int mask_dirichlet;       // 1: Dirichlet rows give 0 (P applied to the output)
    int npt;                  // 7 or 15 (fast path)
    DotIn di;                 // CG_K1 / CG_K2: how the previous launch's dot product arrives
    // hybrid path (3-D grids with natural faces): the marching kernel does the interior nodes, the class-table kernel the
    // boundary nodes of the slab, enumerated analytically: bnd_n > 0 -> one thread per boundary node instead of per index
    long long bnd_n = 0;      // boundary nodes of this slab
    int bnd_full_lo = 0, bnd_full_hi = 0;   // the slab owns global plane 0 / NZ-1 (every node of it is a boundary node)
    int bnd_lkb = 0;          // first local plane that is interior in z
    int bnd_ring = 0;         // boundary nodes of such a plane: 2 NX + 2 (NY - 2)
    int bnd_xcover = 0;       // columns [0, bnd_xcover) of interior rows get p / x from the marching kernel (128-node tiles)
    int fx_hi = 0;            // free_x == 2: the tiles reach the node NX-1 (no extra tile is spent on it: a fifth, one-node tile column
                              // per 513-node row was measured 27 % SLOWER than leaving that column to the class-table pass)
    int bnd_xcols = 2;        // boundary pass: x-face columns of the interior rows it still owns (2 both, 1 only x = NX-1, 0 none)
    int free_x = 0;           // marching kernel: 1 = the x-faces carry unknowns (their CG vectors are not zero: load them); 2 = and the
                              // x-face nodes of interior rows are computed by the marching kernel itself (edge lanes, halved in-face terms)
    // x_in_k2 sessions (marching kernels, K2 = the stencil on the stored p): K1 leaves x alone and records its beta in xd[1]; K2 either leaves x alone too
    // and records its alpha in xd[0] (xmode 0), or updates x for itself and for the K2 before it (xmode 2) — k_fast's XU template parameter
    int x_in_k2 = 0;
    int xmode = 0;
    double* xd = nullptr;
    int lk_lo = 0, lk_hi = 0; // marching kernel: restrict the launch to the local planes [lk_lo, lk_hi) (lk_hi > lk_lo; 0, 0 = all).  The
                              // multi-GPU loop launches K1 on the planes away from the ghosts first (halo exchange in flight)
    int lk_lo2 = 0, lk_hi2 = 0;   // ... and a second, disjoint plane range in the SAME launch (the two edge planes of a slab: one launch, not two)
    const double* ell;        // weighted operators: per-node coefficients, ell[q * ell_stride + idx], q = 15-offset position
    long long ell_stride;
    const double* wface[2] = {nullptr, nullptr};   // weighted marching kernels, weights independent of x: row tables of the x-face 0 / NX-1 where it is natural
    const double* wdinv = nullptr;   // weighted marching kernels, per-lane weights: Jacobi diagonal^-1 of the interior plane class, [j * PX + i]
    int ell_mode;             // 0: one row per node; 1: rows of three planes only (slowest index 0 | interior | last: the weight does not depend on that
                              // coordinate); 2: three rows per grid row (x index 0 | interior | last: the weight does not depend on x) — weighted.hip
};

struct LaunchInfo { int grid = 0; int block = 256; bool dry = false; /* marching kernels: report the grid, launch nothing */ };

int stencil_general(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li);
bool stencil_fast_supported(const GridDev& g, int element_mode);
bool stencil_all_dirichlet(const GridDev& g);
GridDev stencil_flat2d_view(const GridDev& g);
int stencil_fast(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li);
int stencil_max_partials(pdegpu_ctx* ctx);
int stencil_tile3d(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li);     // small 3-D grids: CG_K1 / CG_K2 as one-round tiles
long long stencil_tile3d_grid(const GridDev& g);
int stencil_plane2d(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li);   // 2-D: CG_K1 / CG_K2 without marching

int vec_alloc(pdegpu_ctx* ctx, const GridDev& g, DevVec* v);
void vec_free(DevVec* v);
int vec_zero(pdegpu_ctx* ctx, const GridDev& g, DevVec* v);
// x = Dirichlet value on constrained nodes, `fill` (or nodal[] when non-null, dense host-order device
// array of the slab) elsewhere
int vec_set_bc(pdegpu_ctx* ctx, const GridDev& g, double* x, double fill, const double* nodal_dense, int only_bc);
int vec_extrapolate(pdegpu_ctx* ctx, const GridDev& g, double* x, double* xprev, int first);   // x <- 2x - xprev; xprev <- x
int vec_axpy_flush(pdegpu_ctx* ctx, const GridDev& g, double* x, const double* p, const double* scal, int sr = 0);
int vec_k2(pdegpu_ctx* ctx, const GridDev& g, double* r, const double* q, double* scal, double* partials, LaunchInfo* li,
           const DotIn& di = DotIn());
// which: 0 p.q -> SC_PQ(_LOCAL); 1 r.r -> SC_RR_PREV <- SC_RR, SC_RR(_LOCAL); 2 r.r -> SC_RR(_LOCAL) only (SC_RR_PREV
// already holds the previous value: end of a lazy chunk whose current slot is SC_RR_PREV)
int vec_reduce(pdegpu_ctx* ctx, const double* partials, int n, double* scal, int which, int copy_to_global);
// single-reduction form: {SC_PQ, SC_QQ} <- sums of K1's two partial arrays (n1 each), SC_RR <- sum of K2's n3 partials (n3 < 0, the first
// iteration: <- SC_RR_LOCAL, this slab's share of r_0.r_0 from the residual pass); one launch
int vec_reduce_sr(pdegpu_ctx* ctx, const double* pq_part, const double* qq_part, int n1, const double* rr_part, int n3, double* scal);
int vec_reduce_slot(pdegpu_ctx* ctx, const double* partials, int n, double* scal, int slot);      // scal[slot] <- sum
int vec_norm2(pdegpu_ctx* ctx, const GridDev& g, const double* a, double* partials, LaunchInfo* li);
// weighted.hip
int weighted_assemble(pdegpu_ctx* ctx, const GridDev& g, const pdegpu_weight* w, const double* h, const double* lengths, double cM,
                      double cK, double* ellA, double* ellM, double* dinv, long long stride, int ell_mode);
int vec_k2_pc(pdegpu_ctx* ctx, const GridDev& g, double* r, const double* q, double* z, const double* dinv, double* scal,
              double* partials, int use_alpha, LaunchInfo* li, const DotIn& di = DotIn());
// wmarch.hip: z-marching kernels of the weighted 3-D operators
bool wmarch_supported(const GridDev& g, int cm);
int wmarch_apply(pdegpu_ctx* ctx, const StencilArgs& a, int mode, LaunchInfo* li);
int wmarch_k2(pdegpu_ctx* ctx, const GridDev& g, int cm, double* r, const double* q, const double* dtab, const double* ditab, const double* flo,
              const double* fhi, double* scal, double* partials, LaunchInfo* li, const DotIn& di = DotIn());
int wmarch_rowtab(pdegpu_ctx* ctx, const GridDev& g, const double* ellA, const double* ellM, long long stride, double* rowA, double* rowM, int cls);
int vec_membench(pdegpu_ctx* ctx, const GridDev& g, int which, double* r, const double* q, double* b, double* sink);
int vec_copy(pdegpu_ctx* ctx, const GridDev& g, double* dst, const double* src);
int vec_pack(pdegpu_ctx* ctx, const GridDev& g, const double* v, double* dense);     // pitched -> dense
int vec_unpack(pdegpu_ctx* ctx, const GridDev& g, const double* dense, double* v);   // dense -> pitched

// tridiag.hip: direct solver of the 1-D tools
int tridiag_setup(pdegpu_ctx* ctx, const GridDev& g, const double* tabA, const double* tabM, const double* ellA,
                  const double* ellM, long long ell_stride, double* rows /*device [8][NX]*/);
int tridiag_run(pdegpu_ctx* ctx, int N, const double* rows, double* work /*device [3][N]*/, double* u, double shift, int rhs_given,
                int nsteps, int step0, int total_steps, int stride, double* snaps);

// persist2d.hip: the CG loop of a 2-D system as one cooperative launch (state in registers, two grid barriers per iteration)
struct Persist2d;
bool persist2d_supported(const pdegpu_ctx* ctx, const GridDev& g);
int persist2d_create(pdegpu_ctx* ctx, const GridDev& g, Persist2d** out);
int persist2d_run(pdegpu_ctx* ctx, Persist2d* p, const GridDev& g, const double* w5, double* x, double* r, const double* p_in, double* p_out,
                  double* scal, int iters_max, double target, int* iters_done, int fresh);
void persist2d_free(Persist2d* p);

// comm.hip: the library's RCCL communicator (halo exchange on its own stream, scalar all-reduce on the compute stream)
struct pdegpu_comm;
int comm_rank(const pdegpu_comm* c);
int comm_world(const pdegpu_comm* c);
bool comm_self_loop(const pdegpu_comm* c);               // rehearsal on one GPU: the slab's neighbours are this rank itself
int comm_ready(pdegpu_comm* c, hipStream_t compute);     // the planes to send are final: the exchange stream may start
int comm_group_start(pdegpu_comm* c);
int comm_group_end(pdegpu_comm* c);
int comm_send(pdegpu_comm* c, const double* buf, size_t n, int peer);
int comm_recv(pdegpu_comm* c, double* buf, size_t n, int peer);
int comm_done(pdegpu_comm* c);                           // marks the end of the exchange posted so far
int comm_wait(pdegpu_comm* c, hipStream_t compute);      // later launches on `compute` see the received planes
int comm_allreduce(pdegpu_comm* c, double* buf, size_t n, hipStream_t compute);

int next_poll_chunk(double rr_before, double rr_after, double target, int n_done, int chunk, int cap);   // heat.hip
GridDev make_grid_dev(const pdegpu_grid* grid, const pdegpu_heat* heat, int64_t k0, int64_t nk);
int validate_grid(pdegpu_ctx* ctx, const pdegpu_grid* grid);
