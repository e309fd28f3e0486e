// heat.hip — slab-resident CG session, the one-shot heat solve and the C-ABI entry points of the heat path.
//
// Host-side restatement of the reference's solver bodies (fenics_mcp_server.py:259-318, 391-447, 653-716):
//   steady:     kappa K u = f m                          (one solve, times = [0])
//   transient:  (M + dt kappa K) u^{n+1} = M u^n + dt f m (backward Euler, snapshot per step)
// with Dirichlet rows treated as identity rows.  Where the reference assembles a sparse matrix and runs a
// sparse LU every step, this drives a matrix-free CG on full-grid vectors: Dirichlet nodes carry their
// value in x and zeros in r, p, q, so the lifting A_IB g is exactly the initial residual and no separate
// lifting pass exists.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cctype>
#include <cstdlib>
#include <future>

struct pdegpu_session {
    pdegpu_ctx* ctx = nullptr;
    pdegpu_grid grid{};
    pdegpu_heat heat{};
    GridDev g{};
    GridDev gk{};                  // the grid as the fused kernels see it (= g in 3-D; flat view of a 2-D grid)
    HeatTables tabs{};
    double cM = 0, cK = 0;
    bool fast = false;
    bool tile3d = false;           // small 3-D grid: K1 / K2 through k_tile3d (one load round per tile) instead of marching
    bool plane2d = false;          // 2-D: K1 / K2 through k_plane2d (one load round per tile) instead of the flat view
    bool hybrid = false;           // fast && some boundary node is free: marching kernel (interior) + class-table kernel (boundary nodes)
    int npt = 15;
    double hA[27 * 27];
    DevVec x, r, p[2], q, b;
    double place_ms_default = 0.0, place_ms_best = 0.0;      // tune_x_placement: K1 with the default buffer as x / with the chosen one
    int place_choice = 0;          // which of the six allocations became x (0 = the default)
    DevVec xprev;                  // u^{n-1} of a transient solve (allocated on first use)
    int pcur = 0;
    double* d_scal = nullptr;
    double* d_part = nullptr;       // [3][npart]: per-workgroup partials; the second block takes K2's (lazy loop, single-reduction form), the third K1's Ap.Ap
    int npart = 0;
    double* d_dense = nullptr;
    double* h_pin = nullptr;       // pinned host scalars
    std::vector<hipEvent_t> events;
    int last_iters = 0;            // CG iterations of the previous linear solve of this session (first poll of the next one)
    // per-call kernel timing of the stepwise protocol (pdegpu_session_set_timing): event pairs around every k1 part / k2 launch
    bool timing = false;
    int timing_every = 1;          // event pairs on every timing_every-th iteration only (the records themselves cost stream time at high launch rates)
    long long timing_iter = 0;     // iterations seen since set_timing
    bool timing_now = false;       // the current iteration is a sampled one
    std::vector<hipEvent_t> tev;   // pairs: [2 i] start, [2 i + 1] stop
    std::vector<int> tev_kind;     // 0 K1 (a split K1 contributes several pairs), 1 K2
    size_t tev_used = 0;
    long long timed_k1 = 0, timed_k2 = 0;
    int k1_parts = 0;              // workgroups (partials) launched by the parts of a split K1 so far
    int k2_parts = 0;              // the same for a split K2
    // single-reduction form of the stepwise protocol (pdegpu_session_set_cg_variant 1): d_part = [p.Ap | r.r of K2 | Ap.Ap], npart each
    int variant = 0;
    bool sr_k2_pending = false;    // K2's r.r partials (d_part + npart) have not been summed into SC_RR yet: the next K1's reduce launch does
    int sr_k2_grid = 0;
    bool sr_skip_x = false;        // flush has applied the pending x += alpha p: the next K1 must not
    // x_in_k2 (tunable cg_xdefer; marching sessions whose K2 is the stencil on the stored p): the solution vector is updated by K2, and only by
    // every other one — the K2 in between records its alpha, the next forms both updates from what it reads anyway (k_fast, template XU)
    bool xk2 = false;
    bool xk2_pending = false;      // the latest K2 left its x += alpha p out (flush, or the next K2, makes up for it)
    int xk2_mode = 0;              // StencilArgs.xmode of the current iteration's K2 launches (0 leave out + record, 2 update for two)
    long long timed_k2x = 0;       // timed K2 launches that carried the x update (byte accounting)
    double* d_xd = nullptr;        // {alpha of the K2 that left its update out, beta of the latest K1}
    int rr_slot = SC_RR;           // the scalar pdegpu_session_rr reports (SC_RR_POLL after pdegpu_session_sr_poll)
    Persist2d* p2d = nullptr;      // 2-D sessions: register-resident persistent CG (created on first use)
    // weighted (curvilinear) operators: per-node coefficients + Jacobi-PCG
    bool weighted = false;
    double *ellA = nullptr, *ellM = nullptr, *dinv = nullptr;   // origins (same indexing as the vectors)
    double *ell_base[3] = {nullptr, nullptr, nullptr};
    DevVec z;
    long long ell_stride = 0;
    int ell_mode = 0;              // StencilArgs.ell_mode of the coefficient arrays
    // weighted operators on an all-Dirichlet 3-D box with compact coefficient rows: z-marching kernels (wmarch.hip), no z vector
    bool wfast = false;
    double *rowA = nullptr, *rowM = nullptr;   // ell_mode 2: row tables [NZ * NY][16] (15 weights of the interior column class + D^-1)
    double *rowAf[2] = {nullptr, nullptr}, *rowMf[2] = {nullptr, nullptr};   // ... and of the column classes x = 0 / x = NX-1 where that face is natural
};

// ---------------------------------------------------------------------------------------------
// grid helpers (no GPU needed)
// ---------------------------------------------------------------------------------------------
extern "C" int64_t pdegpu_grid_num_nodes(const pdegpu_grid* grid) {
    if (!grid || grid->dim < 1 || grid->dim > 3) return -1;
    int64_t n = 1;
    for (int r = 0; r < grid->dim; ++r) {
        if (grid->n[r] < 1) return -1;
        n *= (int64_t)grid->n[r] + 1;
    }
    return n;
}

// contiguous, near-equal node-plane ranges; every plane has exactly one owner (dot products are never
// double counted), lower ranks take the remainder.
extern "C" int pdegpu_slab_partition(int64_t n_planes, int n_ranks, int rank, int64_t* k0, int64_t* nk) {
    if (n_planes < 1 || n_ranks < 1 || rank < 0 || rank >= n_ranks || !k0 || !nk) return PDEGPU_EINVAL;
    const int64_t base = n_planes / n_ranks, rem = n_planes % n_ranks;
    *nk = base + (rank < rem ? 1 : 0);
    *k0 = rank * base + std::min<int64_t>(rank, rem);
    return PDEGPU_OK;
}

extern "C" int pdegpu_heat_num_snapshots(const pdegpu_heat* h) {
    if (!h) return PDEGPU_EINVAL;
    if (h->steady) return 1;
    const int stride = h->snapshot_stride < 1 ? 1 : h->snapshot_stride;
    int n = 0;
    for (int s = 0; s <= h->num_steps; ++s)
        if (s % stride == 0 || s == h->num_steps) ++n;
    return n;
}

// Host-only (no GPU): the Dirichlet set the kernels use for this grid and these boundary arguments, node order x fastest — lets the CPU
// test-suite compare it with the oracle's restatement of the reference predicates (fenics_mcp_server.py:233-241, 373-376, 606-628).
extern "C" int pdegpu_dirichlet_set_host(const pdegpu_grid* grid, const pdegpu_heat* heat, uint8_t* mask_out, double* value_out) {
    if (!grid || !heat || !mask_out || grid->dim < 1 || grid->dim > 3) return PDEGPU_EINVAL;
    for (int r = 0; r < grid->dim; ++r)
        if (grid->n[r] < 1) return PDEGPU_EINVAL;
    const int64_t nslow = grid->dim == 3 ? (int64_t)grid->n[2] + 1 : 1;
    const GridDev g = make_grid_dev(grid, heat, 0, nslow);
    size_t e = 0;
    for (int k = 0; k < g.NZ; ++k)
        for (int j = 0; j < g.NY; ++j)
            for (int i = 0; i < g.NX; ++i, ++e) {
                double v = 0.0;
                const bool d = pg_dirichlet(g, i, j, k, &v);
                mask_out[e] = d ? 1 : 0;
                if (value_out) value_out[e] = d ? v : 0.0;
            }
    return PDEGPU_OK;
}

int validate_grid(pdegpu_ctx* ctx, const pdegpu_grid* grid) {
    if (!grid) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "grid is null");
    if (grid->dim < 1 || grid->dim > 3) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "dim must be 1, 2 or 3 (got %d)", grid->dim);
    for (int r = 0; r < grid->dim; ++r) {
        if (grid->n[r] < 1) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "n[%d] must be >= 1 (got %d)", r, grid->n[r]);
        if (!(grid->L[r] > 0.0) || !std::isfinite(grid->L[r]))
            return pdegpu_set_err(ctx, PDEGPU_EINVAL, "L[%d] must be a positive finite length", r);
    }
    if (grid->element_mode != PDEGPU_ELEM_P1SPLIT && grid->element_mode != PDEGPU_ELEM_Q1)
        return pdegpu_set_err(ctx, PDEGPU_EINVAL, "unknown element_mode %d", grid->element_mode);
    return PDEGPU_OK;
}

GridDev make_grid_dev(const pdegpu_grid* grid, const pdegpu_heat* heat, int64_t k0, int64_t nk) {
    GridDev g{};
    g.dim = grid->dim;
    g.NX = grid->n[0] + 1;
    g.NY = grid->dim >= 2 ? grid->n[1] + 1 : 1;
    g.NZ = grid->dim >= 3 ? grid->n[2] + 1 : 1;
    g.PX = (g.NX + 15) / 16 * 16;
    g.PP = (long long)g.PX * g.NY;
    g.k0 = (int)k0;
    g.nkl = (int)nk;
    if (heat) {
        g.bc_mode = heat->bc_mode;
        for (int r = 0; r < 3; ++r) {
            g.bc_present[r] = heat->bc_present[r];
            g.bc_value[r] = heat->bc_value[r];
        }
        // DirichletBC marks FACETS (DOLFIN's default method "topological": a boundary facet is taken when the predicate holds at all of
        // its vertices and at its midpoint; every dof of a taken facet is constrained).  For `other_faces` (fenics_mcp_server.py:614-617,
        // "on the boundary and 0 < x < Lx") a side facet qualifies only if none of its vertices lies on x = 0 or x = Lx: with nx <= 2 cells
        // every side facet touches one of the two, so T_side constrains NOTHING there; from nx = 3 on every side node with 0 < x < Lx is a
        // vertex of a qualifying facet and the per-node test of pg_dirichlet is exact.  left_face / right_face / the uniform predicate mark
        // whole faces and agree with their node sets at any size.
        if (g.bc_mode == PDEGPU_BC_DIRECTIONAL && grid->dim >= 2 && grid->n[0] <= 2) g.bc_present[2] = 0;
    } else {
        g.bc_mode = PDEGPU_BC_DIRECTIONAL;   // nothing constrained
    }
    return g;
}

namespace {

// offsets that carry a non-zero weight in at least one boundary class of a [27][27] class table (k_general loads only those)
unsigned table_mask(const double* tab) {
    unsigned m = 0;
    for (int c = 0; c < 27; ++c)
        for (int o = 0; o < 27; ++o)
            if (tab[c * 27 + o] != 0.0) m |= 1u << o;
    return m;
}

void fast_weights(const double* tab, double* w, bool flat2d = false) {
    const double* c = tab + 13 * 27;   // interior class
    w[0] = c[13]; w[1] = c[14]; w[2] = c[16]; w[3] = c[22];
    w[4] = c[17]; w[5] = c[23]; w[6] = c[25]; w[7] = c[26];
    if (flat2d) {                      // (x, y) -> (x, z): y-coupling moves to the z slot, the xy-diagonal to the xz slot
        w[3] = w[2]; w[5] = w[4];
        w[2] = w[4] = w[6] = w[7] = 0.0;
    }
}

// in-plane weights of a 2-D interior stencil for k_plane2d: w0, wx, wy, wd1 = (+1,+1), wd2 = (-1,+1)
void plane2d_weights(const double* tab, double* w) {
    const double* c = tab + 13 * 27;
    w[0] = c[13]; w[1] = c[14]; w[2] = c[16]; w[3] = c[17]; w[4] = c[15];
    w[5] = w[6] = w[7] = 0.0;
}
bool plane2d_symmetric(const double* tab) {       // the kernel assumes w(-o) = w(o)
    const double* c = tab + 13 * 27;
    const double tol = 1e-13 * std::fabs(c[13]);
    return std::fabs(c[12] - c[14]) <= tol && std::fabs(c[10] - c[16]) <= tol && std::fabs(c[9] - c[17]) <= tol && std::fabs(c[11] - c[15]) <= tol;
}

// Q1 interior stencil: the weight of an offset depends on |dx|, |dy|, |dz| only (checked before the 27-point marching
// kernel, which assumes it, is used)
bool is_sign_symmetric(const double* c /*27 class-13 weights*/) {
    double scale = std::fabs(c[13]);
    for (int o = 0; o < 27; ++o) {
        const int dx = o % 3 - 1, dy = (o / 3) % 3 - 1, dz = o / 9 - 1;
        const int rep = (std::abs(dx) + 1) + 3 * (std::abs(dy) + 1) + 9 * (std::abs(dz) + 1);
        if (std::fabs(c[o] - c[rep]) > 1e-13 * scale) return false;
    }
    return true;
}

bool is_seven_point(const double* w) {
    const double tol = 1e-12 * std::fabs(w[0]);
    return std::fabs(w[4]) <= tol && std::fabs(w[5]) <= tol && std::fabs(w[6]) <= tol && std::fabs(w[7]) <= tol;
}

// hybrid sessions: second launch of an operator application — the boundary nodes of the slab through the class-table
// kernel (same mode, same arguments), its partials appended to the marching kernel's
int hybrid_boundary(pdegpu_session* s, StencilArgs a, int mode, LaunchInfo* li_fast) {
    const GridDev& g = s->g;
    a.g = g;
    a.bnd_full_lo = g.k0 == 0 ? 1 : 0;
    a.bnd_full_hi = g.k0 + g.nkl == g.NZ ? 1 : 0;
    a.bnd_lkb = a.bnd_full_lo;
    const int planes = std::max(0, g.nkl - a.bnd_full_lo - a.bnd_full_hi);
    // free_x == 2: the marching kernel has done the x-face nodes of the interior rows — x = 0 always, x = NX-1 when its tiles reach
    // it (a.fx_hi); otherwise that one column stays here
    const bool fx = a.free_x == 2;
    a.bnd_xcols = fx ? (a.fx_hi ? 0 : 1) : 2;
    // (Round 3 computed that column in the marching K1 / K2 instead — the last lane of a row's last tile holds its value and neighbours as
    // tile-edge values — parity-green and SLOWER: the edge values of all rows of three planes cost the 7-point kernels 134 -> 182 (K1) and
    // 109 -> 138 (K2) VGPRs; 512^3, T_left only: 85-87 -> 67-71 GDoF/s; removed, profiles/r03_natural_face_column_in_marching_kernel.log.)
    // The column x = NX-1 that the tiles do not reach (2^k * 128 + 1 nodes per row) is the expensive part of this pass — one node per 4 KB row, 57 of
    // its 97 us in K1 and 32 of 56 us in K2 at 512^3 (profiles/r02_exp20_*.log).  Where that face is Dirichlet there is nothing to compute on it:
    // the CG vectors r, p vanish there and stay so (only these two passes write them: K1 would store p = 0 + beta 0, the in-place K2 r = 0), x keeps
    // its boundary value; a stored q is zero there since pdegpu_session_begin and nothing but K1 writes it between solves (lift_norm clears its use).
    // So K1 and the in-place K2 leave the column out; every other pass (residual, bare apply) still visits it.
    const bool cg_pass = g.bc_mode == PDEGPU_BC_DIRECTIONAL && a.mask_dirichlet && (mode == APPLY_CG_K1 || (mode == APPLY_CG_K2 && a.out == a.rhs));
    if (fx && a.bnd_xcols == 1 && cg_pass && g.bc_present[1]) a.bnd_xcols = 0;
    a.bnd_ring = 2 * g.NX + a.bnd_xcols * (g.NY - 2);
    a.bnd_xcover = std::min(g.PX, ((g.NX - 1 + 127) / 128) * 128);
    a.bnd_n = (long long)(a.bnd_full_lo + a.bnd_full_hi) * g.NX * g.NY + (long long)planes * a.bnd_ring;
    if (a.bnd_n <= 0) return PDEGPU_OK;
    if (a.bnd_n > 0x7fffffffLL) return pdegpu_set_err(s->ctx, PDEGPU_EUNSUPPORTED, "hybrid path: more than 2^31 boundary nodes");
    const int n1 = li_fast ? li_fast->grid : 0;
    a.partials = a.partials + n1;                 // appended to the marching launch's (K2: wherever the caller keeps its r.r partials)
    if (a.partials2) a.partials2 = a.partials2 + n1;
    LaunchInfo l2;
    PG_TRY(stencil_general(s->ctx, a, mode, &l2));
    if (li_fast) li_fast->grid = n1 + l2.grid;
    return PDEGPU_OK;
}

// weighted marching sessions: where the kernels of wmarch.hip find the coefficients of A (which == 0) or M (which == 1)
void wfast_args(const pdegpu_session* s, int which, StencilArgs* a) {
    a->g = s->g;
    a->ell_mode = s->ell_mode;
    a->ell_stride = s->ell_stride;
    if (s->ell_mode == 1) {                    // compact rows of the plane class "interior" (entry e = PP + j * PX + i), D^-1 of plane 1
        a->ell = (which == 0 ? s->ellA : s->ellM) + s->g.PP;
        a->wdinv = s->dinv + s->g.PP;
    } else {
        a->ell = which == 0 ? s->rowA : s->rowM;
        a->wdinv = nullptr;
        for (int f = 0; f < 2; ++f) a->wface[f] = which == 0 ? s->rowAf[f] : s->rowMf[f];
    }
}

// out = T (in + shift)   with T = A (which==0) or M (which==1); unmasked unless mask
int op_plain(pdegpu_session* s, int which, const double* in, double shift, double* out, int mask) {
    StencilArgs a{};
    a.g = s->g;
    a.in = in;
    a.in_shift = shift;
    a.out = out;
    a.partials = s->d_part;
    a.scal = s->d_scal;
    a.mask_dirichlet = mask;
    const double* htab = which == 0 ? s->hA : s->tabs.h_tabM;
    a.tabA = which == 0 ? s->tabs.d_tabA : s->tabs.d_tabM;
    a.off_mask = table_mask(htab);
    if (s->wfast) {                            // interior rows only: every other node is Dirichlet and no consumer reads its entry
        wfast_args(s, which, &a);
        return wmarch_apply(s->ctx, a, APPLY_PLAIN, nullptr);
    }
    if (s->weighted) {
        a.ell = which == 0 ? s->ellA : s->ellM;
        a.ell_stride = s->ell_stride;
        a.ell_mode = s->ell_mode;
    }
    if (s->tile3d && mask && s->ctx->tile3d_plain) {            // bare SpMV of a tile session (masked: Dirichlet rows give 0)
        a.g = s->g;
        fast_weights(htab, a.w, false);
        a.npt = s->grid.element_mode == PDEGPU_ELEM_Q1 ? 27 : (is_seven_point(a.w) ? 7 : 15);
        return stencil_tile3d(s->ctx, a, APPLY_PLAIN, nullptr);
    }
    if (s->fast) {
        a.g = s->gk;
        fast_weights(htab, a.w, s->gk.flat2d != 0);
        a.npt = s->grid.element_mode == PDEGPU_ELEM_Q1 ? 27 : (is_seven_point(a.w) ? 7 : 15);
        a.free_x = s->hybrid ? (s->ctx->hybrid_fx ? 2 : 1) : 0;
        a.fx_hi = ((s->g.NX - 1) % 128) != 0 ? 1 : 0;
        LaunchInfo lf;
        PG_TRY(stencil_fast(s->ctx, a, APPLY_PLAIN, &lf));
        return s->hybrid ? hybrid_boundary(s, a, APPLY_PLAIN, &lf) : PDEGPU_OK;
    }
    return stencil_general(s->ctx, a, APPLY_PLAIN, nullptr);
}

// out = P (rhs - A in), partials of out.out
int op_residual(pdegpu_session* s, const double* in, const double* rhs, double* out, LaunchInfo* li) {
    StencilArgs a{};
    a.g = s->g;
    a.in = in;
    a.rhs = rhs;
    a.out = out;
    a.partials = s->d_part;
    a.scal = s->d_scal;
    a.mask_dirichlet = 1;
    a.tabA = s->tabs.d_tabA;
    a.off_mask = table_mask(s->hA);
    if (s->wfast) {                            // the partials are those of r . D^-1 r already
        wfast_args(s, 0, &a);
        return wmarch_apply(s->ctx, a, APPLY_RESIDUAL, li);
    }
    if (s->weighted) {
        a.ell = s->ellA;
        a.ell_stride = s->ell_stride;
        a.ell_mode = s->ell_mode;
    }
    if (s->fast) {
        a.g = s->gk;
        fast_weights(s->hA, a.w, s->gk.flat2d != 0);
        a.npt = s->npt;
        a.free_x = s->hybrid ? (s->ctx->hybrid_fx ? 2 : 1) : 0;
        a.fx_hi = ((s->g.NX - 1) % 128) != 0 ? 1 : 0;
        PG_TRY(stencil_fast(s->ctx, a, APPLY_RESIDUAL, li));
        return s->hybrid ? hybrid_boundary(s, a, APPLY_RESIDUAL, li) : PDEGPU_OK;
    }
    return stencil_general(s->ctx, a, APPLY_RESIDUAL, li);
}

// 64 B/DoF variant (K2 recomputes q with the marching stencil): pays for the 7-point operator only — with the
// 15-point operator the second stencil pass costs more than the 8 B/DoF it saves (measured)
// Hybrid sessions too (tunable hybrid_k2): their second stencil pass has a boundary launch of its own — on a slab that owns every plane
// only: K1 forms p_n on the ghost planes for the rows the marching kernel owns, not for the boundary rows, so a slab with neighbours
// keeps the 72-B form (q stored, r and p exchanged).  It also needs hybrid_fx: K2 updates r in place, and the marching kernel stores 0 at
// the nodes of its rows it does not compute — right for Dirichlet nodes, fatal for x-face nodes that the boundary launch still has to read.
bool k2_is_stencil(const pdegpu_session* s) {
    const bool whole = s->g.k0 == 0 && s->g.nkl == s->g.NZ;
    const bool hyb_ok = s->ctx->hybrid_k2 && s->ctx->hybrid_fx && !s->weighted && whole;
    // the 15-point operator (transient steps): the second stencil pass costs about what the 8 B/DoF of q save (1.535 vs 1.555 ms at 512^3) — but it is
    // what lets K2 carry the x update of every other iteration (cg_xdefer: 56 instead of 72 B/DoF, 81.9 -> 98.6 GDoF/s; natural-face sessions
    // 82.8 -> 92.5 / 79.4 -> 85.7), so marching sessions take it by default; the tile forms keep the stored q unless cg_variant = 2 asks
    // (the 27-point operator of the Q1 element mode alike: its K1 stored q too)
    const bool march15 = (s->npt == 15 || s->npt == 27) && s->ctx->cg_xdefer != 0 && !s->tile3d && !s->plane2d && !s->weighted;
    return s->fast && (!s->hybrid || hyb_ok) && ((s->ctx->cg_variant == 1 && (s->npt == 7 || march15)) || (s->ctx->cg_variant == 2 && (s->npt == 7 || s->npt == 15)));
}

// x_in_k2: pure marching sessions (no boundary launch, no tile forms, no weights) in the 64 B/DoF form
bool xk2_eligible(const pdegpu_session* s) {
    // (hybrid sessions: both launches of K1 and of K2 — marching kernel and boundary pass — follow the same protocol)
    if (s->ctx->cg_xdefer == 0 || !s->d_xd) return false;
    // weighted operators whose applications march (wmarch.hip): K2 applies the weighted stencil to the stored p again (z -= alpha D^-1 A p) and carries x
    // (-1: where a vector exceeds 32 MiB — 128^3 cells, everything cache-resident: 44 -> 40 GDoF/s, the second stencil pass costs more than the bytes it
    // saves; 192^3: 54 -> 58, 256^3: 61 -> 77, 512^3: 71 -> 93)
    if (s->wfast) return s->ctx->wfast_xdefer < 0 ? (long long)s->g.nkl * s->g.PP * 8 > (32LL << 20) : s->ctx->wfast_xdefer != 0;
    return k2_is_stencil(s) && s->fast && !s->tile3d && !s->plane2d && !s->weighted;
}
// before the K2 launch(es) of an iteration: leave x out (and record alpha) or update it for this iteration and the one before, alternating
void xk2_begin(pdegpu_session* s) {
    if (!s->xk2) return;
    s->xk2_mode = s->xk2_pending ? 2 : 0;
    s->xk2_pending = !s->xk2_pending;
}
// a new solve (reset_cg / begin / residual): nothing pending, the recorded scalars void
int xk2_reset(pdegpu_session* s) {
    s->xk2 = xk2_eligible(s);
    s->xk2_pending = false;
    s->xk2_mode = 0;
    if (s->d_xd) PG_HIP(s->ctx, hipMemsetAsync(s->d_xd, 0, 4 * sizeof(double), s->ctx->stream));
    return PDEGPU_OK;
}

// lk_lo < lk_hi: one part of a split K1 (marching sessions only) — the launch covers the local planes [lk_lo, lk_hi), appends its
// partials at part_off and leaves the p ping-pong alone (the caller flips it after the last part)
// single-reduction form: how K1 / K2 read the all-reduced block {SC_PQ, SC_RR, SC_QQ} (dot_finish.h)
DotIn sr_dot_in(const pdegpu_session* s) {
    DotIn d;
    d.sr = 1;
    d.skip_x = s->sr_skip_x ? 1 : 0;
    return d;
}

int op_k1(pdegpu_session* s, LaunchInfo* li, const DotIn& di = DotIn(), int lk_lo = 0, int lk_hi = 0, int part_off = 0, int lk_lo2 = 0, int lk_hi2 = 0) {
    StencilArgs a{};
    a.di = di;
    if (di.sr) a.partials2 = s->d_part + 2 * (size_t)s->npart + part_off;
    a.g = s->g;
    a.in = s->weighted ? s->z.v : s->r.v;          // PCG: p = z + beta p with z = D^-1 r
    if (s->weighted) {
        a.ell = s->ellA;
        a.ell_stride = s->ell_stride;
        a.ell_mode = s->ell_mode;
    }
    a.p_old = s->p[s->pcur].v;
    a.p_new = s->p[1 - s->pcur].v;
    a.x = s->x.v;
    a.out = k2_is_stencil(s) ? nullptr : s->q.v;     // 64-B variant: q is recomputed by K2
    a.scal = s->d_scal;
    a.partials = s->d_part + part_off;
    a.mask_dirichlet = 1;
    a.tabA = s->tabs.d_tabA;
    a.off_mask = table_mask(s->hA);
    const bool part = lk_hi > lk_lo;
    if (part) {
        a.lk_lo = lk_lo;
        a.lk_hi = lk_hi;
        a.lk_lo2 = lk_lo2;
        a.lk_hi2 = lk_hi2;
    }
    int rc;
    if (s->wfast) {                            // p = D^-1 r + beta p formed in the kernel: no z vector
        a.in = s->r.v;
        a.out = s->q.v;
        if (s->xk2) {                          // K1 neither touches x nor stores q: K2 applies the stencil again and carries the x update
            a.out = nullptr;
            a.x_in_k2 = 1;
            a.xd = s->d_xd;
        }
        wfast_args(s, 0, &a);
        rc = wmarch_apply(s->ctx, a, APPLY_CG_K1, li);
    } else if (s->plane2d) {
        a.g = s->g;
        plane2d_weights(s->hA, a.w);
        rc = stencil_plane2d(s->ctx, a, APPLY_CG_K1, li);
    } else if (s->tile3d) {
        a.g = s->g;
        fast_weights(s->hA, a.w, false);
        a.npt = s->npt;
        rc = stencil_tile3d(s->ctx, a, APPLY_CG_K1, li);
    } else if (s->fast) {
        a.g = s->gk;
        fast_weights(s->hA, a.w, s->gk.flat2d != 0);
        a.npt = s->npt;
        a.free_x = s->hybrid ? (s->ctx->hybrid_fx ? 2 : 1) : 0;
        a.fx_hi = ((s->g.NX - 1) % 128) != 0 ? 1 : 0;
        if (s->xk2) {                          // K1 leaves the solution vector alone and records its beta
            a.x_in_k2 = 1;
            a.xd = s->d_xd;
        }
        rc = stencil_fast(s->ctx, a, APPLY_CG_K1, li);
        if (rc == PDEGPU_OK && s->hybrid) rc = hybrid_boundary(s, a, APPLY_CG_K1, li);
    } else {
        rc = stencil_general(s->ctx, a, APPLY_CG_K1, li);
    }
    if (rc == PDEGPU_OK && !part) s->pcur ^= 1;
    return rc;
}

// K2: r -= alpha q, partial r.r   (fast path variant 1: q = A p recomputed by the marching stencil)
// part_out: where this launch's r.r partials go (the lazy loop keeps them apart from K1's p.q partials)
// lk_lo < lk_hi: one part of a split K2 (marching sessions whose K2 is the stencil): local planes [lk_lo, lk_hi) and [lk_lo2, lk_hi2)
int op_k2(pdegpu_session* s, LaunchInfo* li, const DotIn& di = DotIn(), double* part_out = nullptr, int lk_lo = 0, int lk_hi = 0, int lk_lo2 = 0, int lk_hi2 = 0) {
    if (!part_out) part_out = s->d_part;
    if (k2_is_stencil(s)) {
        StencilArgs a{};
        a.di = di;
        if (lk_hi > lk_lo) {
            a.lk_lo = lk_lo;
            a.lk_hi = lk_hi;
            a.lk_lo2 = lk_lo2;
            a.lk_hi2 = lk_hi2;
        }
        a.g = s->g;
        a.in = s->p[s->pcur].v;
        a.rhs = s->r.v;
        a.out = s->r.v;
        a.scal = s->d_scal;
        a.partials = part_out;
        a.mask_dirichlet = 1;
        if (s->plane2d) {
            a.g = s->g;
            plane2d_weights(s->hA, a.w);
            return stencil_plane2d(s->ctx, a, APPLY_CG_K2, li);
        }
        if (s->tile3d) {
            a.g = s->g;
            fast_weights(s->hA, a.w, false);
            a.npt = s->npt;
            return stencil_tile3d(s->ctx, a, APPLY_CG_K2, li);
        }
        a.g = s->gk;
        fast_weights(s->hA, a.w, s->gk.flat2d != 0);
        a.npt = s->npt;
        if (s->xk2) {                          // (the caller has chosen this iteration's mode: xk2_begin)
            a.x_in_k2 = 1;
            a.xmode = s->xk2_mode;
            a.xd = s->d_xd;
            a.x = s->x.v;
            a.p_old = s->p[1 - s->pcur].v;     // p of the iteration before: read only when beta cannot rebuild it (restart)
        }
        if (!s->hybrid) return stencil_fast(s->ctx, a, APPLY_CG_K2, li);
        a.free_x = s->ctx->hybrid_fx ? 2 : 1;
        a.fx_hi = ((s->g.NX - 1) % 128) != 0 ? 1 : 0;
        a.tabA = s->tabs.d_tabA;
        a.off_mask = table_mask(s->hA);
        LaunchInfo lf;
        PG_TRY(stencil_fast(s->ctx, a, APPLY_CG_K2, &lf));
        PG_TRY(hybrid_boundary(s, a, APPLY_CG_K2, &lf));
        if (li) *li = lf;
        return PDEGPU_OK;
    }
    if (s->wfast && s->xk2) {
        StencilArgs a{};
        a.di = di;
        a.in = s->p[s->pcur].v;
        a.rhs = s->r.v;                        // (the session's r holds z = D^-1 r)
        a.out = s->r.v;
        a.scal = s->d_scal;
        a.partials = part_out;
        a.mask_dirichlet = 1;
        a.x_in_k2 = 1;
        a.xmode = s->xk2_mode;
        a.xd = s->d_xd;
        a.x = s->x.v;
        a.p_old = s->p[1 - s->pcur].v;
        wfast_args(s, 0, &a);
        return wmarch_apply(s->ctx, a, APPLY_CG_K2, li);
    }
    if (s->wfast) return wmarch_k2(s->ctx, s->g, s->ell_mode, s->r.v, s->q.v, s->ell_mode == 1 ? s->ellA + s->g.PP : s->rowA,
                                   s->ell_mode == 1 ? s->dinv + s->g.PP : nullptr, s->rowAf[0], s->rowAf[1], s->d_scal, part_out, li, di);
    if (s->weighted) return vec_k2_pc(s->ctx, s->g, s->r.v, s->q.v, s->z.v, s->dinv, s->d_scal, part_out, 1, li, di);
    return vec_k2(s->ctx, s->g, s->r.v, s->q.v, s->d_scal, part_out, li, di);
}

// after a RESIDUAL pass: PCG needs z = D^-1 r and rho = r.z in place of r.r
// completes the dot product of the launch described by li: a one-workgroup fixed-order reduce launch (k_reduce); an empty grid
// (a slab without free planes) records the sum 0
int empty_dot(pdegpu_session* s, const LaunchInfo& li, int which) {
    return vec_reduce(s->ctx, s->d_part, li.grid, s->d_scal, which, 1);
}
int residual_finish(pdegpu_session* s, LaunchInfo* li) {
    if (s->weighted && !s->wfast) PG_TRY(vec_k2_pc(s->ctx, s->g, s->r.v, s->q.v, s->z.v, s->dinv, s->d_scal, s->d_part, 0, li));
    return empty_dot(s, *li, 1);
}

// x (including the ghost planes that lie inside the global domain) = Dirichlet values / fill / nodal
int set_bc_ext(pdegpu_session* s, double* x, double fill, const double* nodal_dense, int only_bc) {
    GridDev g2 = s->g;
    if (nodal_dense || s->g.dim < 3) return vec_set_bc(s->ctx, s->g, x, fill, nodal_dense, only_bc);
    const int lo = s->g.k0 > 0 ? 1 : 0;
    const int hi = (s->g.k0 + s->g.nkl < s->g.NZ) ? 1 : 0;
    g2.k0 = s->g.k0 - lo;
    g2.nkl = s->g.nkl + lo + hi;
    return vec_set_bc(s->ctx, g2, x - (long long)lo * s->g.PP, fill, nullptr, only_bc);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" const char* pdegpu_version(void) { return "pdegpu 0.1 (gfx950)"; }

// every tunable of a context: PDEGPU_<KEY> in the environment at pdegpu_create, pdegpu_set_tunable afterwards (table in include/pdegpu.h)
static const char* const kTunableKeys[] = {
    "ell_compact", "wfast", "wfast_ry", "wfast_ry1", "wfast_wg_per_cu", "wfast_one_round", "wfast_k2_mult", "wfast_k2_nt", "wfast_nt", "wfast_ex", "fast_ry", "fast_nt", "fast_ex", "fast_pf", "fast_zlen", "fast_cfg_plain", "fast_cfg_res", "fast_cfg_k1", "fast_cfg_k2", "fast_linear",
    "place_tune", "tile3d_plain", "fast_wg_per_cu", "fast_wg_per_cu_k2", "grid_mult", "k2_grid_mult", "cg_variant", "el_grid_mult", "direct_1d",
    "lazy_dot", "extrapolate", "hybrid_bc", "hybrid_fx", "hybrid_k2", "plane2d", "persist2d", "persist2d_launch", "persist2d_sr", "persist2d_cheb", "persist2d_cheb_ratio", "snapshot_async", "plane2d_ry", "tile3d_max_nodes",
    "fast_minlen", "fast_small_mult", "el_fast", "el_fast_yx", "el_fast_xface", "el_fast_yzface", "el_bnd_last", "el_fast_nt", "el_fast_ry",
    "el_fast_waves", "el_split_min_mb", "el_cg_variant", "el_fast_fit", "el_bnd_gran", "el_xdefer", "cg_xdefer", "wfast_xdefer"};

extern "C" int pdegpu_create(pdegpu_ctx** out, int device_id, void* stream) {
    if (!out) return PDEGPU_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return PDEGPU_ENODEV;
    if (device_id < 0 || device_id >= ndev) return PDEGPU_ENODEV;
    pdegpu_ctx* ctx = new pdegpu_ctx();
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) { delete ctx; return PDEGPU_ENODEV; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { delete ctx; return PDEGPU_ENODEV; }
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    snprintf(ctx->name, sizeof ctx->name, "%s (%s)", prop.name, prop.gcnArchName);
    if (stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(stream);
        ctx->own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return PDEGPU_EHIP; }
        ctx->own_stream = true;
    }
    for (const char* key : kTunableKeys) {
        std::string env = std::string("PDEGPU_") + key;
        for (auto& ch : env) ch = (char)toupper((unsigned char)ch);
        if (const char* v = getenv(env.c_str())) (void)pdegpu_set_tunable(ctx, key, atoi(v));
    }
    *out = ctx;
    return PDEGPU_OK;
}

namespace {
int* tunable_slot(pdegpu_ctx* ctx, const char* key) {
    if (!strcmp(key, "ell_compact")) return &ctx->ell_compact;
    if (!strcmp(key, "wfast")) return &ctx->wfast;
    if (!strcmp(key, "wfast_ry")) return &ctx->wfast_ry;
    if (!strcmp(key, "wfast_wg_per_cu")) return &ctx->wfast_wg_per_cu;
    if (!strcmp(key, "wfast_one_round")) return &ctx->wfast_one_round;
    if (!strcmp(key, "wfast_k2_mult")) return &ctx->wfast_k2_mult;
    if (!strcmp(key, "wfast_ry1")) return &ctx->wfast_ry1;
    if (!strcmp(key, "wfast_k2_nt")) return &ctx->wfast_k2_nt;
    if (!strcmp(key, "wfast_nt")) return &ctx->wfast_nt;
    if (!strcmp(key, "wfast_ex")) return &ctx->wfast_ex;
    if (!strcmp(key, "fast_ry")) return &ctx->fast_ry;
    if (!strcmp(key, "fast_nt")) return &ctx->fast_nt;
    if (!strcmp(key, "fast_ex")) return &ctx->fast_ex;
    if (!strcmp(key, "fast_pf")) return &ctx->fast_pf;
    if (!strcmp(key, "fast_zlen")) return &ctx->fast_zlen;
    if (!strcmp(key, "fast_cfg_plain")) return &ctx->fast_cfg[APPLY_PLAIN];
    if (!strcmp(key, "fast_cfg_res")) return &ctx->fast_cfg[APPLY_RESIDUAL];
    if (!strcmp(key, "fast_cfg_k1")) return &ctx->fast_cfg[APPLY_CG_K1];
    if (!strcmp(key, "fast_cfg_k2")) return &ctx->fast_cfg[APPLY_CG_K2];
    if (!strcmp(key, "fast_linear")) return &ctx->fast_linear;
    if (!strcmp(key, "place_tune")) return &ctx->place_tune;
    if (!strcmp(key, "tile3d_plain")) return &ctx->tile3d_plain;
    if (!strcmp(key, "fast_wg_per_cu")) return &ctx->fast_wg_per_cu;
    if (!strcmp(key, "fast_wg_per_cu_k2")) return &ctx->fast_wg_per_cu_k2;
    if (!strcmp(key, "grid_mult")) return &ctx->grid_mult;
    if (!strcmp(key, "k2_grid_mult")) return &ctx->k2_grid_mult;
    if (!strcmp(key, "cg_variant")) return &ctx->cg_variant;
    if (!strcmp(key, "el_grid_mult")) return &ctx->el_grid_mult;
    if (!strcmp(key, "direct_1d")) return &ctx->direct_1d;
    if (!strcmp(key, "lazy_dot")) return &ctx->lazy_dot;
    if (!strcmp(key, "extrapolate")) return &ctx->extrapolate;
    if (!strcmp(key, "hybrid_bc")) return &ctx->hybrid_bc;
    if (!strcmp(key, "hybrid_fx")) return &ctx->hybrid_fx;
    if (!strcmp(key, "hybrid_k2")) return &ctx->hybrid_k2;
    if (!strcmp(key, "plane2d")) return &ctx->plane2d;
    if (!strcmp(key, "persist2d")) return &ctx->persist2d;
    if (!strcmp(key, "persist2d_sr")) return &ctx->persist2d_sr;
    if (!strcmp(key, "snapshot_async")) return &ctx->snapshot_async;
    if (!strcmp(key, "persist2d_cheb")) return &ctx->persist2d_cheb;
    if (!strcmp(key, "persist2d_cheb_ratio")) return &ctx->persist2d_cheb_ratio;
    if (!strcmp(key, "persist2d_launch")) return &ctx->persist2d_launch;
    if (!strcmp(key, "tile3d_max_nodes")) return &ctx->tile3d_max_nodes;
    if (!strcmp(key, "plane2d_ry")) return &ctx->plane2d_ry;
    if (!strcmp(key, "fast_minlen")) return &ctx->fast_minlen;
    if (!strcmp(key, "fast_small_mult")) return &ctx->fast_small_mult;
    if (!strcmp(key, "el_fast")) return &ctx->el_fast;
    if (!strcmp(key, "el_fast_ry")) return &ctx->el_fast_ry;
    if (!strcmp(key, "el_fast_fit")) return &ctx->el_fast_fit;
    if (!strcmp(key, "el_bnd_gran")) return &ctx->el_bnd_gran;
    if (!strcmp(key, "el_xdefer")) return &ctx->el_xdefer;
    if (!strcmp(key, "cg_xdefer")) return &ctx->cg_xdefer;
    if (!strcmp(key, "wfast_xdefer")) return &ctx->wfast_xdefer;
    if (!strcmp(key, "el_fast_yx")) return &ctx->el_fast_yx;
    if (!strcmp(key, "el_fast_xface")) return &ctx->el_fast_xface;
    if (!strcmp(key, "el_bnd_last")) return &ctx->el_bnd_last;
    if (!strcmp(key, "el_fast_yzface")) return &ctx->el_fast_yzface;
    if (!strcmp(key, "el_fast_nt")) return &ctx->el_fast_nt;
    if (!strcmp(key, "el_fast_waves")) return &ctx->el_fast_waves;
    if (!strcmp(key, "el_split_min_mb")) return &ctx->el_split_min_mb;
    if (!strcmp(key, "el_cg_variant")) return &ctx->el_cg_variant;
    return nullptr;
}
}  // namespace

extern "C" int pdegpu_set_tunable(pdegpu_ctx* ctx, const char* key, int value) {
    if (!ctx || !key) return PDEGPU_EINVAL;
    int* slot = tunable_slot(ctx, key);
    if (!slot) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "unknown tunable '%s'", key);
    if ((!strcmp(key, "grid_mult") || !strcmp(key, "k2_grid_mult") || !strcmp(key, "el_grid_mult")) && (value < 1 || value > 64)) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "grid_mult must be 1..64");
    *slot = value;
    return PDEGPU_OK;
}

extern "C" void pdegpu_destroy(pdegpu_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char* pdegpu_last_error(const pdegpu_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int pdegpu_device_name(const pdegpu_ctx* ctx, char* buf, size_t buflen) {
    if (!ctx || !buf || buflen == 0) return PDEGPU_EINVAL;
    snprintf(buf, buflen, "%s", ctx->name);
    return PDEGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// session
// ---------------------------------------------------------------------------------------------
extern "C" void pdegpu_session_destroy(pdegpu_session* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    vec_free(&s->x); vec_free(&s->r); vec_free(&s->p[0]); vec_free(&s->p[1]); vec_free(&s->q); vec_free(&s->b);
    tables_free(&s->tabs);
    if (s->d_scal) (void)hipFree(s->d_scal);
    if (s->d_xd) (void)hipFree(s->d_xd);
    if (s->d_part) (void)hipFree(s->d_part);
    if (s->d_dense) (void)hipFree(s->d_dense);
    if (s->h_pin) (void)hipHostFree(s->h_pin);
    for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : s->tev) (void)hipEventDestroy(e);
    persist2d_free(s->p2d);
    for (double* b0 : s->ell_base)
        if (b0) (void)hipFree(b0);
    if (s->rowA) (void)hipFree(s->rowA);
    if (s->rowM) (void)hipFree(s->rowM);
    for (int f = 0; f < 2; ++f) {
        if (s->rowAf[f]) (void)hipFree(s->rowAf[f]);
        if (s->rowMf[f]) (void)hipFree(s->rowMf[f]);
    }
    vec_free(&s->z);
    vec_free(&s->xprev);
    delete s;
}

namespace {
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

// average ms of `reps` launches (after one untimed) of the marching K1 (64 B/DoF form) with the roles x, r, p_old, p_new played by the
// session's vectors {x, r, p0, p1, q, b}[roles4[.]]
int time_k1_roles(pdegpu_session* s, const int32_t* roles4, int reps, double* avg_ms_out) {
    pdegpu_ctx* ctx = s->ctx;
    double* bufs[6] = {s->x.v, s->r.v, s->p[0].v, s->p[1].v, s->q.v, s->b.v};
    StencilArgs a{};
    a.g = s->gk;
    a.x = bufs[roles4[0]];
    a.in = bufs[roles4[1]];
    a.p_old = bufs[roles4[2]];
    a.p_new = bufs[roles4[3]];
    a.out = nullptr;
    a.scal = s->d_scal;
    a.partials = s->d_part;
    a.mask_dirichlet = 1;
    fast_weights(s->hA, a.w, s->gk.flat2d != 0);
    a.npt = s->npt;
    int mode = APPLY_CG_K1;
    StencilArgs a1 = a;                        // (x_in_k2: the K1 in front of the timed K2)
    if (s->xk2) {
        // x_in_k2 sessions: the read-modify-write stream of the solution vector is in the K2 that carries the x update — time THAT launch, each time
        // behind the K1 it follows in the loop (on its own, back to back, it does not show the slow placement mode it has there): x = role 0,
        // r (read and written by K2, read by K1) = role 1, p_old = role 2, p_new (K1's output, the stencil's input in K2) = role 3
        mode = APPLY_CG_K2;
        a1.x = nullptr;
        a1.x_in_k2 = 1;
        a1.xd = s->d_xd;
        a.in = bufs[roles4[3]];
        a.rhs = bufs[roles4[1]];
        a.out = bufs[roles4[1]];
        a.x_in_k2 = 1;
        a.xmode = 2;
        a.xd = s->d_xd;
    }
    EventPair ev;
    PG_HIP(ctx, hipEventCreate(&ev.a));
    PG_HIP(ctx, hipEventCreate(&ev.b));
    if (s->xk2) PG_TRY(stencil_fast(ctx, a1, APPLY_CG_K1, nullptr));
    PG_TRY(stencil_fast(ctx, a, mode, nullptr));
    PG_HIP(ctx, hipEventRecord(ev.a, ctx->stream));
    for (int i = 0; i < reps; ++i) {
        if (s->xk2) PG_TRY(stencil_fast(ctx, a1, APPLY_CG_K1, nullptr));
        PG_TRY(stencil_fast(ctx, a, mode, nullptr));
    }
    PG_HIP(ctx, hipEventRecord(ev.b, ctx->stream));
    PG_HIP(ctx, hipEventSynchronize(ev.b));
    float ms = 0;
    PG_HIP(ctx, hipEventElapsedTime(&ms, ev.a, ev.b));
    *avg_ms_out = ms / reps;
    return PDEGPU_OK;
}

// Placement of the solution vector (tunable place_tune, HBM-resident single-operator marching sessions).  K1 streams four vectors — r and p_old
// read, p_new written, x read AND written — and its time depends on where they sit in PHYSICAL memory: the same launch takes 0.85 or 0.91-0.93 ms
// at 512^3 from one set of allocations to the next (profiles/r02_k1_bimodal_between_processes.log).  The six vectors of a session are
// interchangeable before the first begin(), and which of them carries the read-modify-write stream decides most of it
// (profiles/r02_k1_role_probe.log: with the right one 0.845 ms whatever the others do, with any other 0.89-0.92), so: time K1 once with each
// buffer as x, and if one is clearly the best let it be x.  6 x 3 launches on zero-filled vectors (they stay zero: alpha = beta = 1), ~20 ms.
int tune_x_placement(pdegpu_session* s) {
    pdegpu_ctx* ctx = s->ctx;
    const bool hbm = (long long)s->g.nkl * s->g.PP * 8 > (64LL << 20);
    if (!ctx->place_tune || !hbm || !s->fast || s->tile3d || s->plane2d || s->weighted || s->g.dim != 3) return PDEGPU_OK;
    const double one[SC_COUNT] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
    static_assert(SC_COUNT <= 8, "scalar block");
    PG_HIP(ctx, hipMemcpyAsync(s->d_scal, one, SC_COUNT * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (s->xk2) PG_HIP(ctx, hipMemcpyAsync(s->d_xd, one, 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));      // (x_in_k2: the K2 with the x update is timed)
    double t[6];
    for (int c = 0; c < 6; ++c) {
        int32_t roles[4] = {c, 0, 0, 0};
        int k = 1;
        for (int i = 0; i < 6 && k < 4; ++i)
            if (i != c) roles[k++] = i;
        PG_TRY(time_k1_roles(s, roles, 2, &t[c]));
    }
    PG_HIP(ctx, hipMemsetAsync(s->d_scal, 0, SC_COUNT * sizeof(double), ctx->stream));
    if (s->xk2) PG_HIP(ctx, hipMemsetAsync(s->d_xd, 0, 4 * sizeof(double), ctx->stream));
    int best = 0;
    for (int c = 1; c < 6; ++c)
        if (t[c] < t[best]) best = c;
    s->place_ms_default = t[0];
    s->place_ms_best = t[best];
    s->place_choice = 0;
    if (best != 0 && t[best] < 0.98 * t[0]) {           // (a clear win only: 2 % is the noise of two launches)
        DevVec* vecs[6] = {&s->x, &s->r, &s->p[0], &s->p[1], &s->q, &s->b};
        std::swap(*vecs[0], *vecs[best]);
        s->place_choice = best;
    }
    if (getenv("PDEGPU_DEBUG_GEOM"))
        fprintf(stderr, "[pdegpu placement] the launch that updates x, with each buffer as x: %.4f %.4f %.4f %.4f %.4f %.4f ms -> x takes buffer %d\n", t[0], t[1], t[2], t[3], t[4], t[5],
                s->place_choice);
    return PDEGPU_OK;
}
}  // namespace

extern "C" int pdegpu_session_create(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_heat* heat, int64_t k0,
                                     int64_t nk, pdegpu_session** out) {
    if (!ctx) return PDEGPU_EINVAL;
    if (!out || !heat) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "null argument");
    *out = nullptr;
    PG_TRY(validate_grid(ctx, grid));
    PG_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nslow = grid->dim == 3 ? (int64_t)grid->n[2] + 1 : 1;
    if (k0 < 0 || nk < 1 || k0 + nk > nslow)
        return pdegpu_set_err(ctx, PDEGPU_EINVAL, "slab [%lld, %lld) outside the %lld node planes", (long long)k0,
                              (long long)(k0 + nk), (long long)nslow);
    if (heat->bc_mode != PDEGPU_BC_UNIFORM && heat->bc_mode != PDEGPU_BC_DIRECTIONAL)
        return pdegpu_set_err(ctx, PDEGPU_EINVAL, "unknown bc_mode %d", heat->bc_mode);
    if (!(heat->kappa > 0.0)) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "kappa must be > 0");
    if (!heat->steady && !(heat->dt > 0.0)) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "dt must be > 0 for a transient solve");

    pdegpu_session* s = new pdegpu_session();
    s->ctx = ctx;
    s->grid = *grid;
    s->heat = *heat;
    s->g = make_grid_dev(grid, heat, k0, nk);
    s->cM = heat->steady ? 0.0 : 1.0;
    s->cK = heat->steady ? heat->kappa : heat->dt * heat->kappa;
    int rc = tables_build(ctx, grid, &s->tabs);
    if (rc == PDEGPU_OK) rc = tables_combine(ctx, &s->tabs, s->cM, s->cK, s->tabs.d_tabA, s->hA);
    s->fast = heat->fast_path != 0 && stencil_fast_supported(s->g, grid->element_mode);
    s->hybrid = s->fast && !stencil_all_dirichlet(s->g) && ctx->hybrid_bc != 0;
    if (s->fast && !stencil_all_dirichlet(s->g) && !s->hybrid) s->fast = false;
    s->gk = (s->fast && s->g.dim == 2) ? stencil_flat2d_view(s->g) : s->g;
    if (s->fast) {
        double w[8];
        fast_weights(s->hA, w, s->gk.flat2d != 0);
        s->npt = is_seven_point(w) ? 7 : 15;
        if (grid->element_mode == PDEGPU_ELEM_Q1) {
            // 27-point marching kernel: needs the sign symmetry of the Q1 interior weights (stiffness and mass)
            if (is_sign_symmetric(s->hA + 13 * 27) && is_sign_symmetric(s->tabs.h_tabM + 13 * 27) &&
                is_sign_symmetric(s->tabs.h_tabK + 13 * 27)) s->npt = 27;
            else s->fast = s->hybrid = false;
        }
    }
    s->plane2d = s->fast && s->g.dim == 2 && ctx->plane2d != 0 && plane2d_symmetric(s->hA);
    s->tile3d = s->fast && !s->hybrid && s->g.dim == 3 && s->g.k0 == 0 && s->g.nkl == s->g.NZ && ctx->tile3d_max_nodes > 0 &&
                (long long)s->g.NX * s->g.NY * s->g.NZ <= ctx->tile3d_max_nodes && stencil_tile3d_grid(s->g) <= 0x3fffffffLL;
    DevVec* vecs[] = {&s->x, &s->r, &s->p[0], &s->p[1], &s->q, &s->b};
    for (DevVec* v : vecs)
        if (rc == PDEGPU_OK) rc = vec_alloc(ctx, s->g, v);
    int npart = stencil_max_partials(ctx);
    if (s->tile3d && stencil_tile3d_grid(s->g) > npart) npart = (int)stencil_tile3d_grid(s->g);
    if (rc == PDEGPU_OK && hipMalloc(&s->d_scal, SC_COUNT * sizeof(double)) != hipSuccess) rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "hipMalloc scal");
    s->npart = npart;
    if (rc == PDEGPU_OK && hipMalloc(&s->d_part, 3 * (size_t)npart * sizeof(double)) != hipSuccess) rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "hipMalloc partials");
    if (rc == PDEGPU_OK && hipHostMalloc(&s->h_pin, SC_COUNT * sizeof(double)) != hipSuccess) rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "hipHostMalloc");
    if (rc == PDEGPU_OK && hipMemsetAsync(s->d_scal, 0, SC_COUNT * sizeof(double), ctx->stream) != hipSuccess) rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "memset scal");
    if (rc == PDEGPU_OK && hipMemsetAsync(s->d_part, 0, npart * sizeof(double), ctx->stream) != hipSuccess) rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "memset partials");
    if (rc == PDEGPU_OK && hipMalloc(&s->d_xd, 4 * sizeof(double)) != hipSuccess) rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "hipMalloc xd");
    if (rc == PDEGPU_OK) rc = xk2_reset(s);
    if (rc == PDEGPU_OK) rc = tune_x_placement(s);
    if (rc != PDEGPU_OK) { pdegpu_session_destroy(s); return rc; }
    *out = s;
    return PDEGPU_OK;
}

extern "C" int pdegpu_session_begin(pdegpu_session* s, double x_fill) {
    if (!s) return PDEGPU_EINVAL;
    pdegpu_ctx* ctx = s->ctx;
    PG_HIP(ctx, hipSetDevice(ctx->device));
    PG_HIP(ctx, hipMemsetAsync(s->d_scal, 0, SC_COUNT * sizeof(double), ctx->stream));
    PG_TRY(vec_zero(ctx, s->g, &s->p[0]));
    PG_TRY(vec_zero(ctx, s->g, &s->p[1]));
    s->pcur = 0;
    s->sr_k2_pending = s->sr_skip_x = false;
    s->rr_slot = SC_RR;
    PG_TRY(xk2_reset(s));
    PG_TRY(set_bc_ext(s, s->x.v, x_fill, nullptr, 0));
    // constant-source right-hand side b = M (0 + f) = f m  (fenics_mcp_server.py:262, 394, 658)
    PG_TRY(vec_zero(ctx, s->g, &s->q));
    PG_TRY(op_plain(s, 1, s->q.v, s->heat.source, s->b.v, 0));
    return PDEGPU_OK;
}

extern "C" void* pdegpu_session_plane_ptr(pdegpu_session* s, int vec, int64_t which) {
    if (!s || which < -1 || which > s->g.nkl) return nullptr;
    double* base = nullptr;
    switch (vec) {
        case 0: base = s->x.v; break;
        case 1: base = s->r.v; break;
        case 2: base = s->p[s->pcur].v; break;
        case 3: base = s->q.v; break;
        default: return nullptr;
    }
    return base + which * s->g.PP;
}

extern "C" int pdegpu_session_needs_p_halo(const pdegpu_session* s) {
    return (s && k2_is_stencil(s)) ? 0 : 1;
}
extern "C" int64_t pdegpu_session_plane_doubles(const pdegpu_session* s) { return s ? (int64_t)s->g.PP : -1; }
extern "C" double* pdegpu_session_scalars(pdegpu_session* s) { return s ? s->d_scal : nullptr; }

extern "C" int pdegpu_session_residual(pdegpu_session* s) {
    if (!s) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    LaunchInfo li;
    PG_TRY(op_residual(s, s->x.v, s->b.v, s->r.v, &li));
    s->sr_k2_pending = false;      // r is recomputed: whatever a K2 left in its partials is void, scalars[3] is current again
    s->rr_slot = SC_RR;
    return residual_finish(s, &li);
}

namespace {
// event pair around one launch of the stepwise protocol (timing on): returns the index of the start event, -1 when off
int timed_begin(pdegpu_session* s, int kind) {
    if (!s->timing || !s->timing_now) return -1;
    if (s->tev_used + 2 > s->tev.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return -1;
            s->tev.push_back(e);
        }
        s->tev_kind.push_back(kind);
    }
    const int at = (int)s->tev_used;
    s->tev_kind[at / 2] = kind;
    s->tev_used += 2;
    (void)hipEventRecord(s->tev[at], s->ctx->stream);
    return at;
}
void timed_end(pdegpu_session* s, int at) {
    if (at >= 0) (void)hipEventRecord(s->tev[at + 1], s->ctx->stream);
}
// called where an iteration starts (k1 / k1_part(0)): decides whether this iteration's launches are bracketed by events
void timed_iteration(pdegpu_session* s) {
    if (!s->timing) return;
    s->timing_now = (s->timing_iter % s->timing_every) == 0;
    ++s->timing_iter;
}
// a K1 that can be split into plane ranges: marching kernel alone (no boundary pass, no one-round tiles), 3-D
bool k1_splittable(const pdegpu_session* s) { return s->fast && !s->hybrid && !s->tile3d && !s->plane2d && s->g.dim == 3 && !s->weighted; }
// K2 on plane ranges: the marching stencil form of K2 (r -= alpha A p recomputed) on a slab with a neighbour — where it pays: the split costs one
// more launch (7 us) and buys the exchange K2's bulk to hide behind in addition to K1's.  A plane travels in about 8 B/node / 45 GB/s of one xGMI link
// + 20 us; K1's bulk streams 40 B/node at about 6 TB/s.  Slabs whose K1 bulk is shorter than four such transfers split (ranks of 4 and 8 on 512^3 /
// 1024x512x512: 128 and 64 planes), thicker ones hide the exchange behind K1 alone with room to spare and keep K2 in one launch.
bool k2_splittable(const pdegpu_session* s) {
    if (!k1_splittable(s) || !k2_is_stencil(s) || !(s->g.k0 > 0 || s->g.k0 + s->g.nkl < s->g.NZ)) return false;
    const double k1_bulk_us = (double)s->g.nkl * (double)s->g.PP * 40.0 / 6.0e6;
    const double transfer_us = (double)s->g.PP * 8.0 / 45.0e3 + 20.0;
    return k1_bulk_us < 4.0 * transfer_us;
}
}  // namespace

namespace {
// single-reduction form, after the last launch of K1: ONE reduce launch sums K1's p.Ap and Ap.Ap partials and the r.r partials the K2 of the
// iteration before left (on the first iteration: this slab's r_0.r_0 of the residual pass, SC_RR_LOCAL) into the block {SC_PQ, SC_RR, SC_QQ},
// which the multi-GPU host all-reduces as one 24-byte message
int sr_after_k1(pdegpu_session* s, int n1) {
    PG_TRY(vec_reduce_sr(s->ctx, s->d_part, s->d_part + 2 * (size_t)s->npart, n1, s->d_part + s->npart, s->sr_k2_pending ? s->sr_k2_grid : -1, s->d_scal));
    s->sr_k2_pending = false;
    s->sr_skip_x = false;
    s->rr_slot = SC_RR;
    return PDEGPU_OK;
}
}  // namespace

// 0 = classic CG (two scalar reductions per iteration: p.Ap after k1, r.r after k2); 1 = single-reduction form: k1 also sums Ap.Ap, k2's
// r.r stays in its partials, and after k1 the block scalars[2..4] = {p.Ap, r.r, Ap.Ap} is the ONE message to all-reduce per iteration
// (fenics_mcp_server.py:707-716 distributed).  Choose before begin() / residual(); not for weighted (Jacobi-PCG) sessions.
extern "C" int pdegpu_session_set_cg_variant(pdegpu_session* s, int variant) {
    if (!s) return PDEGPU_EINVAL;
    if (variant != 0 && variant != 1) return pdegpu_set_err(s->ctx, PDEGPU_EINVAL, "cg variant must be 0 (classic) or 1 (single reduction)");
    if (variant == 1 && s->weighted) return pdegpu_set_err(s->ctx, PDEGPU_EUNSUPPORTED, "single-reduction CG: not for weighted (preconditioned) sessions");
    s->variant = variant;
    s->sr_k2_pending = false;
    s->sr_skip_x = false;
    s->rr_slot = SC_RR;
    return PDEGPU_OK;
}
extern "C" int pdegpu_session_cg_variant(const pdegpu_session* s) { return s ? s->variant : -1; }

extern "C" int pdegpu_session_k1(pdegpu_session* s) {
    if (!s) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    LaunchInfo li;
    timed_iteration(s);
    const int at = timed_begin(s, 0);
    PG_TRY(op_k1(s, &li, s->variant ? sr_dot_in(s) : DotIn()));
    timed_end(s, at);
    if (s->timing && s->timing_now) ++s->timed_k1;
    if (s->variant) return sr_after_k1(s, li.grid);
    return empty_dot(s, li, 0);
}

extern "C" int pdegpu_session_can_split_k1(const pdegpu_session* s) { return (s && k1_splittable(s)) ? 1 : 0; }

// K1 in two steps for the multi-GPU loop: part 0 = the owned planes whose stencil does not touch a ghost plane (launched while
// the halo exchange is in flight), part 1 = the one or two planes next to the ghosts — ONE launch covering both — + the sum of all
// partials.  Sessions whose K1 cannot be split do nothing in part 0 and the whole K1 in part 1, so a driver may always call 0 then 1.
extern "C" int pdegpu_session_k1_part(pdegpu_session* s, int part) {
    if (!s || (part != 0 && part != 1)) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    if (!k1_splittable(s)) return part == 0 ? PDEGPU_OK : pdegpu_session_k1(s);
    const GridDev& g = s->g;
    const int lo = g.k0 > 0 ? 1 : 0, hi = (g.k0 + g.nkl < g.NZ) ? 1 : 0;      // slab neighbours below / above
    const DotIn di = s->variant ? sr_dot_in(s) : DotIn();
    LaunchInfo li;
    if (part == 0) {
        s->k1_parts = 0;
        timed_iteration(s);
        const int a0 = lo, a1 = g.nkl - hi;
        if (a1 > a0) {
            const int at = timed_begin(s, 0);
            PG_TRY(op_k1(s, &li, di, a0, a1, 0));
            timed_end(s, at);
            s->k1_parts = li.grid;
        }
        return PDEGPU_OK;
    }
    // edge planes: plane 0 (needs ghost -1) and plane nkl-1 (needs ghost nkl); one plane can be both
    int e0[2] = {0, lo ? 1 : 0}, e1[2] = {hi ? std::max(g.nkl - 1, lo ? 1 : 0) : 0, hi ? g.nkl : 0};
    if (e0[1] <= e0[0]) { e0[0] = e1[0]; e0[1] = e1[1]; e1[0] = e1[1] = 0; }      // a single edge goes first
    if (e0[1] > e0[0]) {
        // the partials of all parts share one block of npart doubles: ask for the launch's grid first
        LaunchInfo dry;
        dry.dry = true;
        const int pcur = s->pcur;
        PG_TRY(op_k1(s, &dry, di, e0[0], e0[1], s->k1_parts, e1[0], e1[1]));
        s->pcur = pcur;
        if (s->k1_parts + dry.grid > s->npart)
            return pdegpu_set_err(s->ctx, PDEGPU_EUNSUPPORTED, "split K1: %d + %d workgroups exceed the %d partials of the session", s->k1_parts, dry.grid, s->npart);
        const int at = timed_begin(s, 0);
        PG_TRY(op_k1(s, &li, di, e0[0], e0[1], s->k1_parts, e1[0], e1[1]));
        timed_end(s, at);
        s->k1_parts += li.grid;
    }
    s->pcur ^= 1;
    if (s->timing && s->timing_now) ++s->timed_k1;
    if (s->variant) return sr_after_k1(s, s->k1_parts);
    return vec_reduce(s->ctx, s->d_part, s->k1_parts, s->d_scal, 0, 1);
}

extern "C" int pdegpu_session_k2(pdegpu_session* s) {
    if (!s) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    LaunchInfo li;
    xk2_begin(s);
    const int at = timed_begin(s, (s->xk2 && s->xk2_mode == 2) ? 2 : 1);      // (kind 2: a K2 that carries the x update of an x_in_k2 session)
    if (s->variant) {
        // r -= alpha A p with alpha from the all-reduced block; the r.r partials wait in the second block for the next K1's reduce launch
        PG_TRY(op_k2(s, &li, sr_dot_in(s), s->d_part + s->npart));
        timed_end(s, at);
        if (s->timing && s->timing_now) { ++s->timed_k2; if (s->xk2 && s->xk2_mode == 2) ++s->timed_k2x; }
        s->sr_k2_pending = true;
        s->sr_k2_grid = li.grid;
        return PDEGPU_OK;
    }
    PG_TRY(op_k2(s, &li));
    timed_end(s, at);
    if (s->timing && s->timing_now) { ++s->timed_k2; if (s->xk2 && s->xk2_mode == 2) ++s->timed_k2x; }
    return empty_dot(s, li, 1);
}

// K2 in two steps so that the NEXT iteration's halo exchange can be posted as early as its planes are final: part 0 = the one or two planes next to
// the ghosts (ONE launch) — after it the first / last owned plane of r are those of the new residual and the exchange may start —, part 1 = the planes
// in between + what k2 does after its launch.  The transfer then has K2's bulk AND the next K1's bulk to hide behind (K1's alone: 99 us per rank of
// 8 on 512^3, about what 2 x 2.2 MB over one xGMI link each take).  Sessions whose K2 cannot be split run all of it in part 0 and nothing in part 1,
// so a driver may always call 0, post, 1.
extern "C" int pdegpu_session_can_split_k2(const pdegpu_session* s) { return (s && k2_splittable(s)) ? 1 : 0; }
extern "C" int pdegpu_session_k2_part(pdegpu_session* s, int part) {
    if (!s || (part != 0 && part != 1)) return PDEGPU_EINVAL;
    if (!k2_splittable(s)) return part == 0 ? pdegpu_session_k2(s) : PDEGPU_OK;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    const GridDev& g = s->g;
    const int lo = g.k0 > 0 ? 1 : 0, hi = (g.k0 + g.nkl < g.NZ) ? 1 : 0;
    const DotIn di = s->variant ? sr_dot_in(s) : DotIn();
    double* part_out = s->variant ? s->d_part + s->npart : s->d_part;
    LaunchInfo li;
    if (part == 0) {
        xk2_begin(s);                          // (both parts of the iteration's K2 run in the mode chosen here)
        int e0[2] = {0, lo ? 1 : 0}, e1[2] = {hi ? std::max(g.nkl - 1, lo ? 1 : 0) : 0, hi ? g.nkl : 0};
        if (e0[1] <= e0[0]) { e0[0] = e1[0]; e0[1] = e1[1]; e1[0] = e1[1] = 0; }
        s->k2_parts = 0;
        if (e0[1] > e0[0]) {
            const int at = timed_begin(s, (s->xk2 && s->xk2_mode == 2) ? 2 : 1);      // (kind 2: a K2 that carries the x update of an x_in_k2 session)
            PG_TRY(op_k2(s, &li, di, part_out, e0[0], e0[1], e1[0], e1[1]));
            timed_end(s, at);
            s->k2_parts = li.grid;
        }
        return PDEGPU_OK;
    }
    const int a0 = lo, a1 = g.nkl - hi;
    if (a1 > a0) {
        LaunchInfo dry;
        dry.dry = true;
        PG_TRY(op_k2(s, &dry, di, part_out + s->k2_parts, a0, a1));
        if (s->k2_parts + dry.grid > s->npart)
            return pdegpu_set_err(s->ctx, PDEGPU_EUNSUPPORTED, "split K2: %d + %d workgroups exceed the %d partials of the session", s->k2_parts, dry.grid, s->npart);
        const int at = timed_begin(s, (s->xk2 && s->xk2_mode == 2) ? 2 : 1);      // (kind 2: a K2 that carries the x update of an x_in_k2 session)
        PG_TRY(op_k2(s, &li, di, part_out + s->k2_parts, a0, a1));
        timed_end(s, at);
        s->k2_parts += li.grid;
    }
    if (s->timing && s->timing_now) { ++s->timed_k2; if (s->xk2 && s->xk2_mode == 2) ++s->timed_k2x; }
    if (s->variant) {
        s->sr_k2_pending = true;
        s->sr_k2_grid = s->k2_parts;
        return PDEGPU_OK;
    }
    return vec_reduce(s->ctx, s->d_part, s->k2_parts, s->d_scal, 1, 1);
}

// single-reduction form: r.r of the residual the last k2 left -> scalars[6] (this slab's sum; the multi-GPU host all-reduces that one slot, 8
// bytes, then reads it with pdegpu_session_rr).  For convergence polls only — the iteration itself never needs it.  Without a pending k2
// (right after residual / k1) scalars[3] is current and rr reports that.
extern "C" int pdegpu_session_sr_poll(pdegpu_session* s) {
    if (!s) return PDEGPU_EINVAL;
    if (!s->variant) return pdegpu_set_err(s->ctx, PDEGPU_EINVAL, "sr_poll: the session runs the classic form");
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    if (!s->sr_k2_pending) {
        s->rr_slot = SC_RR;
        return PDEGPU_OK;
    }
    PG_TRY(vec_reduce_slot(s->ctx, s->d_part + s->npart, s->sr_k2_grid, s->d_scal, SC_RR_POLL));
    s->rr_slot = SC_RR_POLL;
    return PDEGPU_OK;
}

// kernel timing of the stepwise protocol (k1 / k1_part / k2): on = 1 starts a fresh measurement; get_stats synchronises and
// ADDS the elapsed kernel time, launch counts and algorithmic bytes since set_timing(1) to *stats (kernel_ms[0] K1, [1] K2)
extern "C" int pdegpu_session_set_timing(pdegpu_session* s, int on) {
    if (!s) return PDEGPU_EINVAL;
    s->timing = on != 0;
    s->timing_every = on > 1 ? on : 1;
    s->timing_iter = 0;
    s->timing_now = false;
    s->tev_used = 0;
    s->timed_k1 = s->timed_k2 = s->timed_k2x = 0;
    return PDEGPU_OK;
}
extern "C" int pdegpu_session_get_stats(pdegpu_session* s, pdegpu_stats* stats) {
    if (!s || !stats) return PDEGPU_EINVAL;
    pdegpu_ctx* ctx = s->ctx;
    PG_HIP(ctx, hipSetDevice(ctx->device));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t at = 0; at + 1 < s->tev_used; at += 2) {
        float ms = 0;
        PG_HIP(ctx, hipEventElapsedTime(&ms, s->tev[at], s->tev[at + 1]));
        const int kind = s->tev_kind[at / 2];
        stats->kernel_ms[kind == 0 ? 0 : (kind == 2 ? 2 : 1)] += ms;
    }
    const double nloc = (double)s->g.nkl * s->g.NY * s->g.NX;
    // x_in_k2 sessions: [0] K1 (24 B/DoF), [1] the K2 launches without the x update (24 B/DoF), [2] those with it (40 B/DoF)
    stats->kernel_launches[0] += s->timed_k1;
    stats->kernel_launches[1] += s->timed_k2 - s->timed_k2x;
    stats->kernel_launches[2] += s->timed_k2x;
    stats->alg_bytes[0] += (s->xk2 ? 24.0 : (k2_is_stencil(s) ? 40.0 : 48.0)) * nloc * (double)s->timed_k1;
    stats->alg_bytes[1] += 24.0 * nloc * (double)(s->timed_k2 - s->timed_k2x);
    stats->alg_bytes[2] += 40.0 * nloc * (double)s->timed_k2x;
    stats->cg_iterations += s->timed_k2;
    stats->path = s->hybrid ? 3 : (s->fast ? 1 : 0);
    s->tev_used = 0;
    s->timed_k1 = s->timed_k2 = s->timed_k2x = 0;
    return PDEGPU_OK;
}

// k1 / k2 / residual already leave this slab's sum in BOTH the local and the global slot (the multi-GPU host then
// all-reduces the global slot in place); this re-copies local -> global, e.g. to undo an all-reduce in tests
extern "C" int pdegpu_session_reduce_local(pdegpu_session* s, int which, int copy_to_global) {
    if (!s || (which != 0 && which != 1)) return PDEGPU_EINVAL;
    if (!copy_to_global) return PDEGPU_OK;
    const int src = which == 0 ? SC_PQ_LOCAL : SC_RR_LOCAL, dst = which == 0 ? SC_PQ : SC_RR;
    PG_HIP(s->ctx, hipMemcpyAsync(s->d_scal + dst, s->d_scal + src, sizeof(double), hipMemcpyDeviceToDevice, s->ctx->stream));
    return PDEGPU_OK;
}

// ---- transient steps on a slab (multi-GPU time loop, fenics_mcp_server.py:702-716): the host exchanges the ghost planes of x
// (vector 0), then  step_rhs: b = M (x + dt f);  lift_norm: r = P (b - A g), g = Dirichlet values, its r.r (the reference norm
// of the step) in the rr slots;  reset_cg + residual start the solve of the step.
extern "C" int pdegpu_session_step_rhs(pdegpu_session* s) {
    if (!s) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    if (s->heat.steady) return pdegpu_set_err(s->ctx, PDEGPU_EINVAL, "step_rhs: the session is a steady system");
    return op_plain(s, 1, s->x.v, s->heat.dt * s->heat.source, s->b.v, 0);
}
extern "C" int pdegpu_session_lift_norm(pdegpu_session* s) {
    if (!s) return PDEGPU_EINVAL;
    pdegpu_ctx* ctx = s->ctx;
    PG_HIP(ctx, hipSetDevice(ctx->device));
    LaunchInfo li;
    PG_TRY(vec_zero(ctx, s->g, &s->q));
    PG_TRY(set_bc_ext(s, s->q.v, 0.0, nullptr, 0));
    PG_TRY(op_residual(s, s->q.v, s->b.v, s->r.v, &li));
    PG_TRY(residual_finish(s, &li));
    s->sr_k2_pending = false;
    s->rr_slot = SC_RR;
    return vec_zero(ctx, s->g, &s->q);
}
extern "C" int pdegpu_session_reset_cg(pdegpu_session* s) {
    if (!s) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    PG_HIP(s->ctx, hipMemsetAsync(s->d_scal, 0, SC_COUNT * sizeof(double), s->ctx->stream));
    s->pcur = 0;      // the old p only ever meets beta = 0 (see linear_solve)
    s->sr_k2_pending = s->sr_skip_x = false;
    s->rr_slot = SC_RR;
    return xk2_reset(s);
}

extern "C" int pdegpu_session_flush(pdegpu_session* s) {
    if (!s) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    if (s->xk2) {
        // x_in_k2: K1 never adds to x, so nothing is pending unless the latest K2 left its update out: x += (its recorded alpha) p
        if (s->xk2_pending) PG_TRY(vec_axpy_flush(s->ctx, s->g, s->x.v, s->p[s->pcur].v, s->d_xd, 2));
        s->xk2_pending = false;
        return PDEGPU_OK;
    }
    PG_TRY(vec_axpy_flush(s->ctx, s->g, s->x.v, s->p[s->pcur].v, s->d_scal, s->variant));
    // the pending update is applied: make the next K1 skip it (alpha_prev = 0) while beta stays valid
    if (s->variant) s->sr_skip_x = true;          // (beta needs p.Ap too in this form: the block stays, the kernel gets a flag)
    else PG_HIP(s->ctx, hipMemsetAsync(s->d_scal + SC_PQ, 0, sizeof(double), s->ctx->stream));
    return PDEGPU_OK;
}

namespace {
// 2-D sessions whose CG loop runs as one cooperative launch (persist2d.hip)
bool use_persist2d(pdegpu_session* s) {
    return s->plane2d && !s->weighted && s->ctx->persist2d != 0 && s->g.dim == 2 && persist2d_supported(s->ctx, s->g);
}
// up to `iters` iterations, stopping early at r.r <= target (target > 0); the scalars and x, r, p of the session are left exactly as
// the launch-per-kernel loop leaves them (x lags by alpha p, p in p[pcur])
int persist2d_iterate(pdegpu_session* s, int iters, double target, int* done, int fresh = 0) {
    if (!s->p2d) PG_TRY(persist2d_create(s->ctx, s->g, &s->p2d));
    double w5[8];
    plane2d_weights(s->hA, w5);
    return persist2d_run(s->ctx, s->p2d, s->g, w5, s->x.v, s->r.v, s->p[s->pcur].v, s->p[s->pcur].v, s->d_scal, iters, target, done, fresh);
}
}  // namespace

extern "C" int pdegpu_session_iterate(pdegpu_session* s, int iters, int time_kernels, pdegpu_stats* stats) {
    if (!s || iters < 0) return PDEGPU_EINVAL;
    pdegpu_ctx* ctx = s->ctx;
    if (s->variant) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "pdegpu_session_iterate runs the classic form: set_cg_variant(0) and begin again");
    PG_HIP(ctx, hipSetDevice(ctx->device));
    if (!time_kernels && iters > 0 && use_persist2d(s)) {
        int done = 0;
        PG_TRY(persist2d_iterate(s, iters, 0.0, &done));
        if (stats) {
            stats->cg_iterations += done;
            stats->path = 1;
        }
        return PDEGPU_OK;
    }
    const size_t need = time_kernels ? (size_t)iters * 4 : 0;
    while (s->events.size() < need) {
        hipEvent_t e;
        PG_HIP(ctx, hipEventCreate(&e));
        s->events.push_back(e);
    }
    LaunchInfo l1, l2;
    int done = 0;
    // lazy dot products (default): no reduce launches inside the chunk — K2's prologue sums K1's p.q partials, the next
    // K1's prologue sums K2's r.r partials (same summation order as k_reduce: bitwise the same scalars); the two r.r
    // slots alternate so that no launch overwrites a scalar its own workgroups still read, and one reduce launch at the
    // end of the chunk leaves the canonical layout (SC_RR current, SC_RR_PREV previous) for the host and for flush.
    const int64_t nslow = s->grid.dim == 3 ? (int64_t)s->grid.n[2] + 1 : 1;
    // (every workgroup of the consuming launch sums ALL partials: only sensible while they are a few thousand)
    const bool lazy = ctx->lazy_dot && s->g.k0 == 0 && s->g.nkl == nslow &&
                      done < iters && !(s->tile3d && stencil_tile3d_grid(s->g) > stencil_max_partials(ctx));
    if (lazy) {
        int cur = SC_RR;
        bool pend = false;
        for (int it = done; it < iters; ++it) {
            DotIn d1;
            if (pend) {
                d1.part = s->d_part + s->npart;
                d1.n = l2.grid;
                d1.s_rr = cur == SC_RR ? SC_RR_PREV : SC_RR;
                d1.s_rr_old = cur;
                cur = d1.s_rr;
            }
            if (time_kernels) PG_HIP(ctx, hipEventRecord(s->events[4 * it + 0], ctx->stream));
            PG_TRY(op_k1(s, &l1, d1));
            if (time_kernels) PG_HIP(ctx, hipEventRecord(s->events[4 * it + 1], ctx->stream));
            if (l1.grid < 1) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "lazy CG loop: empty K1 grid");
            DotIn d2;
            d2.part = s->d_part;
            d2.n = l1.grid;
            d2.s_rr = cur;
            xk2_begin(s);
            if (time_kernels) PG_HIP(ctx, hipEventRecord(s->events[4 * it + 2], ctx->stream));
            PG_TRY(op_k2(s, &l2, d2, s->d_part + s->npart));
            if (time_kernels) PG_HIP(ctx, hipEventRecord(s->events[4 * it + 3], ctx->stream));
            if (l2.grid < 1) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "lazy CG loop: empty K2 grid");
            pend = true;
        }
        PG_TRY(vec_reduce(ctx, s->d_part + s->npart, l2.grid, s->d_scal, cur == SC_RR ? 1 : 2, 1));
        done = iters;
    }
    for (int it = done; it < iters; ++it) {
        if (time_kernels) PG_HIP(ctx, hipEventRecord(s->events[4 * it + 0], ctx->stream));
        PG_TRY(op_k1(s, &l1));
        if (time_kernels) PG_HIP(ctx, hipEventRecord(s->events[4 * it + 1], ctx->stream));
        PG_TRY(empty_dot(s, l1, 0));
        xk2_begin(s);
        if (time_kernels) PG_HIP(ctx, hipEventRecord(s->events[4 * it + 2], ctx->stream));
        PG_TRY(op_k2(s, &l2));
        if (time_kernels) PG_HIP(ctx, hipEventRecord(s->events[4 * it + 3], ctx->stream));
        PG_TRY(empty_dot(s, l2, 1));
    }
    if (stats) {
        stats->cg_iterations += iters;
        stats->path = s->hybrid ? 3 : (s->fast ? 1 : 0);
        if (time_kernels && iters > 0) {
            PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const double nloc = (double)s->g.nkl * s->g.NY * s->g.NX;
            for (int it = 0; it < iters; ++it) {
                float ms = 0;
                PG_HIP(ctx, hipEventElapsedTime(&ms, s->events[4 * it + 0], s->events[4 * it + 1]));
                stats->kernel_ms[0] += ms;
                PG_HIP(ctx, hipEventElapsedTime(&ms, s->events[4 * it + 2], s->events[4 * it + 3]));
                stats->kernel_ms[1] += ms;
            }
            stats->kernel_launches[0] += iters;
            stats->kernel_launches[1] += iters;
            const bool v1 = k2_is_stencil(s);
            // K1: read r, p, x; write p, (q,) x — x_in_k2: read r, p; write p.  K2: read r, q|p; write r — x_in_k2: every other one also reads
            // and writes x (this loop runs pairs: iters / 2 of them, the odd one out is of the kind the pending flag says)
            stats->alg_bytes[0] += (s->xk2 ? 24.0 : (v1 ? 40.0 : 48.0)) * nloc * iters;
            stats->alg_bytes[1] += 24.0 * nloc * iters + (s->xk2 ? 16.0 * nloc * (iters / 2 + ((iters & 1) && !s->xk2_pending ? 1 : 0)) : 0.0);
        }
    }
    return PDEGPU_OK;
}

// ---- the slab loop's two exchanges through the library's own RCCL communicator (comm.hip) ------------------------------------------------
namespace {
// first / last owned plane -> the neighbours' ghost planes.  what: bit 0 the sends, bit 1 the receives (both in real use; a test that runs two
// slabs in one process posts them separately so that sends to self meet their receives in order).  x_planes: the solution vector (the x-halo of a
// backward-Euler step) instead of the CG vectors (r, and p where K1 does not maintain it on the ghost planes).
int session_exchange_post(pdegpu_session* s, pdegpu_comm* c, int what, int x_planes, int lo_rank, int hi_rank) {
    const GridDev& g = s->g;
    const bool lo = g.k0 > 0, hi = g.k0 + g.nkl < g.NZ;
    if (lo_rank < 0) lo_rank = comm_rank(c) - (comm_self_loop(c) ? 0 : 1);
    if (hi_rank < 0) hi_rank = comm_rank(c) + (comm_self_loop(c) ? 0 : 1);
    double* vecs[2];
    int nv = 0;
    if (x_planes) {
        vecs[nv++] = s->x.v;
    } else {
        vecs[nv++] = s->r.v;
        if (!k2_is_stencil(s)) vecs[nv++] = s->p[s->pcur].v;
    }
    const size_t n = (size_t)g.PP;
    for (int v = 0; v < nv; ++v) {
        double* base = vecs[v];
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

extern "C" int pdegpu_session_exchange_post(pdegpu_session* s, pdegpu_comm* c, int what, int x_planes, int lo_rank, int hi_rank) {
    if (!s || !c || (what & ~3) || !what) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    return session_exchange_post(s, c, what, x_planes, lo_rank, hi_rank);
}
// post the exchange: everything enqueued on the session's stream before this call is visible to the sends; returns at once
extern "C" int pdegpu_session_exchange_begin(pdegpu_session* s, pdegpu_comm* c, int x_planes) {
    if (!s || !c) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    const bool any = s->g.k0 > 0 || s->g.k0 + s->g.nkl < s->g.NZ;
    PG_TRY(comm_ready(c, s->ctx->stream));
    if (any) {
        PG_TRY(comm_group_start(c));
        const int rc = session_exchange_post(s, c, 3, x_planes, -1, -1);
        PG_TRY(comm_group_end(c));
        PG_TRY(rc);
    }
    return comm_done(c);
}
// launches enqueued on the session's stream after this call see the received ghost planes (the host does not wait)
extern "C" int pdegpu_session_exchange_end(pdegpu_session* s, pdegpu_comm* c) {
    if (!s || !c) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    return comm_wait(c, s->ctx->stream);
}
// scalars[first .. first + count) summed over the ranks, in place, stream-ordered with the session's kernels
extern "C" int pdegpu_session_allreduce(pdegpu_session* s, pdegpu_comm* c, int first, int count) {
    if (!s || !c || first < 0 || count < 1 || first + count > SC_COUNT) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    return comm_allreduce(c, s->d_scal + first, (size_t)count, s->ctx->stream);
}
// `iters` CG iterations of this rank's slab with the exchange and the scalar reduction(s) issued from here: the loop of pdegpu.dist.SlabCG.iterate
// without the host language in it (fenics_mcp_server.py:707-716 distributed).  Single-reduction form: exchange posted -> K1 away from the ghosts ->
// wait -> K1 edge planes + one reduce -> ONE all-reduce of {p.Ap, r.r, Ap.Ap} -> K2; classic form: 8 bytes after K1 and 8 after K2.
extern "C" int pdegpu_session_iterate_slab(pdegpu_session* s, pdegpu_comm* c, int iters) {
    if (!s || !c || iters < 0) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    const bool multi = comm_world(c) > 1 || comm_self_loop(c);
    bool posted = false;                     // the exchange this iteration's K1 needs went out in the middle of the previous K2
    for (int it = 0; it < iters; ++it) {
        if (multi && !posted) PG_TRY(pdegpu_session_exchange_begin(s, c, 0));
        posted = false;
        PG_TRY(pdegpu_session_k1_part(s, 0));
        if (multi) PG_TRY(pdegpu_session_exchange_end(s, c));
        PG_TRY(pdegpu_session_k1_part(s, 1));
        PG_TRY(comm_allreduce(c, s->d_scal + SC_PQ, s->variant ? 3 : 1, s->ctx->stream));
        if (multi && it + 1 < iters) {       // (the last K2 of a chunk posts nothing: no exchange stays in flight across calls)
            PG_TRY(pdegpu_session_k2_part(s, 0));
            PG_TRY(pdegpu_session_exchange_begin(s, c, 0));
            posted = true;
            PG_TRY(pdegpu_session_k2_part(s, 1));
        } else {
            PG_TRY(pdegpu_session_k2(s));
        }
        if (!s->variant) PG_TRY(comm_allreduce(c, s->d_scal + SC_RR, 1, s->ctx->stream));
    }
    return PDEGPU_OK;
}

extern "C" int pdegpu_session_rr(pdegpu_session* s, double* rr_out) {
    if (!s || !rr_out) return PDEGPU_EINVAL;
    pdegpu_ctx* ctx = s->ctx;
    PG_HIP(ctx, hipSetDevice(ctx->device));
    if (s->variant && s->sr_k2_pending && s->rr_slot != SC_RR_POLL) {
        // single-reduction form: the r.r of the last k2 still sits in its partials
        const int64_t nslow = s->grid.dim == 3 ? (int64_t)s->grid.n[2] + 1 : 1;
        if (s->g.k0 != 0 || s->g.nkl != nslow)
            return pdegpu_set_err(ctx, PDEGPU_EINVAL, "rr after k2 (single-reduction form, slab with neighbours): call pdegpu_session_sr_poll and all-reduce scalars[6] first");
        PG_TRY(pdegpu_session_sr_poll(s));
    }
    PG_HIP(ctx, hipMemcpyAsync(s->h_pin, s->d_scal, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *rr_out = s->h_pin[s->variant ? s->rr_slot : SC_RR];
    return PDEGPU_OK;
}

extern "C" int pdegpu_session_time_apply(pdegpu_session* s, int reps, double* avg_ms_out) {
    if (!s || reps < 1 || !avg_ms_out) return PDEGPU_EINVAL;
    pdegpu_ctx* ctx = s->ctx;
    PG_HIP(ctx, hipSetDevice(ctx->device));
    EventPair ev;
    PG_HIP(ctx, hipEventCreate(&ev.a));
    PG_HIP(ctx, hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    PG_TRY(op_plain(s, 0, s->r.v, 0.0, s->b.v, 1));          // warm-up
    PG_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < reps; ++i) PG_TRY(op_plain(s, 0, s->r.v, 0.0, s->b.v, 1));
    PG_HIP(ctx, hipEventRecord(e1, ctx->stream));
    PG_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0;
    PG_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    *avg_ms_out = ms / reps;
    return PDEGPU_OK;
}

// Diagnostic (tools/role_probe.py): K1 timed with the roles (x, r, p_old, p_new) played by the session's six vectors in the order given by
// roles4 (distinct indices into {x, r, p0, p1, q, b}).  Leaves the vectors in an undefined state: begin() again.
extern "C" int pdegpu_session_debug_time_k1(pdegpu_session* s, const int32_t* roles4, int reps, double* avg_ms_out) {
    if (!s || !roles4 || reps < 1 || !avg_ms_out || !s->fast || s->hybrid || s->tile3d || s->plane2d) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    for (int i = 0; i < 4; ++i) {
        if (roles4[i] < 0 || roles4[i] > 5) return PDEGPU_EINVAL;
        for (int j = 0; j < i; ++j)
            if (roles4[j] == roles4[i]) return PDEGPU_EINVAL;
    }
    return time_k1_roles(s, roles4, reps, avg_ms_out);
}

// in-run memory calibration: average ms of `reps` launches of a streaming kernel over one slab vector set
extern "C" int pdegpu_session_membench(pdegpu_session* s, int which, int reps, double* avg_ms_out) {
    if (!s || reps < 1 || !avg_ms_out || which < 0 || which > 4) return PDEGPU_EINVAL;
    pdegpu_ctx* ctx = s->ctx;
    PG_HIP(ctx, hipSetDevice(ctx->device));
    EventPair ev;
    PG_HIP(ctx, hipEventCreate(&ev.a));
    PG_HIP(ctx, hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    PG_TRY(vec_membench(ctx, s->g, which, s->r.v, s->q.v, s->b.v, s->d_part));
    PG_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < reps; ++i) PG_TRY(vec_membench(ctx, s->g, which, s->r.v, s->q.v, s->b.v, s->d_part));
    PG_HIP(ctx, hipEventRecord(e1, ctx->stream));
    PG_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0;
    PG_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    *avg_ms_out = ms / reps;
    PG_TRY(vec_zero(ctx, s->g, &s->b));        // b was the scratch target: restore "pads and unconstrained garbage are zero"
    return PDEGPU_OK;
}

extern "C" int pdegpu_session_sync(pdegpu_session* s) {
    if (!s) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    PG_HIP(s->ctx, hipStreamSynchronize(s->ctx->stream));
    return PDEGPU_OK;
}

namespace {
int ensure_dense(pdegpu_session* s) {
    if (s->d_dense) return PDEGPU_OK;
    const size_t n = (size_t)s->g.nkl * s->g.NY * s->g.NX;
    PG_HIP(s->ctx, hipMalloc(&s->d_dense, n * sizeof(double)));
    return PDEGPU_OK;
}
int get_vec(pdegpu_session* s, const double* v, double* host_out) {
    PG_TRY(ensure_dense(s));
    const size_t n = (size_t)s->g.nkl * s->g.NY * s->g.NX;
    PG_TRY(vec_pack(s->ctx, s->g, v, s->d_dense));
    PG_HIP(s->ctx, hipMemcpyAsync(host_out, s->d_dense, n * sizeof(double), hipMemcpyDeviceToHost, s->ctx->stream));
    PG_HIP(s->ctx, hipStreamSynchronize(s->ctx->stream));
    return PDEGPU_OK;
}
int put_vec(pdegpu_session* s, const double* host_in, double* v) {
    PG_TRY(ensure_dense(s));
    const size_t n = (size_t)s->g.nkl * s->g.NY * s->g.NX;
    PG_HIP(s->ctx, hipMemcpyAsync(s->d_dense, host_in, n * sizeof(double), hipMemcpyHostToDevice, s->ctx->stream));
    PG_TRY(vec_unpack(s->ctx, s->g, s->d_dense, v));
    PG_HIP(s->ctx, hipStreamSynchronize(s->ctx->stream));
    return PDEGPU_OK;
}
}  // namespace

namespace {
// Snapshots of a transient solve leave the device BESIDE the next step's solve: the field is packed into one of two device buffers on the compute stream,
// and a helper thread copies it to the caller's (pageable) array on a stream of its own — the host-side staging of a pageable copy blocks that thread, not
// the one that issues the next step (config 2: 8.4 MB per step, ~1 ms of every 9.6 ms step).  The TimeSeriesField loops of fenics_mcp_server.py:307-318,
// 436-468, 705-716 append a snapshot per step; nothing reads it before the solve returns.
struct SnapCopier {
    pdegpu_session* s = nullptr;
    hipStream_t cs = nullptr;
    hipEvent_t packed[2] = {nullptr, nullptr};
    double* buf[2] = {nullptr, nullptr};
    std::future<int> fut[2];
    size_t n = 0;
    int turn = 0;
    bool ok = false;

    int init(pdegpu_session* sess) {
        s = sess;
        n = (size_t)s->g.nkl * s->g.NY * s->g.NX;
        if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return PDEGPU_EHIP;
        for (int k = 0; k < 2; ++k)
            if (hipEventCreateWithFlags(&packed[k], hipEventDisableTiming) != hipSuccess || hipMalloc(&buf[k], n * sizeof(double)) != hipSuccess) return PDEGPU_EHIP;
        ok = true;
        return PDEGPU_OK;
    }
    // the copy issued two snapshots ago used the same buffer: it has to be over before the buffer is packed again
    int wait(int k) {
        if (!fut[k].valid()) return PDEGPU_OK;
        const int e = fut[k].get();
        return e == 0 ? PDEGPU_OK : pdegpu_set_err(s->ctx, PDEGPU_EHIP, "snapshot copy: %s", hipGetErrorString((hipError_t)e));
    }
    int push(const double* v, double* host_out) {
        const int k = turn;
        turn ^= 1;
        PG_TRY(wait(k));
        PG_TRY(vec_pack(s->ctx, s->g, v, buf[k]));
        PG_HIP(s->ctx, hipEventRecord(packed[k], s->ctx->stream));
        const int dev = s->ctx->device;
        hipStream_t st = cs;
        hipEvent_t ev = packed[k];
        const double* src = buf[k];
        const size_t bytes = n * sizeof(double);
        auto copy = [dev, st, ev, src, host_out, bytes]() -> int {
            hipError_t e = hipSetDevice(dev);
            if (e == hipSuccess) e = hipStreamWaitEvent(st, ev, 0);
            if (e == hipSuccess) e = hipMemcpyAsync(host_out, src, bytes, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            return (int)e;
        };
        try {
            fut[k] = std::async(std::launch::async, copy);
        } catch (...) {                                         // no thread to be had: the copy runs here, in line
            const int e = copy();
            if (e != 0) return pdegpu_set_err(s->ctx, PDEGPU_EHIP, "snapshot copy: %s", hipGetErrorString((hipError_t)e));
        }
        return PDEGPU_OK;
    }
    int drain() {
        int rc = PDEGPU_OK;
        for (int k = 0; k < 2; ++k) {
            const int r = wait(k);
            if (rc == PDEGPU_OK) rc = r;
        }
        return rc;
    }
    ~SnapCopier() {
        for (int k = 0; k < 2; ++k) {
            if (fut[k].valid()) (void)fut[k].get();
            if (buf[k]) (void)hipFree(buf[k]);
            if (packed[k]) (void)hipEventDestroy(packed[k]);
        }
        if (cs) (void)hipStreamDestroy(cs);
    }
};
}  // namespace

extern "C" int pdegpu_session_get_x(pdegpu_session* s, double* x_out) {
    if (!s || !x_out) return PDEGPU_EINVAL;
    PG_HIP(s->ctx, hipSetDevice(s->ctx->device));
    return get_vec(s, s->x.v, x_out);
}

// ---------------------------------------------------------------------------------------------
// CG on one slab until ||r|| <= rtol * bnorm  (host polls r.r every `chunk` iterations)
// ---------------------------------------------------------------------------------------------
// iterations until the next host poll of r.r: the observed reduction rate of the last chunk predicts how many more
// iterations reach the target (+10 %), clamped to [4, cap]; without a usable rate the chunk doubles.  The host reads 8
// bytes per poll, so a few extra polls near convergence are far cheaper than overshooting by half a chunk.
int next_poll_chunk(double rr_before, double rr_after, double target, int n_done, int chunk, int cap) {
    if (rr_after > 0.0 && rr_before > rr_after && target > 0.0 && rr_after > target && n_done > 0) {
        const double rate = std::log(rr_before / rr_after) / n_done;
        const double need = std::log(rr_after / target) / rate * 1.1 + 1.0;
        if (need < 4.0) return 4;
        if (need < (double)cap) return (int)need;
        return cap;
    }
    return chunk < cap ? std::min(2 * chunk, cap) : cap;
}

namespace {
int cg_solve(pdegpu_session* s, double bnorm2, double rr, double rtol, int max_iter, pdegpu_stats* st, int* iters_out, double* rel_out) {
    pdegpu_ctx* ctx = s->ctx;
    const double target = rtol * rtol * bnorm2;
    int it = 0;
    // consecutive solves of a session (time steps) need nearly the same number of iterations: the first poll of this solve is
    // placed at 85 % of the previous count, the remaining ones by the rate predictor
    int chunk = s->last_iters > 16 ? (int)(0.85 * s->last_iters) : 8;
    EventPair ev;                  // destroyed on every exit path (a failed iterate / rr returns through PG_TRY)
    PG_HIP(ctx, hipEventCreate(&ev.a));
    PG_HIP(ctx, hipEventCreate(&ev.b));
    hipEvent_t e0 = ev.a, e1 = ev.b;
    PG_HIP(ctx, hipEventRecord(e0, ctx->stream));
    const bool persistent = use_persist2d(s);
    bool classic_only = false;      // persistent 2-D loop: the single-reduction form broke down, stay with the classic recurrences
    while (rr > target && it < max_iter && std::isfinite(rr)) {
        if (persistent) {
            // one cooperative launch runs until r.r <= target (every workgroup holds the same r.r): no polling, no overshoot;
            // launches are cut at 8192 iterations only to keep a single launch short
            int done = 0;
            const int req = std::min(8192, max_iter - it);
            const int fresh = (it == 0 && !classic_only) ? 1 : 0;
            PG_TRY(persist2d_iterate(s, req, target, &done, fresh));
            it += done;
            PG_TRY(pdegpu_session_rr(s, &rr));
            if (done < req && rr > target) {
                // The launch left its loop without converging: a breakdown of its recurrences (alpha = 0).  The single-reduction form
                // (Chronopoulos-Gear) loses positivity of its predicted p.Ap earlier than the classic one near tight tolerances, so a
                // fresh single-reduction launch gets ONE retry with the classic recurrences (which restart with beta = 0) before the
                // solve is given up; once iterations have been made the next launch is classic anyway.
                if (fresh) { classic_only = true; continue; }
                if (done == 0) break;
            }
            continue;
        }
        const int n = std::min(chunk, max_iter - it);
        const double rr_before = rr;
        PG_TRY(pdegpu_session_iterate(s, n, 0, nullptr));
        it += n;
        PG_TRY(pdegpu_session_rr(s, &rr));
        chunk = next_poll_chunk(rr_before, rr, target, n, chunk, 64);
    }
    s->last_iters = it;
    PG_TRY(pdegpu_session_flush(s));
    PG_HIP(ctx, hipEventRecord(e1, ctx->stream));
    PG_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0;
    PG_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    if (st) {
        st->cg_iterations += it;
        st->cg_ms += ms;
        st->linear_solves += 1;
    }
    *iters_out = it;
    *rel_out = bnorm2 > 0 ? std::sqrt(rr / bnorm2) : std::sqrt(rr);
    if (!std::isfinite(rr)) return pdegpu_set_err(ctx, PDEGPU_ENOCONV, "CG diverged (non-finite residual) after %d iterations", it);
    if (rr > target) return pdegpu_set_err(ctx, PDEGPU_ENOCONV, "CG did not reach rtol=%g in %d iterations (rel=%g)", rtol, it, *rel_out);
    return PDEGPU_OK;
}

// one linear solve A x = b on the session: reference norm from the lifted right-hand side, CG from x
int linear_solve(pdegpu_session* s, bool x_is_lift_only, pdegpu_stats* st) {
    pdegpu_ctx* ctx = s->ctx;
    LaunchInfo li;
    double bnorm2 = 0;
    PG_HIP(ctx, hipMemsetAsync(s->d_scal, 0, SC_COUNT * sizeof(double), ctx->stream));
    // p of the previous solve is left as it is: with the scalars zeroed the first K1 forms beta = 0 and skips the x-update,
    // so the finite old entries only ever meet a factor 0; they are zero on Dirichlet nodes and pads like every CG vector
    s->pcur = 0;
    PG_TRY(xk2_reset(s));
    if (!x_is_lift_only) {
        // ||P (b - A g)||: g = Dirichlet values, zero elsewhere (held in q for this pass, result in r)
        PG_TRY(vec_zero(ctx, s->g, &s->q));
        PG_TRY(set_bc_ext(s, s->q.v, 0.0, nullptr, 0));
        PG_TRY(op_residual(s, s->q.v, s->b.v, s->r.v, &li));
        PG_TRY(residual_finish(s, &li));
        // the reference norm stays on the device until the first poll reads it together with r.r (one host sync, not two)
        PG_HIP(ctx, hipMemcpyAsync(s->d_scal + SC_BNORM, s->d_scal + SC_RR, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        PG_TRY(vec_zero(ctx, s->g, &s->q));
        PG_HIP(ctx, hipMemsetAsync(s->d_scal, 0, SC_BNORM * sizeof(double), ctx->stream));
    }
    PG_TRY(op_residual(s, s->x.v, s->b.v, s->r.v, &li));
    PG_TRY(residual_finish(s, &li));
    double rr0 = 0;
    PG_TRY(pdegpu_session_rr(s, &rr0));
    bnorm2 = x_is_lift_only ? rr0 : s->h_pin[SC_BNORM];
    int iters = 0;
    double rel = 0;
    int rc = cg_solve(s, bnorm2, rr0, s->heat.rtol, s->heat.max_iter, st, &iters, &rel);
    if (st) {
        st->final_rel_residual = rel;
        if (rc != PDEGPU_OK) st->converged = 0;
    }
    return rc;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// one-shot heat solve (single GPU)
// ---------------------------------------------------------------------------------------------
namespace {
// weighted (curvilinear) operators: per-node coefficients A = cM M_w + cK K_w, M_w, Jacobi diagonal, z vector
int session_make_weighted(pdegpu_session* s, const pdegpu_weight* w) {
    pdegpu_ctx* ctx = s->ctx;
    const long long n = (long long)s->g.nkl * s->g.PP;
    // 3-D grids whose weight does not depend on one coordinate keep three representative rows along it instead of one per node: r^2 sin(theta) does
    // not depend on phi (the slowest index), the cylinder fallback's sqrt(y^2 + z^2), the composite core's factor and the plain weight 1 not on x —
    // 120 B per node and kernel less to read (the weighted K1 moves 168 B per node otherwise), and nz / nx times less memory
    s->ell_mode = 0;
    const bool whole = s->g.k0 == 0 && s->g.nkl == s->g.NZ;
    if (ctx->ell_compact && s->grid.dim == 3 && whole) {
        if (w->kind == PDEGPU_WEIGHT_R2SIN) s->ell_mode = 1;
        else if (w->kind == PDEGPU_WEIGHT_RCYL || w->kind == PDEGPU_WEIGHT_ONE) s->ell_mode = 2;
    }
    const long long ne = s->ell_mode == 1 ? 3LL * s->g.PP : (s->ell_mode == 2 ? 3LL * s->g.NY * s->g.nkl : n);
    const size_t sizes[3] = {(size_t)15 * ne, (size_t)15 * ne, (size_t)n};
    for (int i = 0; i < 3; ++i) PG_HIP(ctx, hipMalloc(&s->ell_base[i], sizes[i] * sizeof(double)));
    if (s->ell_mode) {                         // (entries no representative node writes — plane class 1 of a two-plane grid — must still be finite)
        PG_HIP(ctx, hipMemsetAsync(s->ell_base[0], 0, sizes[0] * sizeof(double), ctx->stream));
        PG_HIP(ctx, hipMemsetAsync(s->ell_base[1], 0, sizes[1] * sizeof(double), ctx->stream));
    }
    s->ellA = s->ell_base[0];
    s->ellM = s->ell_base[1];
    s->dinv = s->ell_base[2];
    s->ell_stride = ne;
    PG_TRY(vec_alloc(ctx, s->g, &s->z));
    double h[3] = {1, 1, 1}, lengths[3] = {0, 0, 0};
    for (int r = 0; r < s->grid.dim; ++r) {
        h[r] = s->grid.L[r] / s->grid.n[r];
        lengths[r] = s->grid.L[r];
    }
    PG_TRY(weighted_assemble(ctx, s->g, w, h, lengths, s->cM, s->cK, s->ellA, s->ellM, s->dinv, s->ell_stride, s->ell_mode));
    s->weighted = true;
    s->wfast = ctx->wfast != 0 && s->ell_mode != 0 && wmarch_supported(s->g, s->ell_mode);
    if (s->wfast && s->ell_mode == 2) {
        const size_t nt = (size_t)16 * s->g.NZ * s->g.NY * sizeof(double);
        PG_HIP(ctx, hipMalloc(&s->rowA, nt));
        PG_HIP(ctx, hipMalloc(&s->rowM, nt));
        PG_TRY(wmarch_rowtab(ctx, s->g, s->ellA, s->ellM, s->ell_stride, s->rowA, s->rowM, 1));
        for (int f = 0; f < 2; ++f) {           // a natural x-face: the rows of its nodes (column class 0 / 2 of the compact rows)
            if (!(s->g.bc_mode == PDEGPU_BC_DIRECTIONAL && !s->g.bc_present[f])) continue;
            PG_HIP(ctx, hipMalloc(&s->rowAf[f], nt));
            PG_HIP(ctx, hipMalloc(&s->rowMf[f], nt));
            PG_TRY(wmarch_rowtab(ctx, s->g, s->ellA, s->ellM, s->ell_stride, s->rowAf[f], s->rowMf[f], f == 0 ? 0 : 2));
        }
    }
    return PDEGPU_OK;
}

int heat_solve_impl(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_heat* heat_in, const pdegpu_weight* weight,
                    const double* u0, double* values_out, double* times_out, pdegpu_stats* stats);
}  // namespace

extern "C" int pdegpu_heat_solve(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_heat* heat_in, const double* u0,
                                 double* values_out, double* times_out, pdegpu_stats* stats) {
    return heat_solve_impl(ctx, grid, heat_in, nullptr, u0, values_out, times_out, stats);
}

// curvilinear tools (:769-1464): the same backward-Euler / steady loop on the weighted forms
extern "C" int pdegpu_heat_solve_weighted_ic(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_weight* weight,
                                             const pdegpu_heat* heat_in, const double* u0, double* values_out, double* times_out,
                                             pdegpu_stats* stats) {
    if (!ctx) return PDEGPU_EINVAL;
    if (!weight) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "null weight");
    if (weight->kind < PDEGPU_WEIGHT_ONE || weight->kind > PDEGPU_WEIGHT_RCYL) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "unknown weight kind %d", weight->kind);
    if (grid && weight->kind == 3 && grid->dim < 2) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "weight r^2 sin(theta) needs dim >= 2");
    if (grid && weight->kind == 4 && grid->dim < 3) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "weight sqrt(y^2+z^2) needs dim == 3");
    if (grid && weight->kind == PDEGPU_WEIGHT_ONE && grid->dim < 3) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "the unit weight (composite core of the box) needs dim == 3");
    if (weight->core_radius > 0.0 && !(weight->core_ratio > 0.0)) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "core_ratio must be > 0");
    if (grid && grid->element_mode != PDEGPU_ELEM_P1SPLIT) return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "weighted forms use the P1 split only");
    return heat_solve_impl(ctx, grid, heat_in, weight, u0, values_out, times_out, stats);
}

// curvilinear tools (:769-1464): the same backward-Euler / steady loop on the weighted forms
extern "C" int pdegpu_heat_solve_weighted(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_weight* weight,
                                          const pdegpu_heat* heat_in, double* values_out, double* times_out, pdegpu_stats* stats) {
    return pdegpu_heat_solve_weighted_ic(ctx, grid, weight, heat_in, nullptr, values_out, times_out, stats);
}


namespace {
// 1-D grids: factor the tridiagonal once, run all backward-Euler steps inside one launch per batch of snapshots
// (tridiag.hip).  `s->x` holds u^0 with the boundary values imposed (transient) or is overwritten (steady / project).
struct TriBuffers {
    double* rows = nullptr;
    double* work = nullptr;
    double* snaps = nullptr;
    ~TriBuffers() {
        if (rows) (void)hipFree(rows);
        if (work) (void)hipFree(work);
        if (snaps) (void)hipFree(snaps);
    }
};

int tri_prepare(pdegpu_session* s, TriBuffers* tb) {
    pdegpu_ctx* ctx = s->ctx;
    const size_t N = (size_t)s->g.NX;
    PG_HIP(ctx, hipMalloc(&tb->rows, 8 * N * sizeof(double)));
    PG_HIP(ctx, hipMalloc(&tb->work, 3 * N * sizeof(double)));
    return tridiag_setup(ctx, s->g, s->tabs.d_tabA, s->tabs.d_tabM, s->weighted ? s->ellA : nullptr, s->weighted ? s->ellM : nullptr,
                         s->ell_stride, tb->rows);
}

int heat_1d_direct(pdegpu_session* s, const pdegpu_heat& heat, double* values_out, double* times_out, pdegpu_stats* st) {
    pdegpu_ctx* ctx = s->ctx;
    const int N = s->g.NX;
    TriBuffers tb;
    PG_TRY(tri_prepare(s, &tb));
    st->path = 2;
    if (heat.steady) {
        PG_HIP(ctx, hipMemsetAsync(s->x.v, 0, (size_t)N * sizeof(double), ctx->stream));      // b = M (0 + f)
        PG_TRY(tridiag_run(ctx, N, tb.rows, tb.work, s->x.v, heat.source, 0, 1, 0, 1, 1, nullptr));
        PG_HIP(ctx, hipMemcpyAsync(values_out, s->x.v, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        times_out[0] = 0.0;
        st->linear_solves += 1;
        return PDEGPU_OK;
    }
    PG_HIP(ctx, hipMemcpyAsync(values_out, s->x.v, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    times_out[0] = 0.0;
    const int stride = heat.snapshot_stride, total = heat.num_steps;
    // snapshots of one launch stay on the device until the launch ends: at most 256 MiB of them per batch
    const long long per_batch_snaps = std::max<long long>(1, (256LL << 20) / ((long long)N * 8));
    const int batch_steps = (int)std::min<long long>(total > 0 ? total : 1, per_batch_snaps * stride);
    PG_HIP(ctx, hipMalloc(&tb.snaps, (size_t)per_batch_snaps * N * sizeof(double)));
    int snap = 1;
    for (int done = 0; done < total;) {
        const int n = std::min(batch_steps, total - done);
        int nsnap = 0;
        for (int q = done + 1; q <= done + n; ++q)
            if (q % stride == 0 || q == total) ++nsnap;
        PG_TRY(tridiag_run(ctx, N, tb.rows, tb.work, s->x.v, heat.dt * heat.source, 0, n, done, total, stride, tb.snaps));
        if (nsnap)
            PG_HIP(ctx, hipMemcpyAsync(values_out + (size_t)snap * N, tb.snaps, (size_t)nsnap * N * sizeof(double), hipMemcpyDeviceToHost,
                                       ctx->stream));
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int q = done + 1; q <= done + n; ++q)
            if (q % stride == 0 || q == total) times_out[snap++] = q * heat.dt;
        done += n;
        st->linear_solves += n;
    }
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PDEGPU_OK;
}
}  // namespace

namespace {
int heat_solve_impl(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_heat* heat_in, const pdegpu_weight* weight,
                    const double* u0, double* values_out, double* times_out, pdegpu_stats* stats) {
    if (!ctx) return PDEGPU_EINVAL;
    if (!heat_in || !values_out || !times_out) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "null argument");
    PG_TRY(validate_grid(ctx, grid));
    pdegpu_heat heat = *heat_in;
    if (weight) heat.fast_path = 0;
    if (heat.rtol <= 0) heat.rtol = 1e-10;
    if (heat.max_iter <= 0) heat.max_iter = 100000;
    if (heat.snapshot_stride < 1) heat.snapshot_stride = 1;
    if (!heat.steady && heat.num_steps < 0) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "num_steps must be >= 0");
    if (!heat.steady && heat.ic_type == PDEGPU_IC_NODAL && !u0) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "ic_type NODAL needs u0");
    pdegpu_stats st{};
    st.converged = 1;
    const int64_t nslow = grid->dim == 3 ? (int64_t)grid->n[2] + 1 : 1;
    pdegpu_session* s = nullptr;
    PG_TRY(pdegpu_session_create(ctx, grid, &heat, 0, nslow, &s));
    if (weight) {
        const int rcw = session_make_weighted(s, weight);
        if (rcw != PDEGPU_OK) {
            pdegpu_session_destroy(s);
            return rcw;
        }
    }
    const size_t N = (size_t)pdegpu_grid_num_nodes(grid);
    st.n_nodes = (int64_t)N;
    st.path = s->wfast ? 4 : (s->hybrid ? 3 : (s->fast ? 1 : 0));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PDEGPU_OK;
    auto fail = [&](int code) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        pdegpu_session_destroy(s);
        if (stats) *stats = st;
        return code;
    };
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(pdegpu_set_err(ctx, PDEGPU_EHIP, "hipEventCreate"));
    (void)hipEventRecord(e0, ctx->stream);

    // free unknowns = nodes that are not Dirichlet (counted on the host; cheap index arithmetic)
    {
        int64_t nfree = 0;
        const GridDev& g = s->g;
        for (int k = 0; k < g.NZ; ++k)
            for (int j = 0; j < g.NY; ++j) {
                // only rows touching a face need the per-node test
                const bool face_row = (g.dim >= 2 && (j == 0 || j == g.NY - 1)) || (g.dim >= 3 && (k == 0 || k == g.NZ - 1));
                if (!face_row) {
                    double v;
                    nfree += g.NX - 2 + (pg_dirichlet(g, 0, j, k, &v) ? 0 : 1) + (g.NX > 1 && pg_dirichlet(g, g.NX - 1, j, k, &v) ? 0 : 1);
                    continue;
                }
                for (int i = 0; i < g.NX; ++i) {
                    double v;
                    if (!pg_dirichlet(g, i, j, k, &v)) ++nfree;
                }
            }
        st.n_unknowns = nfree;
    }

    const bool direct = grid->dim == 1 && ctx->direct_1d != 0;
    if (direct && heat.steady) {
        rc = heat_1d_direct(s, heat, values_out, times_out, &st);
    } else if (heat.steady) {
        rc = pdegpu_session_begin(s, 0.0);
        if (rc == PDEGPU_OK) rc = linear_solve(s, true, &st);
        if (rc == PDEGPU_OK || rc == PDEGPU_ENOCONV) {
            int rc2 = get_vec(s, s->x.v, values_out);
            if (rc2 != PDEGPU_OK) rc = rc2;
            times_out[0] = 0.0;
        }
    } else {
        // u^0: initial condition with the Dirichlet values imposed (fenics_mcp_server.py:296-297, 426, 690-691)
        if (heat.ic_type == PDEGPU_IC_NODAL) {
            rc = ensure_dense(s);
            if (rc == PDEGPU_OK && hipMemcpyAsync(s->d_dense, u0, N * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                rc = pdegpu_set_err(ctx, PDEGPU_EHIP, "upload u0");
            if (rc == PDEGPU_OK) rc = vec_set_bc(ctx, s->g, s->x.v, 0.0, s->d_dense, 0);
        } else {
            rc = set_bc_ext(s, s->x.v, heat.ic_type == PDEGPU_IC_ZERO ? 0.0 : heat.ic_value, nullptr, 0);
        }
        const int stride = heat.snapshot_stride;
        int snap = 0;
        if (direct) {
            if (rc == PDEGPU_OK) rc = heat_1d_direct(s, heat, values_out, times_out, &st);
        } else if (rc == PDEGPU_OK) {
            rc = get_vec(s, s->x.v, values_out);
            times_out[snap++] = 0.0;
        }
        // more than one snapshot to come: they are copied out beside the following steps (snapshot_async = 0: each one in line, as the first is)
        SnapCopier snaps;
        bool async_snaps = false;
        if (!direct && rc == PDEGPU_OK && ctx->snapshot_async && heat.num_steps / stride >= 2) async_snaps = snaps.init(s) == PDEGPU_OK;
        for (int n = 0; !direct && n < heat.num_steps && rc == PDEGPU_OK; ++n) {
            // b = M (u^n + dt f)   — `u_n*v*dx + dt*f*v*dx` (:305, :434, :703)
            rc = op_plain(s, 1, s->x.v, heat.dt * heat.source, s->b.v, 0);
            // CG starts from the linear extrapolation 2 u^n - u^{n-1} (u^n itself on the first step)
            if (rc == PDEGPU_OK && ctx->extrapolate && heat.num_steps > 1) {
                if (!s->xprev.base) rc = vec_alloc(ctx, s->g, &s->xprev);
                if (rc == PDEGPU_OK) rc = vec_extrapolate(ctx, s->g, s->x.v, s->xprev.v, n == 0 ? 1 : 0);
            }
            if (rc == PDEGPU_OK) rc = linear_solve(s, false, &st);
            const int step = n + 1;
            if ((rc == PDEGPU_OK) && (step % stride == 0 || step == heat.num_steps)) {
                rc = async_snaps ? snaps.push(s->x.v, values_out + (size_t)snap * N) : get_vec(s, s->x.v, values_out + (size_t)snap * N);
                times_out[snap++] = step * heat.dt;
            }
        }
        if (async_snaps) {
            const int r = snaps.drain();                      // every snapshot is in the caller's array before the solve returns
            if (rc == PDEGPU_OK) rc = r;
        }
    }
    (void)hipEventRecord(e1, ctx->stream);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    st.solve_ms = ms;
    return fail(rc);
}
}  // namespace

// y = (cM M_w + cK K_w) x for the weighted forms of the curvilinear / composite tools (operator tests, timing).  marching = 1: the z-marching
// kernels of wmarch.hip on the all-Dirichlet box — the free (interior) rows only, every other entry of y is 0; marching = 2: the same with only the
// two x-faces constrained (weights that do not depend on x: the rows of the nodes on the four natural faces are computed too); PDEGPU_EUNSUPPORTED
// where that path does not apply (see wmarch_supported).  marching = 0: every row through the class-table kernels reading the per-node rows.
extern "C" int pdegpu_heat_apply_weighted(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_weight* weight, double cM, double cK, const double* x,
                                          double* y, int marching) {
    if (!ctx) return PDEGPU_EINVAL;
    if (!x || !y || !weight) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "null argument");
    if (marching < 0 || marching > 4) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "marching must be 0 .. 4");
    PG_TRY(validate_grid(ctx, grid));
    if (weight->kind < PDEGPU_WEIGHT_ONE || weight->kind > PDEGPU_WEIGHT_RCYL) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "unknown weight kind %d", weight->kind);
    if (grid->element_mode != PDEGPU_ELEM_P1SPLIT) return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "weighted forms use the P1 split only");
    if ((weight->kind == PDEGPU_WEIGHT_R2SIN && grid->dim < 2) || ((weight->kind == PDEGPU_WEIGHT_RCYL || weight->kind == PDEGPU_WEIGHT_ONE) && grid->dim < 3))
        return pdegpu_set_err(ctx, PDEGPU_EINVAL, "weight kind %d on a %d-D grid", weight->kind, grid->dim);
    pdegpu_heat heat{};
    heat.kappa = 1.0;
    heat.dt = 1.0;
    heat.fast_path = 0;
    heat.bc_mode = marching == 1 ? PDEGPU_BC_UNIFORM : PDEGPU_BC_DIRECTIONAL;
    heat.bc_present[0] = marching ? 1 : 0;
    heat.bc_present[1] = marching == 2 ? 1 : 0;              // 2: T_left and T_right given, the other four faces natural
    heat.bc_present[2] = marching == 4 ? 1 : 0;              // 3: T_left only;  4: T_left and T_side (x = Lx natural)
    const int64_t nslow = grid->dim == 3 ? (int64_t)grid->n[2] + 1 : 1;
    pdegpu_session* s = nullptr;
    PG_TRY(pdegpu_session_create(ctx, grid, &heat, 0, nslow, &s));
    s->cM = cM;
    s->cK = cK;
    int rc = session_make_weighted(s, weight);
    if (rc == PDEGPU_OK && marching && !s->wfast) rc = pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "weighted marching kernels: 3-D grid with >= 2 cells per axis and a weight that does not depend on one coordinate");
    if (rc == PDEGPU_OK) rc = put_vec(s, x, s->x.v);
    if (rc == PDEGPU_OK) rc = vec_zero(ctx, s->g, &s->q);
    if (rc == PDEGPU_OK) rc = op_plain(s, 0, s->x.v, 0.0, s->q.v, marching ? 1 : 0);
    if (rc == PDEGPU_OK) rc = get_vec(s, s->q.v, y);
    pdegpu_session_destroy(s);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// project(): consistent-mass solve M s = rhs, no Dirichlet rows
// ---------------------------------------------------------------------------------------------
extern "C" int pdegpu_mass_solve(pdegpu_ctx* ctx, const pdegpu_grid* grid, const double* rhs, double* s_out, double rtol,
                                 int max_iter, pdegpu_stats* stats) {
    if (!ctx) return PDEGPU_EINVAL;
    if (!rhs || !s_out) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "null argument");
    PG_TRY(validate_grid(ctx, grid));
    pdegpu_heat heat{};
    heat.kappa = 1.0;
    heat.dt = 1.0;
    heat.steady = 0;
    heat.bc_mode = PDEGPU_BC_DIRECTIONAL;      // nothing constrained
    heat.rtol = rtol > 0 ? rtol : 1e-12;
    heat.max_iter = max_iter > 0 ? max_iter : 10000;
    heat.fast_path = 0;
    const int64_t nslow = grid->dim == 3 ? (int64_t)grid->n[2] + 1 : 1;
    pdegpu_session* s = nullptr;
    PG_TRY(pdegpu_session_create(ctx, grid, &heat, 0, nslow, &s));
    // operator = M only
    s->cM = 1.0;
    s->cK = 0.0;
    int rc = tables_combine(ctx, &s->tabs, 1.0, 0.0, s->tabs.d_tabA, s->hA);
    pdegpu_stats st{};
    st.converged = 1;
    st.n_nodes = st.n_unknowns = pdegpu_grid_num_nodes(grid);
    if (rc == PDEGPU_OK && grid->dim == 1 && ctx->direct_1d != 0) {
        // 1-D: tridiagonal M, Thomas recurrence in one launch; the right-hand side goes straight into the scratch row
        auto direct = [&]() -> int {
            const int N = s->g.NX;
            TriBuffers tb;
            PG_TRY(tri_prepare(s, &tb));
            PG_HIP(ctx, hipMemcpyAsync(tb.work + 2 * (size_t)N, rhs, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            PG_TRY(tridiag_run(ctx, N, tb.rows, tb.work, s->x.v, 0.0, 1, 1, 0, 1, 1, nullptr));
            PG_HIP(ctx, hipMemcpyAsync(s_out, s->x.v, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            return PDEGPU_OK;
        };
        rc = direct();
        st.path = 2;
        st.linear_solves = 1;
        pdegpu_session_destroy(s);
        if (stats) *stats = st;
        return rc;
    }
    if (rc == PDEGPU_OK) rc = put_vec(s, rhs, s->b.v);
    if (rc == PDEGPU_OK) rc = vec_zero(ctx, s->g, &s->x);
    if (rc == PDEGPU_OK) rc = linear_solve(s, true, &st);
    if (rc == PDEGPU_OK || rc == PDEGPU_ENOCONV) {
        int rc2 = get_vec(s, s->x.v, s_out);
        if (rc2 != PDEGPU_OK) rc = rc2;
    }
    pdegpu_session_destroy(s);
    if (stats) *stats = st;
    return rc;
}

// ---------------------------------------------------------------------------------------------
// operator access for tests / roofline harness
// ---------------------------------------------------------------------------------------------
extern "C" int pdegpu_heat_tables(pdegpu_ctx* ctx, const pdegpu_grid* grid, double* tabK, double* tabM, double* load) {
    if (!ctx) return PDEGPU_EINVAL;
    PG_TRY(validate_grid(ctx, grid));
    PG_HIP(ctx, hipSetDevice(ctx->device));
    HeatTables t{};
    int rc = tables_build(ctx, grid, &t);
    if (rc == PDEGPU_OK) {
        if (tabK) memcpy(tabK, t.h_tabK, sizeof t.h_tabK);
        if (tabM) memcpy(tabM, t.h_tabM, sizeof t.h_tabM);
        if (load) memcpy(load, t.h_load, sizeof t.h_load);
    }
    tables_free(&t);
    return rc;
}

extern "C" int pdegpu_heat_apply(pdegpu_ctx* ctx, const pdegpu_grid* grid, double cM, double cK, const double* x, double* y,
                                 double* load_out, int fast_path) {
    if (!ctx) return PDEGPU_EINVAL;
    if (!x || !y) return pdegpu_set_err(ctx, PDEGPU_EINVAL, "null argument");
    PG_TRY(validate_grid(ctx, grid));
    pdegpu_heat heat{};
    heat.kappa = 1.0;
    heat.dt = 1.0;
    heat.steady = 0;
    heat.fast_path = fast_path;
    if (fast_path) {                       // fast kernels exist for the all-Dirichlet operator only
        heat.bc_mode = PDEGPU_BC_UNIFORM;
        heat.bc_present[0] = 1;
    } else {
        heat.bc_mode = PDEGPU_BC_DIRECTIONAL;
    }
    const int64_t nslow = grid->dim == 3 ? (int64_t)grid->n[2] + 1 : 1;
    pdegpu_session* s = nullptr;
    PG_TRY(pdegpu_session_create(ctx, grid, &heat, 0, nslow, &s));
    if (fast_path && !s->fast) {
        pdegpu_session_destroy(s);
        return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "fast path needs a 2-D or 3-D P1 grid with >= 2 cells per axis");
    }
    s->cM = cM;
    s->cK = cK;
    int rc = tables_combine(ctx, &s->tabs, cM, cK, s->tabs.d_tabA, s->hA);
    if (rc == PDEGPU_OK) rc = put_vec(s, x, s->x.v);
    // fast path computes the free rows only (Dirichlet rows -> 0); general path applies the full operator
    if (rc == PDEGPU_OK) rc = op_plain(s, 0, s->x.v, 0.0, s->q.v, fast_path ? 1 : 0);
    if (rc == PDEGPU_OK) rc = get_vec(s, s->q.v, y);
    if (rc == PDEGPU_OK && load_out) {
        s->fast = false;                   // load vector on every node: general class-table kernel
        rc = vec_zero(ctx, s->g, &s->r);
        if (rc == PDEGPU_OK) rc = vec_zero(ctx, s->g, &s->b);
        if (rc == PDEGPU_OK) rc = op_plain(s, 1, s->r.v, 1.0, s->b.v, 0);
        if (rc == PDEGPU_OK) rc = get_vec(s, s->b.v, load_out);
    }
    pdegpu_session_destroy(s);
    return rc;
}
