// persist2d.hip — the CG loop of a 2-D heat system as ONE cooperative launch (gfx950).
//
// The 2-D tools (fenics_mcp_server.py:345-468: RectangleMesh, one Dirichlet value on the boundary) solve grids whose whole CG state
// fits in the register files of the chip (config 2: 1025^2 nodes x {x, r, p} = 25 MB against 128 MB of VGPRs), yet the
// launch-per-kernel loop moves that state through L2 / MALL twice per iteration and pays two kernel boundaries plus the
// dependent load rounds of each kernel (21.7 us per iteration at config 2, round 1).  Here every workgroup owns a band of RB
// full-width rows and keeps x, r, p of its nodes in registers for the whole solve; per iteration the only memory traffic is
//   * the first / last row of r of every band (the neighbours' halo: p on the halo rows is recomputed locally from r and the halo's
//     previous p, the same trick the slab kernels use for their ghost planes), and
//   * one 8-byte partial per workgroup and dot product,
// exchanged through TWO grid barriers per iteration (after p.q and after r.r).  x-neighbours inside a band come from DPP wave
// shifts and an LDS hand-off between the band's waves.
//
// Inter-workgroup protocol (cdna_hip_programming.md §6 Guideline 16, MI355X_MICROARCH.md § visibility, first row of the table of
// measured hand-offs: one workgroup per CU, hipMalloc memory): every handed-off byte is stored write-through (8-byte sc1 stores =
// relaxed agent-scope atomic stores), every storing wave drains (s_waitcnt vmcnt(0)) in front of the workgroup barrier, ONE lane
// then adds to a monotonic agent-scope counter and polls it with relaxed sc1 loads; the other waves wait at the workgroup
// barrier that lane joins after its poll has matched; every load of handed-off bytes is an sc1 load (no L1 copy can be stale, so
// no acquire fence is needed).  Spins are bounded: a workgroup that gives up raises a flag every poll loop also watches, so the
// grid always drains; the host then reports the failure.  Nothing depends on dispatch order or placement; the launch is
// cooperative, so a grid that cannot be co-resident is refused at launch instead of deadlocking.
// Sums are formed in a fixed order and every workgroup adds the same partials in the same order: all workgroups hold bitwise
// identical scalars and take the same branch at the convergence test, which therefore needs no host poll at all.
#include "common.h"
#include <cmath>

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<u64*>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ double p2_dpp_lower(double v) {      // lane l <- lane l-1 (lane 0: 0)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double p2_dpp_upper(double v) {      // lane l <- lane l+1 (lane 63: 0)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double p2_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

struct P2dArgs {
    GridDev g;
    double w[5];               // w0, wx, wy, wd1 (offsets +-(1,1)), wd2 (offsets +-(1,-1))
    double* x;
    double* r;
    const double* p_in;
    double* p_out;
    double* scal;              // SC_* block of the session (read at entry, written at exit)
    double* part;              // [2][G] per-workgroup partial sums, by phase parity
    double* pub;               // [2][G][PX]: first / last active row of r of every band
    unsigned* sync;            // SY_* words: barrier counters, failure flag, iterations done (all zeroed by the host before the launch)
    int iters_max;
    double target;             // stop as soon as r.r <= target (0: never)
    int G;                     // workgroups = bands
    int spin_limit;
    unsigned* flags;           // [G] x 16 words: epoch of the rows of r each band has published (single-reduction form)
    int single_reduction;      // 1: Chronopoulos-Gear CG — one grid barrier + one neighbour hand-off per iteration (fresh solves only)
    int cheb_k;                // > 1: Chebyshev polynomial preconditioner of that many terms (k_cg2d_persistent_cheb)
    double cheb_theta, cheb_delta;   // centre / half-width of the interval [lambda_max / ratio, lambda_max] the polynomial is built on
    int barrier_kind;          // 1: XCD-hierarchical (group leaders + generation words; default); 2: sharded counters polled by every
                               // workgroup; 3: hierarchical arrival, every workgroup polls the top counter
};

constexpr int kNW = 8;         // waves per workgroup: 8 x 128 = 1024 columns

// words of the sync block (zeroed by the host before every launch); counters that many workgroups hit sit on lines of their own
enum { SY_FLAT = 0, SY_FAIL = 1, SY_ITERS = 2, SY_POP = 16 /* [8] */, SY_CNT = 32 /* [8] x 16 words */, SY_TOP = 160, SY_GEN = 176 /* [8] x 16 words */,
       SY_WORDS = 320 };

struct BarrierState {
    unsigned epoch = 0;        // hierarchical barriers passed
    int xcc = 0;               // group key of this workgroup: the XCD it runs on (speed only — any partition of the grid is correct)
    unsigned npop = 0;         // workgroups of this group
    unsigned ngroups = 0;      // non-empty groups
    bool leader = false;       // first workgroup of its group to arrive at the census
};

__device__ __forceinline__ bool poll_ge(const P2dArgs& a, const unsigned* word, unsigned want, bool nap) {
    for (int spins = 0;; ++spins) {
        // both loads are in flight together: one memory round trip per poll, not two
        const unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned f = __hip_atomic_load(a.sync + SY_FAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= want) return true;
        if (f != 0u || spins >= a.spin_limit) {
            __hip_atomic_store(a.sync + SY_FAIL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        if (nap) __builtin_amdgcn_s_sleep(1);
    }
}

// census + one flat barrier at kernel start: every workgroup registers with the group of its XCD; afterwards each knows the size
// of its group, whether it leads it, and how many groups exist.  false = the grid failed.
__device__ __forceinline__ bool barrier_init(const P2dArgs& a, BarrierState& bs, int* sShared /* >= 4 ints */) {
    __syncthreads();
    if (threadIdx.x == 0) {
        // HW_REG_XCC_ID (id 20), bits [3:0]: which of the 8 XCDs this workgroup was placed on
        const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
        const unsigned ticket = __hip_atomic_fetch_add(a.sync + SY_POP + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.sync + SY_FLAT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool okb = poll_ge(a, a.sync + SY_FLAT, (unsigned)a.G, true);
        unsigned npop = 0, ngroups = 0;
        if (okb) {
            for (int x = 0; x < 8; ++x) {
                const unsigned c = __hip_atomic_load(a.sync + SY_POP + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ngroups += c > 0u ? 1u : 0u;
                if (x == xcc) npop = c;
            }
        }
        sShared[0] = okb ? 1 : 0;
        sShared[1] = xcc;
        sShared[2] = (int)npop;
        sShared[3] = (int)(ngroups | (ticket == 0u ? 0x100u : 0u));
    }
    __syncthreads();
    bs.xcc = sShared[1];
    bs.npop = (unsigned)sShared[2];
    bs.ngroups = (unsigned)sShared[3] & 0xffu;
    bs.leader = (sShared[3] & 0x100) != 0;
    const bool okb = sShared[0] != 0;
    __syncthreads();
    return okb;
}

// XCD-hierarchical grid barrier (MI355X_MICROARCH.md price list, row barrier-xcd: 4.1 us at 256 workgroups against 7.4 for one flat
// counter — 256 arrivals on ONE word serialise at ~12 ns each): every workgroup arrives on its group's counter; the group leader
// waits for its group, arrives on the top counter, waits for all leaders and publishes the group's generation word, which the
// other workgroups of the group poll.  Payload: sc1 stores drained before the arrival, sc1 loads after the release (see header).
// Sharded form (a.barrier_kind == 2): every workgroup adds to the counter of its group and then polls ALL eight
// group counters itself — eight lanes, one load instruction, one memory round trip per poll — until they sum to G x epoch.  No
// leader hop, no generation word; measured SLOWER than the leader form (config 2: 13.9-14.1 against 12.8-13.6 us per iteration,
// tools/c2_persist.py): 256 pollers on the eight counter lines slow the arrivals down.
__device__ __forceinline__ bool grid_barrier_sharded(const P2dArgs& a, BarrierState& bs, int* sFail) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // every storing wave: its sc1 stores have left
    __syncthreads();
    ++bs.epoch;
    if (threadIdx.x < 64) {                                             // wave 0 polls; the other waves wait at the barrier below
        const int lane = threadIdx.x;
        if (lane == 0) __hip_atomic_fetch_add(a.sync + SY_CNT + 16 * bs.xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)a.G * bs.epoch;
        int fail = 0;
        for (int spins = 0;; ++spins) {
            unsigned v = 0;
            if (lane < 8) v = __hip_atomic_load(a.sync + SY_CNT + 16 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 8) v = __hip_atomic_load(a.sync + SY_FAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 0x40000000u : 0u;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            v = __builtin_amdgcn_readfirstlane(v);
            if (v >= 0x40000000u || spins >= a.spin_limit) {
                if (lane == 0) __hip_atomic_store(a.sync + SY_FAIL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fail = 1;
                break;
            }
            if (v >= want) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) *sFail = fail;
    }
    __syncthreads();
    return *sFail == 0;
}

__device__ __forceinline__ bool grid_barrier(const P2dArgs& a, BarrierState& bs, int* sFail) {
    if (a.barrier_kind == 2) return grid_barrier_sharded(a, bs, sFail);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // every storing wave: its sc1 stores have left
    __syncthreads();
    ++bs.epoch;
    if (threadIdx.x == 0) {
        bool okb = true;
        __hip_atomic_fetch_add(a.sync + SY_CNT + 16 * bs.xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bs.leader) {
            okb = poll_ge(a, a.sync + SY_CNT + 16 * bs.xcc, bs.npop * bs.epoch, false);
            if (okb) {
                __hip_atomic_fetch_add(a.sync + SY_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                okb = poll_ge(a, a.sync + SY_TOP, bs.ngroups * bs.epoch, false);
            }
            if (okb && a.barrier_kind != 3) __hip_atomic_store(a.sync + SY_GEN + 16 * bs.xcc, bs.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (a.barrier_kind == 3) {
            okb = poll_ge(a, a.sync + SY_TOP, bs.ngroups * bs.epoch, true);      // every workgroup watches the top counter: one hop less
        } else {
            okb = poll_ge(a, a.sync + SY_GEN + 16 * bs.xcc, bs.epoch, true);
        }
        *sFail = okb ? 0 : 1;
    }
    __syncthreads();
    return *sFail == 0;
}

// fixed-order sum of the G partials of one phase; identical in every workgroup
__device__ __forceinline__ double sum_partials(const double* part, int G, double* sRed) {
    double v = 0.0;
    for (int e = threadIdx.x; e < G; e += kNW * 64) v += ld_sc1(part + e);
    v = p2_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kNW; ++k) t += sRed[k];
    return t;
}
__device__ __forceinline__ double block_sum(double v, double* sRed) {
    v = p2_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kNW; ++k) t += sRed[k];
    return t;
}

// the two sums of the single-reduction form in ONE pass: one LDS round / one round of sc1 loads instead of two (each value is summed in the order of
// block_sum / sum_partials, so the scalars are bit for bit those of two separate passes)
__device__ __forceinline__ void block_sum2(double va, double vb, double* sRed2 /* [2 * kNW] */, double* oa, double* ob) {
    va = p2_wave_sum(va);
    vb = p2_wave_sum(vb);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        sRed2[threadIdx.x >> 6] = va;
        sRed2[kNW + (threadIdx.x >> 6)] = vb;
    }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
#pragma unroll
    for (int k = 0; k < kNW; ++k) {
        ta += sRed2[k];
        tb += sRed2[kNW + k];
    }
    *oa = ta;
    *ob = tb;
}
__device__ __forceinline__ void sum_partials2(const double* part /* [2][G] */, int G, double* sRed2, double* oa, double* ob) {
    double va = 0.0, vb = 0.0;
    for (int e = threadIdx.x; e < G; e += kNW * 64) {
        const double ta = ld_sc1(part + e), tb = ld_sc1(part + G + e);     // both loads in flight before the first is used
        va += ta;
        vb += tb;
    }
    block_sum2(va, vb, sRed2, oa, ob);
}

template <int RB>
__global__ __launch_bounds__(kNW * 64, 1) void k_cg2d_persistent(P2dArgs a) {
    __shared__ double sRed[kNW];
    __shared__ double sEdge[RB + 2][kNW][2];
    __shared__ int sFail;
    __shared__ int sCensus[4];
    const GridDev& g = a.g;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.x;
    const int i0 = wv * 128 + 2 * lane;
    const int j0 = 1 + b * RB;                                   // first owned row
    const bool col_ok = i0 < g.PX;
    const bool ma = i0 >= 1 && i0 <= g.NX - 2, mb = i0 + 1 <= g.NX - 2;
    const double w0 = a.w[0], wx = a.w[1], wy = a.w[2], wd1 = a.w[3], wd2 = a.w[4];
    int nact = g.NY - 1 - j0;                                    // active rows of this band: j0 .. min(j0 + RB, NY - 1) - 1
    if (nact > RB) nact = RB;
    if (nact < 0) nact = 0;

    // ---- state into registers --------------------------------------------------------------------------------------------
    double2 X[RB], R[RB], P[RB + 2], RH[2];
    auto ld2 = [&](const double* v, int j) {
        const bool ok = col_ok && j >= 0 && j <= g.NY - 1;
        const double2 t = *reinterpret_cast<const double2*>(v + (ok ? (long long)j * g.PX + i0 : 0));
        return ok ? t : make_double2(0.0, 0.0);
    };
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const bool act = q < nact;
        X[q] = act ? ld2(a.x, j0 + q) : make_double2(0.0, 0.0);
        R[q] = act ? ld2(a.r, j0 + q) : make_double2(0.0, 0.0);
        P[q + 1] = act ? ld2(a.p_in, j0 + q) : make_double2(0.0, 0.0);
    }
    // halo rows: the row below the band and the row above its last ACTIVE row (rows of a short last band beyond that are zero)
    P[0] = ld2(a.p_in, j0 - 1);
    RH[0] = ld2(a.r, j0 - 1);
    P[RB + 1] = (nact == RB) ? ld2(a.p_in, j0 + RB) : make_double2(0.0, 0.0);
    RH[1] = (nact == RB) ? ld2(a.r, j0 + RB) : make_double2(0.0, 0.0);
    double pq = a.scal[SC_PQ], rr = a.scal[SC_RR], rr_prev = a.scal[SC_RR_PREV];

    BarrierState bs;
    int it = 0;
    bool ok = barrier_init(a, bs, sCensus);
    unsigned phase = 0;                                          // dot-product phases completed (parity selects the partial buffer)
    while (ok && it < a.iters_max && !(a.target > 0.0 && rr <= a.target)) {
        const double alpha_prev = (pq > 0.0 && rr_prev > 0.0) ? rr_prev / pq : 0.0;
        const double beta = (rr_prev > 0.0) ? rr / rr_prev : 0.0;
        // ---- K1: x += alpha_{n-1} p_{n-1};  p_n = r_n + beta p_{n-1} on the band AND its two halo rows ---------------------
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            X[q].x += alpha_prev * P[q + 1].x;
            X[q].y += alpha_prev * P[q + 1].y;
            P[q + 1].x = R[q].x + beta * P[q + 1].x;
            P[q + 1].y = R[q].y + beta * P[q + 1].y;
        }
        P[0].x = RH[0].x + beta * P[0].x;
        P[0].y = RH[0].y + beta * P[0].y;
        P[RB + 1].x = RH[1].x + beta * P[RB + 1].x;
        P[RB + 1].y = RH[1].y + beta * P[RB + 1].y;
        // x-neighbours across the waves of the band: LDS hand-off of every held row's first / last node
#pragma unroll
        for (int q = 0; q < RB + 2; ++q) {
            if (lane == 0) sEdge[q][wv][0] = P[q].x;
            if (lane == 63) sEdge[q][wv][1] = P[q].y;
        }
        __syncthreads();
        double Lf[RB + 2], Rt[RB + 2];                           // node i0-1 / node i0+2 of every held row
#pragma unroll
        for (int q = 0; q < RB + 2; ++q) {
            double l = p2_dpp_lower(P[q].y), rt = p2_dpp_upper(P[q].x);
            if (lane == 0) l = wv > 0 ? sEdge[q][wv - 1][1] : 0.0;
            if (lane == 63) rt = wv < kNW - 1 ? sEdge[q][wv + 1][0] : 0.0;
            Lf[q] = l;
            Rt[q] = rt;
        }
        // ---- q = A p on the band, p.q ---------------------------------------------------------------------------------------
        double2 Q[RB];
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int c = q + 1;
            const double qa = w0 * P[c].x + wx * (Lf[c] + P[c].y) + wy * (P[c - 1].x + P[c + 1].x) + wd1 * (P[c + 1].y + Lf[c - 1]) +
                              wd2 * (P[c - 1].y + Lf[c + 1]);
            const double qb = w0 * P[c].y + wx * (P[c].x + Rt[c]) + wy * (P[c - 1].y + P[c + 1].y) + wd1 * (Rt[c + 1] + P[c - 1].x) +
                              wd2 * (Rt[c - 1] + P[c + 1].x);
            const bool act = q < nact;
            Q[q] = make_double2((act && ma) ? qa : 0.0, (act && mb) ? qb : 0.0);
            acc += P[c].x * Q[q].x + P[c].y * Q[q].y;
        }
        const double pq_loc = block_sum(acc, sRed);
        const unsigned ph1 = phase & 1u;
        if (threadIdx.x == 0) st_sc1(a.part + (size_t)ph1 * a.G + b, pq_loc);
        ++phase;
        ok = grid_barrier(a, bs, &sFail);
        if (!ok) break;
        pq = sum_partials(a.part + (size_t)ph1 * a.G, a.G, sRed);
        const double alpha = (pq > 0.0) ? rr / pq : 0.0;
        // ---- K2: r -= alpha q, r.r; publish the band's first / last active row of r ----------------------------------------
        acc = 0.0;
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            R[q].x -= alpha * Q[q].x;
            R[q].y -= alpha * Q[q].y;
            acc += R[q].x * R[q].x + R[q].y * R[q].y;
        }
        const double rr_loc = block_sum(acc, sRed);
        const unsigned ph2 = phase & 1u;
        if (threadIdx.x == 0) st_sc1(a.part + (size_t)ph2 * a.G + b, rr_loc);
        ++phase;
        if (col_ok) {
            double2 first = R[0], last = R[0];
#pragma unroll
            for (int q = 1; q < RB; ++q)
                if (q < nact) last = R[q];
            double* pf = a.pub + ((size_t)0 * a.G + b) * g.PX + i0;
            double* pl = a.pub + ((size_t)1 * a.G + b) * g.PX + i0;
            st_sc1(pf, first.x);
            st_sc1(pf + 1, first.y);
            st_sc1(pl, last.x);
            st_sc1(pl + 1, last.y);
        }
        ok = grid_barrier(a, bs, &sFail);
        if (!ok) break;
        const double rr_new = sum_partials(a.part + (size_t)ph2 * a.G, a.G, sRed);
        // halo rows of r for the next p: the band below's LAST row, the band above's FIRST row (boundary rows are zero)
        RH[0] = make_double2(0.0, 0.0);
        RH[1] = make_double2(0.0, 0.0);
        if (col_ok && b > 0) {
            const double* s = a.pub + ((size_t)1 * a.G + (b - 1)) * g.PX + i0;
            RH[0] = make_double2(ld_sc1(s), ld_sc1(s + 1));
        }
        if (col_ok && b + 1 < a.G && nact == RB) {
            const double* s = a.pub + ((size_t)0 * a.G + (b + 1)) * g.PX + i0;
            RH[1] = make_double2(ld_sc1(s), ld_sc1(s + 1));
        }
        rr_prev = rr;
        rr = rr_new;
        ++it;
    }
    // ---- state back to memory (x still lags by alpha p: the session's flush / next K1 applies it) -------------------------------
    if (col_ok) {
#pragma unroll
        for (int q = 0; q < RB; ++q)
            if (q < nact) {
                const long long off = (long long)(j0 + q) * g.PX + i0;
                *reinterpret_cast<double2*>(a.x + off) = X[q];
                *reinterpret_cast<double2*>(a.r + off) = R[q];
                *reinterpret_cast<double2*>(a.p_out + off) = P[q + 1];
            }
    }
    if (b == 0 && threadIdx.x == 0) {
        a.scal[SC_PQ] = pq;
        a.scal[SC_PQ_LOCAL] = pq;
        a.scal[SC_RR] = rr;
        a.scal[SC_RR_LOCAL] = rr;
        a.scal[SC_RR_PREV] = rr_prev;
        a.sync[SY_ITERS] = (unsigned)it;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Single-reduction form (Chronopoulos & Gear 1989): the same Krylov iterates in exact arithmetic, but both dot products of an
// iteration, gamma = r.r and delta = (A r).r, are formed at the same point, so an iteration needs ONE grid barrier; the operator is
// applied to r, whose band-edge rows travel to the two neighbouring bands through a flag hand-off (two workgroups, not the grid):
//   p = r + beta p;  s = w + beta s  (s = A p by recurrence);  x += alpha p;  r -= alpha s;
//   [publish first / last row of r, wait for the neighbours' rows]   w = A r;  gamma' = r.r, delta' = w.r   [grid barrier]
//   beta' = gamma' / gamma;  alpha' = gamma' / (delta' - beta' gamma' / alpha)
// Used for fresh solves only (it starts from p = s = 0); leaves x fully updated (no pending alpha p) and SC_PQ = 0.
template <int RB>
__global__ __launch_bounds__(kNW * 64, 1) void k_cg2d_persistent_sr(P2dArgs a) {
    __shared__ double sRed[2 * kNW];
    __shared__ double sEdge[RB + 2][kNW][2];
    __shared__ int sFail;
    __shared__ int sCensus[4];
    const GridDev& g = a.g;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.x;
    const int i0 = wv * 128 + 2 * lane;
    const int j0 = 1 + b * RB;
    const bool col_ok = i0 < g.PX;
    const bool ma = i0 >= 1 && i0 <= g.NX - 2, mb = i0 + 1 <= g.NX - 2;
    const double w0 = a.w[0], wx = a.w[1], wy = a.w[2], wd1 = a.w[3], wd2 = a.w[4];
    int nact = g.NY - 1 - j0;
    if (nact > RB) nact = RB;
    if (nact < 0) nact = 0;
    double2 X[RB], P[RB], S[RB], W[RB], R[RB + 2];               // R[0], R[RB+1]: halo rows of r
    auto ld2 = [&](const double* v, int j) {
        const bool ok = col_ok && j >= 0 && j <= g.NY - 1;
        const double2 t = *reinterpret_cast<const double2*>(v + (ok ? (long long)j * g.PX + i0 : 0));
        return ok ? t : make_double2(0.0, 0.0);
    };
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const bool act = q < nact;
        X[q] = act ? ld2(a.x, j0 + q) : make_double2(0.0, 0.0);
        R[q + 1] = act ? ld2(a.r, j0 + q) : make_double2(0.0, 0.0);
        P[q] = S[q] = W[q] = make_double2(0.0, 0.0);
    }
    R[0] = ld2(a.r, j0 - 1);
    R[RB + 1] = (nact == RB) ? ld2(a.r, j0 + RB) : make_double2(0.0, 0.0);

    // w = A r on the band, returns the two local sums (r.r, w.r)
    auto apply_r = [&](double* rr_out, double* wr_out) {
#pragma unroll
        for (int q = 0; q < RB + 2; ++q) {
            if (lane == 0) sEdge[q][wv][0] = R[q].x;
            if (lane == 63) sEdge[q][wv][1] = R[q].y;
        }
        __syncthreads();
        double Lf[RB + 2], Rt[RB + 2];
#pragma unroll
        for (int q = 0; q < RB + 2; ++q) {
            double l = p2_dpp_lower(R[q].y), rt = p2_dpp_upper(R[q].x);
            if (lane == 0) l = wv > 0 ? sEdge[q][wv - 1][1] : 0.0;
            if (lane == 63) rt = wv < kNW - 1 ? sEdge[q][wv + 1][0] : 0.0;
            Lf[q] = l;
            Rt[q] = rt;
        }
        double arr = 0.0, awr = 0.0;
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int c = q + 1;
            const double qa = w0 * R[c].x + wx * (Lf[c] + R[c].y) + wy * (R[c - 1].x + R[c + 1].x) + wd1 * (R[c + 1].y + Lf[c - 1]) +
                              wd2 * (R[c - 1].y + Lf[c + 1]);
            const double qb = w0 * R[c].y + wx * (R[c].x + Rt[c]) + wy * (R[c - 1].y + R[c + 1].y) + wd1 * (Rt[c + 1] + R[c - 1].x) +
                              wd2 * (Rt[c - 1] + R[c + 1].x);
            const bool act = q < nact;
            W[q] = make_double2((act && ma) ? qa : 0.0, (act && mb) ? qb : 0.0);
            arr += R[c].x * R[c].x + R[c].y * R[c].y;
            awr += W[q].x * R[c].x + W[q].y * R[c].y;
        }
        block_sum2(arr, awr, sRed, rr_out, wr_out);
    };

    BarrierState bs;
    bool ok = barrier_init(a, bs, sCensus);
    unsigned phase = 0;
    int it = 0;
    // gamma_0 = r.r, delta_0 = (A r).r  (one reduction)
    double gamma = 0.0, delta = 0.0, alpha = 0.0, beta = 0.0;
    if (ok) {
        double lr, lw;
        apply_r(&lr, &lw);
        if (threadIdx.x == 0) {
            st_sc1(a.part + (size_t)(phase & 1u) * 2 * a.G + b, lr);
            st_sc1(a.part + (size_t)(phase & 1u) * 2 * a.G + a.G + b, lw);
        }
        const unsigned ph = phase & 1u;
        ++phase;
        ok = grid_barrier(a, bs, &sFail);
        if (ok) {
            sum_partials2(a.part + (size_t)ph * 2 * a.G, a.G, sRed, &gamma, &delta);
            alpha = delta > 0.0 ? gamma / delta : 0.0;
        }
    }
    unsigned hepoch = 0;                                          // neighbour hand-offs completed
    while (ok && it < a.iters_max && !(a.target > 0.0 && gamma <= a.target) && alpha != 0.0) {
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            P[q].x = R[q + 1].x + beta * P[q].x;
            P[q].y = R[q + 1].y + beta * P[q].y;
            S[q].x = W[q].x + beta * S[q].x;
            S[q].y = W[q].y + beta * S[q].y;
            X[q].x += alpha * P[q].x;
            X[q].y += alpha * P[q].y;
            R[q + 1].x -= alpha * S[q].x;
            R[q + 1].y -= alpha * S[q].y;
        }
        // ---- band-edge rows of the new r to the two neighbouring bands: sc1 rows, drain, ONE lane raises this band's flag ------
        ++hepoch;
        if (col_ok) {
            double2 first = R[1], last = R[1];
#pragma unroll
            for (int q = 1; q < RB; ++q)
                if (q < nact) last = R[q + 1];
            double* pf = a.pub + ((size_t)0 * a.G + b) * g.PX + i0;
            double* pl = a.pub + ((size_t)1 * a.G + b) * g.PX + i0;
            st_sc1(pf, first.x);
            st_sc1(pf + 1, first.y);
            st_sc1(pl, last.x);
            st_sc1(pl + 1, last.y);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x < 64) {                                   // wave 0: lanes 0 / 1 watch the band below / above, lane 2 the failure flag — one
            if (lane == 0) __hip_atomic_store(a.flags + 16 * b, hepoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // round trip per poll for all
            const int nbk = lane == 0 ? b - 1 : b + 1;
            const bool has = lane < 2 && nbk >= 0 && nbk < a.G;
            int fail = 0;
            for (int spins = 0;; ++spins) {
                unsigned v = hepoch, f = 0u;
                if (has) v = __hip_atomic_load(a.flags + 16 * nbk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 2) f = __hip_atomic_load(a.sync + SY_FAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(v >= hepoch)) break;
                if (__any(f != 0u) || spins >= a.spin_limit) {
                    if (lane == 0) __hip_atomic_store(a.sync + SY_FAIL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    fail = 1;
                    break;
                }
            }
            if (lane == 0) sFail = fail;
        }
        __syncthreads();
        if (sFail != 0) { ok = false; break; }
        R[0] = make_double2(0.0, 0.0);
        R[RB + 1] = make_double2(0.0, 0.0);
        if (col_ok && b > 0) {
            const double* sp = a.pub + ((size_t)1 * a.G + (b - 1)) * g.PX + i0;
            R[0] = make_double2(ld_sc1(sp), ld_sc1(sp + 1));
        }
        if (col_ok && b + 1 < a.G && nact == RB) {
            const double* sp = a.pub + ((size_t)0 * a.G + (b + 1)) * g.PX + i0;
            R[RB + 1] = make_double2(ld_sc1(sp), ld_sc1(sp + 1));
        }
        // ---- w = A r, both dot products, ONE grid barrier ------------------------------------------------------------------------
        double lr, lw;
        apply_r(&lr, &lw);
        const unsigned ph = phase & 1u;
        if (threadIdx.x == 0) {
            st_sc1(a.part + (size_t)ph * 2 * a.G + b, lr);
            st_sc1(a.part + (size_t)ph * 2 * a.G + a.G + b, lw);
        }
        ++phase;
        ok = grid_barrier(a, bs, &sFail);
        if (!ok) break;
        double gamma_new, delta_new;
        sum_partials2(a.part + (size_t)ph * 2 * a.G, a.G, sRed, &gamma_new, &delta_new);
        beta = gamma > 0.0 ? gamma_new / gamma : 0.0;
        const double den = delta_new - beta * gamma_new / alpha;
        alpha = den > 0.0 ? gamma_new / den : 0.0;
        gamma = gamma_new;
        ++it;
    }
    if (col_ok) {
#pragma unroll
        for (int q = 0; q < RB; ++q)
            if (q < nact) {
                const long long off = (long long)(j0 + q) * g.PX + i0;
                *reinterpret_cast<double2*>(a.x + off) = X[q];
                *reinterpret_cast<double2*>(a.r + off) = R[q + 1];
                *reinterpret_cast<double2*>(a.p_out + off) = P[q];
            }
    }
    if (b == 0 && threadIdx.x == 0) {
        a.scal[SC_PQ] = 0.0;                                     // x is fully updated: nothing pending for flush / the next K1
        a.scal[SC_PQ_LOCAL] = 0.0;
        a.scal[SC_RR] = gamma;
        a.scal[SC_RR_LOCAL] = gamma;
        a.scal[SC_RR_PREV] = 0.0;                                // a follow-up launch-per-kernel iteration restarts (beta = 0)
        a.sync[SY_ITERS] = (unsigned)it;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Single-reduction PCG with a Chebyshev polynomial preconditioner (cheb_k terms).  What bounds an iteration of the forms above is the grid barrier
// (~4.5 us) and the reduction rounds around it, not the stencil (~1 us) or the neighbour hand-off (profiles/r04_config2_two_row_halo_negative.log):
// z = p(A) r costs cheb_k - 1 more stencil applications with neighbour hand-offs and NO grid barrier, and cuts the number of iterations — of barriers —
// by about cheb_k / 1.2.  p is the Chebyshev polynomial of the interval [lambda_max / ratio, lambda_max] (Saad, Iterative Methods, Alg. 12.1 from z = 0):
// positive on (0, lambda_max], so the preconditioner is SPD whatever the smallest eigenvalue is; lambda_max from Gershgorin's bound on the stencil row.
//   z = p(A) r;  w = A z;  gamma = r.z, delta = w.z, rr = r.r   [ONE grid barrier]   beta = gamma' / gamma;  alpha = gamma' / (delta' - beta gamma' / alpha)
//   p = z + beta p;  s = w + beta s;  x += alpha p;  r -= alpha s
// Every stencil application hands the first / last row of its input to the two neighbouring bands; the rows of consecutive hand-offs alternate between
// two buffers (a band can be at most one hand-off ahead of its neighbour: it needs the neighbour's flag of the current one to go on).
// Fresh solves only; leaves x fully updated, r.r in SC_RR; the iterates differ from plain CG's, the answer at the tolerance does not.
__device__ __forceinline__ void block_sum3(double va, double vb, double vc, double* sRed3 /* [3 * kNW] */, double* oa, double* ob, double* oc) {
    va = p2_wave_sum(va);
    vb = p2_wave_sum(vb);
    vc = p2_wave_sum(vc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        sRed3[threadIdx.x >> 6] = va;
        sRed3[kNW + (threadIdx.x >> 6)] = vb;
        sRed3[2 * kNW + (threadIdx.x >> 6)] = vc;
    }
    __syncthreads();
    double ta = 0.0, tb = 0.0, tc = 0.0;
#pragma unroll
    for (int k = 0; k < kNW; ++k) {
        ta += sRed3[k];
        tb += sRed3[kNW + k];
        tc += sRed3[2 * kNW + k];
    }
    *oa = ta;
    *ob = tb;
    *oc = tc;
}

template <int RB>
__global__ __launch_bounds__(kNW * 64, 1) void k_cg2d_persistent_cheb(P2dArgs a) {
    __shared__ double sRed[3 * kNW];
    __shared__ double sEdge2[2][RB + 2][kNW][2];
    __shared__ int sFail;
    __shared__ int sCensus[4];
    const GridDev& g = a.g;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.x;
    const int i0 = wv * 128 + 2 * lane;
    const int j0 = 1 + b * RB;
    const bool col_ok = i0 < g.PX;
    const bool ma = i0 >= 1 && i0 <= g.NX - 2, mb = i0 + 1 <= g.NX - 2;
    const double w0 = a.w[0], wx = a.w[1], wy = a.w[2], wd1 = a.w[3], wd2 = a.w[4];
    int nact = g.NY - 1 - j0;
    if (nact > RB) nact = RB;
    if (nact < 0) nact = 0;
    double2 X[RB], P[RB], S[RB], R[RB], W[RB], Z[RB], D[RB], T[RB];
    auto ld2 = [&](const double* v, int j) {
        const bool ok = col_ok && j >= 0 && j <= g.NY - 1;
        const double2 t = *reinterpret_cast<const double2*>(v + (ok ? (long long)j * g.PX + i0 : 0));
        return ok ? t : make_double2(0.0, 0.0);
    };
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const bool act = q < nact;
        X[q] = act ? ld2(a.x, j0 + q) : make_double2(0.0, 0.0);
        R[q] = act ? ld2(a.r, j0 + q) : make_double2(0.0, 0.0);
        P[q] = S[q] = W[q] = Z[q] = D[q] = T[q] = make_double2(0.0, 0.0);
    }
    unsigned hepoch = 0;
    const size_t prow = (size_t)a.G * g.PX;
    bool ok = true;

    // out = A v on the band's rows (masked like every CG vector); v's first / last active row goes to the neighbouring bands first (the flag hand-off of
    // the form above; the rows of consecutive hand-offs alternate between two buffers).  A hand-off WITHOUT flags — the published words as the signal: every
    // slot holds an all-ones pattern until its producer lane stores a value, the consumer lane polls its own four words and puts the pattern back, no
    // workgroup barrier — was built and measured: the same iterates, 3.4 instead of 3.6 us per application at 512^2 (bands of 2 rows), not worth its
    // protocol; its 1024^2 figure was spoiled by a build whose arrays sat in scratch memory (see always_inline below).  profiles/r04_config2_chebyshev.log
    // (always_inline: with the body shared between its call sites the arrays it is handed live in scratch memory — 704 bytes per lane, 4x the time)
    auto matvec = [&](const double2* V, double2* out) __attribute__((always_inline)) {
        ++hepoch;
        double* pubp = a.pub + (size_t)(hepoch & 1u) * 2 * prow;
        if (col_ok) {
            double2 first = V[0], last = V[0];
#pragma unroll
            for (int q = 1; q < RB; ++q)
                if (q < nact) last = V[q];
            double* pf = pubp + (size_t)b * g.PX + i0;
            double* pl = pubp + prow + (size_t)b * g.PX + i0;
            st_sc1(pf, first.x);
            st_sc1(pf + 1, first.y);
            st_sc1(pl, last.x);
            st_sc1(pl + 1, last.y);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int fail = 0;
        if (threadIdx.x < 64) {
            if (lane == 0) __hip_atomic_store(a.flags + 16 * b, hepoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int nbk = lane == 0 ? b - 1 : b + 1;
            const bool has = lane < 2 && nbk >= 0 && nbk < a.G;
            for (int spins = 0;; ++spins) {
                unsigned v = hepoch, f = 0u;
                if (has) v = __hip_atomic_load(a.flags + 16 * nbk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 2) f = __hip_atomic_load(a.sync + SY_FAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(v >= hepoch)) break;
                if (__any(f != 0u) || spins >= a.spin_limit) {
                    if (lane == 0) __hip_atomic_store(a.sync + SY_FAIL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    fail = 1;
                    break;
                }
            }
        }
        if (__syncthreads_or(fail)) {
            ok = false;
            return;
        }
        double2 H0 = make_double2(0.0, 0.0), H1 = make_double2(0.0, 0.0);
        if (col_ok && b > 0) {
            const double* sp = pubp + prow + (size_t)(b - 1) * g.PX + i0;
            H0 = make_double2(ld_sc1(sp), ld_sc1(sp + 1));
        }
        if (col_ok && b + 1 < a.G && nact == RB) {
            const double* sp = pubp + (size_t)(b + 1) * g.PX + i0;
            H1 = make_double2(ld_sc1(sp), ld_sc1(sp + 1));
        }
        auto row = [&](int c) { return c == 0 ? H0 : (c == RB + 1 ? H1 : V[c - 1]); };
#pragma unroll
        for (int c = 0; c < RB + 2; ++c) {
            const double2 v = row(c);
            if (lane == 0) sEdge2[hepoch & 1u][c][wv][0] = v.x;
            if (lane == 63) sEdge2[hepoch & 1u][c][wv][1] = v.y;
        }
        __syncthreads();
        double Lf[RB + 2], Rt[RB + 2];
#pragma unroll
        for (int c = 0; c < RB + 2; ++c) {
            const double2 v = row(c);
            double l = p2_dpp_lower(v.y), rt = p2_dpp_upper(v.x);
            if (lane == 0) l = wv > 0 ? sEdge2[hepoch & 1u][c][wv - 1][1] : 0.0;
            if (lane == 63) rt = wv < kNW - 1 ? sEdge2[hepoch & 1u][c][wv + 1][0] : 0.0;
            Lf[c] = l;
            Rt[c] = rt;
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int c = q + 1;
            const double2 vm = row(c - 1), v0 = row(c), vp = row(c + 1);
            const double qa = w0 * v0.x + wx * (Lf[c] + v0.y) + wy * (vm.x + vp.x) + wd1 * (vp.y + Lf[c - 1]) + wd2 * (vm.y + Lf[c + 1]);
            const double qb = w0 * v0.y + wx * (v0.x + Rt[c]) + wy * (vm.y + vp.y) + wd1 * (Rt[c + 1] + vm.x) + wd2 * (Rt[c - 1] + vp.x);
            const bool act = q < nact;
            out[q] = make_double2((act && ma) ? qa : 0.0, (act && mb) ? qb : 0.0);
        }
    };
    // z = p(A) r: cheb_k terms of the Chebyshev iteration for A z = r from z = 0 (T: its residual, D: its direction), then w = A z
    const double theta = a.cheb_theta, cdel = a.cheb_delta, sigma1 = theta / cdel;
    auto precondition_and_apply = [&]() __attribute__((always_inline)) {
        double rho = 1.0 / sigma1;
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            T[q] = R[q];
            D[q] = make_double2(R[q].x / theta, R[q].y / theta);
            Z[q] = D[q];
        }
        for (int i = 1; i < a.cheb_k && ok; ++i) {
            matvec(D, W);                                          // W = A d (scratch here)
            const double rho_new = 1.0 / (2.0 * sigma1 - rho);
            const double c1 = rho_new * rho, c2 = 2.0 * rho_new / cdel;
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                T[q].x -= W[q].x;
                T[q].y -= W[q].y;
                D[q].x = c1 * D[q].x + c2 * T[q].x;
                D[q].y = c1 * D[q].y + c2 * T[q].y;
                Z[q].x += D[q].x;
                Z[q].y += D[q].y;
            }
            rho = rho_new;
        }
        if (ok) matvec(Z, W);
    };
    auto local_sums = [&](double* g_out, double* d_out, double* rr_out) __attribute__((always_inline)) {
        double ag = 0.0, ad = 0.0, ar = 0.0;
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            ag += R[q].x * Z[q].x + R[q].y * Z[q].y;
            ad += W[q].x * Z[q].x + W[q].y * Z[q].y;
            ar += R[q].x * R[q].x + R[q].y * R[q].y;
        }
        block_sum3(ag, ad, ar, sRed, g_out, d_out, rr_out);
    };
    auto global_sums = [&](unsigned ph, double* g_out, double* d_out, double* rr_out) __attribute__((always_inline)) {
        const double* part = a.part + (size_t)ph * 3 * a.G;
        double va = 0.0, vb = 0.0, vc = 0.0;
        for (int e = threadIdx.x; e < a.G; e += kNW * 64) {
            const double ta = ld_sc1(part + e), tb = ld_sc1(part + a.G + e), tc = ld_sc1(part + 2 * a.G + e);
            va += ta;
            vb += tb;
            vc += tc;
        }
        block_sum3(va, vb, vc, sRed, g_out, d_out, rr_out);
    };

    BarrierState bs;
    ok = barrier_init(a, bs, sCensus);
    unsigned phase = 0;
    int it = 0;
    double gamma = 0.0, delta = 0.0, rr = 0.0, alpha = 0.0, beta = 0.0;
    if (ok) precondition_and_apply();
    if (ok) {
        double lg, ld, lr;
        local_sums(&lg, &ld, &lr);
        const unsigned ph = phase & 1u;
        if (threadIdx.x == 0) {
            st_sc1(a.part + (size_t)ph * 3 * a.G + b, lg);
            st_sc1(a.part + (size_t)ph * 3 * a.G + a.G + b, ld);
            st_sc1(a.part + (size_t)ph * 3 * a.G + 2 * a.G + b, lr);
        }
        ++phase;
        ok = grid_barrier(a, bs, &sFail);
        if (ok) {
            global_sums(ph, &gamma, &delta, &rr);
            alpha = delta > 0.0 ? gamma / delta : 0.0;
        }
    }
    while (ok && it < a.iters_max && !(a.target > 0.0 && rr <= a.target) && alpha != 0.0) {
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            P[q].x = Z[q].x + beta * P[q].x;
            P[q].y = Z[q].y + beta * P[q].y;
            S[q].x = W[q].x + beta * S[q].x;
            S[q].y = W[q].y + beta * S[q].y;
            X[q].x += alpha * P[q].x;
            X[q].y += alpha * P[q].y;
            R[q].x -= alpha * S[q].x;
            R[q].y -= alpha * S[q].y;
        }
        precondition_and_apply();
        if (!ok) break;
        double lg, ld, lr;
        local_sums(&lg, &ld, &lr);
        const unsigned ph = phase & 1u;
        if (threadIdx.x == 0) {
            st_sc1(a.part + (size_t)ph * 3 * a.G + b, lg);
            st_sc1(a.part + (size_t)ph * 3 * a.G + a.G + b, ld);
            st_sc1(a.part + (size_t)ph * 3 * a.G + 2 * a.G + b, lr);
        }
        ++phase;
        ok = grid_barrier(a, bs, &sFail);
        if (!ok) break;
        double gamma_new, delta_new, rr_new;
        global_sums(ph, &gamma_new, &delta_new, &rr_new);
        beta = gamma > 0.0 ? gamma_new / gamma : 0.0;
        const double den = delta_new - beta * gamma_new / alpha;
        alpha = den > 0.0 ? gamma_new / den : 0.0;
        gamma = gamma_new;
        rr = rr_new;
        ++it;
    }
    if (col_ok) {
#pragma unroll
        for (int q = 0; q < RB; ++q)
            if (q < nact) {
                const long long off = (long long)(j0 + q) * g.PX + i0;
                *reinterpret_cast<double2*>(a.x + off) = X[q];
                *reinterpret_cast<double2*>(a.r + off) = R[q];
                *reinterpret_cast<double2*>(a.p_out + off) = P[q];
            }
    }
    if (b == 0 && threadIdx.x == 0) {
        a.scal[SC_PQ] = 0.0;
        a.scal[SC_PQ_LOCAL] = 0.0;
        a.scal[SC_RR] = rr;
        a.scal[SC_RR_LOCAL] = rr;
        a.scal[SC_RR_PREV] = 0.0;
        a.sync[SY_ITERS] = (unsigned)it;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
struct Persist2d {
    double* part = nullptr;
    double* pub = nullptr;
    unsigned* sync = nullptr;
    unsigned* flags = nullptr;
    unsigned* h_sync = nullptr;     // pinned copy of the three sync words
    int G = 0, rb = 0;
};

void persist2d_free(Persist2d* p) {
    if (!p) return;
    if (p->part) (void)hipFree(p->part);
    if (p->pub) (void)hipFree(p->pub);
    if (p->sync) (void)hipFree(p->sync);
    if (p->flags) (void)hipFree(p->flags);
    if (p->h_sync) (void)hipHostFree(p->h_sync);
    delete p;
}

// rows per band and band count for this grid, or false when the grid does not fit the register-resident form
static bool persist2d_shape(const pdegpu_ctx* ctx, const GridDev& g, int* rb, int* G) {
    if (g.dim != 2 || g.NX < 3 || g.NY < 3) return false;
    if (g.NX - 1 > kNW * 128) return false;                      // one band spans the columns 0 .. NX-2 with 8 waves (the last column is a
                                                                 // Dirichlet column: CG vectors vanish there and x never changes)
    const int rows = g.NY - 2;
    int need = (rows + ctx->num_cu - 1) / ctx->num_cu;           // one workgroup per CU
    int r = need <= 1 ? 1 : (need <= 2 ? 2 : (need <= 4 ? 4 : (need <= 8 ? 8 : 0)));
    if (r == 0) return false;
    *rb = r;
    *G = (rows + r - 1) / r;
    return *G >= 1 && *G <= ctx->num_cu;
}

bool persist2d_supported(const pdegpu_ctx* ctx, const GridDev& g) {
    int rb, G;
    return persist2d_shape(ctx, g, &rb, &G);
}

int persist2d_create(pdegpu_ctx* ctx, const GridDev& g, Persist2d** out) {
    int rb, G;
    if (!persist2d_shape(ctx, g, &rb, &G)) return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "grid does not fit the persistent 2-D CG");
    Persist2d* p = new Persist2d();
    p->G = G;
    p->rb = rb;
    if (hipMalloc(&p->part, 6 * (size_t)G * sizeof(double)) != hipSuccess || hipMalloc(&p->flags, 16 * (size_t)G * sizeof(unsigned)) != hipSuccess || hipMalloc(&p->pub, 4 * (size_t)G * g.PX * sizeof(double)) != hipSuccess ||
        hipMalloc(&p->sync, SY_WORDS * sizeof(unsigned)) != hipSuccess || hipHostMalloc(&p->h_sync, 64) != hipSuccess) {
        persist2d_free(p);
        return pdegpu_set_err(ctx, PDEGPU_EHIP, "persistent 2-D CG: allocation failed");
    }
    (void)hipMemsetAsync(p->part, 0, 6 * (size_t)G * sizeof(double), ctx->stream);
    (void)hipMemsetAsync(p->pub, 0, 4 * (size_t)G * g.PX * sizeof(double), ctx->stream);
    *out = p;
    return PDEGPU_OK;
}

// up to iters_max CG iterations (stopping early once r.r <= target, target > 0) in one cooperative launch; synchronises and
// returns the number of iterations done.  x, r, p_in / p_out: pitched 2-D vectors of the session.
int persist2d_run(pdegpu_ctx* ctx, Persist2d* p, const GridDev& g, const double* w5, double* x, double* r, const double* p_in, double* p_out,
                  double* scal, int iters_max, double target, int* iters_done, int fresh) {
    P2dArgs a{};
    a.g = g;
    for (int k = 0; k < 5; ++k) a.w[k] = w5[k];
    a.x = x;
    a.r = r;
    a.p_in = p_in;
    a.p_out = p_out;
    a.scal = scal;
    a.part = p->part;
    a.pub = p->pub;
    a.sync = p->sync;
    a.iters_max = iters_max;
    a.target = target;
    a.G = p->G;
    a.spin_limit = 2000000;                                      // ~ seconds: far beyond any barrier of a healthy grid
    a.barrier_kind = (ctx->persist2d == 2 || ctx->persist2d == 3) ? ctx->persist2d : 1;
    a.flags = p->flags;
    a.single_reduction = (fresh && ctx->persist2d_sr) ? 1 : 0;
    a.cheb_k = 0;
    if (a.single_reduction && ctx->persist2d_cheb > 1 && p->rb <= 4) {
        // Gershgorin: no eigenvalue of A exceeds the absolute row sum of the interior stencil (boundary rows have fewer entries)
        const double lmax = std::fabs(w5[0]) + 2.0 * (std::fabs(w5[1]) + std::fabs(w5[2]) + std::fabs(w5[3]) + std::fabs(w5[4]));
        const double lmin = lmax / (double)(ctx->persist2d_cheb_ratio > 1 ? ctx->persist2d_cheb_ratio : 30);
        a.cheb_k = ctx->persist2d_cheb;
        a.cheb_theta = 0.5 * (lmax + lmin);
        a.cheb_delta = 0.5 * (lmax - lmin);
    }
    PG_HIP(ctx, hipMemsetAsync(p->sync, 0, SY_WORDS * sizeof(unsigned), ctx->stream));
    if (a.single_reduction) PG_HIP(ctx, hipMemsetAsync(p->flags, 0, 16 * (size_t)p->G * sizeof(unsigned), ctx->stream));
    void* args[] = {&a};
    const void* fn = nullptr;
    if (a.cheb_k > 1) {
        switch (p->rb) {
            case 1: fn = reinterpret_cast<const void*>(k_cg2d_persistent_cheb<1>); break;
            case 2: fn = reinterpret_cast<const void*>(k_cg2d_persistent_cheb<2>); break;
            default: fn = reinterpret_cast<const void*>(k_cg2d_persistent_cheb<4>); break;
        }
    } else if (a.single_reduction) {
        switch (p->rb) {
            case 1: fn = reinterpret_cast<const void*>(k_cg2d_persistent_sr<1>); break;
            case 2: fn = reinterpret_cast<const void*>(k_cg2d_persistent_sr<2>); break;
            case 4: fn = reinterpret_cast<const void*>(k_cg2d_persistent_sr<4>); break;
            default: fn = reinterpret_cast<const void*>(k_cg2d_persistent_sr<8>); break;
        }
    } else {
        switch (p->rb) {
            case 1: fn = reinterpret_cast<const void*>(k_cg2d_persistent<1>); break;
            case 2: fn = reinterpret_cast<const void*>(k_cg2d_persistent<2>); break;
            case 4: fn = reinterpret_cast<const void*>(k_cg2d_persistent<4>); break;
            default: fn = reinterpret_cast<const void*>(k_cg2d_persistent<8>); break;
        }
    }
    // persist2d_launch = 1: an ordinary launch of the same grid.  The cooperative launch only adds the runtime's co-residency check (G <= num_cu
    // workgroups, one per CU: resident on an otherwise idle device); the kernel's own barriers carry a spin limit and a fail flag, so a grid that
    // is not resident ends in the error below instead of a hang.  For runs under rocprofv3, whose exit handler faults in libhsa-runtime64 after a
    // cooperative launch (profiles/r03_rocprofv3_exit_abort.log).
    if (ctx->persist2d_launch == 1) PG_HIP(ctx, hipLaunchKernel(fn, dim3(p->G), dim3(kNW * 64), args, 0, ctx->stream));
    else PG_HIP(ctx, hipLaunchCooperativeKernel(fn, dim3(p->G), dim3(kNW * 64), args, 0, ctx->stream));
    PG_HIP(ctx, hipMemcpyAsync(p->h_sync, p->sync, 16, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (p->h_sync[1] != 0u) return pdegpu_set_err(ctx, PDEGPU_EHIP, "persistent 2-D CG: a grid barrier timed out (the grid was not co-resident?)");
    if (iters_done) *iters_done = (int)p->h_sync[2];
    return PDEGPU_OK;
}
