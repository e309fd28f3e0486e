// tridiag.hip — direct solver for the 1-D tools (gfx950).
//
// The reference solves every 1-D system with a sparse LU (`solve(a == L, u, bcs)`, fenics_mcp_server.py:265, 311,
// 1538; `project`, :284, 290, 1542, 1546), which for a P1 interval mesh is the Thomas recurrence on a tridiagonal
// matrix.  A 1-D grid has no parallel slack worth a CG (config 1: 1001 unknowns, 35 000 CG iterations of four
// launches each), so the whole backward-Euler loop runs inside ONE launch of ONE workgroup:
//   factor once:  inv_i = 1 / (d_i - a_i c'_{i-1}),  c'_i = c_i inv_i
//   every step:   b = M (u + dt f)  (all 256 lanes),  forward / backward substitution (lane 0, operands staged
//                 through LDS by all lanes so the serial chain only ever waits on its own FMA),  snapshot copy.
// Dirichlet rows are identity rows with the boundary value on the right-hand side — the reference's row replacement.
#include "common.h"

namespace {

constexpr int kCH = 1024;          // elements staged per chunk (3 LDS arrays = 24 KB)

struct TriArgs {
    int N;
    const double* rows;            // [8][N]: aL aD aU | mL mD mU | dirichlet flag | dirichlet value
    double* work;                  // [3][N]: c' | inv | scratch (right-hand side, then the forward sweep)
    double* u;                     // [N] state (in: u^n with boundary values imposed; out: u^{n+steps})
    double shift;                  // b = M (u + shift): dt*f (transient) or f (steady, u = 0)
    int rhs_given;                 // 1: scratch already holds the right-hand side (project / mass solve)
    int nsteps;                    // steps of this launch
    int step0;                     // steps done before this launch
    int total_steps;
    int stride;                    // snapshot every `stride` steps and after the last one
    double* snaps;                 // device [n][N] snapshots written by this launch, in order
};

__global__ void k_tri_setup(GridDev g, const double* tabA, const double* tabM, const double* ellA, const double* ellM,
                            long long stride, double* rows, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double val = 0.0;
    const bool dir = pg_dirichlet(g, i, 0, 0, &val);
    double aL, aD, aU, mL, mD, mU;
    if (ellA) {                    // weighted rows: slot 0 = centre, 1 = x-1, 2 = x+1 (weighted.hip)
        aD = ellA[i];
        aL = ellA[stride + i];
        aU = ellA[2 * stride + i];
        mD = ellM[i];
        mL = ellM[stride + i];
        mU = ellM[2 * stride + i];
    } else {                       // class tables: offsets (dx, 0, 0) = 12, 13, 14
        const int cls = pg_axis_class(i, g.NX) + 3 + 9;
        aL = tabA[cls * 27 + 12];
        aD = tabA[cls * 27 + 13];
        aU = tabA[cls * 27 + 14];
        mL = tabM[cls * 27 + 12];
        mD = tabM[cls * 27 + 13];
        mU = tabM[cls * 27 + 14];
    }
    if (i == 0) aL = mL = 0.0;
    if (i == N - 1) aU = mU = 0.0;
    if (dir) {
        aL = 0.0;
        aD = 1.0;
        aU = 0.0;
    }
    rows[i] = aL;
    rows[(size_t)N + i] = aD;
    rows[2 * (size_t)N + i] = aU;
    rows[3 * (size_t)N + i] = mL;
    rows[4 * (size_t)N + i] = mD;
    rows[5 * (size_t)N + i] = mU;
    rows[6 * (size_t)N + i] = dir ? 1.0 : 0.0;
    rows[7 * (size_t)N + i] = val;
}

// One workgroup.  Serial recurrences run on lane 0 of wave 0 over LDS-staged chunks, 8 elements per register batch.
__global__ __launch_bounds__(256) void k_tri_solve(TriArgs a) {
    __shared__ double sA[kCH], sB[kCH], sC[kCH];
    const int N = a.N;
    const int tid = threadIdx.x;
    const double* aL = a.rows;
    const double* aD = a.rows + (size_t)N;
    const double* aU = a.rows + 2 * (size_t)N;
    const double* mL = a.rows + 3 * (size_t)N;
    const double* mD = a.rows + 4 * (size_t)N;
    const double* mU = a.rows + 5 * (size_t)N;
    const double* dflag = a.rows + 6 * (size_t)N;
    const double* dval = a.rows + 7 * (size_t)N;
    double* cp = a.work;
    double* inv = a.work + (size_t)N;
    double* scr = a.work + 2 * (size_t)N;

    // ---- factorisation (Thomas): inv_i, c'_i --------------------------------------------------------------------
    {
        double carry = 0.0;        // c'_{i-1}
        for (int base = 0; base < N; base += kCH) {
            const int n = min(kCH, N - base);
            for (int t = tid; t < n; t += 256) {
                sA[t] = aL[base + t];
                sB[t] = aD[base + t];
                sC[t] = aU[base + t];
            }
            __syncthreads();
            if (tid == 0) {
                for (int t = 0; t < n; t += 8) {
                    double ra[8], rb[8], rc[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        ra[q] = sA[t + q];
                        rb[q] = sB[t + q];
                        rc[q] = sC[t + q];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (t + q < n) {
                            const double iv = 1.0 / (rb[q] - ra[q] * carry);
                            carry = rc[q] * iv;
                            sB[t + q] = iv;
                            sC[t + q] = carry;
                        }
                }
            }
            __syncthreads();
            for (int t = tid; t < n; t += 256) {
                inv[base + t] = sB[t];
                cp[base + t] = sC[t];
            }
            __syncthreads();
        }
    }

    int snap = 0;
    for (int st = 0; st < a.nsteps; ++st) {
        // ---- right-hand side b = M (u + shift), Dirichlet rows = boundary value --------------------------------
        if (!a.rhs_given) {
            for (int i = tid; i < N; i += 256) {
                double b;
                if (dflag[i] != 0.0) {
                    b = dval[i];
                } else {
                    const double ul = i > 0 ? a.u[i - 1] + a.shift : 0.0;
                    const double uc = a.u[i] + a.shift;
                    const double ur = i < N - 1 ? a.u[i + 1] + a.shift : 0.0;
                    b = mL[i] * ul + mD[i] * uc + mU[i] * ur;
                }
                scr[i] = b;
            }
        }
        __threadfence_block();
        __syncthreads();
        // ---- forward substitution: y_i = (b_i - a_i y_{i-1}) inv_i ----------------------------------------------
        {
            double carry = 0.0;
            for (int base = 0; base < N; base += kCH) {
                const int n = min(kCH, N - base);
                for (int t = tid; t < n; t += 256) {
                    sA[t] = aL[base + t];
                    sB[t] = inv[base + t];
                    sC[t] = scr[base + t];
                }
                __syncthreads();
                if (tid == 0) {
                    for (int t = 0; t < n; t += 8) {
                        double ra[8], rb[8], rc[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            ra[q] = sA[t + q];
                            rb[q] = sB[t + q];
                            rc[q] = sC[t + q];
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (t + q < n) {
                                carry = (rc[q] - ra[q] * carry) * rb[q];
                                sC[t + q] = carry;
                            }
                    }
                }
                __syncthreads();
                for (int t = tid; t < n; t += 256) scr[base + t] = sC[t];
                __syncthreads();
            }
        }
        __threadfence_block();
        __syncthreads();
        // ---- backward substitution: u_i = y_i - c'_i u_{i+1} ----------------------------------------------------
        {
            double carry = 0.0;
            const int nchunks = (N + kCH - 1) / kCH;
            for (int c = nchunks - 1; c >= 0; --c) {
                const int base = c * kCH;
                const int n = min(kCH, N - base);
                for (int t = tid; t < n; t += 256) {
                    sA[t] = cp[base + t];
                    sC[t] = scr[base + t];
                }
                __syncthreads();
                if (tid == 0) {
                    for (int t = ((n - 1) / 8) * 8; t >= 0; t -= 8) {
                        double ra[8], rc[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            ra[q] = sA[t + q];
                            rc[q] = sC[t + q];
                        }
#pragma unroll
                        for (int q = 7; q >= 0; --q)
                            if (t + q < n) {
                                carry = rc[q] - ra[q] * carry;
                                sC[t + q] = carry;
                            }
                    }
                }
                __syncthreads();
                for (int t = tid; t < n; t += 256) a.u[base + t] = sC[t];
                __syncthreads();
            }
        }
        __threadfence_block();
        __syncthreads();
        // ---- snapshot ------------------------------------------------------------------------------------------
        const int step = a.step0 + st + 1;
        if (a.snaps && (step % a.stride == 0 || step == a.total_steps)) {
            double* dst = a.snaps + (size_t)snap * N;
            for (int i = tid; i < N; i += 256) dst[i] = a.u[i];
            ++snap;
        }
    }
}

}  // namespace

int tridiag_setup(pdegpu_ctx* ctx, const GridDev& g, const double* tabA, const double* tabM, const double* ellA,
                  const double* ellM, long long ell_stride, double* rows) {
    const int N = g.NX;
    hipLaunchKernelGGL(k_tri_setup, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, g, tabA, tabM, ellA, ellM,
                       ell_stride, rows, N);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}

int tridiag_run(pdegpu_ctx* ctx, int N, const double* rows, double* work, double* u, double shift, int rhs_given, int nsteps,
                int step0, int total_steps, int stride, double* snaps) {
    TriArgs a{};
    a.N = N;
    a.rows = rows;
    a.work = work;
    a.u = u;
    a.shift = shift;
    a.rhs_given = rhs_given;
    a.nsteps = nsteps;
    a.step0 = step0;
    a.total_steps = total_steps;
    a.stride = stride < 1 ? 1 : stride;
    a.snaps = snaps;
    hipLaunchKernelGGL(k_tri_solve, dim3(1), dim3(256), 0, ctx->stream, a);
    PG_HIP(ctx, hipGetLastError());
    return PDEGPU_OK;
}
