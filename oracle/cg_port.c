/* CPU ORACLE, C port (test infrastructure, NOT product code): matrix-free CG for the constant-coefficient
 * heat operator A = cM*M + cK*K of the reference's P1 discretisation on a uniform box with every boundary
 * node Dirichlet (fenics_mcp_server.py:626-628, 657-661, 702-709), OpenMP over the host cores.
 *
 * It restates the SAME algorithm the GPU path runs (classic CG on the free nodes, node stencil from the
 * closed-form P1 weights checked in tests/test_oracle_known_answers.py), not the reference's sparse LU:
 * it exists (1) as a second, independent checker of the CG iterates the HIP kernels produce and (2) as the
 * same-algorithm CPU baseline bench.py times on the GPU node's own host cores ("kind": "port").
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Parity status: unpinned by the reference (no fixtures exist; see oracle/fem_p1.py header).
 *
 * Grid: (NX,NY,NZ) nodes, x fastest, dense (no padding).  w[8] = interior stencil weights
 * {centre, +-x, +-y, +-z, +-(1,1,0), +-(1,0,1), +-(0,1,1), +-(1,1,1)}.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

static double now_s(void) { return omp_get_wtime(); }

static void apply(const double* restrict x, double* restrict y, int NX, int NY, int NZ, const double* w, int full15) {
    const long sx = 1, sy = NX, sz = (long)NX * NY;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k < NZ - 1; ++k)
        for (int j = 1; j < NY - 1; ++j) {
            const long base = (long)k * sz + (long)j * sy;
            for (int i = 1; i < NX - 1; ++i) {
                const long c = base + i;
                double s = w[0] * x[c] + w[1] * (x[c - sx] + x[c + sx]) + w[2] * (x[c - sy] + x[c + sy]) +
                           w[3] * (x[c - sz] + x[c + sz]);
                if (full15)
                    s += w[4] * (x[c + sx + sy] + x[c - sx - sy]) + w[5] * (x[c + sx + sz] + x[c - sx - sz]) +
                         w[6] * (x[c + sy + sz] + x[c - sy - sz]) + w[7] * (x[c + sx + sy + sz] + x[c - sx - sy - sz]);
                y[c] = s;
            }
        }
}

static double dot(const double* a, const double* b, long n) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (long i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* Runs `iters` CG iterations for A x = b, b = rhs_const on free nodes, starting from x0 = bc_value on the
 * boundary and 0 inside.  rr_hist[0..iters] receives r.r after each iteration (rr_hist[0] = initial).
 * x_out (nullable) receives the iterate.  Returns the wall time of the iteration loop in seconds. */
double cgport_run(int NX, int NY, int NZ, const double* w, int full15, double rhs_const, double bc_value, int iters,
                  int nthreads, double* x_out, double* rr_hist) {
    if (nthreads > 0) omp_set_num_threads(nthreads);
    const long n = (long)NX * NY * NZ;
    double* x = (double*)malloc(n * sizeof(double));
    double* r = (double*)malloc(n * sizeof(double));
    double* p = (double*)malloc(n * sizeof(double));
    double* q = (double*)malloc(n * sizeof(double));
    if (!x || !r || !p || !q) { free(x); free(r); free(p); free(q); return -1.0; }
    const long sz = (long)NX * NY;
    /* first touch in the distribution the sweeps use: the vector updates run over the flat index with schedule(static) and the
     * operator over collapse(2)(k, j) with schedule(static), i.e. both give thread t the t-th contiguous slice of the arrays
     * (a static loop over the 513 planes alone would hand 5 planes each to 103 of 128 threads and leave the pages of the
     * other threads' slices on the wrong NUMA node) */
#pragma omp parallel for schedule(static)
    for (long c = 0; c < n; ++c) {
        const long e = c % sz;
        const int k = (int)(c / sz), j = (int)(e / NX), i = (int)(e % NX);
        const int bnd = (i == 0) | (i == NX - 1) | (j == 0) | (j == NY - 1) | (k == 0) | (k == NZ - 1);
        x[c] = bnd ? bc_value : 0.0;
        r[c] = p[c] = q[c] = 0.0;
    }
    /* r = P(b - A x0); Dirichlet rows stay 0 */
    apply(x, q, NX, NY, NZ, w, full15);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k < NZ - 1; ++k)
        for (int j = 1; j < NY - 1; ++j)
            for (int i = 1; i < NX - 1; ++i) {
                const long c = (long)k * sz + (long)j * NX + i;
                r[c] = rhs_const - q[c];
                p[c] = r[c];
            }
#pragma omp parallel for schedule(static)
    for (long c = 0; c < n; ++c) q[c] = 0.0;
    double rr = dot(r, r, n);
    if (rr_hist) rr_hist[0] = rr;
    const double t0 = now_s();
    for (int it = 0; it < iters; ++it) {
        apply(p, q, NX, NY, NZ, w, full15);
        const double pq = dot(p, q, n);
        const double alpha = pq > 0.0 ? rr / pq : 0.0;
        double rr_new = 0.0;
#pragma omp parallel for reduction(+ : rr_new) schedule(static)
        for (long c = 0; c < n; ++c) {
            x[c] += alpha * p[c];
            r[c] -= alpha * q[c];
            rr_new += r[c] * r[c];
        }
        const double beta = rr > 0.0 ? rr_new / rr : 0.0;
#pragma omp parallel for schedule(static)
        for (long c = 0; c < n; ++c) p[c] = r[c] + beta * p[c];
        rr = rr_new;
        if (rr_hist) rr_hist[it + 1] = rr;
    }
    const double t1 = now_s();
    if (x_out) memcpy(x_out, x, n * sizeof(double));
    free(x); free(r); free(p); free(q);
    return t1 - t0;
}

int cgport_max_threads(void) { return omp_get_max_threads(); }
