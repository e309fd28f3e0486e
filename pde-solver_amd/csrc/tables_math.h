// tables_math.h — macro-element integration and class-table fold, shared by the device kernel
// (tables.hip) and its host twin (pdegpu_heat_tables_host, used by the CPU test-suite).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include "../../include/pdegpu.h"



// stiffness / mass entry (pa, pb = positions inside the simplex) of one P1 simplex with the given
// local macro-cell vertices; exact integration (constant gradients).
__host__ __device__ inline void simplex_entry(int dim, const int* verts, const double* h, int pa, int pb, double* k, double* m) {
    double J[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};   // columns = edge vectors from vertex 0
    for (int c = 0; c < dim; ++c)
        for (int r = 0; r < dim; ++r) {
            int b1 = (verts[c + 1] >> r) & 1, b0 = (verts[0] >> r) & 1;
            J[r][c] = (b1 - b0) * h[r];
        }
    double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                 J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
    double inv[3][3];
    inv[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det;
    inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
    inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
    inv[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
    inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
    inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
    inv[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
    inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
    inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
    // grad lambda_p (p>=1) = row p-1 of J^-1; grad lambda_0 = -sum
    double ga[3] = {0, 0, 0}, gb[3] = {0, 0, 0};
    for (int r = 0; r < dim; ++r) {
        if (pa == 0) { for (int p = 0; p < dim; ++p) ga[r] -= inv[p][r]; } else ga[r] = inv[pa - 1][r];
        if (pb == 0) { for (int p = 0; p < dim; ++p) gb[r] -= inv[p][r]; } else gb[r] = inv[pb - 1][r];
    }
    const double fact = dim == 1 ? 1.0 : (dim == 2 ? 2.0 : 6.0);
    const double vol = fabs(det) / fact;
    double gg = 0;
    for (int r = 0; r < dim; ++r) gg += ga[r] * gb[r];
    *k = vol * gg;
    *m = vol * (pa == pb ? 2.0 : 1.0) / ((dim + 1) * (dim + 2));
}

// macro-element entry (a, b) for the reference's P1 space on the simplex split of one cell
__host__ __device__ inline void macro_p1(int dim, const double* h, int a, int b, double* k, double* m) {
    // DOLFIN 2019.1.0 BoxMesh: six Kuhn tets sharing the v0-v7 diagonal; RectangleMesh diagonal="right"
    constexpr int kKuhn[6][4] = {{0, 1, 3, 7}, {0, 1, 7, 5}, {0, 5, 7, 4}, {0, 3, 2, 7}, {0, 6, 4, 7}, {0, 2, 6, 7}};
    constexpr int kTri[2][3] = {{0, 1, 3}, {0, 2, 3}};
    double ks = 0, ms = 0;
    const int nsimp = dim == 1 ? 1 : (dim == 2 ? 2 : 6);
    for (int s = 0; s < nsimp; ++s) {
        int verts[4] = {0, 1, 0, 0};
        if (dim == 2) { for (int p = 0; p < 3; ++p) verts[p] = kTri[s][p]; }
        if (dim == 3) { for (int p = 0; p < 4; ++p) verts[p] = kKuhn[s][p]; }
        int pa = -1, pb = -1;
        for (int p = 0; p <= dim; ++p) {
            if (verts[p] == a) pa = p;
            if (verts[p] == b) pb = p;
        }
        if (pa < 0 || pb < 0) continue;
        double k1, m1;
        simplex_entry(dim, verts, h, pa, pb, &k1, &m1);
        ks += k1;
        ms += m1;
    }
    *k = ks;
    *m = ms;
}

// macro-element entry for the multilinear (Q1) element, 2-point Gauss per axis (exact on a box)
__host__ __device__ inline void macro_q1(int dim, const double* h, int a, int b, double* k, double* m) {
    const double gp[2] = {0.5 - 0.28867513459481288225, 0.5 + 0.28867513459481288225};
    double vol = 1.0;
    for (int r = 0; r < dim; ++r) vol *= h[r];
    const int nq = 1 << dim;
    double ks = 0, ms = 0;
    for (int q = 0; q < nq; ++q) {
        double Na = 1, Nb = 1, fa[3], fb[3];
        for (int r = 0; r < dim; ++r) {
            double xi = gp[(q >> r) & 1];
            fa[r] = ((a >> r) & 1) ? xi : 1.0 - xi;
            fb[r] = ((b >> r) & 1) ? xi : 1.0 - xi;
            Na *= fa[r];
            Nb *= fb[r];
        }
        double gg = 0;
        for (int r = 0; r < dim; ++r) {
            double da = (((a >> r) & 1) ? 1.0 : -1.0) / h[r], db = (((b >> r) & 1) ? 1.0 : -1.0) / h[r];
            for (int e = 0; e < dim; ++e)
                if (e != r) { da *= fa[e]; db *= fb[e]; }
            gg += da * db;
        }
        ks += gg;
        ms += Na * Nb;
    }
    const double wq = vol / nq;
    *k = ks * wq;
    *m = ms * wq;
}


// macro-element entry (a,b) for the chosen element family
__host__ __device__ inline void macro_entry(int dim, int element_mode, const double* h, int a, int b, double* k, double* m) {
    if (element_mode == PDEGPU_ELEM_Q1) macro_q1(dim, h, a, b, k, m);
    else macro_p1(dim, h, a, b, k, m);
}

// fold the 8x8 macro-element matrices into the coupling weight of (boundary class, neighbour offset):
// sum over the incident cells that exist for that class and contain both nodes.
__host__ __device__ inline void fold_entry(int dim, const double (*sK)[8], const double (*sM)[8], int cls, int o,
                                           double* wk_out, double* wm_out) {
    const int c[3] = {cls % 3, (cls / 3) % 3, cls / 9};
    const int d[3] = {o % 3 - 1, (o / 3) % 3 - 1, o / 9 - 1};
    double wk = 0, wm = 0;
    for (int s = 0; s < 8; ++s) {            // incident cell: node is local vertex bit sb of cell (idx - sb)
        bool ok = true;
        int la = 0, lb = 0;
        for (int r = 0; r < 3; ++r) {
            const int sb = (s >> r) & 1;
            if (r >= dim) { ok &= (sb == 0) & (d[r] == 0); continue; }
            if (c[r] == 0 && sb == 1) ok = false;   // low face: no cell on the low side
            if (c[r] == 2 && sb == 0) ok = false;   // high face: no cell on the high side
            const int nb = sb + d[r];
            if (nb < 0 || nb > 1) ok = false;
            la |= sb << r;
            lb |= (nb & 1) << r;
        }
        if (!ok) continue;
        wk += sK[la][lb];
        wm += sM[la][lb];
    }
    *wk_out = wk;
    *wm_out = wm;
}

// ---------------------------------------------------------------------------------------------
// linear elasticity:  a(u,v) = int sigma(u):eps(v),  sigma = lam tr(eps) I + 2 mu eps
// (fenics_mcp_server.py:1653-1667, 1808-1817, 1677, 1827) on the same simplex split.
// ---------------------------------------------------------------------------------------------
// the 15 neighbour offsets coupled by the Kuhn / "right"-diagonal split (0, +-axes, same-sign diagonals);
// index = position in this list, value = offset id o = (dx+1) + 3(dy+1) + 9(dz+1)
#define PG_NOFF15 15
__host__ __device__ inline int pg_off15(int q) {
    constexpr int kOff[15] = {13, 12, 14, 10, 16, 4, 22, 9, 17, 3, 23, 1, 25, 0, 26};
    return kOff[q];
}

// vertices of simplex s of the macro cell
__host__ __device__ inline void pg_simplex_verts(int dim, int s, int* verts) {
    constexpr int kKuhn[6][4] = {{0, 1, 3, 7}, {0, 1, 7, 5}, {0, 5, 7, 4}, {0, 3, 2, 7}, {0, 6, 4, 7}, {0, 2, 6, 7}};
    constexpr int kTri[2][3] = {{0, 1, 3}, {0, 2, 3}};
    verts[0] = 0; verts[1] = 1; verts[2] = 0; verts[3] = 0;
    if (dim == 2) { for (int p = 0; p < 3; ++p) verts[p] = kTri[s][p]; }
    if (dim == 3) { for (int p = 0; p < 4; ++p) verts[p] = kKuhn[s][p]; }
}
__host__ __device__ inline int pg_num_simplices(int dim) { return dim == 1 ? 1 : (dim == 2 ? 2 : 6); }

// constant gradients of the barycentric basis of simplex `verts` (g[p][r]) and its volume
__host__ __device__ inline void simplex_grads(int dim, const int* verts, const double* h, double (*g)[3], double* vol) {
    double J[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int c = 0; c < dim; ++c)
        for (int r = 0; r < dim; ++r) J[r][c] = (((verts[c + 1] >> r) & 1) - ((verts[0] >> r) & 1)) * h[r];
    const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                       J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
    double inv[3][3];
    inv[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det;
    inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
    inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
    inv[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
    inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
    inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
    inv[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
    inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
    inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
    for (int r = 0; r < 3; ++r) {
        g[0][r] = 0.0;
        for (int p = 1; p <= 3; ++p) g[p][r] = 0.0;
    }
    for (int r = 0; r < dim; ++r) {
        for (int p = 1; p <= dim; ++p) {
            g[p][r] = inv[p - 1][r];
            g[0][r] -= inv[p - 1][r];
        }
    }
    *vol = fabs(det) / (dim == 1 ? 1.0 : (dim == 2 ? 2.0 : 6.0));
}

// 3x3 (dim x dim used) block of the macro-element elasticity matrix between local vertices la, lb
__host__ __device__ inline void macro_elasticity_block(int dim, const double* h, double lam, double mu, int la, int lb,
                                                        double* blk /*[9], row-major a*3+b*/) {
    for (int e = 0; e < 9; ++e) blk[e] = 0.0;
    const int ns = pg_num_simplices(dim);
    for (int s = 0; s < ns; ++s) {
        int verts[4];
        pg_simplex_verts(dim, s, verts);
        int pa = -1, pb = -1;
        for (int p = 0; p <= dim; ++p) {
            if (verts[p] == la) pa = p;
            if (verts[p] == lb) pb = p;
        }
        if (pa < 0 || pb < 0) continue;
        double g[4][3], vol;
        simplex_grads(dim, verts, h, g, &vol);
        double gg = 0.0;
        for (int r = 0; r < dim; ++r) gg += g[pa][r] * g[pb][r];
        for (int a = 0; a < dim; ++a)
            for (int b = 0; b < dim; ++b)
                blk[a * 3 + b] += vol * (lam * g[pa][a] * g[pb][b] + mu * g[pa][b] * g[pb][a] + (a == b ? mu * gg : 0.0));
    }
}

// class-table entry: block coupling node of class `cls` to its neighbour at 15-list position q
// (sum over the incident cells that exist and contain both nodes) — same fold as the scalar tables
__host__ __device__ inline void fold_elasticity_block(int dim, const double* h, double lam, double mu, int cls, int q,
                                                       double* blk /*[9]*/) {
    const int o = pg_off15(q);
    const int c[3] = {cls % 3, (cls / 3) % 3, cls / 9};
    const int d[3] = {o % 3 - 1, (o / 3) % 3 - 1, o / 9 - 1};
    for (int e = 0; e < 9; ++e) blk[e] = 0.0;
    for (int s = 0; s < 8; ++s) {
        bool ok = true;
        int la = 0, lb = 0;
        for (int r = 0; r < 3; ++r) {
            const int sb = (s >> r) & 1;
            if (r >= dim) { ok &= (sb == 0) & (d[r] == 0); continue; }
            if (c[r] == 0 && sb == 1) ok = false;
            if (c[r] == 2 && sb == 0) ok = false;
            const int nb = sb + d[r];
            if (nb < 0 || nb > 1) ok = false;
            la |= sb << r;
            lb |= (nb & 1) << r;
        }
        if (!ok) continue;
        double b1[9];
        macro_elasticity_block(dim, h, lam, mu, la, lb, b1);
        for (int e = 0; e < 9; ++e) blk[e] += b1[e];
    }
}

// ---------------------------------------------------------------------------------------------
// weighted (curvilinear) heat forms, fenics_mcp_server.py:769-1464:  a = w u v + dt kappa w grad u . grad v with
// w an Expression of degree 1 or 2 = its P1 / P2 interpolant on every simplex, integrated exactly.
// The operator is no longer translation invariant, so every node gets its own 15 coupling weights.
// ---------------------------------------------------------------------------------------------

__host__ __device__ inline double pg_weight_eval(int kind, const double* x) {
    if (kind == PDEGPU_WEIGHT_ONE) return 1.0;
    if (kind == PDEGPU_WEIGHT_R) return x[0];
    if (kind == PDEGPU_WEIGHT_R2) return x[0] * x[0];
    if (kind == PDEGPU_WEIGHT_RCYL) return sqrt(x[1] * x[1] + x[2] * x[2]);
    return x[0] * x[0] * sin(x[1]);
}
__host__ __device__ inline int pg_weight_degree(int kind) { return (kind == PDEGPU_WEIGHT_R || kind == PDEGPU_WEIGHT_ONE) ? 1 : 2; }

// position q in the 15-offset list of offset id o (or -1)
__host__ __device__ inline int pg_q_of_o(int o) {
    constexpr int kQ[27] = {13, 11, -1, 9, 5, -1, -1, -1, -1, 7, 3, -1, 1, 0, 2, -1, 4, 8, -1, -1, -1, -1, 6, 10, -1, 12, 14};
    return kQ[o];
}

// int over a unit-volume d-simplex of prod lambda_i^{e_i} = d! prod e_i! / (sum e_i + d)!
__host__ __device__ inline double pg_mono(int d, const int* e) {
    constexpr double kFact[10] = {1, 1, 2, 6, 24, 120, 720, 5040, 40320, 362880};
    int s = 0;
    double num = kFact[d];
    for (int i = 0; i <= d; ++i) {
        s += e[i];
        num *= kFact[e[i]];
    }
    return num / kFact[s + d];
}

// row of node (i, j, k): rowK[15], rowM[15] (positions of the 15-offset list) and the load entry int w phi_i
__host__ __device__ inline void weighted_row(int dim, const int* nn /*node counts*/, const double* origin, const double* h,
                                             const double* lengths, int kind, double core_radius, double core_ratio, int i,
                                             int j, int k, double* rowK, double* rowM, double* load) {
    for (int q = 0; q < PG_NOFF15; ++q) { rowK[q] = 0.0; rowM[q] = 0.0; }
    *load = 0.0;
    const int idx3[3] = {i, j, k};
    const int deg = pg_weight_degree(kind);
    const int ns = pg_num_simplices(dim);
    for (int s = 0; s < (1 << dim); ++s) {               // incident cell: node is local vertex `s` of cell (idx - bits(s))
        int c0[3] = {0, 0, 0};
        bool ok = true;
        for (int r = 0; r < dim; ++r) {
            c0[r] = idx3[r] - ((s >> r) & 1);
            ok &= (c0[r] >= 0) & (c0[r] < nn[r] - 1);
        }
        if (!ok) continue;
        for (int t = 0; t < ns; ++t) {
            int verts[4];
            pg_simplex_verts(dim, t, verts);
            int pa = -1;
            for (int p = 0; p <= dim; ++p)
                if (verts[p] == s) pa = p;
            if (pa < 0) continue;
            double g[4][3], vol;
            simplex_grads(dim, verts, h, g, &vol);
            // vertex coordinates and the interpolated weight as sum_t coef_t * prod lambda^{exp_t}
            // vertex coordinates exactly as the mesh generators form them, a + (b - a) * i / n (mul, div, add: the same
            // IEEE operations as the oracle, so borderline `r < core_radius` tests agree bit for bit)
            double X[4][3];
            for (int p = 0; p <= dim; ++p)
                for (int r = 0; r < 3; ++r) {
                    X[p][r] = r < dim ? origin[r] + (lengths[r] * (double)(c0[r] + ((verts[p] >> r) & 1))) / (double)(nn[r] - 1) : 0.0;
                }
            double coef[14];
            int expo[14][4];
            int nt = 0;
            for (int p = 0; p <= dim; ++p) {
                const double wv = pg_weight_eval(kind, X[p]);
                if (deg == 1) {
                    coef[nt] = wv;
                    for (int e = 0; e < 4; ++e) expo[nt][e] = (e == p);
                    ++nt;
                } else {                                  // vertex basis of P2: 2 l^2 - l
                    coef[nt] = 2.0 * wv;
                    for (int e = 0; e < 4; ++e) expo[nt][e] = 2 * (e == p);
                    ++nt;
                    coef[nt] = -wv;
                    for (int e = 0; e < 4; ++e) expo[nt][e] = (e == p);
                    ++nt;
                }
            }
            if (deg == 2)
                for (int a = 0; a <= dim; ++a)
                    for (int b = a + 1; b <= dim; ++b) {   // edge basis of P2: 4 l_a l_b, value at the edge midpoint
                        double mid[3];
                        for (int r = 0; r < 3; ++r) mid[r] = 0.5 * (X[a][r] + X[b][r]);
                        coef[nt] = 4.0 * pg_weight_eval(kind, mid);
                        for (int e = 0; e < 4; ++e) expo[nt][e] = (e == a) + (e == b);
                        ++nt;
                    }
            // composite core: DG0 diffusivity, a simplex is core iff all vertices and the midpoint are inside (:541-550)
            double kfac = 1.0;
            if (core_radius > 0.0) {
                bool core = true;
                double mid[3] = {0, 0, 0};
                for (int p = 0; p <= dim; ++p) {
                    core &= sqrt(X[p][1] * X[p][1] + X[p][2] * X[p][2]) < core_radius;
                    for (int r = 0; r < 3; ++r) mid[r] += X[p][r] / (dim + 1);
                }
                core &= sqrt(mid[1] * mid[1] + mid[2] * mid[2]) < core_radius;
                if (core) kfac = core_ratio;
            }
            double w_int = 0.0, w_a = 0.0;
            for (int u = 0; u < nt; ++u) {
                int e[4] = {expo[u][0], expo[u][1], expo[u][2], expo[u][3]};
                w_int += coef[u] * pg_mono(dim, e);
                e[pa] += 1;
                w_a += coef[u] * pg_mono(dim, e);
            }
            *load += vol * w_a;
            for (int p = 0; p <= dim; ++p) {
                int o = 13;                               // offset id of vertex p relative to this node
                const int mult[3] = {1, 3, 9};
                for (int r = 0; r < dim; ++r) o += mult[r] * (((verts[p] >> r) & 1) - ((s >> r) & 1));
                const int q = pg_q_of_o(o);
                double gg = 0.0;
                for (int r = 0; r < dim; ++r) gg += g[pa][r] * g[p][r];
                double w_ab = 0.0;
                for (int u = 0; u < nt; ++u) {
                    int e[4] = {expo[u][0], expo[u][1], expo[u][2], expo[u][3]};
                    e[pa] += 1;
                    e[p] += 1;
                    w_ab += coef[u] * pg_mono(dim, e);
                }
                rowK[q] += kfac * vol * w_int * gg;
                rowM[q] += vol * w_ab;
            }
        }
    }
}
