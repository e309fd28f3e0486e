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
