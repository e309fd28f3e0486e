/* A C host of the drop-in boundary (include/pdegpu.h): the steady branch of _solve_heat_3d_raw (fenics_mcp_server.py:657-661) on a
 * BoxMesh(1, 1, 1; n, n, n) with T = 0 on the whole boundary and f = 1, solved on the GPU by the library; prints the centre value and
 * the CG statistics.  No Python, no torch: what a compiled host links.
 *   gcc -std=c99 -I include examples/box_heat.c -L pde-solver_amd/lib -lpdegpu -Wl,-rpath,$PWD/pde-solver_amd/lib -o box_heat && ./box_heat 64 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pdegpu.h"

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 64;
    if (n < 2 || n % 2) {
        fprintf(stderr, "cells per axis must be even and >= 2\n");
        return 2;
    }
    pdegpu_ctx* ctx = NULL;
    if (pdegpu_create(&ctx, 0, NULL) != PDEGPU_OK) {
        fprintf(stderr, "no usable gfx950 device (there is no CPU fallback)\n");
        return 1;
    }
    pdegpu_grid grid;
    memset(&grid, 0, sizeof grid);
    grid.dim = 3;
    for (int a = 0; a < 3; ++a) {
        grid.n[a] = n;
        grid.L[a] = 1.0;
    }
    grid.element_mode = PDEGPU_ELEM_P1SPLIT;
    pdegpu_heat heat;
    memset(&heat, 0, sizeof heat);
    heat.kappa = 1.0;
    heat.steady = 1;
    heat.num_steps = 1;
    heat.source = 1.0;
    heat.ic_type = PDEGPU_IC_ZERO;
    heat.bc_mode = PDEGPU_BC_UNIFORM;
    heat.bc_present[0] = 1;
    heat.snapshot_stride = 1;
    heat.rtol = 1e-10;
    heat.max_iter = 100000;
    heat.fast_path = 1;
    const long long N = (long long)(n + 1) * (n + 1) * (n + 1);
    const int n_snap = pdegpu_heat_num_snapshots(&heat);
    double* values = malloc((size_t)n_snap * (size_t)N * sizeof(double));
    double* times = malloc((size_t)n_snap * sizeof(double));
    pdegpu_stats st;
    memset(&st, 0, sizeof st);
    const int rc = pdegpu_heat_solve(ctx, &grid, &heat, NULL, values, times, &st);
    if (rc != PDEGPU_OK) {
        fprintf(stderr, "pdegpu_heat_solve failed (%d): %s\n", rc, pdegpu_last_error(ctx));
        return 1;
    }
    const long long c = (long long)(n / 2) * (n + 1) * (n + 1) + (long long)(n / 2) * (n + 1) + n / 2;
    printf("cells %d^3  unknowns %lld  snapshots %d  centre %.12e  iterations %lld  converged %d  rel_residual %.3e  solve_ms %.3f\n", n,
           (long long)st.n_unknowns, n_snap, values[(size_t)(n_snap - 1) * (size_t)N + (size_t)c], (long long)st.cg_iterations, st.converged,
           st.final_rel_residual, st.solve_ms);
    free(values);
    free(times);
    pdegpu_destroy(ctx);
    return 0;
}
