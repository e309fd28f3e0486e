#define _DEFAULT_SOURCE
/* A C host of the slab loop with the communication issued by the library (include/pdegpu.h, csrc/comm.hip): the CG of the steady solve of
 * fenics_mcp_server.py:657-661 on this rank's slab of node planes, halo exchange + ONE 24-byte all-reduce per iteration on the library's own RCCL
 * communicator, a chunk of iterations per C call, one host synchronisation per chunk.  RANK / WORLD_SIZE / PDEGPU_COMM_ID_FILE come from the
 * launcher (one process per GPU; rank 0 writes the 128-byte id to the file, the others wait for it — any bootstrap works: MPI_Bcast, a socket);
 * without them: one rank.
 *   gcc -std=c99 -I include examples/slab_loop.c -L pde-solver_amd/lib -lpdegpu -Wl,-rpath,$PWD/pde-solver_amd/lib -o slab_loop && ./slab_loop 96 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "pdegpu.h"

#define CK(call)                                                                                      \
    do {                                                                                              \
        const int rc__ = (call);                                                                      \
        if (rc__ != PDEGPU_OK) {                                                                      \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, ctx ? pdegpu_last_error(ctx) : "");  \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 96;
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), dev = env_int("LOCAL_RANK", 0);
    pdegpu_ctx* ctx = NULL;
    if (pdegpu_create(&ctx, dev, NULL) != PDEGPU_OK) {
        fprintf(stderr, "rank %d: no usable gfx950 device %d (there is no CPU fallback)\n", rank, dev);
        return 1;
    }
    /* the unique id: made on rank 0, carried by a file here */
    unsigned char id[128];
    const char* idfile = getenv("PDEGPU_COMM_ID_FILE");
    if (rank == 0) {
        CK(pdegpu_comm_unique_id(id));
        if (world > 1) {
            char tmp[4096];
            snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
            FILE* f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0 || rename(tmp, idfile) != 0) return 1;
        }
    } else {
        FILE* f = NULL;
        for (int tries = 0; tries < 600 && !(f = fopen(idfile, "rb")); ++tries) usleep(100000);
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) return 1;
        fclose(f);
    }
    pdegpu_comm* comm = NULL;
    CK(pdegpu_comm_create(ctx, rank, world, id, &comm));

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
    heat.source = 1.0;
    heat.bc_mode = PDEGPU_BC_UNIFORM;
    heat.bc_present[0] = 1;
    heat.fast_path = 1;
    int64_t k0 = 0, nk = 0;
    CK(pdegpu_slab_partition(n + 1, world, rank, &k0, &nk));
    pdegpu_session* s = NULL;
    CK(pdegpu_session_create(ctx, &grid, &heat, k0, nk, &s));
    CK(pdegpu_session_set_cg_variant(s, 1));                 /* single-reduction recurrences: ONE scalar message per iteration */
    CK(pdegpu_session_begin(s, 0.0));
    CK(pdegpu_session_residual(s));                          /* r = P (b - A x); this slab's r.r in scalars[3] */
    CK(pdegpu_session_allreduce(s, comm, 3, 1));
    double rr0 = 0.0, rr = 0.0;
    CK(pdegpu_session_rr(s, &rr0));
    const double target = 1e-16 * rr0;                       /* ||r|| <= 1e-8 ||r_0|| */
    int iterations = 0;
    rr = rr0;
    while (rr > target && iterations < 20000) {
        CK(pdegpu_session_iterate_slab(s, comm, 16));        /* enqueues 16 x {exchange, K1 around it, all-reduce, K2}; returns at once */
        iterations += 16;
        CK(pdegpu_session_sr_poll(s));                       /* the r.r of the last K2: this slab's share into scalars[6] ... */
        CK(pdegpu_session_allreduce(s, comm, 6, 1));         /* ... summed over the ranks ... */
        CK(pdegpu_session_rr(s, &rr));                       /* ... and read: the one host synchronisation of the chunk */
    }
    CK(pdegpu_session_flush(s));                             /* the pending x += alpha p */
    if (rank == 0) printf("cells %d^3  ranks %d  planes of rank 0: %lld  iterations %d  rr/rr0 %.3e\n", n, world, (long long)nk, iterations, rr / rr0);
    pdegpu_session_destroy(s);
    pdegpu_comm_destroy(comm);
    pdegpu_destroy(ctx);
    return 0;
}
