/* pdegpu.h — C-ABI of libpdegpu.so: the MI355X (gfx950) structured-grid FEM solve path that replaces
 * DOLFIN's assemble()+solve() behind the reference's box heat / static elasticity tools.
 *
 * Every entry point below is what a foreign-function binding of the reference's solve path binds.
 * The reference interface each one replaces is cited as file:line into the reference repository
 * (fenics_mcp_server.py unless stated otherwise).
 *
 * Conventions: plain C types only; returns 0 on success, a negative PDEGPU_E* code on failure (never
 * throws, never writes to stdout — stdout is the reference server's JSON-RPC channel, :222-225);
 * pdegpu_last_error() returns the message of the last failure on that context; the caller owns every
 * output buffer (host memory, C-contiguous fp64); the library owns all device memory inside the
 * context.  One context = one GPU = one process rank (one process per GPU).
 */
#ifndef PDEGPU_H
#define PDEGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDEGPU_OK            0
#define PDEGPU_EINVAL       (-1)   /* bad argument (shape, enum, null pointer) */
#define PDEGPU_EHIP         (-2)   /* a HIP runtime call failed */
#define PDEGPU_ENOCONV      (-3)   /* CG hit max_iter before reaching rtol */
#define PDEGPU_ENODEV       (-4)   /* no usable gfx950 device */
#define PDEGPU_EUNSUPPORTED (-5)   /* valid request outside the accelerated path */

/* element_mode: which constant macro-element matrix is integrated on the device.
 * P1SPLIT = the reference's discretisation, FunctionSpace(mesh,"P",1) on IntervalMesh/RectangleMesh/
 * BoxMesh (:229-230, :369-370, :533-535): each grid cell split into 1 interval / 2 triangles / 6 Kuhn
 * tetrahedra.  Q1 = multilinear element on the same nodes (BASELINE.json wording; differs by O(h^2)). */
#define PDEGPU_ELEM_P1SPLIT 0
#define PDEGPU_ELEM_Q1      1

/* initial condition (:276-292, :408-423, :672-687) */
#define PDEGPU_IC_CONSTANT 0
#define PDEGPU_IC_ZERO     1
#define PDEGPU_IC_NODAL    2       /* caller supplies nodal values u0[N] (used for the projected sine/cosine IC) */

/* Dirichlet set selection for heat */
#define PDEGPU_BC_UNIFORM     0    /* every boundary node = bc_value[0]            (:373-376, :626-628) */
#define PDEGPU_BC_DIRECTIONAL 1    /* x==0 -> bc_value[0], x==Lx -> bc_value[1], other boundary nodes
                                      with 0<x<Lx -> bc_value[2]; bc_present[i]==0 leaves that part
                                      natural (:233-241 for 1-D, :606-623 for the 3-D box).  DirichletBC marks
                                      facets, so with n[0] <= 2 cells bc_value[2] constrains nothing (every
                                      side facet touches x = 0 or x = Lx; see make_grid_dev) */

#define PDEGPU_QUANTITY_STRESS 0
#define PDEGPU_QUANTITY_STRAIN 1

typedef struct pdegpu_ctx pdegpu_ctx;

/* structured grid = the reference's IntervalMesh/RectangleMesh/BoxMesh arguments (:229, :369, :533,
 * :1516, :1648, :1803): dim, cells per axis, box lengths.  Unused axes: n=0, L=0. */
typedef struct pdegpu_grid {
    int32_t dim;            /* 1, 2 or 3 */
    int32_t n[3];           /* cells per axis (nx, ny, nz) */
    double  L[3];           /* Lx, Ly, Lz */
    int32_t element_mode;   /* PDEGPU_ELEM_* */
    int32_t reserved;
} pdegpu_grid;

/* arguments of _solve_heat_{1,2,3}d_raw (:204-219, :345-361, :475-500), box branch */
typedef struct pdegpu_heat {
    double  kappa;              /* diffusivity */
    double  dt;
    int32_t num_steps;
    int32_t steady;             /* 1: one steady solve, times=[0] (:259-269) */
    double  source;             /* constant f (0 when source_type == "none") */
    int32_t ic_type;            /* PDEGPU_IC_* */
    int32_t bc_mode;            /* PDEGPU_BC_* */
    double  ic_value;           /* T_initial */
    double  bc_value[3];
    int32_t bc_present[3];
    int32_t snapshot_stride;    /* store every stride-th state (state 0 and the last one always); 1 = all,
                                   as the reference does (:694-696, :711-713) */
    double  rtol;               /* CG: stop when ||r|| <= rtol * ||b - A g|| */
    int32_t max_iter;           /* per linear solve */
    int32_t fast_path;          /* 1 (default): use the fused z-marching kernels when the problem
                                   qualifies; 0: force the general class-table kernels */
} pdegpu_heat;

/* arguments of _solve_elasticity_{1,2,3}d_static (:1470-1477, :1593-1604, :1749-1762) */
typedef struct pdegpu_elasticity {
    double  E, nu;
    double  area;               /* 1-D bar only (:1474) */
    double  body[3];            /* body force; 1-D uses body[0] */
    int32_t plane_stress;       /* 2-D only (:1603, :1658-1665) */
    int32_t quantity;           /* PDEGPU_QUANTITY_* */
    double  rtol;
    int32_t max_iter;
    int32_t reserved;
} pdegpu_elasticity;

typedef struct pdegpu_stats {
    int64_t n_nodes;            /* grid nodes */
    int64_t n_unknowns;         /* free scalar unknowns of the main linear system */
    int64_t cg_iterations;      /* total over all linear solves of the call */
    int32_t linear_solves;
    int32_t converged;          /* 1 if every solve reached rtol */
    double  final_rel_residual; /* of the last solve */
    double  solve_ms;           /* device time of the whole call (hipEvent) */
    double  cg_ms;              /* device time inside CG iterations */
    double  kernel_ms[4];       /* summed hipEvent time: [0] operator apply (SpMV, fused or not),
                                   [1] vector update(s), [2] right-hand-side/other, [3] transfers.  Stepwise timing of a session whose
                                   solution vector is updated by K2 (tunable cg_xdefer): [1] the K2 launches without the x update,
                                   [2] those that carry it */
    int64_t kernel_launches[4];
    double  alg_bytes[4];       /* algorithmic HBM bytes of the launches counted in kernel_ms */
    int32_t path;               /* 0 general class-table kernels, 1 fused z-marching kernels */
    int32_t reserved;
} pdegpu_stats;

/* ---- context ---------------------------------------------------------------------------------- */
/* One context per GPU.  device_id is the HIP ordinal; stream is a hipStream_t passed as void* (NULL =
 * the library creates its own non-blocking stream). */
int         pdegpu_create(pdegpu_ctx** out, int device_id, void* stream);
void        pdegpu_destroy(pdegpu_ctx* ctx);
const char* pdegpu_last_error(const pdegpu_ctx* ctx);
int         pdegpu_device_name(const pdegpu_ctx* ctx, char* buf, size_t buflen);
const char* pdegpu_version(void);
/* Performance tunables of a context: pdegpu_set_tunable(ctx, key, value), or PDEGPU_<KEY> in the environment at pdegpu_create.
 * Results do not depend on them beyond the solver tolerance; every non-default value is a parity-tested alternative or a tuning aid.
 * Measured negatives are not kept as switches (round 3 removed hipGraph replay, in-launch dot completion, residency caps,
 * non-temporal input loads, the 48-node / two-nodes-per-lane elasticity tiles and the allocation modes: DESIGN.md §4b, profiles/).
 *
 * key                | default | meaning
 * -------------------+---------+----------------------------------------------------------------------------------------------
 * heat, z-marching kernels (3-D; 2-D through the flat view)
 * fast_ry            | 0       | rows per thread (1|2|4|8); 0 = 4 for the bare apply / residual pass and the K1 that leaves x alone (cg_xdefer),
 *                    |         | 2 for the other CG kernels
 * fast_nt            | -1      | non-temporal hints: bit 0 stores, bit 1 loads of x; -1 = both where a slab vector exceeds 64 MiB
 * fast_ex            | 1       | CG kernels hand tile-edge x-neighbours between the workgroup's waves through LDS
 * fast_pf            | -1      | deeper prefetch (plane k+2 requested before plane k is computed): -1 = bare apply / residual pass,
 *                    |         | 1 = every one-vector mode, 0 = off
 * fast_linear        | 1       | a workgroup takes four consecutive tiles of the row-major tile order (0 = wxt x wyt blocks)
 * fast_wg_per_cu     | 0       | workgroups per CU the z-chunking aims for; 0 = 2 on HBM-resident grids, occupancy API otherwise
 * fast_wg_per_cu_k2  | 0       | the same for the stencil K2 alone (0 = follow fast_wg_per_cu)
 * fast_zlen          | 0       | forced planes per z-chunk (0 = fitted chunk policy)
 * fast_minlen        | 1       | shortest z-chunk on grids too small to fill the device
 * fast_small_mult    | 1       | resident rounds of workgroups such grids may launch
 * fast_cfg_plain|res|k1|k2 | -1 | per-mode word ry + 16 wg_per_cu + 256 ex + 512 pf; -1 = the global tunables
 * cg_variant         | 1       | 1 = K2 applies the marching stencil to the stored p again (64 B/DoF; 56 with cg_xdefer): the 7-point operator, and
 *                    |         | the 15-point operator of transient runs on pure marching sessions with cg_xdefer; 2 = the 15-point operator
 *                    |         | everywhere; 0 = K1 stores q, flat K2 (72 B/DoF)
 * cg_xdefer          | 1       | marching sessions in the 64 B/DoF form: the solution vector is updated by K2 (which holds p_n, r_n and alpha_n) instead
 *                    |         | of the next K1, and only by every other K2 — the one in between records its alpha, the next forms both updates
 *                    |         | with p_(n-1) = (p_n - r_n) / beta_n.  K1 24 B/DoF, K2 24 | 40 B/DoF: 56 instead of 64 B/DoF per iteration
 * lazy_dot           | 1       | single-slab loops: the next kernel's prologue sums the previous launch's partials (no reduce launches)
 * place_tune         | 1       | HBM-resident marching sessions time K1 once with each of their six interchangeable vectors as the
 *                    |         | solution vector and keep the fastest (~18 launches at creation; profiles/r03_placement_*.log)
 * grid_mult          | 8       | flat kernels: workgroups per CU
 * k2_grid_mult       | 2       | flat K2: workgroups per CU
 * extrapolate        | 1       | transient solves start CG from 2 u^n - u^(n-1)
 * heat, other kernel forms
 * tile3d_max_nodes   | 3000000 | 3-D grids up to this many nodes run K1 / K2 as one-round tiles (0 = always march)
 * tile3d_plain       | 1       | ... and the bare masked apply too
 * plane2d            | 1       | 2-D grids: K1 / K2 as one-round tiles (0 = flat view of the marching kernels)
 * plane2d_ry         | 4       | rows per wavefront tile of that form (2|4|8)
 * persist2d          | 1       | 2-D grids up to 1024 cells wide: the whole CG loop in ONE cooperative launch, state in registers
 *                    |         | (2 / 3 = other grid-barrier forms)
 * persist2d_launch   | 0       | 1 = that kernel is launched as an ordinary kernel on the same occupancy-sized grid (its own barriers have a spin
 *                    |         | limit and a fail flag): for runs under rocprofv3, whose exit handler faults inside libhsa-runtime64 after a
 *                    |         | cooperative launch (profiles/r03_rocprofv3_exit_abort.log)
 * persist2d_sr       | 1       | that launch uses the single-reduction recurrences (one grid barrier + one neighbour hand-off)
 * persist2d_cheb     | 8       | ... and a Chebyshev polynomial preconditioner of that many terms (bands of <= 4 rows; one grid barrier per 8 stencil
 *                    |         | applications instead of one per application; the iteration count reported is that of the OUTER iterations); 0 = plain CG
 * persist2d_cheb_ratio | 100   | ... built on [lambda_max / ratio, lambda_max], lambda_max = Gershgorin's bound of the stencil row
 * snapshot_async     | 1       | transient solves: the snapshot of a step leaves the device beside the next step (helper thread, second stream)
 * hybrid_bc          | 1       | natural faces: marching kernel on the interior + class-table kernel on the boundary nodes
 * hybrid_fx          | 1       | ... with the natural x-face rows inside the marching kernel
 * hybrid_k2          | 1       | ... steady sessions on one slab use the 64 B/DoF form
 * direct_1d          | 1       | 1-D grids: Thomas recurrence inside one launch (0 = CG, diagnostic)
 * elasticity
 * wfast              | 1       | weighted 3-D operators on a box whose boundary nodes are all constrained, compact rows: z-marching kernels with the weights
 *                    |         | in registers (weight independent of the marching coordinate) or SGPRs (independent of x), 72 B/DoF per PCG
 *                    |         | iteration (csrc/wmarch.hip); 0 = class-table kernels reading the rows (88 B/DoF, 45 loads per node)
 * wfast_xdefer       | -1      | (-1 = where one vector exceeds 32 MiB; smaller grids are cache-resident and keep the flat K2) those sessions' K2 applies the weighted stencil to the stored p again (z -= alpha D^-1 A p: K1 stores no q) and updates x for
 *                    |         | two iterations at once, every other iteration (as cg_xdefer does for the plain operators): 56 instead of 72 B/DoF; 0 = flat K2
 * wfast_ry           | 2       | rows per wavefront column of those kernels where the weights are wave-uniform (1|2)
 * wfast_wg_per_cu    | 0       | workgroups per CU their z-chunking aims for (0 = occupancy API)
 * wfast_one_round    | 1       | HBM-resident grids: the z-chunking with the most workgroups that are all resident at once, where one fills 1.6 per CU
 * wfast_k2_mult      | 4       | workgroups per CU of their K2 (a wavefront per grid row)
 * wfast_ex           | -1      | those kernels hand tile-edge x-neighbours between the workgroup's waves through LDS (0 = 2-lane global loads;
 *                    |         | -1 = per-lane weights always, wave-uniform weights on cache-resident grids: where it measured faster)
 * wfast_nt           | -1      | non-temporal stores of q / p / x (and loads of x) in those marching kernels: 1|0, -1 = where one vector exceeds 64 MiB
 * wfast_k2_nt        | -1      | non-temporal loads / stores in that K2: 1|0, -1 = where one vector exceeds 64 MiB
 * ell_compact        | 1       | weighted (curvilinear / composite) 3-D operators whose weight does not depend on one coordinate store the coefficient
 *                    |         | rows of three representative planes / columns instead of one row per node (0 = one row per node)
 * el_fast            | 1       | 3-D: z-marching block stencil (0 = thread-per-node class-table kernel)
 * el_fast_ry         | 2       | rows per wavefront column (1|2)
 * el_fast_yx         | 1       | y-halo rows handed between the workgroup's waves through LDS
 * el_fast_xface      | 1       | x-face nodes inside the marching kernel
 * el_fast_yzface     | 3       | bit 0: y-face rows, bit 1: the planes 0 / NZ-1 as well (the class-table pass keeps the box edges)
 * el_fast_nt         | -1      | non-temporal stores of the output: 1|0, -1 = where the vectors are HBM-resident
 * el_fast_waves      | 0       | wavefront target of the marching part; 0 = 4096, 8192 on HBM-resident grids
 * el_xdefer          | 1       | single-reduction PCG of a whole solve: x += alpha p is left out of every other update pass and made up for by the next
 *                    |         | one from what that pass reads anyway (6 instead of 7 vector passes on average); 0 = in every pass
 * el_bnd_gran        | 64      | boundary nodes a boundary workgroup of the marching apply takes per pass where only edges and corners are left to them (64|256)
 * el_fast_fit        | 1       | a launch between one and two resident rounds of workgroups is cut into as many z-chunks as fill ONE round, of two
 *                    |         | lengths (config 5: 19 chunks of 4 / 3 planes = 766 workgroups on 768 slots instead of 21 x 3 = 846; 0 = equal chunks)
 * el_cg_variant      | 1       | PCG recurrences: 1 = single-reduction form (the r- and p-updates fused into ONE pass of 7 vector reads /
 *                    |         | writes, one scalar reduction per iteration; same iteration counts), 0 = classic (p-update, apply, K2)
 * el_split_min_mb    | 32      | slab loop: k1 is split around the halo exchange where one vector of the slab holds at least this many
 *                    |         | MiB (two more launches, ~9 us: config 5's slabs ran 26 -> 34 us per iteration at N = 8 with the split)
 * el_bnd_last        | 0       | boundary workgroups take the last block indices
 * el_grid_mult       | 4       | flat / class-table kernels: workgroups per CU
 */
int         pdegpu_set_tunable(pdegpu_ctx* ctx, const char* key, int value);
/* Host-only diagnostic (needs no GPU): launch geometry of the 7-point z-marching kernels for a single-slab 3-D grid of nx x ny x nz NODES on
 * a device with num_cu compute units and resident_wg_per_cu resident workgroups of that kernel per CU (gfx950 build: PLAIN 2, K1 3, K2 4),
 * default tunables.  mode: 0 bare apply, 1 residual, 2 CG K1, 3 CG K2.  out4 = {rows per thread, workgroups per z-chunk, z-chunks, planes per
 * chunk}.  The chunk policy is fitted to measurements; tests pin the geometry of BASELINE's grids through this call. */
int         pdegpu_marching_geometry(int num_cu, int nx_nodes, int ny_nodes, int nz_nodes, int mode, int resident_wg_per_cu, int32_t* out4);

/* ---- heat: replaces mesh+FunctionSpace+DirichletBC+solve(a==L) of _solve_heat_{1,2,3}d_raw
 *      (:229-318, :369-447, :533-716).  Node order of every array: x fastest, then y, then z
 *      (N = (nx+1)(ny+1)(nz+1)); the reference exposes its dof order only through coordinates.
 *      values_out [n_snap][N], times_out [n_snap], n_snap = pdegpu_heat_num_snapshots(...);
 *      u0 = nodal initial values [N] when ic_type == PDEGPU_IC_NODAL, else NULL.
 *      Transient solves (snapshot_async = 1, the default): while step n + 1 is solved, a helper thread of the library copies the snapshot of step n
 *      into values_out on a stream of its own; every snapshot is in place when the call returns (also when it returns an error), none of the
 *      caller's memory is touched afterwards, and the caller must not read values_out before that.  The fields are those of in-line copies. */
int pdegpu_heat_num_snapshots(const pdegpu_heat* h);
int pdegpu_heat_solve(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_heat* heat,
                      const double* u0, double* values_out, double* times_out, pdegpu_stats* stats);

/* ---- curvilinear heat tools (_solve_heat_1d_cylindrical_raw :769, _1d_spherical :925, _2d_cylindrical :1063,
 *      _2d_spherical :1195, _3d_spherical :1333): the same P1 discretisation on IntervalMesh / RectangleMesh / BoxMesh
 *      in COORDINATE space [origin, origin + L] with one scalar weight w multiplying the mass and the stiffness
 *      form; w enters as the P1 / P2 interpolant of an Expression of degree 1 / 2 and is integrated exactly.
 *      Constant initial condition only (:872-875). */
#define PDEGPU_WEIGHT_ONE   0      /* w = 1             Constant(1.0): the plain box with a composite core (:537-572, :647) */
#define PDEGPU_WEIGHT_R     1      /* w = x0            Expression("x[0]", degree=1)               (:802, :1100) */
#define PDEGPU_WEIGHT_R2    2      /* w = x0^2          Expression("x[0] * x[0]", degree=2)        (:965)        */
#define PDEGPU_WEIGHT_R2SIN 3      /* w = x0^2 sin(x1)  Expression("x[0] * x[0] * sin(x[1])", 2)   (:1230, :1371) */
#define PDEGPU_WEIGHT_RCYL  4      /* w = sqrt(x1^2+x2^2), degree 2: the BoxMesh fallback of the 3-D cylinder (:523-529, :645) */
typedef struct pdegpu_weight {
    int32_t kind;               /* PDEGPU_WEIGHT_* */
    int32_t reserved;
    double  origin[3];          /* lower corner of the mesh in coordinate space (r_inner, 0, 0) */
    double  core_radius;        /* composite core (:537-572): a simplex whose vertices and midpoint all lie within   */
    double  core_ratio;         /* core_radius of the x-axis has diffusivity core_ratio * kappa; core_radius <= 0: none */
} pdegpu_weight;
int pdegpu_heat_solve_weighted(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_weight* weight,
                               const pdegpu_heat* heat, double* values_out, double* times_out, pdegpu_stats* stats);
/* y = (cM M_w + cK K_w) x — the weighted forms `w*u*v*dx`, `w*dot(grad(u), grad(v))*dx` of :805-806, :968-969, :1103-1104, :1233-1234,
 * :1374-1375 and of the box / cylinder branch :702-703 with r_expr and the DG0 kappa — applied matrix-free (operator tests, timing).
 * marching = 1: the z-marching kernels (csrc/wmarch.hip) on the box with every boundary node constrained: interior rows only, the rest of y is 0;
 * marching = 2: the same with only the faces x = 0 and x = Lx constrained (T_left, T_right of :582-604; weights that do not depend on x): the rows of
 * the nodes on the four natural faces too; marching = 3: only x = 0 constrained (T_left alone); marching = 4: x = 0 and the four side faces
 * constrained, x = Lx natural (T_left + T_side).  PDEGPU_EUNSUPPORTED where they do not apply (2-D / 1-D grids, weights depending on all
 * coordinates, fewer than 2 cells on an axis, natural faces with a weight that depends on x).
 * marching = 0: every row, class-table kernels reading one coefficient row per node. */
int pdegpu_heat_apply_weighted(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_weight* weight, double cM, double cK,
                               const double* x, double* y, int marching);
/* the same with a nodal initial field u0 [N] (heat->ic_type == PDEGPU_IC_NODAL): the projected sine / cosine initial conditions
 * of solve_heat_3D on its cylinder (BoxMesh fallback) and composite-core branches (:672-687) */
int pdegpu_heat_solve_weighted_ic(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_weight* weight,
                                  const pdegpu_heat* heat, const double* u0, double* values_out, double* times_out,
                                  pdegpu_stats* stats);
/* host twin of the per-node weighted assembly (CPU tests): rowsK / rowsM [N][15] (15-offset order
 * {0, -x, +x, -y, +y, -z, +z, -(xy), +(xy), -(xz), +(xz), -(yz), +(yz), -(xyz), +(xyz)}), load [N] */
int pdegpu_weighted_rows_host(const pdegpu_grid* grid, const pdegpu_weight* weight, double* rowsK, double* rowsM,
                              double* load);

/* ---- project(f, V) with a consistent mass matrix (:284, :415, :679, :1542, :1714, :1862):
 *      solves M s = rhs on the device, rhs[N] is the assembled load vector int f phi_i. */
int pdegpu_mass_solve(pdegpu_ctx* ctx, const pdegpu_grid* grid, const double* rhs, double* s_out,
                      double rtol, int max_iter, pdegpu_stats* stats);

/* ---- operator access for tests and the roofline harness -------------------------------------- */
/* y = (cM*M + cK*K) x on the full grid, no Dirichlet elimination; x, y: host [N].  Also returns the
 * load vector m_i = int phi_i in load_out when non-NULL. */
int pdegpu_heat_apply(pdegpu_ctx* ctx, const pdegpu_grid* grid, double cM, double cK,
                      const double* x, double* y, double* load_out, int fast_path);

/* class tables integrated on the device: tabK/tabM [27 classes][27 offsets], load [27] */
int pdegpu_heat_tables(pdegpu_ctx* ctx, const pdegpu_grid* grid, double* tabK, double* tabM, double* load);
/* host twin of the same integration (same source, compiled for the CPU): lets the CPU-only test-suite
 * check the macro-element math against the oracle's assembled matrices without a GPU */
int pdegpu_heat_tables_host(const pdegpu_grid* grid, double* tabK, double* tabM, double* load);

/* ---- elasticity: replaces _solve_elasticity_{1,2,3}d_static's VectorFunctionSpace assembly, clamp at
 *      x = 0, solve, per-cell von Mises and project() (:1516-1564, :1648-1718, :1803-1866).
 *      field_out [N] = projected von Mises stress|strain (1-D: axial stress|strain);
 *      disp_out  [N][dim] nullable. */
int pdegpu_elasticity_solve(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el,
                            double* field_out, double* disp_out, pdegpu_stats* stats);
/* y = K u on the full grid (parity tests).  use_mfma: 0 = vector block stencil, every row of the operator; 1 = element-by-element MFMA (3-D);
 * 2 = the launch the PCG loop issues: as 0, but the rows of the clamped face x = 0 give 0 */
int pdegpu_elasticity_apply(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el,
                            const double* u, double* y, int use_mfma);
/* average device time (ms) of `reps` applications of the elasticity operator: use_mfma = 0 vector block stencil,
 * 1 element-by-element v_mfma_f64_16x16x4_f64 (3-D) */
int pdegpu_elasticity_time_apply(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el, int use_mfma,
                                 int reps, double* avg_ms_out);
/* host twin of the device fold of the d x d block class table, tab [27 classes][15 offsets][3x3] (CPU tests) */
int pdegpu_elasticity_tables_host(const pdegpu_grid* grid, double lam, double mu, double* tab);

/* ---- slab-resident elasticity PCG session (3-D): same protocol as the heat session below.  Vectors: 0 x, 1 r,
 *      2 p (current), 3 q; each has 3 component arrays (z = D^-1 r is never stored: the p-update forms it).  Before every k1 the
 *      host exchanges the first / last owned plane of r (3 component planes, contiguous each: no pack kernel) with the slab neighbours —
 *      p on the ghost planes is maintained by k1 itself (same formula, same global scalars as the neighbour); after k1 it all-reduces scalars[0]
 *      (p.q), after k2 scalars[1..2] (r.z, r.r) — in place, 8 / 16 bytes.  begin() needs no exchange (x = 0). */
typedef struct pdegpu_el_session pdegpu_el_session;
int     pdegpu_el_session_create(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el, int64_t k0, int64_t nk,
                                 pdegpu_el_session** out);
void    pdegpu_el_session_destroy(pdegpu_el_session* s);
int     pdegpu_el_session_begin(pdegpu_el_session* s);      /* b = body m; x = 0; r = P b; z; local r.z, r.r */
int     pdegpu_el_session_flush(pdegpu_el_session* s);      /* pending x += a p */
int     pdegpu_el_session_scalars_host(pdegpu_el_session* s, double* out4);   /* synchronises */
int     pdegpu_el_session_get_u(pdegpu_el_session* s, double* u_out /* host [nk * plane nodes][3] */);
/* The PCG recurrences of a session (default: the context's el_cg_variant tunable).  0 = classic: the host all-reduces scalars[0] (p.Ap) after k1 and
 * scalars[1..2] (r.z, r.r) after k2.  1 = single-reduction form: the residual update r -= a q moves INTO k1, fused with the p-update (7 vector
 * passes instead of 8, one launch less); k2 and flush do nothing; after k1 / k1_part(1) the block scalars[0..3] = {p.Ap, r.z, r.r, Ap.D^-1 Ap} is
 * the ONE 32-byte message of the iteration; beta comes from the predicted r.z = a^2 (Ap.D^-1 Ap) - r.z (exact arithmetic: the same iterates; the
 * true r.z of every residual forms the next alpha).  On a slab with neighbours r changes inside k1, so its halo exchange moves there too:
 * k1_pre (update pass of the owned planes) -> exchange the first / last owned plane of r -> k1_part(0) -> k1_part(1); k1_pre is a no-op in the
 * classic form, so a driver may always call it.  x always matches the residual scalars[2] reports.  Choose before begin(). */
int     pdegpu_el_session_set_cg_variant(pdegpu_el_session* s, int variant);
int     pdegpu_el_session_cg_variant(const pdegpu_el_session* s);
/* von Mises + project() (:1840-1866) on one GPU from a displacement field gathered on the host, disp [N][dim] */
int     pdegpu_elasticity_postprocess(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_elasticity* el, const double* disp,
                                      double* field_out);

/* ---- slab-resident CG session: the unit the benchmark and the multi-GPU driver run --------------
 * (whole solves and chunks of iterations here; the single steps of an iteration — k1 / k2 and their parts, plane pointers, the exchange
 *  building blocks — are the driver / rehearsal API of pdegpu_stepwise.h)
 * A session owns the device vectors of ONE slab of a 3-D (or 2-D) heat system
 *      A = cM*M + cK*K  with the Dirichlet set of `heat`,
 * for global node planes [k0, k0+nk) of the slowest axis (z in 3-D, y in 2-D), plus one ghost plane on
 * each side.  Single GPU: k0 = 0, nk = n_slowest+1.  The multi-GPU host loop (one process per GPU)
 * interleaves these calls with its RCCL halo exchange / all-reduce; every call only enqueues work on
 * the context stream (no host synchronisation) unless stated.  */
typedef struct pdegpu_session pdegpu_session;

int  pdegpu_session_create(pdegpu_ctx* ctx, const pdegpu_grid* grid, const pdegpu_heat* heat,
                           int64_t k0, int64_t nk, pdegpu_session** out);
void pdegpu_session_destroy(pdegpu_session* s);
/* set x to the Dirichlet values on constrained nodes and `fill` elsewhere; b = M*(fill_rhs) i.e. the
 * right-hand side of a constant load (steady: b = source*m) is prepared by session_begin. */
int  pdegpu_session_begin(pdegpu_session* s, double x_fill);
int  pdegpu_session_residual(pdegpu_session* s);            /* r = P(b - A x), rr_local */
/* The CG recurrences of the stepwise protocol.  variant 0 (default): classic CG — the host all-reduces pq_global after k1 and rr_global after
 * k2 (two 8-byte messages per iteration).  variant 1: single-reduction form (one message per iteration): k1 additionally sums
 * Ap.Ap, k2 leaves its r.r in per-workgroup partials, and k1 / k1_part(1) end with ONE reduce launch that fills scalars[2..4] =
 * {p.Ap, r.r (of the residual k1 started from), Ap.Ap}; the host all-reduces these 24 bytes in place after k1 and nothing after k2.  alpha =
 * r.r / p.Ap as before; beta is formed from the predicted |r_{n+1}|^2 = alpha^2 Ap.Ap - r.r (exact arithmetic: the same iterates; the true
 * r.r of every residual still enters the next alpha).  Convergence polls: sr_poll sums the last k2's partials into scalars[6] (this slab's
 * share: all-reduce that slot, 8 bytes), then pdegpu_session_rr reports it; on a slab without neighbours rr does this by itself.  Choose the
 * variant before begin() / residual(); pdegpu_session_iterate (the native single-slab loop) runs the classic form only. */
int  pdegpu_session_set_cg_variant(pdegpu_session* s, int variant);
int  pdegpu_session_cg_variant(const pdegpu_session* s);
/* kernel timing of the stepwise protocol: set_timing(s, 1) starts a fresh measurement (hipEvent pairs around every k1 / k1_part
 * / k2 launch on the context stream), get_stats synchronises and ADDS kernel_ms[0] (k1), kernel_ms[1] (k2), launch counts and
 * algorithmic bytes since then to *stats.  set_timing(s, k), k > 1: only every k-th iteration is bracketed (an event record costs
 * stream time of its own — at the 5000 iterations per second of a thin slab the pairs of EVERY launch add ~5 % to the loop). */
int  pdegpu_session_set_timing(pdegpu_session* s, int on);
int  pdegpu_session_get_stats(pdegpu_session* s, pdegpu_stats* stats);
/* transient steps on a slab — the time loop of _solve_heat_3d_raw (:702-716) distributed over slabs.  Per step the host
 * exchanges the ghost planes of x (vector 0) and calls step_rhs (b = M (x + dt f)), lift_norm (r = P (b - A g) with g = the
 * Dirichlet values; its r.r, the reference norm of the step, lands in the rr slots to be all-reduced and read), reset_cg,
 * residual, then iterates k1 / k2 as for the steady system and flushes. */
int  pdegpu_session_step_rhs(pdegpu_session* s);
int  pdegpu_session_lift_norm(pdegpu_session* s);
int  pdegpu_session_reset_cg(pdegpu_session* s);
int  pdegpu_session_flush(pdegpu_session* s);               /* pending x += alpha*p: call before x is read.  Between iterations x lags the residual by
                                                                * one update (deferred into the next K1) or — cg_xdefer, where K2 updates x for two
                                                                * iterations at once — by none or one; after flush it is the iterate r belongs to */
/* run `iters` full CG iterations on a single slab without the host in the loop (k1,k2 back to back).
 * If time_kernels != 0, brackets every launch with hipEvents and accumulates stats->kernel_ms. */
int  pdegpu_session_iterate(pdegpu_session* s, int iters, int time_kernels, pdegpu_stats* stats);
int  pdegpu_session_rr(pdegpu_session* s, double* rr_out);  /* synchronises; global r.r */
/* roofline harness: average device time (ms, hipEvents on the context stream) of `reps` launches of the bare
 * operator application b = A r (the matrix-free SpMV alone: read one vector, write one, 16 B/DoF).
 * Overwrites b; call after the solve. */
int  pdegpu_session_time_apply(pdegpu_session* s, int reps, double* avg_ms_out);
/* in-run memory calibration over the session's own vectors (padded slab, nk*PP doubles each):
 * which = 0 copy (1R+1W), 1 read-only (1R), 2 write-only (1W), 3 triad (2R+1W), 4 in-place update (2R+1W). */
int  pdegpu_session_membench(pdegpu_session* s, int which, int reps, double* avg_ms_out);
int  pdegpu_session_get_x(pdegpu_session* s, double* x_out /* host [nk][plane nodes] */);
int  pdegpu_session_sync(pdegpu_session* s);

/* ---- the slab loop's communication issued by the library itself (csrc/comm.hip) ------------------------------------------------------------
 * The reference has no multi-process code (its only MPI-adjacent line is fenics_mcp_server.py:141); a host that distributes the solve of
 * :702-716 over slabs may drive the exchange / all-reduce itself on the pointers above (pdegpu/dist.py does, through torch.distributed), or hand
 * the library an RCCL communicator and let it run whole iterations: no host-language call path in the loop (tools/dist_overhead.py: 14 us of
 * C-ABI calls per iteration against 124 us with torch.distributed's all-reduce and point-to-point batch).  RCCL is resolved at run time from the copy
 * the process already holds (else librccl.so.1 of the loader's path): libpdegpu.so does not link it.
 *   unique_id: 128 bytes (ncclUniqueId) made on one rank and carried to the others by the host's own bootstrap (MPI_Bcast, torch.distributed
 *   broadcast of a byte tensor, a file); create: collective over all `world` ranks, one process per GPU, the context's device.
 * Stream protocol of an exchange: exchange_begin makes the communicator's own stream wait for everything enqueued so far on the context stream,
 * posts ONE group of sends / receives there (first owned plane -> rank-1's upper ghost, last owned plane -> rank+1's lower ghost; heat: r, plus p
 * where k1 does not maintain it on the ghosts, or x with x_planes != 0 for a transient step's right-hand side; elasticity: the three component planes
 * of r) and returns; exchange_end makes LATER launches on the context stream wait for their arrival — whatever is enqueued in between overlaps the
 * transfer.  allreduce sums scalars[first .. first+count) over the ranks in place on the context stream.  iterate_slab = `iters` times
 * {exchange_begin, k1_part(0), exchange_end, k1_part(1), all-reduce, k2 (, all-reduce)} in the session's CG variant — heat single-reduction form:
 * 4 launches + ONE 24-byte collective per iteration; elasticity: k1_pre first, ONE 32-byte collective.  Nothing here synchronises the host.
 * exchange_post is the building block (what: bit 0 the sends, bit 1 the receives; lo_rank / hi_rank < 0 = rank -/+ 1) for hosts that group several
 * sessions' messages themselves, bracketed by pdegpu_comm_group_begin / _end. */
typedef struct pdegpu_comm pdegpu_comm;
int  pdegpu_comm_unique_id(void* id128_out);
int  pdegpu_comm_create(pdegpu_ctx* ctx, int rank, int world, const void* id128, pdegpu_comm** out);
void pdegpu_comm_destroy(pdegpu_comm* c);
int  pdegpu_comm_rank(const pdegpu_comm* c);
int  pdegpu_comm_world(const pdegpu_comm* c);
int  pdegpu_session_iterate_slab(pdegpu_session* s, pdegpu_comm* c, int iters);
/* convergence polls of a host that runs chunks of iterate_slab: sr_poll (single-reduction form) sums this slab's share of the r.r the last k2 left
 * into scalars[6]; allreduce sums scalars[first .. first+count) over the ranks in place on the context stream; pdegpu_session_rr then reads it */
int  pdegpu_session_sr_poll(pdegpu_session* s);
int  pdegpu_session_allreduce(pdegpu_session* s, pdegpu_comm* c, int first, int count);
int  pdegpu_el_session_allreduce(pdegpu_el_session* s, pdegpu_comm* c, int first, int count);
int  pdegpu_el_session_iterate_slab(pdegpu_el_session* s, pdegpu_comm* c, int iters);

/* ---- CPU-visible helpers that need no GPU (used by `-m "not gpu"` tests) ---------------------- */
int64_t pdegpu_grid_num_nodes(const pdegpu_grid* grid);
/* the Dirichlet set (mask_out [N] 0/1, value_out [N] nullable) the kernels use for `grid` with the boundary arguments of `heat`
 * (bc_mode, bc_present, bc_value) — the DirichletBC predicates of :233-241, :373-376, :606-628 with the facet rule above */
int     pdegpu_dirichlet_set_host(const pdegpu_grid* grid, const pdegpu_heat* heat, uint8_t* mask_out, double* value_out);
int     pdegpu_slab_partition(int64_t n_planes, int n_ranks, int rank, int64_t* k0, int64_t* nk);

#ifdef __cplusplus
}
#endif
#endif /* PDEGPU_H */
