/*
 * avs.h -- C ABI of the MI355X-native adaptive variational viscosity hot path.
 *
 * Drop-in boundary.  The reference is a Houdini GAS micro-solver whose only
 * entry point is HDK_AdaptiveViscosity::solveGasSubclass (HDK_AdaptiveViscosity.cpp:126,
 * "cpp:" below).  The proprietary HDK cannot be compiled against here, so the seam sits
 * INSIDE that function: everything between cpp:418 ("Precompute stress gradients") and
 * cpp:653 (end of "Solve Linear System") is replaced by the calls declared in this header.
 * INTEGRATION.md shows the shim a maintainer adds on the Houdini side.
 *
 * Conventions
 *   - plain C types only; the library never throws; every call returns avs_status and
 *     avs_last_error() gives a thread-local message for the last failure.
 *   - all grids are dense, x fastest ("i + rx*(j + ry*k)"), with the sample resolutions of
 *     SIM_RawField::init (HDK_Utilities.h:13-16; cpp:314-321, 370-392):
 *       centre (nx,ny,nz)>>level; face a: +1 on a; edge a: +1 on the two other axes.
 *     HDK's exint (int64) index voxels are int32 here, fp32 label voxels are int8.
 *   - pointers are host or device memory as told by avs_memspace; the library copies on
 *     every set_* (the caller keeps ownership, as Houdini does: cpp:29-34, no state survives).
 *   - one avs_ctx is used by one host thread at a time; different contexts are independent.
 *   - base resolution must be a power of two per axis (the reference pads to powers of two,
 *     HDK_OctreeGrid.cpp:18-24; power-of-two input means no padding).
 */
#ifndef AVS_H
#define AVS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVS_MAX_LEVELS 8
#define AVS_EDGE_STENCIL_CAP 32   /* getEdgeStressFaces emits <= 4 slots x 8 faces, cpp:1789-1907 */
#define AVS_CENTER_STENCIL_CAP 8  /* getCenterStressFaces emits <= 2 x 4 faces, cpp:1925-1962 */
#define AVS_EDGE_BOUNDARY_CAP 4
#define AVS_CENTER_BOUNDARY_CAP 2

typedef enum {
    AVS_OK = 0,
    AVS_EINVAL = 1,    /* bad argument / inconsistent inputs (reference: addError + return false, cpp:152-229) */
    AVS_ENOMEM = 2,
    AVS_EHIP = 3,      /* a HIP runtime call failed */
    AVS_ERCCL = 4,     /* an RCCL call failed */
    AVS_EINTERNAL = 5, /* a reference assert() would have fired (e.g. foundSelf, cpp:2436) */
    AVS_ESTATE = 6     /* call sequence violated (e.g. solve before assemble) */
} avs_status;

typedef enum { AVS_MEM_HOST = 0, AVS_MEM_DEVICE = 1 } avs_memspace;

/* HDK_OctreeGrid::OctreeCellLabel, HDK_OctreeGrid.h:33-39 */
typedef enum { AVS_INACTIVE = 0, AVS_ACTIVE = 1, AVS_UP = 2, AVS_DOWN = 3 } avs_cell_label;

/* index sentinels, HDK_Utilities.h:18-21; values >= 0 are DOF / stress ids */
#define AVS_UNASSIGNED (-1)
#define AVS_SOLIDBOUNDARY (-2)
#define AVS_OUTSIDE (-3)

/* index pyramids created at cpp:337-393 */
typedef enum {
    AVS_INDEX_VELOCITY = 0, /* octreeVelocityIndices[level][axis]  (face lattice) */
    AVS_INDEX_EDGE = 1,     /* edgeStressIndices[level][axis]      (edge lattice) */
    AVS_INDEX_CENTER = 2    /* centerStressIndices[level]          (centre lattice; axis ignored) */
} avs_index_kind;

/* fp32 level-0 fields read by the hot path */
typedef enum {
    AVS_FIELD_CENTER_WEIGHTS = 0, /* centerIntegrationWeights (cpp:239)            centre */
    AVS_FIELD_EDGE_WEIGHTS = 1,   /* edgeIntegrationWeights[axis] (cpp:240)        edge a */
    AVS_FIELD_FACE_WEIGHTS = 2,   /* "faceWeights" vector field (cpp:144)          face a */
    AVS_FIELD_VISCOSITY = 3,      /* "viscosity" (cpp:203)                         centre */
    AVS_FIELD_DENSITY = 4,        /* GAS_NAME_DENSITY (cpp:218)                    centre */
    AVS_FIELD_VELOCITY = 5,       /* GAS_NAME_VELOCITY component (cpp:139)         face a */
    AVS_FIELD_SOLID_VELOCITY = 6  /* GAS_NAME_COLLISIONVELOCITY component (cpp:142) face a */
} avs_field_kind;

typedef struct {
    int32_t nx, ny, nz;             /* level-0 resolution, powers of two */
    double dx;                      /* level-0 voxel size (getVoxelSize().maxComponent(), cpp:242) */
    double dt;                      /* timestep (cpp:130) */
    int32_t levels;                 /* octreeLabels.getOctreeLevels() (cpp:275), 1..AVS_MAX_LEVELS */
    int32_t use_enhanced_gradients; /* getUseEnhancedGradients() (cpp:438) */
    int32_t device;                 /* HIP device ordinal */
    void *stream;                   /* hipStream_t to enqueue on; NULL = the library creates one */
    /* Resolution of the simulation grid the level-0 SCALAR fields live on (weights, viscosity, density, velocities),
     * i.e. before HDK_OctreeGrid::init stretched the octree grid to powers of two (oct.cpp:13-24).  0 = nx, ny, nz.
     * Labels and index pyramids always use nx, ny, nz.  Samples of the padded lattice outside this grid belong to
     * INACTIVE cells (oct.cpp:375-379) and are never read by an active row; the library fills them (viscosity /
     * density by border replication, everything else with 0).  The pre-pass takes the same field_n* in its own descriptor;
     * avs_set_regular_index_field / avs_transfer_to_regular_grid exchange arrays on the simulation grid's face lattices. */
    int32_t field_nx, field_ny, field_nz;
    /* AVS_PRECISION_F64 (0, default): SolveType = fpreal64, the build the reference ships.  AVS_PRECISION_F32: the system the
     * reference builds with USESINGLEPRECISION (util.h:25-37: SolveType = fpreal32): every triplet narrowed to float where
     * Eigen::Triplet<SolveType> is constructed (cpp:2447, 2768), duplicates summed in float (setFromTriplets, cpp:613-614), the
     * right-hand side updated in float steps (cpp:2456, 2772), the initial guess narrowed at its store (cpp:2371).  Matrix, rhs and
     * x0 are then float VALUES in the same fp64 arrays.  Single-GPU solves can iterate on FLOAT vectors with float scalars, as Eigen's
     * float CG does (round 5; dot products: a thread's terms in float, everything across threads in double -- Eigen's vectorised
     * reduction order is not reproduced): AVS_OPTION_F32_VECTORS = 1 always, -1 (default) for systems the CU-resident loop does not
     * take; partitioned solves and the other cases iterate in fp64 on the float system (at least as accurate, same stopping rule).
     * The solution is a float vector either way (Eigen::VectorXf). */
    int32_t precision;
} avs_desc;
enum { AVS_PRECISION_F64 = 0, AVS_PRECISION_F32 = 1 };

typedef struct {
    int32_t iterations; /* solver.iterations() (cpp:629) */
    int32_t converged;  /* 1 if |r|^2 < tol^2 |b|^2 was reached; non-convergence is NOT an error (cpp:645-652) */
    double error;       /* solver.error() = sqrt(|r|^2/|b|^2) (cpp:630) */
    double rhs_norm2;
    int64_t n;          /* octreeVelocityDOFCount */
    int64_t nnz;
    double solve_ms;    /* HIP-event time of the PCG loop */
    double spmv_ms;     /* mean HIP-event time of one SpMV launch during this solve (0 if not sampled) */
    int32_t resident;   /* 1 = the iterations ran in the CU-resident loop (one cooperative launch: matrix words in the register files,
                           vector slices in LDS -- systems of up to ~1 M rows (~1.9 M with the rows that do not fit streamed from memory) in the packed
                           single-dictionary form, single-GPU solves and
                           partitioned solves whose rank has its GPU for itself; AVS_CG_RESIDENT=0 keeps the launch-per-phase loops) */
    int32_t cancelled;  /* 1 = avs_cancel ended the loop before convergence / max_iterations (converged is 0; x holds the last iterate) */
} avs_solve_info;

typedef struct {
    int64_t n_velocity, n_edge, n_center; /* DOF / stress counts */
    int64_t nnz;                          /* after duplicate merging */
    int64_t raw_triplets;                 /* before merging (= size of the reference's triplet list) */
    double stencil_ms, guess_ms, system_ms, csr_ms; /* HIP-event times of the phases */
} avs_assembly_info;

typedef struct avs_ctx avs_ctx;

const char *avs_last_error(void);
const char *avs_version(void);
/* ABI revision of this header: bumped whenever a struct layout or the set of exported entries changes (2: round 5 -- avs_matrix_format
 * carries struct_size, avs_solve_info.cancelled, avs_cancel; the measurement entries moved to libavs_probe.so in round 4). */
#define AVS_ABI_VERSION 2
int32_t avs_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Context.  Replaces the stack-allocated temporaries of solveGasSubclass (cpp:429-470, 507-551).
 * ---------------------------------------------------------------------------------------- */
avs_status avs_create(const avs_desc *desc, avs_ctx **out);
void avs_destroy(avs_ctx *ctx);

/* ------------------------------------------------------------------------------------------
 * Inputs of the hot path (produced by cpp:233-416 in the reference).
 * ---------------------------------------------------------------------------------------- */
/* octreeLabels.getGridLabels(level), HDK_OctreeGrid.h:144-148 */
avs_status avs_set_labels(avs_ctx *ctx, int32_t level, const int8_t *labels, avs_memspace where);
/* one grid of the three index pyramids, cpp:337-393 */
avs_status avs_set_index_field(avs_ctx *ctx, avs_index_kind kind, int32_t level, int32_t axis,
                               const int32_t *indices, avs_memspace where);
/* octreeVelocityDOFCount / edgeStressDOFCount / centerStressDOFCount, cpp:355-357, 395-408.
 * Validated against the uploaded grids (max id + 1). */
avs_status avs_set_dof_counts(avs_ctx *ctx, int64_t n_velocity, int64_t n_edge, int64_t n_center);
/* data == NULL selects the constant-field fast path (field()->isConstant, cpp:2090, 2248, 2501) */
avs_status avs_set_scalar_field(avs_ctx *ctx, avs_field_kind kind, int32_t axis, const float *data,
                                float constant, avs_memspace where);

/* ------------------------------------------------------------------------------------------
 * Hot path part 1: assembly (cpp:418-594 + setFromTriplets cpp:613-614).
 * avs_assemble runs all phases; the single phases are exported for parity tests and for
 * hosts that want Houdini perf-monitor scopes per phase (cpp:441, 473, 516, 554).
 * ---------------------------------------------------------------------------------------- */
avs_status avs_build_stencils(avs_ctx *ctx);      /* buildEdgeStressStencils + buildCenterStress{Stencils,Weights}, cpp:443-498 */
avs_status avs_build_initial_guess(avs_ctx *ctx); /* buildVelocityMapping, cpp:518-528 */
avs_status avs_build_system(avs_ctx *ctx);        /* buildOctreeSystemFromStencils + triplet merge, cpp:577-593, 613-614 */
avs_status avs_assemble(avs_ctx *ctx, avs_assembly_info *info /* may be NULL */);

/* ------------------------------------------------------------------------------------------
 * Hot path part 2: solve.  Eigen::ConjugateGradient<SparseMatrix<SolveType>, Lower|Upper>
 * with DiagonalPreconditioner, solveWithGuess(rhs, restrictedVelocity) (cpp:618-630).
 * ---------------------------------------------------------------------------------------- */
avs_status avs_solve(avs_ctx *ctx, double tolerance, int32_t max_iterations, avs_solve_info *info);
/* User interrupt: the reference polls UT_Interrupt::opInterrupt() inside its loops (cpp:2528; HDK_OctreeGrid.cpp:584-588).  Callable from
 * ANY thread while another thread is inside avs_solve / avs_dist_solve on the same context: the running loop ends at its next poll of
 * the device state (every 32 iterations of the launch-per-phase loops; the CU-resident loop, one cooperative launch of at most
 * max_iterations x ~20-70 us, cannot be interrupted: it runs to its end, and the request is consumed when it returns -- reported as
 * cancelled = 1, converged = 0 if the launch stopped at max_iterations without converging, otherwise the solve is simply done), the solve
 * returns AVS_OK with converged = 0 and cancelled = 1, and the request is consumed.  A request that is already pending when avs_solve
 * starts skips the CU-resident loop: the launch-per-phase loop consumes it at its first poll (0 iterations, cancelled = 1).  A request
 * that finds no solve running cancels the NEXT one -- a host whose watcher thread may call avs_cancel right after the solve returned
 * calls avs_cancel_clear once the watcher has been joined (shim/HDK_AdaptiveViscosity_avs.cpp does).  In a partitioned solve every rank
 * must be cancelled (the ranks leave the loop in the same iteration: the request travels with the CG sums). */
avs_status avs_cancel(avs_ctx *ctx);
avs_status avs_cancel_clear(avs_ctx *ctx);   /* drop a pending request (no solve may be running on the context) */

/* Solver options of the context (set any time before avs_solve / avs_dist_solve; the default is what the USEEIGEN build does).
 * AVS_OPTION_PRECONDITIONER: AVS_PRECONDITIONER_JACOBI (0, default: Eigen's DiagonalPreconditioner, cpp:618) or
 * AVS_PRECONDITIONER_NONE (1): plain CG, what the build WITHOUT USEEIGEN asks of HDK's UT_SparseMatrixRowT::solveConjugateGradient
 * (cpp:638-642: the preconditioner argument is nullptr).  That routine is closed source: the recurrence is standard CG, the stopping
 * rule used here stays Eigen's (|r|^2 < tol^2 |b|^2), and the result is only pinned to that extent. */
typedef enum {
    AVS_OPTION_PRECONDITIONER = 0,
    AVS_OPTION_RESIDENT_LOOP = 1, /* 1 (default): the CU-resident PCG loop where a system fits the chip; 0: always the launch-per-phase loops */
    AVS_OPTION_TRANSPORT = 2,     /* multi-GPU: AVS_USE_TRANSPORT_AUTO (direct after its connect-time self-test, else RCCL), _RCCL, _DIRECT */
    AVS_OPTION_PARANOID = 3,      /* multi-GPU: 1 = every round's halo segments are re-added by the reader and compared with the sender's checksum */
    AVS_OPTION_GRAPH_REPLAY = 4,  /* 1 (default): chunks of iterations replay a captured hipGraph */
    AVS_OPTION_BRICK_FORM = 5,    /* brick-structured SpMV form: AVS_BRICK_AUTO (systems of >= 2 M rows, kept where the tiles are full enough -- a
                                   * structural rule, so the same input always runs the same kernel), _NEVER, _ALWAYS, _TUNE (auto, decided by timing
                                   * both forms at the assembly: host-synchronous and not reproducible from run to run); takes effect at the next avs_assemble */
    AVS_OPTION_FUSED_SCALAR_STEPS = 6, /* 1 (default): the CG scalar steps ride in the vector kernels; 0: one reduction launch per step */
    AVS_OPTION_RELOAD_ENVIRONMENT = 7, /* any value: take the AVS_* environment variables again (they are read once, at avs_create; tools and tests) */
    AVS_OPTION_F32_VECTORS = 8,   /* AVS_PRECISION_F32 contexts, single-GPU solves: 1 = the iteration runs on float vectors with float scalars --
                                   * SolveType = fpreal32 through Eigen::ConjugateGradient (util.h:25-37, cpp:613-630) --; 0 = fp64 iteration on
                                   * the float system; -1 (default) = float vectors where the system is too large for the CU-resident loop
                                   * (the bandwidth of the vectors is what an iteration costs there), the resident fp64 loop where it fits the
                                   * chip (faster).  Takes effect at the next avs_assemble (the brick form's walk is laid out for the kernel
                                   * that will run). */
    AVS_OPTION_FUSED_VECTOR_UPDATE = 9  /* single-GPU launch-per-phase loop: r -= alpha t and x += alpha p ; p = z + beta p as ONE launch with a grid barrier
                                   * in between -- the new r stays in registers / LDS, 7.25 n instead of 8.5 n doubles per iteration; same sums in the
                                   * same order: iteration counts and solution bits do not depend on it.  0 never, 1 wherever a system qualifies
                                   * (>= 524,288 and <= 8,388,608 rows, a device with >= 256 CUs), -1 only systems larger than the Infinity Cache.
                                   * A barrier that is not passed within AVS_PCG_FUSED_TIMEOUT_MS (2000; the GPU shared with other work) redoes
                                   * the solve with the two launches.  Environment: AVS_PCG_FUSE_VECTORS. */
} avs_solver_option;
enum { AVS_USE_TRANSPORT_AUTO = 0, AVS_USE_TRANSPORT_RCCL = 1, AVS_USE_TRANSPORT_DIRECT = 2 };
enum { AVS_BRICK_AUTO = -1, AVS_BRICK_NEVER = 0, AVS_BRICK_ALWAYS = 1, AVS_BRICK_TUNE = 2 };
enum { AVS_PRECONDITIONER_JACOBI = 0, AVS_PRECONDITIONER_NONE = 1 };
avs_status avs_set_solver_option(avs_ctx *ctx, avs_solver_option option, int32_t value);

/* ------------------------------------------------------------------------------------------
 * Outputs.  avs_get_solution is what cpp:661-707 consumes (viscositySolution).
 * The others exist for parity tests.  Any pointer may be NULL to skip that array.
 * ---------------------------------------------------------------------------------------- */
avs_status avs_get_assembly_info(avs_ctx *ctx, avs_assembly_info *info);
/* storage the solver's SpMV streams for the assembled matrix (all forms are lossless: the same products
 * in the same order as plain CSR).  12 B/non-zero: fp64 value + int32 column (what Eigen holds, cpp:613);
 * 6 B: uint16 code into a table of the matrix's distinct values + int32 column; 4 B: code and column
 * packed in one word, when bits(n) + bits(table) <= 32.  With more than 2048 distinct values the 6-B form keeps one
 * dictionary per SpMV tile (tile_local_tables): the values of a 512-row tile repeat even when the matrix as a whole
 * has 10^4..10^5 distinct ones (smoothly varying viscosity), and a tile's table fits in LDS.  When code and column do not fit
 * one word directly (many values, or more than 2^25 columns) the column is stored tile-relative (column_windows): 4 B again. */
typedef struct avs_matrix_format {
    int32_t struct_size;        /* IN: sizeof(avs_matrix_format) of the caller's header -- only that many bytes are written (the struct has grown
                                 * between revisions; a caller built against an older header keeps working) */
    int32_t reordered;          /* 1 = rows/columns renumbered brick-major inside the solver */
    int32_t value_table_size;   /* distinct values; 0 = not value-indexed (> 65536 distinct values) */
    int32_t column_bits;        /* > 0 = packed form, columns in the low bits */
    int32_t bytes_per_nonzero;  /* 12, 6 or 4 */
    int32_t tile_local_tables;  /* 1 = one dictionary per 512-row SpMV tile (value_table_size = total entries, 8 B each, streamed once) */
    int32_t column_windows;     /* 1 = 4-B words code | window slot | offset: a tile's columns lie in <= 64 windows of 2^14 ids (+ 256 B per tile) */
    int32_t brick_tiles;        /* > 0 = the single-GPU loop multiplies with the brick-structured form (csrc/avs_brick.hip): rows of one 8^3
                                 * brick of fine cells stored as geometric row patterns (one 8-B descriptor per row), the rest as 4-B words */
    int32_t brick_patterns;     /* distinct row patterns of the whole matrix */
    int64_t brick_pattern_rows; /* rows stored as patterns */
    int64_t brick_bytes;        /* bytes the brick kernel streams per launch for the matrix (descriptors, runs, pattern lists, words) */
    int32_t brick_walk;         /* tile walk of the persistent workgroups: 0 one contiguous eighth of the tiles per XCD, 1 chunks dealt to the XCDs in turn */
    int32_t brick_value_codes;  /* 1 = variable-viscosity variant: the patterns carry the geometry only, every pattern row streams its own 2-B value codes
                                 * into a per-tile value table (round 5) */
    int32_t fused_vector_update; /* 1 = the last avs_solve ran the two vector kernels of an iteration as one launch (AVS_OPTION_FUSED_VECTOR_UPDATE, round 6) */
    int32_t fused_vector_faults; /* launches of it on this context whose grid barrier timed out (the solve was redone with the two launches) */
} avs_matrix_format;
avs_status avs_get_matrix_format(avs_ctx *ctx, avs_matrix_format *fmt);
avs_status avs_get_solution(avs_ctx *ctx, double *x, int64_t n, avs_memspace where);
avs_status avs_get_initial_guess(avs_ctx *ctx, double *x0, int64_t n, avs_memspace where);
/* hands the context a solution vector (reference DOF numbering, n_velocity entries) for the post-solve transfer: a hosted multi-GPU
 * group's host program sums the per-rank vectors of avs_dist_get_solution itself and gives every rank the whole vector back; also a
 * velocity computed elsewhere (tests: the same vector through two transfer paths).  The reference keeps its solution in `solution`
 * between cpp:645 and cpp:655-707. */
avs_status avs_set_solution(avs_ctx *ctx, const double *x, int64_t n, avs_memspace where);
avs_status avs_get_csr(avs_ctx *ctx, int32_t *row_ptr /* n+1 */, int32_t *col, double *val,
                       double *rhs, avs_memspace where);
/* SoA stencil records: entry k of stencil s lives at [k * count + s].
 * edge:   count = n_edge,     cap AVS_EDGE_STENCIL_CAP,   boundary cap AVS_EDGE_BOUNDARY_CAP
 * centre: count = 3*n_center (list id = cell id + n_center*axis, cpp:2186), weights n_center */
avs_status avs_get_edge_stencils(avs_ctx *ctx, int32_t *cnt, int32_t *idx, double *coef,
                                 int32_t *bcnt, double *bval, double *weight, avs_memspace where);
avs_status avs_get_center_stencils(avs_ctx *ctx, int32_t *cnt, int32_t *idx, double *coef,
                                   int32_t *bcnt, double *bval, double *weight, avs_memspace where);

/* ------------------------------------------------------------------------------------------
 * Post-solve transfer (SURVEY 8(f) "next #2"): what cpp:655-707 does with viscositySolution --
 * setOctreeVelocity (cpp:2779-2813), HDK_OctreeVectorFieldInterpolator (node values, T-junction
 * aware; HDK_OctreeVectorFieldInterpolator.cpp:118-845) and applyVelocitiesToRegularGrid
 * (cpp:2815-2894): the regular MAC-grid velocity Houdini reads back.
 * ---------------------------------------------------------------------------------------- */
/* regularVelocityIndices[axis] (cpp:303-329): >= 0 regular DOF, AVS_SOLIDBOUNDARY, else untouched face; on the face lattice
 * of the SIMULATION grid (field_n*, + 1 on `axis`) */
avs_status avs_set_regular_index_field(avs_ctx *ctx, int32_t axis, const int32_t *indices, avs_memspace where);
/* out_*: face lattices of the simulation grid (field_n*); faces that are not regular DOFs keep the input velocity */
avs_status avs_transfer_to_regular_grid(avs_ctx *ctx, float *out_x, float *out_y, float *out_z, avs_memspace where);
/* The same transfer as an IN-PLACE update, which is what the reference does to `vel` (cpp:655-707: only the faces named by
 * regularVelocityIndices are assigned).  vel_x/y/z: DEVICE arrays on the simulation grid's face lattices that already hold the velocity
 * field given to avs_set_scalar_field(AVS_FIELD_VELOCITY): only the faces the transfer changes are written -- on a sparse scene nine
 * tiles in ten are neither read nor written.  (Simulation grids that HDK_OctreeGrid::init had to pad are written face by face as by
 * avs_transfer_to_regular_grid.) */
avs_status avs_transfer_to_regular_grid_in_place(avs_ctx *ctx, float *vel_x, float *vel_y, float *vel_z);
/* interpolator node grids after all passes, (n+1)^3 per level: labels (0 inactive, 1 active), values fp32 */
avs_status avs_get_node_grid(avs_ctx *ctx, int32_t level, int8_t *labels, float *vx, float *vy, float *vz, avs_memspace where);

/* ------------------------------------------------------------------------------------------
 * Seam A: only the solve (replaces cpp:611-643).  CSR with int32 row pointers/columns, fp64
 * values; x_inout holds the initial guess on entry and the solution on return.
 * ---------------------------------------------------------------------------------------- */
avs_status avs_pcg_csr(int64_t n, const int32_t *row_ptr, const int32_t *col, const double *val,
                       const double *b, double *x_inout, double tolerance, int32_t max_iterations,
                       avs_memspace where, int32_t device, void *stream, avs_solve_info *info);

/* (measurement / test entries -- single SpMV launches, kernel sweeps, stream probes -- live in include/avs_probe.h and libavs_probe.so) */

/* ------------------------------------------------------------------------------------------
 * Pre-pass on the device (SURVEY 8(f) "next #1/#4"): what solveGasSubclass computes BEFORE the hot
 * path -- integration weights (cpp:712-766), refinement mask (cpp:815-867), octree label pyramid
 * (HDK_OctreeGrid.cpp:4-243), classification (cpp:1087-1443) and serial numbering in HDK tile
 * order (cpp:1445-1715) -- from the liquid / solid SDFs (centre lattice, fp32, x fastest).
 * ---------------------------------------------------------------------------------------- */
typedef struct avs_prepass avs_prepass;
typedef struct {
    int32_t nx, ny, nz;         /* powers of two */
    double dx;
    int32_t desired_levels;     /* getOctreeLevels(), cpp:263 */
    int32_t n_super;            /* getNumberSuperSamples(), cpp:756 (default 3) */
    double extrapolation_scale; /* getExtrapolation(), cpp:243 (default 0.5) */
    int32_t device;
    void *stream;
    /* Resolution of the SIMULATION grid the SDFs are given on when it is not a power of two per axis: nx, ny, nz are then the
     * octree grid HDK_OctreeGrid::init stretches it to (smallest powers of two that contain it, oct.cpp:10-24), cells outside
     * the simulation grid stay INACTIVE (oct.cpp:375-379).  0 = nx, ny, nz.  The SDFs are read with clamped coordinates outside
     * their grid (SIM_RawField border behaviour); the liquid must not touch the border of the simulation grid.  All getters
     * below return arrays on the (padded) octree lattices. */
    int32_t field_nx, field_ny, field_nz;
} avs_prepass_desc;
typedef struct {
    int32_t levels;                       /* after capping, HDK_OctreeGrid.cpp:198-211; 0 = no liquid */
    int64_t n_velocity, n_edge, n_center; /* cpp:395-408 */
    int64_t n_regular;                    /* regularVelocityDOFcount, cpp:323 */
    double weights_ms, octree_ms, classify_ms, number_ms;
} avs_prepass_info;
avs_status avs_prepass_create(const avs_prepass_desc *desc, avs_prepass **out);
/* Slab-local pre-pass (round 6; SURVEY 8(e): "each GPU assembles the rows it owns from its slab of the pyramids plus a halo").  The
 * reference's loops are row-local (classification / numbering cpp:1087-1715, octree oct.cpp:395-565), so a rank of a partitioned solve
 * needs them on its slab only.  cuts[0 .. world]: fine-cell coordinates along cut_axis, cuts[0] = 0, cuts[world] = the axis' extent,
 * ascending; rank r owns the faces whose position lies in [cuts[r], cuts[r + 1]).  avs_prepass_run then computes weights, mask, labels,
 * classification and ids only inside the rank's WINDOW: at level l the slab +- 12 level-l cells, rounded out to whole 16-entry tiles
 * (labels on as much more as the coarser levels' windows depend on).  The ids are the reference's GLOBAL ids (cpp:1566-1593 numbers tile
 * after tile through the whole lattice): every tile is counted by the rank that owns its first plane, `allreduce` sums the per-tile counts
 * of all ranks (ONE call per run, a few MB of int32 on the device, in place, enqueued on / synchronised with `stream`), and every rank scans
 * the same array.  The levels cap (oct.cpp:198-211) rides in the same buffer.  Everything inside the window is bit-identical to the
 * single-rank pre-pass; lattice entries outside it are unspecified.  world == 1 or cuts == NULL switches the mode off. */
typedef avs_status (*avs_allreduce_i32_fn)(int32_t *device_data, int64_t count, void *stream, void *user);
avs_status avs_prepass_set_slab(avs_prepass *pp, int32_t cut_axis, const int32_t *cuts, int32_t world_size, int32_t rank,
                                avs_allreduce_i32_fn allreduce, void *user);
/* the window of the last run along the cut axis: entries [lo[l], hi[l]) of the level-l lattices (hi == the level's cell count: to the end) */
avs_status avs_prepass_get_window(avs_prepass *pp, int32_t *lo, int32_t *hi /* AVS_MAX_LEVELS each */, int64_t *n_window /* 3: velocity, edge, centre DOFs inside it */);
void avs_prepass_destroy(avs_prepass *pp);
/* (AVS_MEM_DEVICE arrays on a power-of-two grid are read in place while the call runs -- no copy; host arrays and padded grids are staged) */
avs_status avs_prepass_run(avs_prepass *pp, const float *liquid_sdf, const float *solid_sdf /* NULL: none */, avs_memspace where);
avs_status avs_prepass_get_info(avs_prepass *pp, avs_prepass_info *info);
avs_status avs_prepass_get_labels(avs_prepass *pp, int32_t level, int8_t *out, avs_memspace where);
avs_status avs_prepass_get_mask(avs_prepass *pp, int8_t *out, avs_memspace where);
avs_status avs_prepass_get_index(avs_prepass *pp, avs_index_kind kind, int32_t level, int32_t axis, int32_t *out, avs_memspace where);
/* regularVelocityIndices[axis] (cpp:1445-1512) */
avs_status avs_prepass_get_regular_index(avs_prepass *pp, int32_t axis, int32_t *out, avs_memspace where);
avs_status avs_prepass_get_weights(avs_prepass *pp, avs_field_kind kind /* CENTER / EDGE / FACE weights */, int32_t axis, float *out,
                                   avs_memspace where);
/* hands labels, index pyramids, DOF counts, regular-grid indices and the three weight fields to a solve context created with
 * levels == info.levels on the same device.  BY REFERENCE (round 5; it used to copy ~12 full-size lattices per level set: 78 GB at
 * 1024^3): the context holds the pre-pass's allocations; the pre-pass keeps two per lattice and its next avs_prepass_run fills the set no
 * context references, so what a context was given stays as it is until its next avs_prepass_apply, and avs_prepass_destroy may be called
 * while contexts still use what they were lent.  An avs_set_* call on the context replaces a lent lattice by the context's own copy.
 * Frame after frame on one avs_prepass object the run also skips what its allocations already hold from their last filling (weight
 * bricks far from the surface whose constant has not changed, index tiles outside the last occupancy): outputs identical to a fresh
 * object's, bit for bit; AVS_PREPASS_TEMPORAL=0 in the environment at avs_prepass_create switches it off. */
avs_status avs_prepass_apply(avs_prepass *pp, avs_ctx *ctx);


/* dof -> lattice location tables built by the library from the index pyramids:
 * 4 x int32 per DOF = (level | axis << 8, i, j, k).  kind selects velocity / edge / centre. */
avs_status avs_get_dof_table(avs_ctx *ctx, avs_index_kind kind, int32_t *table, avs_memspace where);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU (the reference is single-process, so nothing here has a counterpart).
 * One process per GPU.  The octree is cut into spatial slabs along one axis (cuts on multiples of
 * 2^(levels-1) fine cells); every rank owns the rows of the faces inside its slab, renumbered
 * locally as [owned | halo grouped by owner]; per CG iteration the halo entries of p travel by
 * RCCL send/recv and the CG scalars by RCCL all-reduce, both enqueued on the solver's stream.
 *
 * Part 1 -- host-side planner (pure integer work on host arrays, usable without a GPU).
 * ---------------------------------------------------------------------------------------- */
typedef struct avs_plan avs_plan;
typedef struct {
    int64_t n_own, n_halo, nnz_local, n_send;
    int32_t n_peers;
} avs_plan_sizes;

/* owner rank of every velocity DOF: slabs along cut_axis balanced by row nnz */
avs_status avs_plan_owners(int64_t n, const int32_t *dof_table, const int32_t *row_ptr, int32_t levels,
                           int32_t cut_axis, int32_t extent_fine, int32_t world_size, int32_t *owner_out);
/* local structures of `rank`: owned rows, halo columns, local CSR pattern, send lists */
avs_status avs_plan_create(int64_t n, const int32_t *row_ptr, const int32_t *col, const int32_t *owner, int32_t rank,
                           int32_t world_size, avs_plan **out);
avs_status avs_plan_get_sizes(const avs_plan *plan, avs_plan_sizes *sizes);
/* own_global[n_own], halo_global[n_halo] (grouped by owner rank, ascending id), row_ptr_local[n_own+1],
 * col_local[nnz_local] (local ids: < n_own owned, else n_own + halo position), val_src[nnz_local]
 * (position of the entry in the global val array), peers[n_peers], send_counts / recv_counts[n_peers],
 * send_idx[n_send] (owned local ids, peer after peer).  Any pointer may be NULL. */
avs_status avs_plan_get_arrays(const avs_plan *plan, int32_t *own_global, int32_t *halo_global, int32_t *row_ptr_local,
                               int32_t *col_local, int32_t *val_src, int32_t *peers, int32_t *send_counts,
                               int32_t *recv_counts, int32_t *send_idx);
void avs_plan_destroy(avs_plan *plan);

/* ------------------------------------------------------------------------------------------
 * Part 2 -- device side.  Transport is RCCL (one process per GPU; the unique id is created on rank
 * 0 and broadcast by the host program, e.g. through torch.distributed) or, for single-GPU testing,
 * an in-process group of "virtual ranks" (one avs_ctx + one host thread each, same device).
 * ---------------------------------------------------------------------------------------- */
#define AVS_UNIQUE_ID_BYTES 128
avs_status avs_dist_get_unique_id(uint8_t id[AVS_UNIQUE_ID_BYTES]);
avs_status avs_dist_init(avs_ctx *ctx, const uint8_t id[AVS_UNIQUE_ID_BYTES], int32_t rank, int32_t world_size);

/* Hosted group: neither RCCL nor an in-process group -- the HOST PROGRAM carries the set-up data between the ranks (MPI, gloo,
 * files ...).  Only the direct transport (below) works in such a group.  Sequence on every rank: avs_dist_init_hosted,
 * avs_dist_assemble (or avs_assemble + avs_dist_partition), avs_dist_export_blob, exchange so that every rank holds all
 * blobs in rank order, avs_dist_import_blobs, avs_dist_solve.  avs_dist_get_solution then returns this rank's owned entries
 * (zeros elsewhere): the caller adds the per-rank vectors. */
#define AVS_DIST_BLOB_BYTES 512
avs_status avs_dist_init_hosted(avs_ctx *ctx, int32_t rank, int32_t world_size);
avs_status avs_dist_export_blob(avs_ctx *ctx, uint8_t blob[AVS_DIST_BLOB_BYTES]);
avs_status avs_dist_import_blobs(avs_ctx *ctx, const uint8_t *blobs /* world_size x AVS_DIST_BLOB_BYTES, rank order */);

typedef struct avs_local_group avs_local_group;
avs_status avs_local_group_create(int32_t world_size, avs_local_group **out);
void avs_local_group_destroy(avs_local_group *group);
avs_status avs_dist_init_local(avs_ctx *ctx, avs_local_group *group, int32_t rank);

/* Every rank calls it after avs_assemble on an identical (replicated) pyramid: keeps its slab of
 * the system (cut along `cut_axis`, -1 = longest axis) and builds halo / send lists.  The plan is built on the
 * device from the CSR in HBM; AVS_DIST_PLAN=host runs the host planner above on a downloaded copy instead
 * (identical arrays, kept as the reference for tests). */
avs_status avs_dist_partition(avs_ctx *ctx, int32_t cut_axis);
/* Distributed assembly ("each GPU assembles the rows it owns"): replaces avs_assemble + avs_dist_partition.
 * Every rank builds the cheap index-only pieces for the whole octree (dof tables, stress stencils, restriction, raw
 * triplet counts, brick-major permutation), picks the same slab cuts (balanced by raw triplet counts), then assembles,
 * sorts and compresses ONLY its own rows, and derives halo / send lists from them (the pattern is symmetric).  No
 * global matrix exists afterwards: avs_get_csr / avs_solve are unavailable, avs_dist_solve / avs_dist_get_solution work
 * as after avs_dist_partition.  info->nnz is the LOCAL non-zero count. */
avs_status avs_dist_assemble(avs_ctx *ctx, int32_t cut_axis, avs_assembly_info *info /* may be NULL */);
/* Slab-local path (round 6): the pre-pass of THIS rank's window + the assembly of its rows, nothing swept over the whole octree.
 *   avs_dist_bind_prepass(ctx, pp, cut_axis, cuts)   pp's next runs are slab-local (avs_prepass_set_slab with this rank, this group's
 *                                                     all-reduce -- RCCL or the in-process group; a hosted group passes its own callback
 *                                                     to avs_prepass_set_slab instead); cuts == NULL switches it off
 *   avs_prepass_run(pp, ...) ; avs_prepass_apply(pp, ctx) ; avs_dist_assemble(ctx, ...)      on every rank, collectively
 * The cuts must exist BEFORE anything is counted: take them from the previous frame -- avs_dist_get_cuts(ctx, 1, ...) returns the
 * cuts the last assembly's per-plane weights (summed over the ranks; a hosted group has no library-side exchange: it gets the cuts it
 * used) suggest for the next one, which = 0 the cuts it used -- or start from equal slabs.  avs_dist_assemble on such a context builds stencils for the stresses within 4 cells (of their level) of the
 * slab, the rank's rows (bit-identical to the reference's rows), halo and send lists; for the same cuts every array equals what the
 * replicated-index avs_dist_assemble produces.  Three lookup arrays indexed by DOF id (6 B per DOF, filled by memset) are the only
 * global-sized work.  avs_transfer_to_regular_grid(_in_place) on a slab-local context scatters and samples the DOFs of the rank's window
 * and writes the regular-grid faces whose position along the cut axis lies in the rank's slab (the others keep the input velocity);
 * it reads the WHOLE solution vector: avs_dist_get_solution first (it gathers it on every rank; hosted group: avs_set_solution).
 * Entries that need the whole pyramid (avs_assemble, avs_dist_partition) report AVS_ESTATE on a slab-local context. */
avs_status avs_dist_bind_prepass(avs_ctx *ctx, avs_prepass *pp, int32_t cut_axis, const int32_t *cuts /* world_size + 1, or NULL */);
avs_status avs_dist_get_cuts(avs_ctx *ctx, int32_t which, int32_t *cut_axis /* may be NULL */, int32_t *cuts /* world_size + 1 */);
avs_status avs_dist_get_plan_sizes(avs_ctx *ctx, avs_plan_sizes *sizes);
/* rows per SpMV tile (workgroup) of the solver's default kernel */
int32_t avs_spmv_tile_rows(void);
/* number of SpMV tiles that read no halo column (they run while the halo is in flight) / that do */
avs_status avs_dist_get_overlap_tiles(avs_ctx *ctx, int32_t *interior, int32_t *boundary);
/* the plan's arrays, copied to host memory (parity tests; sizes from avs_dist_get_plan_sizes / _overlap_tiles;
 * any pointer may be NULL): own_global[n_own], row_ptr_local[n_own+1], col_local[nnz_local], send_idx[n_send],
 * peers / send_counts / recv_counts[n_peers], tiles_interior[], tiles_boundary[] */
avs_status avs_dist_get_plan_arrays(avs_ctx *ctx, int32_t *own_global, int32_t *row_ptr_local, int32_t *col_local,
                                    int32_t *send_idx, int32_t *peers, int32_t *send_counts, int32_t *recv_counts,
                                    int32_t *tiles_interior, int32_t *tiles_boundary);
avs_status avs_dist_solve(avs_ctx *ctx, double tolerance, int32_t max_iterations, avs_solve_info *info);
/* how the partitioned solve of this context communicates (filled after avs_dist_partition / avs_dist_assemble) */
typedef enum {
    AVS_TRANSPORT_RCCL = 0,   /* k_pack + ncclSend/ncclRecv halo exchange, ncclAllReduce of the CG scalars */
    AVS_TRANSPORT_DIRECT = 1  /* peer-mapped comm blocks (IPC handles / peer access over xGMI): boundary entries are stored
                                 straight into the neighbour's halo area, CG scalars by a flag-based all-gather; no RCCL call
                                 inside the iteration */
} avs_dist_transport;
typedef struct {
    int32_t world_size;
    int32_t rccl_ranks;                /* ncclCommCount of the communicator; 0 = no RCCL communicator (in-process / hosted group) */
    int32_t transport;                 /* avs_dist_transport the next avs_dist_solve will use */
    int32_t graph_replay;              /* 1 = full chunks of iterations are replayed from a captured hipGraph */
    int32_t launches_per_iteration;    /* kernel launches per CG iteration of the loop in use */
    int32_t collectives_per_iteration; /* RCCL calls per CG iteration */
    int32_t selftest_rounds;           /* direct transport: pattern rounds run over the connected comm blocks before the first solve
                                          (AVS_DIST_SELFTEST_ROUNDS, default 64; 0 = not run) */
    int32_t paranoid;                  /* 1 = every round's halo segments are checked against the sender's checksum (AVS_DIST_PARANOID=1) */
    int64_t selftest_bad_entries;      /* all-gathered count of wrong halo entries / checksums in those rounds (0 = passed, -1 = not run,
                                          -2 = a wait timed out); a non-zero count retires the direct transport for the plan */
} avs_dist_info;
avs_status avs_dist_get_info(avs_ctx *ctx, avs_dist_info *info);
/* gathers the full solution (global DOF order) on every rank */
avs_status avs_dist_get_solution(avs_ctx *ctx, double *x, int64_t n, avs_memspace where);

#ifdef __cplusplus
}
#endif
#endif /* AVS_H */
