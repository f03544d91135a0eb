/*
 * avs_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the adaptive variational viscosity hot path of
 * rgoldade/AdaptiveViscositySolver (reference @ 2024_08_07), plus the pre-pass
 * that produces the hot path's inputs.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (adaptiveviscositysolver_amd/, include/avs.h) never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, no tests and cannot
 * be built here (it needs the proprietary Houdini HDK and Eigen; neither is in
 * /root/reference).  This restatement is therefore pinned by (i) line-by-line
 * citations into the reference, (ii) the reference's own debug invariants
 * restated in tests/, (iii) maths-derived known answers.
 *
 * Reference citations use: cpp: = Source/HDK_AdaptiveViscosity.cpp,
 * oct.cpp:/oct.h: = Source/HDK_OctreeGrid.{cpp,h}, util.h: = Source/HDK_Utilities.h
 */
#ifndef AVS_ORACLE_H
#define AVS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 8
#define ORC_EDGE_CAP 32   /* max entries of one edge-stress stencil (cpp:1789-1907) */
#define ORC_CENTER_CAP 8  /* max entries of one centre-stress stencil (cpp:1925-1962) */
#define ORC_EDGE_BCAP 4
#define ORC_CENTER_BCAP 2

/* octree cell labels, oct.h:33-39 */
enum { ORC_INACTIVE = 0, ORC_ACTIVE = 1, ORC_UP = 2, ORC_DOWN = 3 };
/* index sentinels, util.h:18-21 */
enum { ORC_FLUID = 0, ORC_UNASSIGNED = -1, ORC_SOLIDBOUNDARY = -2, ORC_OUTSIDE = -3 };

/* field kinds for orc_set_field / orc_get_field (fp32, dense, x fastest) */
enum {
    ORC_F_LIQUID = 0,      /* liquid SDF, centre lattice                    */
    ORC_F_SOLID = 1,       /* solid SDF (positive inside solid), centre     */
    ORC_F_VISCOSITY = 2,   /* centre                                        */
    ORC_F_DENSITY = 3,     /* centre                                        */
    ORC_F_VELOCITY = 4,    /* + axis : regular-grid face velocity           */
    ORC_F_SOLIDVEL = 7,    /* + axis : solid velocity, face lattice         */
    ORC_F_FACEW = 10,      /* + axis : FLIP face weights ("surfaceweights") */
    ORC_F_CENTERW = 13,    /* centre integration weights                    */
    ORC_F_EDGEW = 14       /* + axis : edge integration weights             */
};

/* index-grid kinds (int32, dense, x fastest, per level) */
enum { ORC_I_VELOCITY = 0, ORC_I_EDGE = 1, ORC_I_CENTER = 2 };

typedef struct orc_ctx orc_ctx;

orc_ctx *orc_create(int nx, int ny, int nz, double dx, double dt, int desired_levels,
                    int use_enhanced_gradients);
void orc_destroy(orc_ctx *c);

/* inputs; ptr == NULL sets the field to the constant `cval` (HDK constant-field fast path) */
int orc_set_field(orc_ctx *c, int kind, const float *data, float cval);
int orc_get_field(orc_ctx *c, int kind, float *out);
int64_t orc_field_size(orc_ctx *c, int kind);

/* ---- pre-pass (inputs of the hot path; cpp:712-1715, oct.cpp:4-920) ---- */
int orc_build_weights(orc_ctx *c, int n_super, int also_face_weights);
int orc_build_octree(orc_ctx *c, double extrapolation_scale);
int orc_build_indices(orc_ctx *c, double extrapolation_scale);

/* direct setters so the hot path can run on externally produced inputs */
int orc_set_levels(orc_ctx *c, int levels);
int orc_set_labels(orc_ctx *c, int level, const int8_t *labels);
int orc_set_index(orc_ctx *c, int kind, int level, int axis, const int32_t *idx);
int orc_finalize_indices(orc_ctx *c); /* recount DOFs + rebuild dof tables after orc_set_index */

int orc_levels(orc_ctx *c);
int64_t orc_grid_size(orc_ctx *c, int kind, int level, int axis, int res_out[3]); /* kind: 0 vel,1 edge,2 center/labels */
int orc_get_labels(orc_ctx *c, int level, int8_t *out);
int orc_get_mask(orc_ctx *c, int8_t *out);
int orc_get_index(orc_ctx *c, int kind, int level, int axis, int32_t *out);
int64_t orc_count(orc_ctx *c, int kind);
/* dof -> packed location: out[4*d+0] = level | axis<<8, then i, j, k */
int orc_get_dof_table(orc_ctx *c, int kind, int32_t *out);

/* ---- hot path (cpp:418-653) ---- */
int orc_build_stencils(orc_ctx *c);      /* cpp:429-499 */
int orc_build_initial_guess(orc_ctx *c); /* cpp:507-529 */
int orc_assemble(orc_ctx *c);            /* cpp:537-594 + setFromTriplets cpp:613-614 */

/* stencil read-back (SoA: entry k of stencil s at [k*count + s]) */
int orc_get_edge_stencils(orc_ctx *c, int32_t *cnt, int32_t *idx, double *coef, int32_t *bcnt,
                          double *bval, double *weight);
int orc_get_center_stencils(orc_ctx *c, int32_t *cnt, int32_t *idx, double *coef, int32_t *bcnt,
                            double *bval, double *weight);
int orc_get_initial_guess(orc_ctx *c, double *x0);
int64_t orc_nnz(orc_ctx *c);
int64_t orc_raw_triplets(orc_ctx *c);
int orc_get_csr(orc_ctx *c, int64_t *row_ptr, int32_t *col, double *val, double *rhs);

/* ---- solve: Jacobi-PCG exactly in the order of Eigen::ConjugateGradient (cpp:618-630) ---- */
typedef struct {
    int iterations;
    double error;        /* sqrt(|r|^2 / |b|^2)         */
    double rhs_norm2;
    double seconds;      /* wall-clock of the CG loop   */
    double spmv_seconds; /* accumulated SpMV time       */
    int threads;
} orc_pcg_info;

int orc_pcg_csr(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *val,
                const double *b, double *x_inout, double tol, int max_iters, int threads,
                orc_pcg_info *info);
/* separate thread counts for the SpMV and for the vector operations ("eigen_faithful" = (T, 1), see the .c file) */
int orc_pcg_csr_ex(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *val,
                   const double *b, double *x, double tol, int max_iters, int spmv_threads, int vec_threads,
                   orc_pcg_info *info);
int orc_spmv_csr(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *val,
                 const double *x, double *y, int threads);
/* SolveType of the reference (util.h:25-37): 0 = fpreal64 (default), 1 = fpreal32 (USESINGLEPRECISION): triplets narrowed to float where
 * Eigen::Triplet<SolveType> is built, duplicates summed in float, rhs updated in float steps, initial guess narrowed at its store, the
 * CG run with float scalars and vectors.  Call before orc_build_initial_guess / orc_assemble.  Outputs stay double arrays (float values). */
int orc_set_precision(orc_ctx *c, int f32);
float orc_dot_f32(const float *a, const float *b, int64_t n); /* a.dot(b) of two VectorXf in the order of Eigen's SSE2 linear vectorised redux */
int orc_set_f32_serial_dots(int on); /* float CG: 1 = left-to-right float dots instead of Eigen's SSE2 redux order (thread-local) */
/* 0 = Jacobi (default), 1 = plain CG: the build without USEEIGEN hands HDK's solveConjugateGradient no preconditioner (cpp:638-642) */
int orc_set_preconditioner(orc_ctx *c, int none);
int orc_solve(orc_ctx *c, double tol, int max_iters, int threads, double *x_out, orc_pcg_info *info);

/* ---- post-solve transfer (cpp:655-707) ---------------------------------
 * regular-grid face classification + numbering (cpp:1087-1165, 1445-1512), then setOctreeVelocity
 * (cpp:2779-2813), HDK_OctreeVectorFieldInterpolator (interp.h:30-138, interp.cpp:118-845) and
 * applyVelocitiesToRegularGrid (cpp:2815-2894).  Positions are handled in exact index space (every
 * sample point is a lattice point); where HDK's fp32 position round trip would decide a tie (a face
 * lying exactly on a cell boundary) the forward cell is taken.  PARITY UNPINNED like the rest. */
int orc_build_regular_indices(orc_ctx *c, double extrapolation_scale);
int64_t orc_regular_count(orc_ctx *c);
int orc_get_regular_index(orc_ctx *c, int axis, int32_t *out);
int orc_set_regular_index(orc_ctx *c, int axis, const int32_t *idx);
/* out[a]: face lattice a of the base grid, starts as a copy of the input velocity */
int orc_transfer_to_regular_grid(orc_ctx *c, const double *solution, float *out_x, float *out_y, float *out_z);
/* node grids of the interpolator after all passes, for parity tests: labels int8, values fp32 */
int orc_get_node_grid(orc_ctx *c, int level, int8_t *labels, float *vx, float *vy, float *vz);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
