/*
 * avs_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See avs_oracle.h.
 *
 * PARITY UNPINNED (no reference vectors exist; see header).  Every function cites the
 * reference lines it restates.  Compile with -ffp-contract=off: products and sums must
 * round exactly once each so that the HIP path can be compared bit-for-bit where the
 * operation order is the same.
 *
 * Conventions restated from SURVEY.md App. A (HDK behaviour, documented but unpinned):
 *   - dense grids, x fastest; lattices: centre (nx,ny,nz); face a (+1 on a);
 *     edge a (+1 on the two other axes)                                (util.h:13-16)
 *   - a level-l grid has resolution base>>l and voxel size dx*2^l      (oct.cpp:58-70)
 *   - HDK voxel arrays are stored as 16^3 tiles; "for each tile, for each voxel" visits
 *     tiles x-fastest and voxels inside a tile x-fastest.  That order defines DOF
 *     numbering                                                        (cpp:1566-1593)
 *   - SIM_RawField::getValue(pos) = trilinear interpolation on the field's own lattice,
 *     clamped at the border.  All positions the reference samples at are lattice points
 *     of some level, i.e. exact multiples of half a fine cell, so here sampling is done
 *     in exact index space (weights are 0 or 1/2), fp32 arithmetic, x then y then z.
 */
#include "avs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16

typedef struct {
    float *data;
    float cval;
    int is_const;
} fieldf;

struct orc_ctx {
    int n[3];
    double dx, dt;
    int desired_levels, levels, enhanced;
    int f32; /* SolveType = fpreal32 (USESINGLEPRECISION, util.h:25-37): see orc_set_precision */
    int no_precond; /* plain CG (the build without USEEIGEN passes no preconditioner, cpp:638-642): see orc_set_preconditioner */

    fieldf liquid, solid, visc, dens, vel[3], solidvel[3], facew[3], centerw, edgew[3];

    int8_t *mask;
    int8_t *labels[ORC_MAX_LEVELS];
    int32_t *vidx[ORC_MAX_LEVELS][3], *eidx[ORC_MAX_LEVELS][3], *cidx[ORC_MAX_LEVELS];
    int64_t nvel, nedge, ncenter;
    int32_t *vdof, *edof, *cdof; /* 4 ints per dof: level|axis<<8, i, j, k */

    /* stencils, SoA */
    int32_t *e_cnt, *e_idx, *e_bcnt;
    double *e_coef, *e_bval, *e_w;
    int32_t *c_cnt, *c_idx, *c_bcnt; /* 3*ncenter lists */
    double *c_coef, *c_bval, *c_w;   /* c_w: ncenter */

    int32_t *ridx[3];
    int64_t nregular;
    int8_t *nlab[ORC_MAX_LEVELS];
    float *nval[ORC_MAX_LEVELS][3];

    double *x0, *rhs;
    int64_t *row_ptr;
    int32_t *col;
    double *val;
    int64_t nnz, nraw;
};

/* ------------------------------------------------------------------------------------ */
/* small helpers                                                                        */
/* ------------------------------------------------------------------------------------ */
static inline void cell_res(const orc_ctx *c, int l, int r[3])
{
    r[0] = c->n[0] >> l;
    r[1] = c->n[1] >> l;
    r[2] = c->n[2] >> l;
}
static inline void face_res(const orc_ctx *c, int l, int a, int r[3])
{
    cell_res(c, l, r);
    r[a] += 1;
}
static inline void edge_res(const orc_ctx *c, int l, int a, int r[3])
{
    cell_res(c, l, r);
    r[(a + 1) % 3] += 1;
    r[(a + 2) % 3] += 1;
}
static inline size_t lin(const int r[3], const int p[3])
{
    return (size_t)p[0] + (size_t)r[0] * ((size_t)p[1] + (size_t)r[1] * (size_t)p[2]);
}
static inline size_t vol(const int r[3]) { return (size_t)r[0] * (size_t)r[1] * (size_t)r[2]; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int floordiv2(int v) { return v >> 1; } /* arithmetic shift == floor(v/2) */

static int is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* fp32 lerp, one rounding per operation (build with -ffp-contract=off) */
static inline float lerp32(float a, float b, float t)
{
    float s = 1.0f - t;
    float pa = a * s;
    float pb = b * t;
    return pa + pb;
}

/* lattice descriptor: off2[a] = 1 when samples sit at cell centres along a, 0 when on nodes */
static const int OFF_CENTER[3] = {1, 1, 1};
static inline void off_face(int a, int o[3])
{
    o[0] = o[1] = o[2] = 1;
    o[a] = 0;
}
static inline void off_edge(int a, int o[3])
{
    o[0] = o[1] = o[2] = 0;
    o[a] = 1;
}

/* Sample field F (lattice offsets off2, resolution r) at the point whose coordinates, in
 * units of HALF a fine cell, are P2.  Restates SIM_RawField::getValue (HDK, unpinned). */
static float sample_f32(const fieldf *F, const int r[3], const int off2[3], const int P2[3])
{
    if (F->is_const)
        return F->cval;
    int i0[3], i1[3];
    float t[3];
    for (int a = 0; a < 3; ++a) {
        int q2 = P2[a] - off2[a];
        int fl = floordiv2(q2);
        t[a] = (q2 - 2 * fl) ? 0.5f : 0.0f;
        i0[a] = clampi(fl, 0, r[a] - 1);
        i1[a] = clampi(fl + 1, 0, r[a] - 1);
    }
    const float *d = F->data;
#define AT(ii, jj, kk) d[(size_t)(ii) + (size_t)r[0] * ((size_t)(jj) + (size_t)r[1] * (size_t)(kk))]
    float c00 = lerp32(AT(i0[0], i0[1], i0[2]), AT(i1[0], i0[1], i0[2]), t[0]);
    float c10 = lerp32(AT(i0[0], i1[1], i0[2]), AT(i1[0], i1[1], i0[2]), t[0]);
    float c01 = lerp32(AT(i0[0], i0[1], i1[2]), AT(i1[0], i0[1], i1[2]), t[0]);
    float c11 = lerp32(AT(i0[0], i1[1], i1[2]), AT(i1[0], i1[1], i1[2]), t[0]);
#undef AT
    float c0 = lerp32(c00, c10, t[1]);
    float c1 = lerp32(c01, c11, t[1]);
    return lerp32(c0, c1, t[2]);
}

/* positions (half-fine-cell units) of level-l samples */
static inline void pos2_center(int l, const int p[3], int P2[3])
{
    for (int a = 0; a < 3; ++a)
        P2[a] = (2 * p[a] + 1) << l;
}
static inline void pos2_face(int l, int axis, const int p[3], int P2[3])
{
    for (int a = 0; a < 3; ++a)
        P2[a] = (a == axis) ? ((2 * p[a]) << l) : ((2 * p[a] + 1) << l);
}
static inline void pos2_edge(int l, int axis, const int p[3], int P2[3])
{
    for (int a = 0; a < 3; ++a)
        P2[a] = (a == axis) ? ((2 * p[a] + 1) << l) : ((2 * p[a]) << l);
}

/* topology helpers, util.h:46-217 */
static inline void face_to_cell(const int f[3], int axis, int dir, int c[3])
{
    c[0] = f[0]; c[1] = f[1]; c[2] = f[2];
    if (dir == 0) --c[axis];
}
static inline void cell_to_face(const int c[3], int axis, int dir, int f[3])
{
    f[0] = c[0]; f[1] = c[1]; f[2] = c[2];
    if (dir == 1) ++f[axis];
}
static inline void face_to_edge(const int f[3], int faceAxis, int edgeAxis, int dir, int e[3])
{
    e[0] = f[0]; e[1] = f[1]; e[2] = f[2];
    if (dir == 1) ++e[3 - faceAxis - edgeAxis];
}
static inline void edge_to_face(const int e[3], int edgeAxis, int faceAxis, int dir, int f[3])
{
    f[0] = e[0]; f[1] = e[1]; f[2] = e[2];
    if (dir == 0) --f[3 - faceAxis - edgeAxis];
}
static inline void edge_to_cell(const int e[3], int edgeAxis, int ci, int c[3])
{
    c[0] = e[0]; c[1] = e[1]; c[2] = e[2];
    for (int o = 0; o < 2; ++o)
        if (!(ci & (1 << o)))
            --c[(edgeAxis + 1 + o) % 3];
}
static inline void cell_to_edge(const int c[3], int edgeAxis, int ei, int e[3])
{
    e[0] = c[0]; e[1] = c[1]; e[2] = c[2];
    for (int o = 0; o < 2; ++o)
        if (ei & (1 << o))
            ++e[(edgeAxis + 1 + o) % 3];
}
/* oct.h:94-142 */
static inline void child_face(const int f[3], int axis, int ci, int o[3])
{
    o[0] = 2 * f[0]; o[1] = 2 * f[1]; o[2] = 2 * f[2];
    if (ci & 1) ++o[(axis + 1) % 3];
    if (ci & 2) ++o[(axis + 2) % 3];
}
static inline void child_edge(const int e[3], int edgeAxis, int ci, int o[3])
{
    o[0] = 2 * e[0]; o[1] = 2 * e[1]; o[2] = 2 * e[2];
    if (ci > 0) ++o[edgeAxis];
}
static inline void child_edge_in_face(const int f[3], int faceAxis, int edgeAxis, int ci, int o[3])
{
    o[0] = 2 * f[0]; o[1] = 2 * f[1]; o[2] = 2 * f[2];
    if (ci == 1) ++o[edgeAxis];
    ++o[3 - faceAxis - edgeAxis];
}

static void field_free(fieldf *f)
{
    free(f->data);
    f->data = NULL;
}

/* ------------------------------------------------------------------------------------ */
/* context                                                                              */
/* ------------------------------------------------------------------------------------ */
orc_ctx *orc_create(int nx, int ny, int nz, double dx, double dt, int desired_levels,
                    int use_enhanced_gradients)
{
    /* Power-of-two base resolution: the reference pads the level-0 grid up to powers of
     * two (oct.cpp:18-24); with power-of-two input no padding happens. */
    if (!is_pow2(nx) || !is_pow2(ny) || !is_pow2(nz) || desired_levels < 1 ||
        desired_levels > ORC_MAX_LEVELS)
        return NULL;
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    if (!c) return NULL;
    c->n[0] = nx; c->n[1] = ny; c->n[2] = nz;
    c->dx = dx;
    c->dt = dt;
    c->desired_levels = desired_levels;
    c->enhanced = use_enhanced_gradients;
    /* level cap, oct.cpp:32-40 */
    int lv = desired_levels;
    for (int a = 0; a < 3; ++a) {
        int lg = 0;
        while ((1 << (lg + 1)) <= c->n[a]) ++lg;
        if (lg < lv) lv = lg;
    }
    if (lv < 1) lv = 1;
    c->levels = lv;
    fieldf *fs[] = {&c->liquid, &c->solid, &c->visc, &c->dens, &c->vel[0], &c->vel[1], &c->vel[2],
                    &c->solidvel[0], &c->solidvel[1], &c->solidvel[2], &c->facew[0], &c->facew[1],
                    &c->facew[2], &c->centerw, &c->edgew[0], &c->edgew[1], &c->edgew[2]};
    for (size_t i = 0; i < sizeof(fs) / sizeof(fs[0]); ++i) {
        fs[i]->is_const = 1;
        fs[i]->cval = 0.f;
    }
    c->solid.cval = -1.f; /* far from any solid (positive inside solid, cpp:1157) */
    c->visc.cval = 1.f;
    c->dens.cval = 1.f;
    for (int a = 0; a < 3; ++a) c->facew[a].cval = 1.f;
    return c;
}

static void free_indices(orc_ctx *c)
{
    for (int l = 0; l < ORC_MAX_LEVELS; ++l) {
        for (int a = 0; a < 3; ++a) {
            free(c->vidx[l][a]); c->vidx[l][a] = NULL;
            free(c->eidx[l][a]); c->eidx[l][a] = NULL;
        }
        free(c->cidx[l]); c->cidx[l] = NULL;
    }
    free(c->vdof); free(c->edof); free(c->cdof);
    c->vdof = c->edof = c->cdof = NULL;
}
static void free_stencils(orc_ctx *c)
{
    free(c->e_cnt); free(c->e_idx); free(c->e_bcnt); free(c->e_coef); free(c->e_bval); free(c->e_w);
    free(c->c_cnt); free(c->c_idx); free(c->c_bcnt); free(c->c_coef); free(c->c_bval); free(c->c_w);
    c->e_cnt = c->e_idx = c->e_bcnt = NULL; c->e_coef = c->e_bval = c->e_w = NULL;
    c->c_cnt = c->c_idx = c->c_bcnt = NULL; c->c_coef = c->c_bval = c->c_w = NULL;
}
static void free_system(orc_ctx *c)
{
    free(c->x0); free(c->rhs); free(c->row_ptr); free(c->col); free(c->val);
    c->x0 = c->rhs = NULL; c->row_ptr = NULL; c->col = NULL; c->val = NULL;
}

void orc_destroy(orc_ctx *c)
{
    if (!c) return;
    field_free(&c->liquid); field_free(&c->solid); field_free(&c->visc); field_free(&c->dens);
    field_free(&c->centerw);
    for (int a = 0; a < 3; ++a) {
        field_free(&c->vel[a]); field_free(&c->solidvel[a]); field_free(&c->facew[a]);
        field_free(&c->edgew[a]);
    }
    free(c->mask);
    for (int l = 0; l < ORC_MAX_LEVELS; ++l) free(c->labels[l]);
    for (int a = 0; a < 3; ++a) free(c->ridx[a]);
    for (int l = 0; l < ORC_MAX_LEVELS; ++l) { free(c->nlab[l]); for (int a = 0; a < 3; ++a) free(c->nval[l][a]); }
    free_indices(c);
    free_stencils(c);
    free_system(c);
    free(c);
}

static fieldf *field_of(orc_ctx *c, int kind, int r[3])
{
    if (kind == ORC_F_LIQUID) { cell_res(c, 0, r); return &c->liquid; }
    if (kind == ORC_F_SOLID) { cell_res(c, 0, r); return &c->solid; }
    if (kind == ORC_F_VISCOSITY) { cell_res(c, 0, r); return &c->visc; }
    if (kind == ORC_F_DENSITY) { cell_res(c, 0, r); return &c->dens; }
    if (kind >= ORC_F_VELOCITY && kind < ORC_F_VELOCITY + 3) { face_res(c, 0, kind - ORC_F_VELOCITY, r); return &c->vel[kind - ORC_F_VELOCITY]; }
    if (kind >= ORC_F_SOLIDVEL && kind < ORC_F_SOLIDVEL + 3) { face_res(c, 0, kind - ORC_F_SOLIDVEL, r); return &c->solidvel[kind - ORC_F_SOLIDVEL]; }
    if (kind >= ORC_F_FACEW && kind < ORC_F_FACEW + 3) { face_res(c, 0, kind - ORC_F_FACEW, r); return &c->facew[kind - ORC_F_FACEW]; }
    if (kind == ORC_F_CENTERW) { cell_res(c, 0, r); return &c->centerw; }
    if (kind >= ORC_F_EDGEW && kind < ORC_F_EDGEW + 3) { edge_res(c, 0, kind - ORC_F_EDGEW, r); return &c->edgew[kind - ORC_F_EDGEW]; }
    return NULL;
}

int64_t orc_field_size(orc_ctx *c, int kind)
{
    int r[3];
    if (!field_of(c, kind, r)) return -1;
    return (int64_t)vol(r);
}

int orc_set_field(orc_ctx *c, int kind, const float *data, float cval)
{
    int r[3];
    fieldf *f = field_of(c, kind, r);
    if (!f) return 1;
    free(f->data);
    f->data = NULL;
    if (!data) {
        f->is_const = 1;
        f->cval = cval;
        return 0;
    }
    size_t n = vol(r);
    f->data = (float *)malloc(n * sizeof(float));
    if (!f->data) return 2;
    memcpy(f->data, data, n * sizeof(float));
    f->is_const = 0;
    return 0;
}

int orc_get_field(orc_ctx *c, int kind, float *out)
{
    int r[3];
    fieldf *f = field_of(c, kind, r);
    if (!f) return 1;
    size_t n = vol(r);
    if (f->is_const)
        for (size_t i = 0; i < n; ++i) out[i] = f->cval;
    else
        memcpy(out, f->data, n * sizeof(float));
    return 0;
}

static inline float fget(const fieldf *f, const int r[3], const int p[3])
{
    return f->is_const ? f->cval : f->data[lin(r, p)];
}

/* ------------------------------------------------------------------------------------ */
/* pre-pass 1: integration weights  (cpp:712-791; arithmetic lives in HDK                */
/* SIM_RawField::computeSDFWeightsSampled -- UNPINNED, definition below)                 */
/*                                                                                      */
/* weight(sample) = fraction of n^3 sub-samples of the voxel-sized box centred on the    */
/* sample whose trilinearly interpolated liquid SDF is < 0.  Sub-sample s along one axis */
/* sits at centre + ((s+0.5)/n - 0.5) voxels.  The constants (integer cell offset and    */
/* fp32 fraction) are derived in double from that expression and rounded once.           */
/* ------------------------------------------------------------------------------------ */
static void subsample_consts(int n, int target_off2, int s, int *di, float *fr)
{
    double d = ((target_off2 ? 0.5 : 0.0) - 0.5) + (((double)s + 0.5) / (double)n - 0.5);
    double fl = floor(d);
    *di = (int)fl;
    *fr = (float)(d - fl);
}

static void weights_for_lattice(const orc_ctx *c, const int toff2[3], const int tr[3], int n,
                                float *out)
{
    int sr[3];
    cell_res(c, 0, sr);
    const float *sdf = c->liquid.data;
    int di[3][16];
    float fr[3][16];
    for (int a = 0; a < 3; ++a)
        for (int s = 0; s < n; ++s)
            subsample_consts(n, toff2[a], s, &di[a][s], &fr[a][s]);
    const float n3 = (float)(n * n * n);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < tr[2]; ++k)
        for (int j = 0; j < tr[1]; ++j)
            for (int i = 0; i < tr[0]; ++i) {
                int p[3] = {i, j, k};
                int count = 0;
                if (c->liquid.is_const) {
                    count = (c->liquid.cval < 0.f) ? n * n * n : 0;
                } else {
                    /* exact shortcut: interpolation is sign preserving, so an all-negative
                     * (all non-negative) neighbourhood gives n^3 (0) */
                    int lo[3], hi[3], allneg = 1, allpos = 1;
                    for (int a = 0; a < 3; ++a) {
                        lo[a] = clampi(p[a] + di[a][0], 0, sr[a] - 1);
                        hi[a] = clampi(p[a] + di[a][n - 1] + 1, 0, sr[a] - 1);
                    }
                    for (int kk = lo[2]; kk <= hi[2] && (allneg || allpos); ++kk)
                        for (int jj = lo[1]; jj <= hi[1]; ++jj)
                            for (int ii = lo[0]; ii <= hi[0]; ++ii) {
                                float v = sdf[(size_t)ii + (size_t)sr[0] * ((size_t)jj + (size_t)sr[1] * (size_t)kk)];
                                if (v < 0.f) allpos = 0; else allneg = 0;
                            }
                    if (allneg) count = n * n * n;
                    else if (allpos) count = 0;
                    else {
                        for (int sz = 0; sz < n; ++sz)
                            for (int sy = 0; sy < n; ++sy)
                                for (int sx = 0; sx < n; ++sx) {
                                    int s3[3] = {sx, sy, sz};
                                    int i0[3], i1[3];
                                    float t[3];
                                    for (int a = 0; a < 3; ++a) {
                                        int b = p[a] + di[a][s3[a]];
                                        i0[a] = clampi(b, 0, sr[a] - 1);
                                        i1[a] = clampi(b + 1, 0, sr[a] - 1);
                                        t[a] = fr[a][s3[a]];
                                    }
#define AT(ii, jj, kk) sdf[(size_t)(ii) + (size_t)sr[0] * ((size_t)(jj) + (size_t)sr[1] * (size_t)(kk))]
                                    float c00 = lerp32(AT(i0[0], i0[1], i0[2]), AT(i1[0], i0[1], i0[2]), t[0]);
                                    float c10 = lerp32(AT(i0[0], i1[1], i0[2]), AT(i1[0], i1[1], i0[2]), t[0]);
                                    float c01 = lerp32(AT(i0[0], i0[1], i1[2]), AT(i1[0], i0[1], i1[2]), t[0]);
                                    float c11 = lerp32(AT(i0[0], i1[1], i1[2]), AT(i1[0], i1[1], i1[2]), t[0]);
#undef AT
                                    float c0 = lerp32(c00, c10, t[1]);
                                    float c1 = lerp32(c01, c11, t[1]);
                                    float v = lerp32(c0, c1, t[2]);
                                    if (v < 0.f) ++count;
                                }
                    }
                }
                out[lin(tr, p)] = (float)count / n3;
            }
}

static int alloc_field(fieldf *f, size_t n)
{
    free(f->data);
    f->data = (float *)malloc(n * sizeof(float));
    f->is_const = 0;
    return f->data ? 0 : 2;
}

/* cpp:748-766 (solid weights are never applied: getter-name mismatch) */
int orc_build_weights(orc_ctx *c, int n_super, int also_face_weights)
{
    if (n_super < 1 || n_super > 16) return 1;
    int r[3], o[3];
    cell_res(c, 0, r);
    if (alloc_field(&c->centerw, vol(r))) return 2;
    weights_for_lattice(c, OFF_CENTER, r, n_super, c->centerw.data);
    for (int a = 0; a < 3; ++a) {
        edge_res(c, 0, a, r);
        off_edge(a, o);
        if (alloc_field(&c->edgew[a], vol(r))) return 2;
        weights_for_lattice(c, o, r, n_super, c->edgew[a].data);
    }
    if (also_face_weights) {
        /* the FLIP solver's "surfaceweights" input (cpp:144); synthetic scenes use the same
         * super-sampled liquid fraction at face samples */
        for (int a = 0; a < 3; ++a) {
            face_res(c, 0, a, r);
            off_face(a, o);
            if (alloc_field(&c->facew[a], vol(r))) return 2;
            weights_for_lattice(c, o, r, n_super, c->facew[a].data);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* pre-pass 2: refinement mask (cpp:815-867) + octree (oct.cpp:4-243, 310-920)           */
/* ------------------------------------------------------------------------------------ */
int orc_build_octree(orc_ctx *c, double extrapolation_scale)
{
    int r0[3];
    cell_res(c, 0, r0);
    size_t n0 = vol(r0);
    const double dx = c->dx;
    const double extrapolation = dx * extrapolation_scale;         /* cpp:243 */
    const double inner = dx * fmax(2., 0.);                        /* cpp:259-261, A.5 item 4 */
    const double outer = 3. * dx;                                  /* cpp:262 */

    free(c->mask);
    c->mask = (int8_t *)malloc(n0);
    if (!c->mask) return 2;
    for (int k = 0; k < r0[2]; ++k)
        for (int j = 0; j < r0[1]; ++j)
            for (int i = 0; i < r0[0]; ++i) {
                int p[3] = {i, j, k};
                double sdf = (double)fget(&c->liquid, r0, p);
                int m;
                if (sdf > 0 && sdf < outer) m = 0;
                else if (sdf <= 0.) {
                    if (sdf > -inner) m = 0;
                    else {
                        /* solidSurface.getValue(cell centre) == the aligned voxel */
                        double s = (double)fget(&c->solid, r0, p);
                        m = (s > (-inner - extrapolation)) ? 0 : -1;
                    }
                } else m = 1;
                c->mask[lin(r0, p)] = (int8_t)m;
            }

    /* level count, oct.cpp:32-40 */
    int L = c->desired_levels;
    for (int a = 0; a < 3; ++a) {
        int lg = 0;
        while ((1 << (lg + 1)) <= c->n[a]) ++lg;
        if (lg < L) L = lg;
    }
    if (L < 1) L = 1;
    for (int l = 0; l < ORC_MAX_LEVELS; ++l) { free(c->labels[l]); c->labels[l] = NULL; }
    for (int l = 0; l < L; ++l) {
        int r[3];
        cell_res(c, l, r);
        c->labels[l] = (int8_t *)calloc(vol(r), 1); /* INACTIVE, oct.cpp:59,69 */
        if (!c->labels[l]) return 2;
    }
    /* base labels, oct.cpp:383-388 */
    for (size_t i = 0; i < n0; ++i)
        c->labels[0][i] = c->mask[i] == 0 ? ORC_ACTIVE : (c->mask[i] < 0 ? ORC_UP : ORC_INACTIVE);

    for (int l = 0; l < L - 1; ++l) {
        int r[3], rp[3];
        cell_res(c, l, r);
        cell_res(c, l + 1, rp);
        int8_t *lab = c->labels[l], *par = c->labels[l + 1];
        /* pass 1 (oct.cpp:395-565): UP with an ACTIVE sibling -> ACTIVE; ACTIVE -> parent DOWN.
         * Sibling groups never straddle 16^3 tiles, so the constant-tile shortcuts do not
         * change the result; flips are monotone so visiting order does not matter. */
        for (int k = 0; k < rp[2]; ++k)
            for (int j = 0; j < rp[1]; ++j)
                for (int i = 0; i < rp[0]; ++i) {
                    int any = 0;
                    for (int ci = 0; ci < 8; ++ci) {
                        int q[3] = {2 * i + (ci & 1), 2 * j + ((ci >> 1) & 1), 2 * k + ((ci >> 2) & 1)};
                        if (lab[lin(r, q)] == ORC_ACTIVE) any = 1;
                    }
                    if (!any) continue;
                    for (int ci = 0; ci < 8; ++ci) {
                        int q[3] = {2 * i + (ci & 1), 2 * j + ((ci >> 1) & 1), 2 * k + ((ci >> 2) & 1)};
                        if (lab[lin(r, q)] == ORC_UP) lab[lin(r, q)] = ORC_ACTIVE;
                    }
                    int pp[3] = {i, j, k};
                    par[lin(rp, pp)] = ORC_DOWN; /* oct.cpp:120 */
                }
        /* pass 2 (oct.cpp:657-754): DOWN list first, then ACTIVE list (oct.cpp:145,162) */
        for (int k = 0; k < r[2]; ++k)
            for (int j = 0; j < r[1]; ++j)
                for (int i = 0; i < r[0]; ++i) {
                    int p[3] = {i, j, k};
                    if (lab[lin(r, p)] == ORC_DOWN) {
                        int pp[3] = {i / 2, j / 2, k / 2};
                        par[lin(rp, pp)] = ORC_DOWN;
                    }
                }
        for (int k = 0; k < r[2]; ++k)
            for (int j = 0; j < r[1]; ++j)
                for (int i = 0; i < r[0]; ++i) {
                    int p[3] = {i, j, k};
                    if (lab[lin(r, p)] != ORC_ACTIVE) continue;
                    for (int a = 0; a < 3; ++a)
                        for (int d = 0; d < 2; ++d) {
                            int q[3] = {i, j, k};
                            q[a] += d ? 1 : -1;
                            if (q[a] < 0 || q[a] >= r[a]) continue;
                            if (lab[lin(r, q)] == ORC_UP) {
                                int pp[3] = {q[0] / 2, q[1] / 2, q[2] / 2};
                                par[lin(rp, pp)] = ORC_ACTIVE;
                            }
                        }
                }
        /* pass 3 (oct.cpp:757-840): UP with INACTIVE parent -> parent UP */
        for (int k = 0; k < r[2]; ++k)
            for (int j = 0; j < r[1]; ++j)
                for (int i = 0; i < r[0]; ++i) {
                    int p[3] = {i, j, k};
                    if (lab[lin(r, p)] != ORC_UP) continue;
                    int pp[3] = {i / 2, j / 2, k / 2};
                    if (par[lin(rp, pp)] == ORC_INACTIVE) par[lin(rp, pp)] = ORC_UP;
                }
    }
    /* top level, oct.cpp:843-875 */
    {
        int r[3];
        cell_res(c, L - 1, r);
        size_t n = vol(r);
        for (size_t i = 0; i < n; ++i)
            if (c->labels[L - 1][i] == ORC_UP) c->labels[L - 1][i] = ORC_ACTIVE;
    }
    /* cap at the first level without ACTIVE cells, oct.cpp:198-211 */
    int capped = 0;
    for (; capped < L; ++capped) {
        int r[3];
        cell_res(c, capped, r);
        size_t n = vol(r);
        int has = 0;
        for (size_t i = 0; i < n && !has; ++i) has = c->labels[capped][i] == ORC_ACTIVE;
        if (!has) break;
    }
    for (int l = capped; l < L; ++l) { free(c->labels[l]); c->labels[l] = NULL; }
    c->levels = capped;
    return capped > 0 ? 0 : 3;
}

/* ------------------------------------------------------------------------------------ */
/* pre-pass 3: classification + numbering (cpp:886-1715)                                 */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int tr[3]; /* tiles per axis */
    uint8_t *occ;
} tilemask;

static int tm_init(tilemask *t, const int r[3])
{
    for (int a = 0; a < 3; ++a) t->tr[a] = (r[a] + TILE - 1) / TILE;
    t->occ = (uint8_t *)calloc((size_t)t->tr[0] * t->tr[1] * t->tr[2], 1);
    return t->occ ? 0 : 2;
}
static inline void tm_mark(tilemask *t, const int p[3])
{
    t->occ[(size_t)(p[0] / TILE) + (size_t)t->tr[0] * ((size_t)(p[1] / TILE) + (size_t)t->tr[1] * (size_t)(p[2] / TILE))] = 1;
}
static inline int tm_get(const tilemask *t, const int p[3])
{
    return t->occ[(size_t)(p[0] / TILE) + (size_t)t->tr[0] * ((size_t)(p[1] / TILE) + (size_t)t->tr[1] * (size_t)(p[2] / TILE))];
}

/* serial sweep in HDK tile order (cpp:1566-1593): FLUID -> running counter; also records
 * dof -> (level, axis, i, j, k) */
static int64_t number_grid(int32_t *g, const int r[3], int64_t next)
{
    int tr[3];
    for (int a = 0; a < 3; ++a) tr[a] = (r[a] + TILE - 1) / TILE;
    for (int tz = 0; tz < tr[2]; ++tz)
        for (int ty = 0; ty < tr[1]; ++ty)
            for (int tx = 0; tx < tr[0]; ++tx) {
                int z1 = (tz + 1) * TILE < r[2] ? (tz + 1) * TILE : r[2];
                int y1 = (ty + 1) * TILE < r[1] ? (ty + 1) * TILE : r[1];
                int x1 = (tx + 1) * TILE < r[0] ? (tx + 1) * TILE : r[0];
                for (int k = tz * TILE; k < z1; ++k)
                    for (int j = ty * TILE; j < y1; ++j)
                        for (int i = tx * TILE; i < x1; ++i) {
                            size_t o = (size_t)i + (size_t)r[0] * ((size_t)j + (size_t)r[1] * (size_t)k);
                            if (g[o] == ORC_FLUID) g[o] = (int32_t)next++;
                        }
            }
    return next;
}

static int build_dof_table(int32_t **tab, int64_t n, int32_t *const *grids /*[levels*naxes]*/,
                           const orc_ctx *c, int kind)
{
    free(*tab);
    *tab = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * 4 * sizeof(int32_t));
    if (!*tab) return 2;
    int naxes = kind == ORC_I_CENTER ? 1 : 3;
    for (int l = 0; l < c->levels; ++l)
        for (int a = 0; a < naxes; ++a) {
            int r[3];
            if (kind == ORC_I_VELOCITY) face_res(c, l, a, r);
            else if (kind == ORC_I_EDGE) edge_res(c, l, a, r);
            else cell_res(c, l, r);
            const int32_t *g = grids[l * naxes + a];
            for (int k = 0; k < r[2]; ++k)
                for (int j = 0; j < r[1]; ++j)
                    for (int i = 0; i < r[0]; ++i) {
                        int32_t id = g[(size_t)i + (size_t)r[0] * ((size_t)j + (size_t)r[1] * (size_t)k)];
                        if (id >= 0) {
                            if (id >= n) return 4;
                            int32_t *t = *tab + 4 * (size_t)id;
                            t[0] = l | (a << 8);
                            t[1] = i; t[2] = j; t[3] = k;
                        }
                    }
        }
    return 0;
}

static int rebuild_dof_tables(orc_ctx *c)
{
    int32_t *gv[ORC_MAX_LEVELS * 3], *ge[ORC_MAX_LEVELS * 3], *gc[ORC_MAX_LEVELS];
    for (int l = 0; l < c->levels; ++l) {
        for (int a = 0; a < 3; ++a) {
            gv[l * 3 + a] = c->vidx[l][a];
            ge[l * 3 + a] = c->eidx[l][a];
        }
        gc[l] = c->cidx[l];
    }
    int rc;
    if ((rc = build_dof_table(&c->vdof, c->nvel, gv, c, ORC_I_VELOCITY))) return rc;
    if ((rc = build_dof_table(&c->edof, c->nedge, ge, c, ORC_I_EDGE))) return rc;
    if ((rc = build_dof_table(&c->cdof, c->ncenter, gc, c, ORC_I_CENTER))) return rc;
    return 0;
}

static int32_t *alloc_idx(const int r[3])
{
    size_t n = vol(r);
    int32_t *g = (int32_t *)malloc(n * sizeof(int32_t));
    if (g)
        for (size_t i = 0; i < n; ++i) g[i] = ORC_UNASSIGNED; /* makeConstant(HDK_UNASSIGNED) */
    return g;
}

int orc_build_indices(orc_ctx *c, double extrapolation_scale)
{
    if (!c->labels[0]) return 3;
    const int L = c->levels;
    const double extrapolation = c->dx * extrapolation_scale;
    free_indices(c);
    int r0[3];
    cell_res(c, 0, r0);
    const double occ_sdf = 2. * c->dx; /* cpp:907 */

    /* ---- velocity faces: cpp:1514-1561 + classifyOctreeVelocityFacesPartial cpp:1167-1323 */
    for (int l = 0; l < L; ++l) {
        int cr[3];
        cell_res(c, l, cr);
        const int8_t *lab = c->labels[l];
        for (int axis = 0; axis < 3; ++axis) {
            int fr[3];
            face_res(c, l, axis, fr);
            int32_t *g = c->vidx[l][axis] = alloc_idx(fr);
            if (!g) return 2;
            tilemask tm;
            if (tm_init(&tm, fr)) return 2;
            /* occupied tiles: cpp:887-943 (level 0: liquid SDF < 2dx), cpp:946-1000 (ACTIVE cells) */
            for (int k = 0; k < cr[2]; ++k)
                for (int j = 0; j < cr[1]; ++j)
                    for (int i = 0; i < cr[0]; ++i) {
                        int p[3] = {i, j, k};
                        int hit = (l == 0) ? ((double)fget(&c->liquid, r0, p) < occ_sdf)
                                           : (lab[lin(cr, p)] == ORC_ACTIVE);
                        if (!hit) continue;
                        for (int d = 0; d < 2; ++d) {
                            int f[3];
                            cell_to_face(p, axis, d, f);
                            tm_mark(&tm, f);
                        }
                    }
            int foff[3];
            off_face(axis, foff);
            for (int k = 0; k < fr[2]; ++k)
                for (int j = 0; j < fr[1]; ++j)
                    for (int i = 0; i < fr[0]; ++i) {
                        int f[3] = {i, j, k};
                        if (!tm_get(&tm, f)) continue; /* constant tile: stays UNASSIGNED (cpp:1197) */
                        int bc[3], fc[3];
                        face_to_cell(f, axis, 0, bc);
                        face_to_cell(f, axis, 1, fc);
                        size_t o = lin(fr, f);
                        if (bc[axis] < 0 || fc[axis] >= cr[axis]) { /* cpp:1210-1215 */
                            if (l == 0) g[o] = ORC_OUTSIDE;
                            continue;
                        }
                        int bl = lab[lin(cr, bc)], fl = lab[lin(cr, fc)];
                        if (l == 0) {
                            if (bl == ORC_ACTIVE && fl == ORC_ACTIVE) { /* cpp:1232-1272 */
                                int active = 0;
                                if (fget(&c->centerw, r0, bc) > 0.f || fget(&c->centerw, r0, fc) > 0.f)
                                    active = 1;
                                for (int ea = 0; ea < 3 && !active; ++ea) {
                                    if (ea == axis) continue;
                                    int er[3];
                                    edge_res(c, 0, ea, er);
                                    for (int d = 0; d < 2; ++d) {
                                        int e[3];
                                        face_to_edge(f, axis, ea, d, e);
                                        if (fget(&c->edgew[ea], er, e) > 0.f) { active = 1; break; }
                                    }
                                }
                                if (active) {
                                    int P2[3];
                                    pos2_face(0, axis, f, P2);
                                    float s = sample_f32(&c->solid, r0, OFF_CENTER, P2);
                                    g[o] = ((double)s > -extrapolation) ? ORC_SOLIDBOUNDARY : ORC_FLUID;
                                } else g[o] = ORC_OUTSIDE;
                            } else if (bl == ORC_INACTIVE || fl == ORC_INACTIVE) g[o] = ORC_OUTSIDE;
                            else if ((bl == ORC_UP && fl == ORC_ACTIVE) || (bl == ORC_ACTIVE && fl == ORC_UP))
                                g[o] = ORC_FLUID;
                        } else { /* cpp:1301-1319 */
                            if ((bl == ORC_ACTIVE && fl == ORC_ACTIVE) || (bl == ORC_UP && fl == ORC_ACTIVE) ||
                                (bl == ORC_ACTIVE && fl == ORC_UP))
                                g[o] = ORC_FLUID;
                        }
                    }
            free(tm.occ);
        }
    }
    /* serial numbering, cpp:1566-1593 */
    int64_t next = 0;
    for (int l = 0; l < L; ++l)
        for (int axis = 0; axis < 3; ++axis) {
            int fr[3];
            face_res(c, l, axis, fr);
            next = number_grid(c->vidx[l][axis], fr, next);
        }
    c->nvel = next;

    /* ---- edge stresses: cpp:1596-1663 + classifyEdgeStressesPartial cpp:1325-1405 */
    for (int l = 0; l < L; ++l) {
        int cr[3];
        cell_res(c, l, cr);
        const int8_t *lab = c->labels[l];
        for (int axis = 0; axis < 3; ++axis) {
            int er[3];
            edge_res(c, l, axis, er);
            int32_t *g = c->eidx[l][axis] = alloc_idx(er);
            if (!g) return 2;
            tilemask tm;
            if (tm_init(&tm, er)) return 2;
            for (int k = 0; k < cr[2]; ++k) /* cpp:1003-1057 */
                for (int j = 0; j < cr[1]; ++j)
                    for (int i = 0; i < cr[0]; ++i) {
                        int p[3] = {i, j, k};
                        if (lab[lin(cr, p)] != ORC_ACTIVE) continue;
                        for (int ei = 0; ei < 4; ++ei) {
                            int e[3];
                            cell_to_edge(p, axis, ei, e);
                            tm_mark(&tm, e);
                        }
                    }
            for (int k = 0; k < er[2]; ++k)
                for (int j = 0; j < er[1]; ++j)
                    for (int i = 0; i < er[0]; ++i) {
                        int e[3] = {i, j, k};
                        if (!tm_get(&tm, e)) continue;
                        size_t o = lin(er, e);
                        int active = 0;
                        for (int ci = 0; ci < 4; ++ci) { /* cpp:1361-1380 */
                            int q[3];
                            edge_to_cell(e, axis, ci, q);
                            if (q[0] < 0 || q[1] < 0 || q[2] < 0 || q[0] >= cr[0] || q[1] >= cr[1] || q[2] >= cr[2]) {
                                g[o] = ORC_OUTSIDE;
                                break; /* NB: isStressActive keeps whatever it was (cpp:1368-1369) */
                            }
                            int lb = lab[lin(cr, q)];
                            if (lb == ORC_DOWN) { active = 0; break; }
                            else if (lb == ORC_ACTIVE) active = 1;
                        }
                        if (active) {
                            if (l == 0) g[o] = (fget(&c->edgew[axis], er, e) > 0.f) ? ORC_FLUID : ORC_OUTSIDE;
                            else g[o] = ORC_FLUID;
                        }
                    }
            free(tm.occ);
        }
    }
    next = 0;
    for (int l = 0; l < L; ++l)
        for (int axis = 0; axis < 3; ++axis) {
            int er[3];
            edge_res(c, l, axis, er);
            next = number_grid(c->eidx[l][axis], er, next);
        }
    c->nedge = next;

    /* ---- centre stresses: cpp:1665-1715 + classifyCenterStressesPartial cpp:1407-1443 */
    next = 0;
    for (int l = 0; l < L; ++l) {
        int cr[3];
        cell_res(c, l, cr);
        int32_t *g = c->cidx[l] = alloc_idx(cr);
        if (!g) return 2;
        size_t n = vol(cr);
        for (size_t o = 0; o < n; ++o)
            if (c->labels[l][o] == ORC_ACTIVE &&
                (l != 0 || (c->centerw.is_const ? c->centerw.cval : c->centerw.data[o]) > 0.f))
                g[o] = ORC_FLUID;
        next = number_grid(g, cr, next);
    }
    c->ncenter = next;
    return rebuild_dof_tables(c);
}

/* ---- direct setters ----------------------------------------------------------------- */
int orc_set_levels(orc_ctx *c, int levels)
{
    if (levels < 1 || levels > ORC_MAX_LEVELS) return 1;
    for (int a = 0; a < 3; ++a)
        if ((c->n[a] >> (levels - 1)) < 1) return 1;
    c->levels = levels;
    return 0;
}
int orc_set_labels(orc_ctx *c, int level, const int8_t *labels)
{
    if (level < 0 || level >= c->levels) return 1;
    int r[3];
    cell_res(c, level, r);
    free(c->labels[level]);
    c->labels[level] = (int8_t *)malloc(vol(r));
    if (!c->labels[level]) return 2;
    memcpy(c->labels[level], labels, vol(r));
    return 0;
}
int orc_set_index(orc_ctx *c, int kind, int level, int axis, const int32_t *idx)
{
    if (level < 0 || level >= c->levels || axis < 0 || axis > 2) return 1;
    int r[3];
    int32_t **slot;
    if (kind == ORC_I_VELOCITY) { face_res(c, level, axis, r); slot = &c->vidx[level][axis]; }
    else if (kind == ORC_I_EDGE) { edge_res(c, level, axis, r); slot = &c->eidx[level][axis]; }
    else if (kind == ORC_I_CENTER) { cell_res(c, level, r); slot = &c->cidx[level]; }
    else return 1;
    free(*slot);
    *slot = (int32_t *)malloc(vol(r) * sizeof(int32_t));
    if (!*slot) return 2;
    memcpy(*slot, idx, vol(r) * sizeof(int32_t));
    return 0;
}
static int64_t max_plus_one(const int32_t *g, size_t n)
{
    int64_t m = 0;
    for (size_t i = 0; i < n; ++i)
        if (g[i] >= m) m = (int64_t)g[i] + 1;
    return m;
}
int orc_finalize_indices(orc_ctx *c)
{
    int64_t nv = 0, ne = 0, nc = 0;
    for (int l = 0; l < c->levels; ++l) {
        int r[3];
        for (int a = 0; a < 3; ++a) {
            if (!c->vidx[l][a] || !c->eidx[l][a]) return 3;
            face_res(c, l, a, r);
            int64_t m = max_plus_one(c->vidx[l][a], vol(r));
            if (m > nv) nv = m;
            edge_res(c, l, a, r);
            m = max_plus_one(c->eidx[l][a], vol(r));
            if (m > ne) ne = m;
        }
        if (!c->cidx[l] || !c->labels[l]) return 3;
        cell_res(c, l, r);
        int64_t m = max_plus_one(c->cidx[l], vol(r));
        if (m > nc) nc = m;
    }
    c->nvel = nv; c->nedge = ne; c->ncenter = nc;
    return rebuild_dof_tables(c);
}

int orc_levels(orc_ctx *c) { return c->levels; }
int64_t orc_grid_size(orc_ctx *c, int kind, int level, int axis, int res_out[3])
{
    int r[3];
    if (kind == ORC_I_VELOCITY) face_res(c, level, axis, r);
    else if (kind == ORC_I_EDGE) edge_res(c, level, axis, r);
    else cell_res(c, level, r);
    if (res_out) { res_out[0] = r[0]; res_out[1] = r[1]; res_out[2] = r[2]; }
    return (int64_t)vol(r);
}
int orc_get_labels(orc_ctx *c, int level, int8_t *out)
{
    if (level < 0 || level >= c->levels || !c->labels[level]) return 1;
    int r[3];
    cell_res(c, level, r);
    memcpy(out, c->labels[level], vol(r));
    return 0;
}
int orc_get_mask(orc_ctx *c, int8_t *out)
{
    if (!c->mask) return 1;
    int r[3];
    cell_res(c, 0, r);
    memcpy(out, c->mask, vol(r));
    return 0;
}
int orc_get_index(orc_ctx *c, int kind, int level, int axis, int32_t *out)
{
    if (level < 0 || level >= c->levels) return 1;
    int r[3];
    const int32_t *g;
    if (kind == ORC_I_VELOCITY) { face_res(c, level, axis, r); g = c->vidx[level][axis]; }
    else if (kind == ORC_I_EDGE) { edge_res(c, level, axis, r); g = c->eidx[level][axis]; }
    else { cell_res(c, level, r); g = c->cidx[level]; }
    if (!g) return 1;
    memcpy(out, g, vol(r) * sizeof(int32_t));
    return 0;
}
int64_t orc_count(orc_ctx *c, int kind)
{
    return kind == ORC_I_VELOCITY ? c->nvel : (kind == ORC_I_EDGE ? c->nedge : c->ncenter);
}
int orc_get_dof_table(orc_ctx *c, int kind, int32_t *out)
{
    const int32_t *t = kind == ORC_I_VELOCITY ? c->vdof : (kind == ORC_I_EDGE ? c->edof : c->cdof);
    int64_t n = orc_count(c, kind);
    if (!t) return 1;
    memcpy(out, t, (size_t)n * 4 * sizeof(int32_t));
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* hot path part 1: stress stencils                                                     */
/* ------------------------------------------------------------------------------------ */
static inline int32_t vget(const orc_ctx *c, int l, int a, const int f[3])
{
    int r[3];
    face_res(c, l, a, r);
    return c->vidx[l][a][lin(r, f)];
}
static inline int32_t eget(const orc_ctx *c, int l, int a, const int e[3])
{
    int r[3];
    edge_res(c, l, a, r);
    return c->eidx[l][a][lin(r, e)];
}
static inline int lget(const orc_ctx *c, int l, const int p[3])
{
    int r[3];
    cell_res(c, l, r);
    return c->labels[l][lin(r, p)];
}

typedef struct {
    int cnt, bcnt;
    int32_t idx[ORC_EDGE_CAP];
    double coef[ORC_EDGE_CAP];
    double bval[ORC_EDGE_BCAP];
    int overflow;
} stencil;

static inline void st_push(stencil *s, int32_t id, double co)
{
    if (s->cnt < ORC_EDGE_CAP) { s->idx[s->cnt] = id; s->coef[s->cnt] = co; s->cnt++; }
    else s->overflow = 1;
}

/* getEdgeStressFaces, cpp:1717-1908 */
static void edge_stress_faces(const orc_ctx *c, int level, int axis, const int edge[3], stencil *s)
{
    s->cnt = s->bcnt = 0;
    s->overflow = 0;
    const double dx = c->dx * (double)(1 << level); /* cpp:1733 */
    int atTransition[3] = {0, 0, 0}, faceOutside[3] = {0, 0, 0};
    float gdx[3] = {0.f, 0.f, 0.f}; /* UT_Vector3 is fp32, cpp:1738 */

    for (int fa = 0; fa < 3; ++fa) { /* cpp:1740-1787 */
        if (fa == axis) continue;
        int fr[3];
        face_res(c, level, fa, fr);
        const int ga = 3 - fa - axis;
        for (int dir = 0; dir < 2; ++dir) {
            int face[3];
            edge_to_face(edge, axis, fa, dir, face);
            if (face[ga] < 0 || face[ga] >= fr[ga]) {
                gdx[ga] = (float)((double)gdx[ga] + .5 * dx);
                faceOutside[ga] = 1;
            } else {
                int32_t vi = c->vidx[level][fa][lin(fr, face)];
                if (vi >= 0) gdx[ga] = (float)((double)gdx[ga] + .5 * dx);
                else if (vi == ORC_OUTSIDE || vi == ORC_SOLIDBOUNDARY) {
                    gdx[ga] = (float)((double)gdx[ga] + .5 * dx);
                    faceOutside[ga] = 1;
                } else { /* UNASSIGNED */
                    gdx[ga] = (float)((double)gdx[ga] + dx);
                    if (c->enhanced) atTransition[ga] = 1;
                }
            }
        }
    }
    for (int fa = 0; fa < 3; ++fa) { /* cpp:1789-1907 */
        if (fa == axis) continue;
        int fr[3];
        face_res(c, level, fa, fr);
        const int ga = 3 - fa - axis;
        for (int dir = 0; dir < 2; ++dir) {
            int face[3];
            edge_to_face(edge, axis, fa, dir, face);
            const double sign = (dir == 0) ? -1 : 1;
            if (face[ga] < 0 || face[ga] >= fr[ga]) continue;
            const int32_t vi = c->vidx[level][fa][lin(fr, face)];
            const double g = (double)gdx[ga];
            if (vi >= 0) {
                if (atTransition[ga] && !faceOutside[ga]) { /* cpp:1814-1824 */
                    int sib[3] = {face[0], face[1], face[2]};
                    sib[axis] += (edge[axis] % 2 == 0) ? 1 : -1;
                    int32_t si = c->vidx[level][fa][lin(fr, sib)];
                    st_push(s, si, .25 * sign / g);
                    st_push(s, vi, .25 * sign / g);
                } else st_push(s, vi, .5 * sign / g); /* cpp:1827 */
            } else if (vi == ORC_UNASSIGNED) {
                if (level + 1 >= c->levels) { /* the reference would index past its level array here (cpp:1853, 1888) */
                    s->overflow = 2;
                    continue;
                }
                if (edge[fa] % 2 != 0) { /* dangling edge, cpp:1835-1884 */
                    for (int oi = 0; oi < 2; ++oi) {
                        int offs = oi == 0 ? -1 : 1;
                        int of[3] = {face[0], face[1], face[2]};
                        of[fa] += offs;
                        int pf[3] = {of[0] / 2, of[1] / 2, of[2] / 2};
                        int32_t pi = vget(c, level + 1, fa, pf);
                        if (pi >= 0) st_push(s, pi, .25 * sign / g);
                        else if (pi == ORC_UNASSIGNED) {
                            for (int ci = 0; ci < 4; ++ci) {
                                int cf[3];
                                child_face(pf, fa, ci, cf);
                                int32_t cvi = c->vidx[level][fa][lin(fr, cf)];
                                if (cvi >= 0) st_push(s, cvi, .0625 * sign / g);
                                else s->overflow = 2; /* reference: assert(false) cpp:1878 */
                            }
                        }
                    }
                } else { /* cpp:1886-1894 */
                    int pf[3] = {face[0] / 2, face[1] / 2, face[2] / 2};
                    int32_t pi = vget(c, level + 1, fa, pf);
                    if (pi < 0) s->overflow = 2; /* reference: assert cpp:1891 */
                    st_push(s, pi, .5 * sign / g);
                }
            } else if (vi == ORC_SOLIDBOUNDARY) { /* cpp:1896-1905; quirk A.5.1: component = edge axis */
                int P2[3], off[3], vr[3];
                pos2_face(level, fa, face, P2);
                off_face(axis, off);
                face_res(c, 0, axis, vr);
                double lv = (double)sample_f32(&c->solidvel[axis], vr, off, P2);
                if (s->bcnt < ORC_EDGE_BCAP) s->bval[s->bcnt++] = .5 * sign * lv / g;
            }
        }
    }
}

/* edgeOctreeVolumes, cpp:2004-2057 */
static double edge_octree_volume(const orc_ctx *c, int level, int axis, const int edge[3])
{
    const double dx = (double)(1 << level);
    float vdx[3] = {0.f, 0.f, 0.f};
    vdx[axis] = (float)dx;
    for (int fa = 0; fa < 3; ++fa) {
        if (fa == axis) continue;
        int fr[3];
        face_res(c, level, fa, fr);
        const int ga = 3 - fa - axis;
        for (int dir = 0; dir < 2; ++dir) {
            int face[3];
            edge_to_face(edge, axis, fa, dir, face);
            if (face[ga] < 0 || face[ga] >= fr[ga]) vdx[ga] = (float)((double)vdx[ga] + .5 * dx);
            else {
                int32_t vi = c->vidx[level][fa][lin(fr, face)];
                if (vi >= 0 || vi == ORC_OUTSIDE || vi == ORC_SOLIDBOUNDARY)
                    vdx[ga] = (float)((double)vdx[ga] + .5 * dx);
                else vdx[ga] = (float)((double)vdx[ga] + dx);
            }
        }
    }
    float v = vdx[0] * vdx[1];
    v = v * vdx[2];
    return (double)v;
}

/* getCenterStressFaces, cpp:1910-1963 */
static void center_stress_faces(const orc_ctx *c, int level, int axis, const int cell[3], stencil *s)
{
    s->cnt = s->bcnt = 0;
    s->overflow = 0;
    const double dx = c->dx * (double)(1 << level); /* cpp:1923 */
    for (int dir = 0; dir < 2; ++dir) {
        int face[3];
        cell_to_face(cell, axis, dir, face);
        const double sign = (dir == 0) ? -1 : 1;
        const int32_t vi = vget(c, level, axis, face);
        if (vi >= 0) st_push(s, vi, sign / dx);
        else if (vi == ORC_UNASSIGNED) {
            for (int ci = 0; ci < 4; ++ci) {
                if (level == 0) { s->overflow = 2; break; } /* reference: assert(level > 0) */
                int cf[3];
                child_face(face, axis, ci, cf);
                int32_t cvi = vget(c, level - 1, axis, cf);
                if (cvi < 0) s->overflow = 2;
                st_push(s, cvi, .25 * sign / dx);
            }
        } else if (vi == ORC_SOLIDBOUNDARY) {
            int P2[3], off[3], vr[3];
            pos2_face(level, axis, face, P2);
            off_face(axis, off);
            face_res(c, 0, axis, vr);
            double lv = (double)sample_f32(&c->solidvel[axis], vr, off, P2);
            if (s->bcnt < ORC_CENTER_BCAP) s->bval[s->bcnt++] = sign * lv / dx;
        }
    }
}

int orc_build_stencils(orc_ctx *c)
{
    if (!c->vdof || !c->edof || !c->cdof) return 3;
    free_stencils(c);
    const int64_t ne = c->nedge, nc = c->ncenter;
    const size_t se = (size_t)(ne > 0 ? ne : 1), sc = (size_t)(nc > 0 ? nc : 1);
    c->e_cnt = (int32_t *)calloc(se, sizeof(int32_t));
    c->e_bcnt = (int32_t *)calloc(se, sizeof(int32_t));
    c->e_idx = (int32_t *)malloc(se * ORC_EDGE_CAP * sizeof(int32_t));
    c->e_coef = (double *)calloc(se * ORC_EDGE_CAP, sizeof(double));
    c->e_bval = (double *)calloc(se * ORC_EDGE_BCAP, sizeof(double));
    c->e_w = (double *)calloc(se, sizeof(double));
    c->c_cnt = (int32_t *)calloc(sc * 3, sizeof(int32_t));
    c->c_bcnt = (int32_t *)calloc(sc * 3, sizeof(int32_t));
    c->c_idx = (int32_t *)malloc(sc * 3 * ORC_CENTER_CAP * sizeof(int32_t));
    c->c_coef = (double *)calloc(sc * 3 * ORC_CENTER_CAP, sizeof(double));
    c->c_bval = (double *)calloc(sc * 3 * ORC_CENTER_BCAP, sizeof(double));
    c->c_w = (double *)calloc(sc, sizeof(double));
    if (!c->e_cnt || !c->e_bcnt || !c->e_idx || !c->e_coef || !c->e_bval || !c->e_w || !c->c_cnt ||
        !c->c_bcnt || !c->c_idx || !c->c_coef || !c->c_bval || !c->c_w)
        return 2;
    memset(c->e_idx, 0xff, se * ORC_EDGE_CAP * sizeof(int32_t));
    memset(c->c_idx, 0xff, sc * 3 * ORC_CENTER_CAP * sizeof(int32_t));
    int bad = 0;
    int r0[3];
    cell_res(c, 0, r0);

    /* buildEdgeStressStencilsPartial, cpp:2059-2160 */
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t id = 0; id < ne; ++id) {
        const int32_t *t = c->edof + 4 * id;
        const int level = t[0] & 0xff, axis = t[0] >> 8;
        const int edge[3] = {t[1], t[2], t[3]};
        stencil s;
        edge_stress_faces(c, level, axis, edge, &s);
        bad |= s.overflow;
        c->e_cnt[id] = s.cnt;
        c->e_bcnt[id] = s.bcnt;
        for (int k = 0; k < s.cnt; ++k) {
            c->e_idx[(size_t)k * ne + id] = s.idx[k];
            c->e_coef[(size_t)k * ne + id] = s.coef[k];
        }
        for (int k = 0; k < s.bcnt; ++k) c->e_bval[(size_t)k * ne + id] = s.bval[k];
        double w;
        if (level == 0) { /* cpp:2125-2138 */
            int er[3];
            edge_res(c, 0, axis, er);
            w = (double)fget(&c->edgew[axis], er, edge);
            if (w == 1.) w = edge_octree_volume(c, level, axis, edge);
        } else w = edge_octree_volume(c, level, axis, edge);
        if (c->visc.is_const) w *= (double)c->visc.cval; /* cpp:2144-2145 */
        else {
            int P2[3];
            pos2_edge(level, axis, edge, P2);
            w *= (double)sample_f32(&c->visc, r0, OFF_CENTER, P2); /* cpp:2148-2150 */
        }
        c->e_w[id] = 4. * c->dt * w; /* cpp:2155 */
    }

    /* buildCenterStressStencilsPartial cpp:2162-2221, buildCenterStressWeightsPartial cpp:2223-2289 */
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t id = 0; id < nc; ++id) {
        const int32_t *t = c->cdof + 4 * id;
        const int level = t[0] & 0xff;
        const int cell[3] = {t[1], t[2], t[3]};
        for (int axis = 0; axis < 3; ++axis) {
            stencil s;
            center_stress_faces(c, level, axis, cell, &s);
            bad |= s.overflow;
            const size_t sid = (size_t)id + (size_t)nc * axis; /* cpp:2186, 2207 */
            const size_t n3 = (size_t)nc * 3;
            c->c_cnt[sid] = s.cnt;
            c->c_bcnt[sid] = s.bcnt;
            for (int k = 0; k < s.cnt && k < ORC_CENTER_CAP; ++k) {
                c->c_idx[(size_t)k * n3 + sid] = s.idx[k];
                c->c_coef[(size_t)k * n3 + sid] = s.coef[k];
            }
            for (int k = 0; k < s.bcnt; ++k) c->c_bval[(size_t)k * n3 + sid] = s.bval[k];
        }
        double w;
        if (level == 0) w = (double)fget(&c->centerw, r0, cell); /* cpp:2271-2272 */
        else {
            double d = (double)(1 << level);
            w = d * d * d; /* cpp:2241-2245 */
        }
        if (c->visc.is_const) w *= (double)c->visc.cval;
        else {
            int P2[3];
            pos2_center(level, cell, P2);
            w *= (double)sample_f32(&c->visc, r0, OFF_CENTER, P2);
        }
        c->c_w[id] = 2. * c->dt * w; /* cpp:2284 */
    }
    return bad ? 10 + bad : 0;
}

/* ------------------------------------------------------------------------------------ */
/* hot path: initial guess by restriction, buildVelocityMappingPartial cpp:2291-2402     */
/* ------------------------------------------------------------------------------------ */
static void restrict_rec(const orc_ctx *c, int axis, const int face[3], float weight, int level,
                         const int vr[3], double *acc)
{
    if (level == 0) {
        int f[3]; /* the reference reads out of bounds here (A.5 item 9); the oracle clamps */
        for (int a = 0; a < 3; ++a) f[a] = clampi(face[a], 0, vr[a] - 1);
        double lw = (double)weight; /* const fpreal localWeight, cpp:2355 */
        *acc += lw * (double)fget(&c->vel[axis], vr, f); /* cpp:2373 */
        return;
    }
    static const double inAxis[3] = {1. / 16., 1. / 8., 1. / 16.}; /* cpp:2323 */
    for (int ci = 0; ci < 4; ++ci) {
        int cf[3];
        child_face(face, axis, ci, cf);
        for (int o = -1; o < 2; ++o) {
            int af[3] = {cf[0], cf[1], cf[2]};
            af[axis] += o;
            float w = (float)(inAxis[o + 1] * (double)weight); /* fpreal32 myWeight, cpp:2318, 2390 */
            restrict_rec(c, axis, af, w, level - 1, vr, acc);
        }
    }
}

int orc_build_initial_guess(orc_ctx *c)
{
    if (!c->vdof) return 3;
    free(c->x0);
    c->x0 = (double *)calloc((size_t)(c->nvel > 0 ? c->nvel : 1), sizeof(double));
    if (!c->x0) return 2;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t id = 0; id < c->nvel; ++id) {
        const int32_t *t = c->vdof + 4 * id;
        const int level = t[0] & 0xff, axis = t[0] >> 8;
        const int face[3] = {t[1], t[2], t[3]};
        int vr[3];
        face_res(c, 0, axis, vr);
        double acc = 0;
        /* depth-first recursion visits the leaves in the same order as the reference's FIFO
         * queue pops them (both are lexicographic in (child, offset) per level) */
        restrict_rec(c, axis, face, 1.f, level, vr, &acc);
        c->x0[id] = c->f32 ? (double)(float)acc : acc; /* initialGuess(octreeFaceIndex) = restrictedVelocity (a VectorXf element), cpp:2371 */
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* hot path part 2: row assembly, buildOctreeSystemFromStencilsPartial cpp:2459-2777     */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int64_t n;        /* entries written (or counted) */
    int32_t *col;     /* NULL on the counting pass */
    double *val;
    double diag, rhs;
    int bad;
    int f32; /* Eigen::Triplet<float> / VectorXf semantics */
} rowacc;
static inline double solve_type(double x, int f32) { return f32 ? (double)(float)x : x; }

/* applyToMatrix, cpp:2404-2457 */
static void apply_to_matrix(rowacc *ra, double coefficient, int32_t vi, int cnt, const int32_t *idx,
                            const double *coef, size_t stride, int bcnt, const double *bval,
                            size_t bstride)
{
    int found = 0;
    for (int i = 0; i < cnt; ++i)
        if (idx[(size_t)i * stride] == vi) {
            coefficient *= coef[(size_t)i * stride];
            found = 1;
            break;
        }
    if (!found) ra->bad = 1; /* reference: assert(foundSelf) cpp:2436 */
    for (int i = 0; i < cnt; ++i) {
        double element = coefficient * coef[(size_t)i * stride];
        if (idx[(size_t)i * stride] == vi) ra->diag += element;
        else {
            if (ra->col) {
                ra->col[ra->n] = idx[(size_t)i * stride];
                ra->val[ra->n] = solve_type(element, ra->f32); /* Eigen::Triplet<SolveType>(row, col, element), cpp:2447 */
            }
            ra->n++;
        }
    }
    for (int i = 0; i < bcnt; ++i) ra->rhs = solve_type(ra->rhs - coefficient * bval[(size_t)i * bstride], ra->f32); /* cpp:2456 */
}

static inline void apply_edge(const orc_ctx *c, rowacc *ra, int32_t vi, int32_t eid)
{
    const size_t ne = (size_t)c->nedge;
    apply_to_matrix(ra, c->e_w[eid], vi, c->e_cnt[eid], c->e_idx + eid, c->e_coef + eid, ne,
                    c->e_bcnt[eid], c->e_bval + eid, ne);
}
static inline void apply_center(const orc_ctx *c, rowacc *ra, int32_t vi, int32_t cid, int axis)
{
    const size_t n3 = (size_t)c->ncenter * 3;
    const size_t sid = (size_t)cid + (size_t)c->ncenter * axis;
    int cnt = c->c_cnt[sid];
    if (cnt > ORC_CENTER_CAP) cnt = ORC_CENTER_CAP;
    apply_to_matrix(ra, c->c_w[cid], vi, cnt, c->c_idx + sid, c->c_coef + sid, n3, c->c_bcnt[sid],
                    c->c_bval + sid, n3);
}

/* faceOctreeVolumes, cpp:1965-2002 */
static double face_octree_volume(const orc_ctx *c, int level, int axis, const int face[3], int *bad)
{
    int cr[3];
    cell_res(c, level, cr);
    const double dx = (double)(1 << level);
    double g = 0;
    for (int dir = 0; dir < 2; ++dir) {
        int cell[3];
        face_to_cell(face, axis, dir, cell);
        if (cell[axis] < 0 || cell[axis] >= cr[axis]) g += .5 * dx;
        else {
            int lb = c->labels[level][lin(cr, cell)];
            if (lb == ORC_ACTIVE || lb == ORC_INACTIVE) g += .5 * dx;
            else {
                int pc[3] = {cell[0] / 2, cell[1] / 2, cell[2] / 2};
                if (level + 1 < c->levels && lget(c, level + 1, pc) == ORC_ACTIVE) g += dx;
                else *bad = 1; /* reference: assert(false) cpp:1996 */
            }
        }
    }
    return dx * dx * g;
}

static void assemble_row(const orc_ctx *c, int32_t vi, rowacc *ra)
{
    const int32_t *t = c->vdof + 4 * (size_t)vi;
    const int level = t[0] & 0xff, axis = t[0] >> 8;
    const int face[3] = {t[1], t[2], t[3]};
    int cr[3], fr[3];
    cell_res(c, level, cr);
    face_res(c, level, axis, fr);

    for (int dir = 0; dir < 2; ++dir) { /* cpp:2547-2650 */
        int cell[3];
        face_to_cell(face, axis, dir, cell);
        if (cell[axis] < 0 || cell[axis] >= cr[axis]) continue;
        int sc[3], sl;
        if (c->labels[level][lin(cr, cell)] == ORC_ACTIVE) {
            sc[0] = cell[0]; sc[1] = cell[1]; sc[2] = cell[2];
            sl = level;
        } else {
            sc[0] = cell[0] / 2; sc[1] = cell[1] / 2; sc[2] = cell[2] / 2;
            sl = level + 1;
            if (sl >= c->levels) { ra->bad = 1; continue; }
        }
        {
            int scr[3];
            cell_res(c, sl, scr);
            int32_t ci = c->cidx[sl][lin(scr, sc)];
            if (ci >= 0) apply_center(c, ra, vi, ci, axis); /* cpp:2576-2599 */
        }
        for (int fa = 0; fa < 3; ++fa) { /* cpp:2614-2649 */
            if (fa == axis) continue;
            for (int fd = 0; fd < 2; ++fd) {
                int af[3];
                cell_to_face(sc, fa, fd, af);
                if (vget(c, sl, fa, af) == ORC_UNASSIGNED) {
                    const int ea = 3 - fa - axis;
                    if (sl == 0) { ra->bad = 1; continue; } /* reference indexes level -1 here */
                    for (int ii = 0; ii < 2; ++ii) {
                        int e[3];
                        child_edge_in_face(af, fa, ea, ii, e);
                        int32_t ei = eget(c, sl - 1, ea, e);
                        if (ei >= 0) apply_edge(c, ra, vi, ei);
                    }
                }
            }
        }
    }
    for (int ea = 0; ea < 3; ++ea) { /* cpp:2652-2745 */
        if (ea == axis) continue;
        for (int dir = 0; dir < 2; ++dir) {
            int e[3];
            face_to_edge(face, axis, ea, dir, e);
            const int32_t ei = eget(c, level, ea, e);
            if (ei >= 0) {
                if (c->enhanced) { /* cpp:2664-2697 */
                    const int ta = 3 - ea - axis;
                    int af[3] = {face[0], face[1], face[2]};
                    af[ta] += (dir == 0) ? -1 : 1;
                    if (af[ta] >= 0 && af[ta] < fr[ta]) {
                        if (c->vidx[level][axis][lin(fr, af)] == ORC_UNASSIGNED) {
                            int se[3] = {e[0], e[1], e[2]};
                            se[ea] += (e[ea] % 2 == 0) ? 1 : -1;
                            int32_t sei = eget(c, level, ea, se);
                            if (sei < 0) ra->bad = 1; /* assert cpp:2680 */
                            else apply_edge(c, ra, vi, sei);
                        }
                    }
                }
                apply_edge(c, ra, vi, ei); /* cpp:2698-2712 */
            } else if (ei == ORC_UNASSIGNED) { /* cpp:2714-2742 */
                if (level == 0) continue; /* reference: assert(level > 0); unoccupied-tile quirk A.1 */
                for (int ci = 0; ci < 2; ++ci) {
                    int ce[3];
                    child_edge(e, ea, ci, ce);
                    int32_t cei = eget(c, level - 1, ea, ce);
                    if (cei >= 0) apply_edge(c, ra, vi, cei);
                }
            }
        }
    }
    /* mass term, cpp:2748-2772 */
    double fw;
    int bad = 0;
    if (level == 0) {
        int vr[3];
        face_res(c, 0, axis, vr);
        fw = (double)fget(&c->facew[axis], vr, face);
        if (fw == 1.) fw = face_octree_volume(c, level, axis, face, &bad);
    } else fw = face_octree_volume(c, level, axis, face, &bad);
    if (bad) ra->bad = 1;
    if (c->dens.is_const) fw *= (double)c->dens.cval;
    else {
        int P2[3], r0[3];
        cell_res(c, 0, r0);
        pos2_face(level, axis, face, P2);
        fw *= (double)sample_f32(&c->dens, r0, OFF_CENTER, P2);
    }
    if (ra->col) {
        ra->col[ra->n] = vi;
        ra->val[ra->n] = solve_type(fw + ra->diag, ra->f32); /* cpp:2768 */
    }
    ra->n++;
    ra->rhs = solve_type(ra->rhs + fw * c->x0[vi], ra->f32); /* cpp:2772 */
}

/* Eigen::SparseMatrix::setFromTriplets (cpp:613-614; Eigen is not vendored -- any 3.3/3.4):
 * duplicates of one (row, col) are summed left to right in triplet order, then each column is
 * sorted by row.  A is stored column-major in Eigen; CSR of the same matrix is built here. */
static int64_t compress_row(int64_t n, int32_t *col, double *val, int f32)
{
    /* stable insertion sort by column */
    for (int64_t i = 1; i < n; ++i) {
        int32_t cc = col[i];
        double vv = val[i];
        int64_t j = i - 1;
        while (j >= 0 && col[j] > cc) {
            col[j + 1] = col[j];
            val[j + 1] = val[j];
            --j;
        }
        col[j + 1] = cc;
        val[j + 1] = vv;
    }
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (m > 0 && col[m - 1] == col[i]) val[m - 1] = f32 ? (double)((float)val[m - 1] + (float)val[i]) : val[m - 1] + val[i];
        else {
            col[m] = col[i];
            val[m] = val[i];
            ++m;
        }
    }
    return m;
}

int orc_assemble(orc_ctx *c)
{
    if (!c->e_cnt || !c->x0) return 3;
    const int64_t n = c->nvel;
    free(c->rhs); free(c->row_ptr); free(c->col); free(c->val);
    c->rhs = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    c->row_ptr = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    c->col = NULL; c->val = NULL;
    if (!c->rhs || !c->row_ptr) return 2;
    int64_t *rawptr = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    if (!rawptr) return 2;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t i = 0; i < n; ++i) {
        rowacc ra = {0, NULL, NULL, 0., 0., 0, c->f32};
        assemble_row(c, (int32_t)i, &ra);
        rawptr[i + 1] = ra.n;
        bad |= ra.bad;
    }
    for (int64_t i = 0; i < n; ++i) rawptr[i + 1] += rawptr[i];
    const int64_t nraw = rawptr[n];
    c->nraw = nraw;
    int32_t *rc = (int32_t *)malloc((size_t)(nraw > 0 ? nraw : 1) * sizeof(int32_t));
    double *rv = (double *)malloc((size_t)(nraw > 0 ? nraw : 1) * sizeof(double));
    if (!rc || !rv) { free(rawptr); free(rc); free(rv); return 2; }
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t i = 0; i < n; ++i) {
        rowacc ra = {0, rc + rawptr[i], rv + rawptr[i], 0., 0., 0, c->f32};
        assemble_row(c, (int32_t)i, &ra);
        c->rhs[i] = ra.rhs;
        c->row_ptr[i + 1] = compress_row(ra.n, rc + rawptr[i], rv + rawptr[i], c->f32);
        bad |= ra.bad;
    }
    for (int64_t i = 0; i < n; ++i) c->row_ptr[i + 1] += c->row_ptr[i];
    c->nnz = c->row_ptr[n];
    c->col = (int32_t *)malloc((size_t)(c->nnz > 0 ? c->nnz : 1) * sizeof(int32_t));
    c->val = (double *)malloc((size_t)(c->nnz > 0 ? c->nnz : 1) * sizeof(double));
    if (!c->col || !c->val) { free(rawptr); free(rc); free(rv); return 2; }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        int64_t m = c->row_ptr[i + 1] - c->row_ptr[i];
        memcpy(c->col + c->row_ptr[i], rc + rawptr[i], (size_t)m * sizeof(int32_t));
        memcpy(c->val + c->row_ptr[i], rv + rawptr[i], (size_t)m * sizeof(double));
    }
    free(rawptr); free(rc); free(rv);
    return bad ? 10 : 0;
}

int orc_get_edge_stencils(orc_ctx *c, int32_t *cnt, int32_t *idx, double *coef, int32_t *bcnt,
                          double *bval, double *weight)
{
    if (!c->e_cnt) return 3;
    size_t ne = (size_t)c->nedge;
    if (cnt) memcpy(cnt, c->e_cnt, ne * sizeof(int32_t));
    if (idx) memcpy(idx, c->e_idx, ne * ORC_EDGE_CAP * sizeof(int32_t));
    if (coef) memcpy(coef, c->e_coef, ne * ORC_EDGE_CAP * sizeof(double));
    if (bcnt) memcpy(bcnt, c->e_bcnt, ne * sizeof(int32_t));
    if (bval) memcpy(bval, c->e_bval, ne * ORC_EDGE_BCAP * sizeof(double));
    if (weight) memcpy(weight, c->e_w, ne * sizeof(double));
    return 0;
}
int orc_get_center_stencils(orc_ctx *c, int32_t *cnt, int32_t *idx, double *coef, int32_t *bcnt,
                            double *bval, double *weight)
{
    if (!c->c_cnt) return 3;
    size_t nc = (size_t)c->ncenter, n3 = nc * 3;
    if (cnt) memcpy(cnt, c->c_cnt, n3 * sizeof(int32_t));
    if (idx) memcpy(idx, c->c_idx, n3 * ORC_CENTER_CAP * sizeof(int32_t));
    if (coef) memcpy(coef, c->c_coef, n3 * ORC_CENTER_CAP * sizeof(double));
    if (bcnt) memcpy(bcnt, c->c_bcnt, n3 * sizeof(int32_t));
    if (bval) memcpy(bval, c->c_bval, n3 * ORC_CENTER_BCAP * sizeof(double));
    if (weight) memcpy(weight, c->c_w, nc * sizeof(double));
    return 0;
}
int orc_get_initial_guess(orc_ctx *c, double *x0)
{
    if (!c->x0) return 3;
    memcpy(x0, c->x0, (size_t)c->nvel * sizeof(double));
    return 0;
}
int64_t orc_nnz(orc_ctx *c) { return c->nnz; }
int64_t orc_raw_triplets(orc_ctx *c) { return c->nraw; }
int orc_get_csr(orc_ctx *c, int64_t *row_ptr, int32_t *col, double *val, double *rhs)
{
    if (!c->row_ptr) return 3;
    if (row_ptr) memcpy(row_ptr, c->row_ptr, ((size_t)c->nvel + 1) * sizeof(int64_t));
    if (col) memcpy(col, c->col, (size_t)c->nnz * sizeof(int32_t));
    if (val) memcpy(val, c->val, (size_t)c->nnz * sizeof(double));
    if (rhs) memcpy(rhs, c->rhs, (size_t)c->nvel * sizeof(double));
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* post-solve transfer                                                                  */
/* ------------------------------------------------------------------------------------ */
/* buildRegularVelocityIndices, cpp:1445-1512 + classifyRegularVelocityFacesPartial cpp:1087-1165 */
int orc_build_regular_indices(orc_ctx *c, double extrapolation_scale)
{
    const double extrapolation = c->dx * extrapolation_scale;
    const double occ_sdf = 2. * c->dx;
    int r0[3];
    cell_res(c, 0, r0);
    int64_t next = 0;
    for (int axis = 0; axis < 3; ++axis) {
        int fr[3];
        face_res(c, 0, axis, fr);
        free(c->ridx[axis]);
        int32_t *g = c->ridx[axis] = alloc_idx(fr);
        if (!g) return 2;
        tilemask tm;
        if (tm_init(&tm, fr)) return 2;
        for (int k = 0; k < r0[2]; ++k)
            for (int j = 0; j < r0[1]; ++j)
                for (int i = 0; i < r0[0]; ++i) {
                    int p[3] = {i, j, k};
                    if (!((double)fget(&c->liquid, r0, p) < occ_sdf)) continue;
                    for (int d = 0; d < 2; ++d) {
                        int f[3];
                        cell_to_face(p, axis, d, f);
                        tm_mark(&tm, f);
                    }
                }
        for (int k = 0; k < fr[2]; ++k)
            for (int j = 0; j < fr[1]; ++j)
                for (int i = 0; i < fr[0]; ++i) {
                    int f[3] = {i, j, k};
                    if (!tm_get(&tm, f)) continue;
                    int bc[3], fc[3];
                    face_to_cell(f, axis, 0, bc);
                    face_to_cell(f, axis, 1, fc);
                    if (bc[axis] < 0 || fc[axis] >= r0[axis]) continue; /* cpp:1124-1125 */
                    int active = fget(&c->centerw, r0, bc) > 0.f || fget(&c->centerw, r0, fc) > 0.f;
                    for (int ea = 0; ea < 3 && !active; ++ea) {
                        if (ea == axis) continue;
                        int er[3];
                        edge_res(c, 0, ea, er);
                        for (int d = 0; d < 2; ++d) {
                            int e[3];
                            face_to_edge(f, axis, ea, d, e);
                            if (fget(&c->edgew[ea], er, e) > 0.f) { active = 1; break; }
                        }
                    }
                    if (active) {
                        int P2[3];
                        pos2_face(0, axis, f, P2);
                        float s = sample_f32(&c->solid, r0, OFF_CENTER, P2);
                        g[lin(fr, f)] = ((double)s > -extrapolation) ? ORC_SOLIDBOUNDARY : ORC_FLUID;
                    }
                }
        free(tm.occ);
        next = number_grid(g, fr, next); /* cpp:1486-1509: one counter over the three axes */
    }
    c->nregular = next;
    return 0;
}
int64_t orc_regular_count(orc_ctx *c) { return c->nregular; }
int orc_get_regular_index(orc_ctx *c, int axis, int32_t *out)
{
    if (axis < 0 || axis > 2 || !c->ridx[axis]) return 1;
    int fr[3];
    face_res(c, 0, axis, fr);
    memcpy(out, c->ridx[axis], vol(fr) * sizeof(int32_t));
    return 0;
}
int orc_set_regular_index(orc_ctx *c, int axis, const int32_t *idx)
{
    if (axis < 0 || axis > 2) return 1;
    int fr[3];
    face_res(c, 0, axis, fr);
    free(c->ridx[axis]);
    c->ridx[axis] = (int32_t *)malloc(vol(fr) * sizeof(int32_t));
    if (!c->ridx[axis]) return 2;
    memcpy(c->ridx[axis], idx, vol(fr) * sizeof(int32_t));
    return 0;
}

static inline void node_res(const orc_ctx *c, int l, int r[3])
{
    cell_res(c, l, r);
    r[0] += 1; r[1] += 1; r[2] += 1;
}
/* HDKnodeToFace, util.h:187-203 */
static inline void node_to_face(const int n[3], int faceAxis, int fi, int f[3])
{
    f[0] = n[0]; f[1] = n[1]; f[2] = n[2];
    for (int o = 0; o < 2; ++o)
        if (!(fi & (1 << o))) --f[(faceAxis + 1 + o) % 3];
}

typedef struct {
    float *vel[ORC_MAX_LEVELS][3]; /* octreeVelocity[level][axis], cpp:664-691 */
    float *nw[ORC_MAX_LEVELS][3];  /* nodeWeights */
    int32_t *nf[ORC_MAX_LEVELS];   /* nodeFlags */
} postwork;

static inline int32_t vidx_clamped(const orc_ctx *c, int l, int a, const int f[3])
{
    int r[3], q[3];
    face_res(c, l, a, r);
    for (int d = 0; d < 3; ++d) q[d] = clampi(f[d], 0, r[d] - 1); /* the reference reads out of bounds near the domain border */
    return c->vidx[l][a][lin(r, q)];
}
static inline float vel_clamped(const orc_ctx *c, const postwork *w, int l, int a, const int f[3])
{
    int r[3], q[3];
    face_res(c, l, a, r);
    for (int d = 0; d < 3; ++d) q[d] = clampi(f[d], 0, r[d] - 1);
    return w->vel[l][a][lin(r, q)];
}

/* interpSPGrid, interp.cpp:660-845; P2 = sample position in half fine cells (exact) */
static double interp_sp_grid(const orc_ctx *c, const postwork *w, const int P2[3], int axis)
{
    const int L = c->levels;
    int cell[3] = {floordiv2(P2[0]), floordiv2(P2[1]), floordiv2(P2[2])}; /* floor(indexPoint), node lattice level 0 */
    for (int level = 0; level < L; ++level) {
        int cr[3];
        cell_res(c, level, cr);
        int cc[3] = {clampi(cell[0], 0, cr[0] - 1), clampi(cell[1], 0, cr[1] - 1), clampi(cell[2], 0, cr[2] - 1)};
        if (c->labels[level][lin(cr, cc)] == ORC_ACTIVE) {
            const double scale = (double)(1 << (level + 1)); /* half fine cells per level-`level` cell */
            double ifp[3];
            int face[3];
            for (int a = 0; a < 3; ++a) {
                ifp[a] = (double)P2[a] / scale - (a == axis ? 0. : .5); /* posToIndex on the face lattice */
                face[a] = (int)floor(ifp[a]);
            }
            int transition = 0;
            for (int fi = 0; fi < 8 && !transition; ++fi) { /* HDKcellToNode(face, fi), interp.cpp:683-698 */
                int nf[3] = {face[0] + (fi & 1), face[1] + ((fi >> 1) & 1), face[2] + ((fi >> 2) & 1)};
                if (vidx_clamped(c, level, axis, nf) == ORC_UNASSIGNED) transition = 1;
            }
            if (!transition) { /* interp.cpp:700-728 */
                double iw[3];
                for (int a = 0; a < 3; ++a) {
                    iw[a] = ifp[a] - (double)face[a];
                    iw[a] = iw[a] < 0. ? 0. : (iw[a] > 1. ? 1. : iw[a]);
                }
                double v = 0;
                for (int fi = 0; fi < 8; ++fi) {
                    int nf[3] = {face[0] + (fi & 1), face[1] + ((fi >> 1) & 1), face[2] + ((fi >> 2) & 1)};
                    double wt = 1.;
                    for (int a = 0; a < 3; ++a) wt *= (nf[a] - face[a] == 0) ? (1. - iw[a]) : iw[a];
                    v += wt * (double)vel_clamped(c, w, level, axis, nf);
                }
                return v;
            }
            /* interp.cpp:730-836 */
            double ciw = (double)P2[axis] / scale - (double)cell[axis];
            ciw = ciw < 0. ? 0. : (ciw > 1. ? 1. : ciw);
            const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
            double fiv[2] = {0., 0.};
            for (int dir = 0; dir < 2; ++dir) {
                int af[3];
                cell_to_face(cell, axis, dir, af);
                int fl = level;
                if (vidx_clamped(c, level, axis, af) == ORC_UNASSIGNED && level > 0) { /* project onto a child face */
                    const double cs = (double)(1 << level);
                    const double cip1 = (double)P2[a1] / cs, cip2 = (double)P2[a2] / cs; /* node lattice of level-1 */
                    for (int ci = 0; ci < 4; ++ci) {
                        int cf[3];
                        child_face(af, axis, ci, cf);
                        if ((double)cf[a1] <= cip1 && (double)cf[a2] <= cip2 && (double)(cf[a1] + 1) >= cip1 && (double)(cf[a2] + 1) >= cip2) {
                            fl = level - 1;
                            af[0] = cf[0]; af[1] = cf[1]; af[2] = cf[2];
                            break;
                        }
                    }
                }
                const double ns = (double)(1 << (fl + 1));
                double inp1 = (double)P2[a1] / ns, inp2 = (double)P2[a2] / ns;
                double fw[2] = {inp1 - floor(inp1), inp2 - floor(inp2)};
                const double fvel = (double)vel_clamped(c, w, fl, axis, af);
                int nr[3];
                node_res(c, fl, nr);
                double avg = 0;
                for (int ni = 0; ni < 4; ++ni) { /* HDKfaceToNode, util.h:133-149 */
                    int nd[3] = {af[0], af[1], af[2]};
                    if (ni & 1) ++nd[a1];
                    if (ni & 2) ++nd[a2];
                    double wt = 1.;
                    wt *= (nd[a1] - af[a1] == 0) ? (1. - fw[0]) : fw[0];
                    wt *= (nd[a2] - af[a2] == 0) ? (1. - fw[1]) : fw[1];
                    int q[3] = {clampi(nd[0], 0, nr[0] - 1), clampi(nd[1], 0, nr[1] - 1), clampi(nd[2], 0, nr[2] - 1)};
                    const double nv = (double)c->nval[fl][axis][lin(nr, q)];
                    avg += nv;
                    fiv[dir] += nv * wt;
                }
                double m = fw[0];
                double m2 = fw[1];
                double m3 = 1. - fw[0], m4 = 1 - fw[1];
                double mm = m3 < m4 ? m3 : m4;          /* SYSmin(1 - w0, 1 - w1) */
                mm = m2 < mm ? m2 : mm;                 /* SYSmin(w1, ...) */
                mm = m < mm ? m : mm;                   /* SYSmin(w0, ...) */
                fiv[dir] += 2. * (fvel - .25 * avg) * mm;
            }
            return (1. - ciw) * fiv[0] + ciw * fiv[1];
        }
        cell[0] = cell[0] / 2; cell[1] = cell[1] / 2; cell[2] = cell[2] / 2; /* getParentCell */
    }
    return 0.; /* reference: assert(false) */
}

int orc_transfer_to_regular_grid(orc_ctx *c, const double *solution, float *out_x, float *out_y, float *out_z)
{
    if (!c->vdof || !c->ridx[0] || !c->ridx[1] || !c->ridx[2]) return 3;
    const int L = c->levels;
    postwork w;
    memset(&w, 0, sizeof(w));
    int rc = 0;
    /* setOctreeVelocity, cpp:2779-2813 (fp32 fields, 0 elsewhere) */
    for (int l = 0; l < L; ++l) {
        int nr[3];
        node_res(c, l, nr);
        free(c->nlab[l]);
        c->nlab[l] = (int8_t *)calloc(vol(nr), 1);
        w.nf[l] = (int32_t *)calloc(vol(nr), sizeof(int32_t));
        for (int a = 0; a < 3; ++a) {
            int fr[3];
            face_res(c, l, a, fr);
            w.vel[l][a] = (float *)calloc(vol(fr), sizeof(float));
            free(c->nval[l][a]);
            c->nval[l][a] = (float *)calloc(vol(nr), sizeof(float));
            w.nw[l][a] = (float *)calloc(vol(nr), sizeof(float));
            if (!w.vel[l][a] || !c->nval[l][a] || !w.nw[l][a]) rc = 2;
            if (rc) continue;
            size_t n = vol(fr);
            for (size_t o = 0; o < n; ++o) {
                int32_t id = c->vidx[l][a][o];
                if (id >= 0) w.vel[l][a][o] = (float)solution[id];
            }
        }
        if (!c->nlab[l] || !w.nf[l]) rc = 2;
    }
    if (rc) goto done;
    /* setActiveNodes (interp.cpp:118-188) + sampleActiveNodes (interp.cpp:190-286) */
    for (int l = 0; l < L; ++l) {
        int nr[3];
        node_res(c, l, nr);
        const double weight = (double)(1 << (L - l - 1));
        for (int k = 0; k < nr[2]; ++k)
            for (int j = 0; j < nr[1]; ++j)
                for (int i = 0; i < nr[0]; ++i) {
                    int node[3] = {i, j, k};
                    int active = 0, inactive = 0;
                    for (int fa = 0; !inactive && fa < 3; ++fa) {
                        int fr[3];
                        face_res(c, l, fa, fr);
                        const int b1 = (fa + 1) % 3, b2 = (fa + 2) % 3;
                        for (int fi = 0; fi < 4; ++fi) {
                            int f[3];
                            node_to_face(node, fa, fi, f);
                            if (f[b1] < 0 || f[b2] < 0 || f[b1] >= fr[b1] || f[b2] >= fr[b2]) { inactive = 1; continue; }
                            int32_t vi = c->vidx[l][fa][lin(fr, f)];
                            if (vi >= 0) active = 1;
                            else if (vi == ORC_SOLIDBOUNDARY || vi == ORC_OUTSIDE) { inactive = 1; break; }
                        }
                    }
                    if (!(active && !inactive)) continue;
                    const size_t no = lin(nr, node);
                    c->nlab[l][no] = 1;
                    int32_t flag = 0;
                    for (int fa = 0; fa < 3; ++fa) {
                        int fr[3];
                        face_res(c, l, fa, fr);
                        double av = 0., aw = 0.;
                        for (int fi = 0; fi < 4; ++fi) {
                            int f[3];
                            node_to_face(node, fa, fi, f);
                            int32_t vi = c->vidx[l][fa][lin(fr, f)]; /* active nodes have all 12 faces in bounds */
                            if (vi >= 0) {
                                av += weight * (double)w.vel[l][fa][lin(fr, f)];
                                aw += weight;
                                flag += 1 << (fa * 4 + fi);
                            } else if (vi != ORC_UNASSIGNED) {
                                aw += weight;
                                flag += 1 << (fa * 4 + fi);
                            }
                        }
                        c->nval[l][fa][no] = (float)av;
                        w.nw[l][fa][no] = (float)aw;
                    }
                    w.nf[l][no] = flag;
                }
    }
    /* bubbleActiveNodeValues, interp.cpp:288-355 */
    for (int l = 0; l < L - 1; ++l) {
        int nr[3], pr[3];
        node_res(c, l, nr);
        node_res(c, l + 1, pr);
        for (int k = 0; k < nr[2]; k += 2)
            for (int j = 0; j < nr[1]; j += 2)
                for (int i = 0; i < nr[0]; i += 2) {
                    int node[3] = {i, j, k}, par[3] = {i / 2, j / 2, k / 2};
                    const size_t no = lin(nr, node), po = lin(pr, par);
                    if (c->nlab[l][no] != 1 || c->nlab[l + 1][po] != 1) continue;
                    w.nf[l + 1][po] = w.nf[l][no] + w.nf[l + 1][po];
                    for (int a = 0; a < 3; ++a) {
                        w.nw[l + 1][a][po] = (float)((double)w.nw[l][a][no] + (double)w.nw[l + 1][a][po]);
                        c->nval[l + 1][a][po] = (float)((double)c->nval[l][a][no] + (double)c->nval[l + 1][a][po]);
                    }
                    c->nlab[l][no] = 2; /* DEPENDENTNODE */
                }
    }
    /* finishIncompleteNodes, interp.cpp:357-567 */
    for (int l = 0; l < L - 1; ++l) {
        int nr[3];
        node_res(c, l, nr);
        const double weight = (double)(1 << (L - l - 1));
        for (int k = 0; k < nr[2]; ++k)
            for (int j = 0; j < nr[1]; ++j)
                for (int i = 0; i < nr[0]; ++i) {
                    int node[3] = {i, j, k};
                    const size_t no = lin(nr, node);
                    if (c->nlab[l][no] != 1) continue;
                    int32_t flag = w.nf[l][no];
                    if (flag == 0xFFF) continue;
                    int32_t temp = flag;
                    for (int bit = 0; flag != 0xFFF && bit < 12; ++bit, temp >>= 1) {
                        if (temp & 1) continue;
                        const int fa = bit / 4, fi = bit % 4;
                        int face[3];
                        node_to_face(node, fa, fi, face);
                        int found = 0;
                        if (node[fa] % 2 == 0) {
                            int pf[3] = {face[0] / 2, face[1] / 2, face[2] / 2};
                            int pr[3];
                            face_res(c, l + 1, fa, pr);
                            if (c->vidx[l + 1][fa][lin(pr, pf)] >= 0) {
                                double ghost = (double)w.vel[l + 1][fa][lin(pr, pf)];
                                double v = (double)c->nval[l][fa][no];
                                v += weight * ghost;
                                c->nval[l][fa][no] = (float)v;
                                double ww = (double)w.nw[l][fa][no];
                                ww += weight;
                                w.nw[l][fa][no] = (float)ww;
                                flag += 1 << bit;
                                found = 1;
                            }
                        }
                        if (!found) {
                            int cell[3] = {face[0], face[1], face[2]}; /* HDKfaceToCell(face, faceAxis, 1) */
                            int sl = l;
                            for (;;) {
                                int cr[3];
                                cell_res(c, sl, cr);
                                int cc[3] = {clampi(cell[0], 0, cr[0] - 1), clampi(cell[1], 0, cr[1] - 1), clampi(cell[2], 0, cr[2] - 1)};
                                if (c->labels[sl][lin(cr, cc)] == ORC_ACTIVE || sl + 1 >= L) break;
                                cell[0] /= 2; cell[1] /= 2; cell[2] /= 2;
                                ++sl;
                            }
                            /* fractional position of the face along faceAxis inside the level-sl cell */
                            const double ip = (double)((int64_t)face[fa] << l) / (double)(1 << sl);
                            const double iw = ip - floor(ip);
                            double ghost = 0.;
                            for (int dir = 0; dir < 2; ++dir) {
                                int of[3];
                                cell_to_face(cell, fa, dir, of);
                                const double lw = dir == 0 ? 1. - iw : iw;
                                int32_t oi = vidx_clamped(c, sl, fa, of);
                                if (oi >= 0) ghost += lw * (double)vel_clamped(c, &w, sl, fa, of);
                                else if (oi == ORC_UNASSIGNED && sl > 0) {
                                    for (int ci = 0; ci < 4; ++ci) {
                                        int cf[3];
                                        child_face(of, fa, ci, cf);
                                        if (vidx_clamped(c, sl - 1, fa, cf) >= 0)
                                            ghost += .25 * lw * (double)vel_clamped(c, &w, sl - 1, fa, cf);
                                    }
                                }
                            }
                            double v = (double)c->nval[l][fa][no];
                            v += weight * ghost;
                            c->nval[l][fa][no] = (float)v;
                            double ww = (double)w.nw[l][fa][no];
                            ww += weight;
                            w.nw[l][fa][no] = (float)ww;
                            flag += 1 << bit;
                        }
                    }
                    w.nf[l][no] = flag;
                }
    }
    /* normalizeActiveNodes, interp.cpp:569-613 */
    for (int l = 0; l < L; ++l) {
        int nr[3];
        node_res(c, l, nr);
        size_t n = vol(nr);
        for (size_t o = 0; o < n; ++o)
            if (c->nlab[l][o] == 1)
                for (int a = 0; a < 3; ++a)
                    c->nval[l][a][o] = (float)((double)c->nval[l][a][o] / (double)w.nw[l][a][o]);
    }
    /* distributeNodeValuesDown, interp.cpp:615-658 */
    for (int l = L - 2; l >= 0; --l) {
        int nr[3], pr[3];
        node_res(c, l, nr);
        node_res(c, l + 1, pr);
        for (int k = 0; k < nr[2]; ++k)
            for (int j = 0; j < nr[1]; ++j)
                for (int i = 0; i < nr[0]; ++i) {
                    int node[3] = {i, j, k};
                    const size_t no = lin(nr, node);
                    if (c->nlab[l][no] != 2) continue;
                    int par[3] = {i / 2, j / 2, k / 2};
                    for (int a = 0; a < 3; ++a) c->nval[l][a][no] = c->nval[l + 1][a][lin(pr, par)];
                    c->nlab[l][no] = 1;
                }
    }
    /* applyVelocitiesToRegularGrid, cpp:2815-2894 */
    {
        float *outs[3] = {out_x, out_y, out_z};
        for (int axis = 0; axis < 3; ++axis) {
            int fr[3], off[3];
            face_res(c, 0, axis, fr);
            off_face(axis, off);
            orc_get_field(c, ORC_F_VELOCITY + axis, outs[axis]);
#pragma omp parallel for collapse(2) schedule(static)
            for (int k = 0; k < fr[2]; ++k)
                for (int j = 0; j < fr[1]; ++j)
                    for (int i = 0; i < fr[0]; ++i) {
                        int f[3] = {i, j, k};
                        const size_t o = lin(fr, f);
                        const int32_t ri = c->ridx[axis][o];
                        int P2[3];
                        pos2_face(0, axis, f, P2);
                        if (ri >= 0) {
                            const int32_t oi = c->vidx[0][axis][o];
                            if (oi >= 0) outs[axis][o] = (float)solution[oi];
                            else if (oi == ORC_SOLIDBOUNDARY) outs[axis][o] = sample_f32(&c->solidvel[axis], fr, off, P2);
                            else if (oi == ORC_UNASSIGNED) outs[axis][o] = (float)interp_sp_grid(c, &w, P2, axis);
                        } else if (ri == ORC_SOLIDBOUNDARY)
                            outs[axis][o] = sample_f32(&c->solidvel[axis], fr, off, P2);
                    }
        }
    }
done:
    for (int l = 0; l < L; ++l) {
        free(w.nf[l]);
        for (int a = 0; a < 3; ++a) { free(w.vel[l][a]); free(w.nw[l][a]); }
    }
    return rc;
}

int orc_get_node_grid(orc_ctx *c, int level, int8_t *labels, float *vx, float *vy, float *vz)
{
    if (level < 0 || level >= c->levels || !c->nlab[level]) return 1;
    int nr[3];
    node_res(c, level, nr);
    if (labels) memcpy(labels, c->nlab[level], vol(nr));
    float *o[3] = {vx, vy, vz};
    for (int a = 0; a < 3; ++a)
        if (o[a]) memcpy(o[a], c->nval[level][a], vol(nr) * sizeof(float));
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* solve.  Eigen::ConjugateGradient<SparseMatrix<double>, Lower|Upper> with the default  */
/* DiagonalPreconditioner (cpp:618-630).  Eigen is NOT in /root/reference (FindEIGEN3    */
/* .cmake:18-27, no pinned version); the published algorithm of Eigen 3.3/3.4            */
/* IterativeLinearSolvers/ConjugateGradient.h is restated below.  Dot      */
/* products are plain left-to-right sums (Eigen's are packet-wise; unpinnable).          */
/* ------------------------------------------------------------------------------------ */
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int orc_spmv_csr(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *val,
                 const double *x, double *y, int threads)
{
    /* Eigen's column-major product adds A(i,j)*x(j) into y(i) for j ascending, starting from
     * zero: the same order as a CSR row walk with sorted columns. */
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.;
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) s += val[k] * x[col[k]];
        y[i] = s;
    }
    return 0;
}

static double dot_(int64_t n, const double *a, const double *b, int threads)
{
    double s = 0.;
#ifdef _OPENMP
    if (threads > 1) {
        /* Deterministic for a given team size: every thread sums a fixed contiguous share, the shares are added in thread
         * order.  (An OpenMP `reduction` combines the partial sums in arrival order: the iteration count of a 1271-iteration
         * solve then wandered between 1271 and 1277 from run to run, and with it the parity tests that compare counts.) */
        enum { MAXT = 1024 };
        double part[MAXT];
        int team = 1;
        if (threads > MAXT) threads = MAXT;
#pragma omp parallel num_threads(threads)
        {
            const int t = omp_get_thread_num(), T = omp_get_num_threads();
            const int64_t i0 = n * t / T, i1 = n * (t + 1) / T;
            double ps = 0.;
            for (int64_t i = i0; i < i1; ++i) ps += a[i] * b[i];
            part[t] = ps;
            if (t == 0) team = T;
        }
        for (int t = 0; t < team; ++t) s += part[t];
        return s;
    }
#endif
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* parallel first-touch copy: pages land on the NUMA node of the thread that will stream them
 * (schedule(static) everywhere), otherwise a many-core host runs the baseline off one node */
static void *numa_copy(const void *src, size_t bytes, int threads)
{
    char *dst = (char *)malloc(bytes ? bytes : 1);
    if (!dst) return NULL;
    const int64_t chunks = (int64_t)((bytes + 4095) / 4096);
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t c = 0; c < chunks; ++c) {
        size_t o = (size_t)c * 4096, m = bytes - o < 4096 ? bytes - o : 4096;
        memcpy(dst + o, (const char *)src + o, m);
    }
    return dst;
}

static int orc_pcg_csr_impl(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *val,
                            const double *b, double *x, double tol, int max_iters, int spmv_threads, int threads,
                            orc_pcg_info *info);
static _Thread_local int orc_no_precond_tls = 0; /* set by orc_solve around its call (the public PCG entry points keep their signatures) */

int orc_pcg_csr(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *val,
                const double *b, double *x, double tol, int max_iters, int threads,
                orc_pcg_info *info)
{
    return orc_pcg_csr_ex(n, row_ptr, col, val, b, x, tol, max_iters, threads, threads, info);
}

/* spmv_threads: OpenMP threads of the row-parallel SpMV; vec_threads: threads of the dots / AXPYs.
 * Eigen parallelises only the sparse product (Lower|Upper with OpenMP, reference CMakeLists.txt:27-32):
 * "eigen_faithful" = (T, 1); "all_parallel" = (T, T). */
int orc_pcg_csr_ex(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *val,
                   const double *b, double *x, double tol, int max_iters, int spmv_threads, int vec_threads,
                   orc_pcg_info *info)
{
    if (spmv_threads < 1) spmv_threads = 1;
    if (vec_threads < 1) vec_threads = 1;
    const int threads = spmv_threads;
    if (threads == 1 || n < (1 << 20))
        return orc_pcg_csr_impl(n, row_ptr, col, val, b, x, tol, max_iters, spmv_threads, vec_threads, info);
    /* large parallel runs (the bench's cpu_baseline): NUMA-local copies of the system.  Row blocks of
     * the static schedule own contiguous nnz ranges only approximately; good enough for streaming. */
    const int64_t nnz = row_ptr[n];
    int64_t *rp = (int64_t *)numa_copy(row_ptr, ((size_t)n + 1) * sizeof(int64_t), threads);
    int32_t *cl = (int32_t *)numa_copy(col, (size_t)nnz * sizeof(int32_t), threads);
    double *vl = (double *)numa_copy(val, (size_t)nnz * sizeof(double), threads);
    double *bb = (double *)numa_copy(b, (size_t)n * sizeof(double), threads);
    double *xx = (double *)numa_copy(x, (size_t)n * sizeof(double), threads);
    int rc = 2;
    if (rp && cl && vl && bb && xx) {
        rc = orc_pcg_csr_impl(n, rp, cl, vl, bb, xx, tol, max_iters, spmv_threads, vec_threads, info);
        memcpy(x, xx, (size_t)n * sizeof(double));
    }
    free(rp); free(cl); free(vl); free(bb); free(xx);
    return rc;
}

static int orc_pcg_csr_impl(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *val,
                            const double *b, double *x, double tol, int max_iters, int spmv_threads, int threads,
                            orc_pcg_info *info)
{
    if (threads < 1) threads = 1; /* vector ops */
    if (spmv_threads < 1) spmv_threads = 1;
    double *r = (double *)malloc((size_t)n * sizeof(double));
    double *p = (double *)malloc((size_t)n * sizeof(double));
    double *z = (double *)malloc((size_t)n * sizeof(double));
    double *tmp = (double *)malloc((size_t)n * sizeof(double));
    double *invd = (double *)malloc((size_t)n * sizeof(double));
    if (!r || !p || !z || !tmp || !invd) { free(r); free(p); free(z); free(tmp); free(invd); return 2; }
    if (spmv_threads > 1) { /* first touch by the threads that stream the rows */
#pragma omp parallel for schedule(static) num_threads(spmv_threads)
        for (int64_t i = 0; i < n; ++i) { r[i] = 0.; p[i] = 0.; z[i] = 0.; tmp[i] = 0.; invd[i] = 0.; }
    }
    /* DiagonalPreconditioner::factorize: invdiag = 1/A(j,j) if != 0 else 1 */
#pragma omp parallel for schedule(static) num_threads(spmv_threads) if (spmv_threads > 1)
    for (int64_t i = 0; i < n; ++i) {
        double d = 0.;
        int have = 0;
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k)
            if (col[k] == i) { d = val[k]; have = 1; }
        invd[i] = (have && d != 0. && !orc_no_precond_tls) ? 1. / d : 1.;
    }
    double t0 = now_s(), tspmv = 0.;
    int iters = 0;
    double err = 0.;
    orc_spmv_csr(n, row_ptr, col, val, x, tmp, spmv_threads);
    for (int64_t i = 0; i < n; ++i) r[i] = b[i] - tmp[i];
    double rhsNorm2 = dot_(n, b, b, threads);
    if (rhsNorm2 == 0.) {
        for (int64_t i = 0; i < n; ++i) x[i] = 0.;
        iters = 0; err = 0.;
        goto done;
    }
    {
        const double considerAsZero = 2.2250738585072014e-308;
        double threshold = tol * tol * rhsNorm2;
        if (threshold < considerAsZero) threshold = considerAsZero;
        double residualNorm2 = dot_(n, r, r, threads);
        if (residualNorm2 < threshold) {
            iters = 0;
            err = sqrt(residualNorm2 / rhsNorm2);
            goto done;
        }
        for (int64_t i = 0; i < n; ++i) p[i] = invd[i] * r[i];
        double absNew = dot_(n, r, p, threads);
        int i = 0;
        while (i < max_iters) {
            double ts = now_s();
            orc_spmv_csr(n, row_ptr, col, val, p, tmp, spmv_threads);
            tspmv += now_s() - ts;
            double alpha = absNew / dot_(n, p, tmp, threads);
            if (threads > 1) {
#pragma omp parallel for schedule(static) num_threads(threads)
                for (int64_t k = 0; k < n; ++k) { x[k] += alpha * p[k]; r[k] -= alpha * tmp[k]; }
            } else {
                for (int64_t k = 0; k < n; ++k) x[k] += alpha * p[k];
                for (int64_t k = 0; k < n; ++k) r[k] -= alpha * tmp[k];
            }
            residualNorm2 = dot_(n, r, r, threads);
            if (residualNorm2 < threshold) break;
            if (threads > 1) {
#pragma omp parallel for schedule(static) num_threads(threads)
                for (int64_t k = 0; k < n; ++k) z[k] = invd[k] * r[k];
            } else
                for (int64_t k = 0; k < n; ++k) z[k] = invd[k] * r[k];
            double absOld = absNew;
            absNew = dot_(n, r, z, threads);
            double beta = absNew / absOld;
            if (threads > 1) {
#pragma omp parallel for schedule(static) num_threads(threads)
                for (int64_t k = 0; k < n; ++k) p[k] = z[k] + beta * p[k];
            } else
                for (int64_t k = 0; k < n; ++k) p[k] = z[k] + beta * p[k];
            i++;
        }
        err = sqrt(residualNorm2 / rhsNorm2);
        iters = i;
    }
done:
    if (info) {
        info->iterations = iters;
        info->error = err;
        info->rhs_norm2 = rhsNorm2;
        info->seconds = now_s() - t0;
        info->spmv_seconds = tspmv;
        info->threads = spmv_threads;
    }
    free(r); free(p); free(z); free(tmp); free(invd);
    return 0;
}

/* Eigen's float reductions (a.dot(b), v.squaredNorm() of VectorXf): the linear vectorised redux of Eigen 3.3 / 3.4 (Core/Redux.h,
 * redux_impl<..., LinearVectorizedTraversal, NoUnrolling>) with the packet the reference's build has -- its CMakeLists.txt sets no
 * -march flag, so x86-64's baseline SSE2: 4 floats.  Two packet accumulators take alternate packets (a product, then an add, per lane:
 * no FMA in SSE2), are added lane by lane, a last odd packet is added, the four lanes are folded as (l0 + l2) + (l1 + l3) (predux of
 * Packet4f), and the scalar tail follows.  Eigen is not in /root/reference (a system dependency, no version pinned): this restates the
 * published algorithm; vectors from Eigen's allocator are 16-byte aligned, so the aligned range starts at element 0.
 * orc_set_f32_serial_dots(1) switches to plain left-to-right float sums (what rounds 3-4 of this oracle used). */
static _Thread_local int orc_f32_serial_dots_tls = 0;
int orc_set_f32_serial_dots(int on) { orc_f32_serial_dots_tls = on ? 1 : 0; return 0; }
static float dot_f32_eigen(const float *a, const float *b, int64_t n)
{
    if (orc_f32_serial_dots_tls) {
        float s = 0.f;
        for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
        return s;
    }
    const int64_t P = 4;
    const int64_t size2 = (n / (2 * P)) * (2 * P), size1 = (n / P) * P;
    float res;
    if (size1 == 0) {
        if (n == 0) return 0.f;
        res = a[0] * b[0];
        for (int64_t i = 1; i < n; ++i) res += a[i] * b[i];
        return res;
    }
    float p0[4], p1[4];
    for (int l = 0; l < 4; ++l) p0[l] = a[l] * b[l];
    if (size1 > P) {
        for (int l = 0; l < 4; ++l) p1[l] = a[P + l] * b[P + l];
        for (int64_t i = 2 * P; i < size2; i += 2 * P)
            for (int l = 0; l < 4; ++l) {
                p0[l] += a[i + l] * b[i + l];
                p1[l] += a[i + P + l] * b[i + P + l];
            }
        for (int l = 0; l < 4; ++l) p0[l] += p1[l];
        if (size1 > size2)
            for (int l = 0; l < 4; ++l) p0[l] += a[size2 + l] * b[size2 + l];
    }
    res = (p0[0] + p0[2]) + (p0[1] + p0[3]);
    for (int64_t i = size1; i < n; ++i) res += a[i] * b[i];
    return res;
}

/* (exported for tests/test_oracle_known_answers.py: the reduction order against a numpy restatement of Redux.h) */
float orc_dot_f32(const float *a, const float *b, int64_t n) { return dot_f32_eigen(a, b, n); }

/* Eigen::ConjugateGradient<SparseMatrix<float>, Lower|Upper> (USESINGLEPRECISION): the algorithm above with every scalar and vector a float;
 * row sums left to right in float, dots in Eigen's reduction order (dot_f32_eigen).  The GPU side folds its dots in another order (a
 * thread's terms in float, the rest in double), so it is compared to the SOLUTION and the iteration count within a tolerance.  The
 * system arrives as float values in double arrays. */
static int orc_pcg_csr_f32(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *val, const double *b, double *x,
                           double tol_d, int max_iters, orc_pcg_info *info)
{
    const int64_t nnz = row_ptr[n];
    float *v = (float *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(float));
    float *xf = (float *)malloc((size_t)(n > 0 ? n : 1) * 6 * sizeof(float));
    if (!v || !xf) { free(v); free(xf); return 2; }
    float *r = xf + n, *p = r + n, *z = p + n, *tmp = z + n, *invd = tmp + n;
    for (int64_t k = 0; k < nnz; ++k) v[k] = (float)val[k];
    for (int64_t i = 0; i < n; ++i) xf[i] = (float)x[i];
    const float tol = (float)tol_d;
    const double t0 = now_s();
    for (int64_t i = 0; i < n; ++i) {
        float d = 0.f;
        int have = 0;
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k)
            if (col[k] == i) { d = v[k]; have = 1; }
        invd[i] = (have && d != 0.f && !orc_no_precond_tls) ? 1.f / d : 1.f;
    }
#define SPMV_F(src, dst) for (int64_t i_ = 0; i_ < n; ++i_) { float s_ = 0.f; for (int64_t k_ = row_ptr[i_]; k_ < row_ptr[i_ + 1]; ++k_) s_ += v[k_] * (src)[col[k_]]; (dst)[i_] = s_; }
#define DOT_F(a_, b_, out_) { out_ = dot_f32_eigen((a_), (b_), n); }
    int iters = 0;
    float err = 0.f, rhsNorm2, residualNorm2;
    SPMV_F(xf, tmp);
    for (int64_t i = 0; i < n; ++i) r[i] = (float)b[i] - tmp[i];
    for (int64_t i = 0; i < n; ++i) z[i] = (float)b[i]; /* (z is free until the loop) */
    DOT_F(z, z, rhsNorm2);
    if (rhsNorm2 == 0.f) {
        for (int64_t i = 0; i < n; ++i) xf[i] = 0.f;
    } else {
        float threshold = tol * tol * rhsNorm2;
        if (threshold < 1.17549435e-38f) threshold = 1.17549435e-38f;
        DOT_F(r, r, residualNorm2);
        if (residualNorm2 >= threshold) {
            for (int64_t i = 0; i < n; ++i) p[i] = invd[i] * r[i];
            float absNew;
            DOT_F(r, p, absNew);
            int i = 0;
            while (i < max_iters) {
                SPMV_F(p, tmp);
                float pq;
                DOT_F(p, tmp, pq);
                const float alpha = absNew / pq;
                for (int64_t k = 0; k < n; ++k) xf[k] += alpha * p[k];
                for (int64_t k = 0; k < n; ++k) r[k] -= alpha * tmp[k];
                DOT_F(r, r, residualNorm2);
                if (residualNorm2 < threshold) break;
                for (int64_t k = 0; k < n; ++k) z[k] = invd[k] * r[k];
                const float absOld = absNew;
                DOT_F(r, z, absNew);
                const float beta = absNew / absOld;
                for (int64_t k = 0; k < n; ++k) p[k] = z[k] + beta * p[k];
                i++;
            }
            iters = i;
        }
        err = sqrtf(residualNorm2 / rhsNorm2);
    }
#undef SPMV_F
#undef DOT_F
    for (int64_t i = 0; i < n; ++i) x[i] = (double)xf[i];
    if (info) {
        info->iterations = iters;
        info->error = (double)err;
        info->rhs_norm2 = (double)rhsNorm2;
        info->seconds = now_s() - t0;
        info->spmv_seconds = 0.;
        info->threads = 1;
    }
    free(v); free(xf);
    return 0;
}

/* 0 = SolveType fpreal64 (default), 1 = fpreal32: set before orc_build_initial_guess / orc_assemble */
int orc_set_precision(orc_ctx *c, int f32)
{
    if (!c || (f32 != 0 && f32 != 1)) return 1;
    c->f32 = f32;
    return 0;
}

/* 0 = Jacobi (Eigen's DiagonalPreconditioner, default), 1 = none (plain CG) */
int orc_set_preconditioner(orc_ctx *c, int none)
{
    if (!c || (none != 0 && none != 1)) return 1;
    c->no_precond = none;
    return 0;
}

static int orc_solve_inner(orc_ctx *c, double tol, int max_iters, int threads, double *x_out, orc_pcg_info *info);
int orc_solve(orc_ctx *c, double tol, int max_iters, int threads, double *x_out, orc_pcg_info *info)
{
    if (!c->row_ptr || !c->x0) return 3;
    orc_no_precond_tls = c->no_precond;
    const int rc = orc_solve_inner(c, tol, max_iters, c->no_precond ? 1 : threads, x_out, info); /* (the flag is thread-local: serial solve) */
    orc_no_precond_tls = 0;
    return rc;
}
static int orc_solve_inner(orc_ctx *c, double tol, int max_iters, int threads, double *x_out, orc_pcg_info *info)
{
    if (c->f32) {
        memcpy(x_out, c->x0, (size_t)c->nvel * sizeof(double));
        return orc_pcg_csr_f32(c->nvel, c->row_ptr, c->col, c->val, c->rhs, x_out, tol, max_iters, info);
    }
    memcpy(x_out, c->x0, (size_t)c->nvel * sizeof(double)); /* solveWithGuess(rhs, guess) cpp:627 */
    return orc_pcg_csr(c->nvel, c->row_ptr, c->col, c->val, c->rhs, x_out, tol, max_iters, threads, info);
}
