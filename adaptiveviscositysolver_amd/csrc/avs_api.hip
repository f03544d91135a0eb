// avs_api.hip -- the extern "C" surface declared in include/avs.h.
// Context management, input upload, phase orchestration, read-back.  No kernels here.
#include <dlfcn.h>

#include <cmath>
#include <cstdlib>
#include <mutex>
#include <new>

#include "avs_internal.hpp"

namespace avs {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        if (const char *e = getenv("AVS_ROCTX"))
            if (atoi(e) == 0) return;
        for (const char *lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void *h = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
const Roctx &roctx()
{
    static Roctx r;
    return r;
}
} // namespace
void roctx_push(const char *name)
{
    if (roctx().push) (void)roctx().push(name);
}
void roctx_pop()
{
    if (roctx().pop) (void)roctx().pop();
}

// phase drivers implemented in avs_assembly.hip
avs_status build_dof_tables(avs_ctx *c);
avs_status build_system(avs_ctx *c);

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static void grid_res(const avs_desc &d, int kind /*0 face,1 edge,2 centre*/, int level, int axis, int r[3])
{
    r[0] = d.nx >> level;
    r[1] = d.ny >> level;
    r[2] = d.nz >> level;
    if (kind == 0) r[axis] += 1;
    else if (kind == 1) {
        r[0] += (axis != 0);
        r[1] += (axis != 1);
        r[2] += (axis != 2);
    }
}
static size_t vol3(const int r[3]) { return (size_t)r[0] * (size_t)r[1] * (size_t)r[2]; }


} // namespace avs

using namespace avs;

namespace avs {
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
static double env_dbl(const char *name, double dflt) { const char *e = getenv(name); return e ? atof(e) : dflt; }
// the only reader of the environment (besides AVS_ROCTX, a process-wide tracing switch, and the probe build's test hooks)
Options options_from_env()
{
    Options o;
    o.resident = env_int("AVS_CG_RESIDENT", 1) != 0;
    if (const char *e = getenv("AVS_DIST_TRANSPORT")) o.transport = strcmp(e, "rccl") == 0 ? 1 : (strcmp(e, "direct") == 0 ? 2 : 0);
    o.paranoid = env_int("AVS_DIST_PARANOID", 0) != 0;
    o.graph = env_int("AVS_PCG_GRAPH", 1) != 0;
    if (const char *e = getenv("AVS_BRICK")) { const int v = atoi(e); o.brick = v < 0 ? -1 : (v > 2 ? 1 : v); } // <0 auto, 0 never, 1 always, 2 tune
    o.brick_interleave = env_int("AVS_BRICK_INTERLEAVE", 1) != 0;
    o.brick_shift = env_int("AVS_BRICK_SHIFT", 3);
    o.value_index = env_int("AVS_VALUE_INDEX", 1) != 0;
    o.value_pack = env_int("AVS_VALUE_PACK", 1) != 0;
    o.tile_tables = env_int("AVS_TILE_TABLES", 1) != 0;
    o.column_windows = env_int("AVS_COLUMN_WINDOWS", 1) != 0;
    o.brick_min_regular = env_dbl("AVS_BRICK_MIN_REGULAR", 0.6);
    o.brick_timing = getenv("AVS_BRICK_TIMING") != nullptr;
    o.brick_plan = env_int("AVS_BRICK_PLAN", 1) != 0;
    o.brick_value_codes = env_int("AVS_BRICK_VALUE_CODES", 1) != 0;
    if (const char *e = getenv("AVS_BRICK_COST")) {
        BrickCost &k = o.brick_cost;
        (void)sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf", &k.tile, &k.row, &k.run, &k.word, &k.etile, &k.quad);
    }
    o.fuse_beta = env_int("AVS_PCG_FUSE_BETA", 1) != 0;
    { const int v = env_int("AVS_PCG_FUSE_VECTORS", 0); o.fuse_vectors = v < 0 ? -1 : (v > 0 ? 1 : 0); }
    o.fused_timeout_ms = env_int("AVS_PCG_FUSED_TIMEOUT_MS", 2000);
    { const int v = env_int("AVS_F32_VECTORS", -1); o.f32_vectors = v < 0 ? -1 : (v > 0 ? 1 : 0); }
    o.prepass_temporal = env_int("AVS_PREPASS_TEMPORAL", 1) != 0;
    { const int v = env_int("AVS_POST_DOF_SAMPLE", -1); o.post_dof_sample = v < 0 ? -1 : (v > 0 ? 1 : 0); }
    o.resident_cus = env_int("AVS_CG_RESIDENT_CUS", 0);
    o.resident_equal_lanes = getenv("AVS_CG_RESIDENT_EQUAL_LANES") != nullptr;
    o.resident_max_global = env_int("AVS_CG_RESIDENT_MAX_GLOBAL", 3);
    o.resident_max_quads = env_int("AVS_CG_RESIDENT_MAX_QUADS", 0);
    o.resident_no_stream = getenv("AVS_CG_RESIDENT_NO_STREAM") != nullptr;
    if (const char *e = getenv("AVS_CG_RESIDENT_REMAP_CHUNK")) o.resident_remap_chunk = atoll(e);
    o.resident_lane_fill = env_dbl("AVS_CG_RESIDENT_LANE_FILL", 0.90);
    o.resident_remote_cost = env_dbl("AVS_CG_RESIDENT_REMOTE_COST", 3.0);
    o.resident_stream_cost = env_dbl("AVS_CG_RESIDENT_STREAM_COST", 1.5);
    o.resident_coherent_fill = env_int("AVS_CG_RESIDENT_COHERENT_FILL", 1);
    o.resident_timers = env_int("AVS_CG_RESIDENT_TIMERS", 0);
    o.resident_verbose = env_int("AVS_CG_RESIDENT_VERBOSE", 0);
    if (const char *e = getenv("AVS_DIST_CG")) o.dist_standard_cg = strcmp(e, "standard") == 0;
    o.dist_overlap = env_int("AVS_DIST_OVERLAP", 1) != 0;
    o.dist_loopback = env_int("AVS_DIST_LOOPBACK", 0) != 0;
    if (const char *e = getenv("AVS_DIST_PLAN")) o.dist_host_plan = strcmp(e, "host") == 0;
    o.dist_selftest_rounds = env_int("AVS_DIST_SELFTEST_ROUNDS", 64);
    o.dist_split_rows = env_int("AVS_DIST_SPLIT_ROWS", 1) != 0;
    o.dist_plane_shift = env_int("AVS_DIST_PLANE_SHIFT", -1);
    o.trace_phases = env_int("AVS_TRACE_PHASES", 0);
    if (const char *e = getenv("AVS_DIST_TIMEOUT_MS")) o.dist_timeout_ms = atoll(e) > 0 ? atoll(e) : 0;
    return o;
}
static thread_local const Options *tl_opt = nullptr;
static thread_local Options tl_fallback;
static thread_local bool tl_fallback_ready = false;
const Options &cur_opt()
{
    if (tl_opt) return *tl_opt;
    // an entry point without a context (avs_pcg_csr, measurement entries): the environment as it is when the entry starts
    // (OptScope(nullptr) re-reads it), not once per thread for the life of the process
    if (!tl_fallback_ready) { tl_fallback = options_from_env(); tl_fallback_ready = true; }
    return tl_fallback;
}
OptScope::OptScope(const ::avs_ctx *c) : prev(tl_opt)
{
    if (c) tl_opt = &c->opt;
    else if (!prev) { tl_fallback = options_from_env(); tl_fallback_ready = true; }
    // hipGetLastError() is per THREAD, not per library: the host application (or torch in the tests: pointer-attribute probes of host
    // memory leave "invalid argument" behind) may have left an error that is none of ours.  Every outermost entry of the C ABI starts clean,
    // so that the checks behind our own launches report our own launches.
    if (!prev) (void)hipGetLastError();
}
OptScope::~OptScope() { tl_opt = prev; }

static thread_local std::atomic<int> *tl_cancel = nullptr;
CancelScope::CancelScope(std::atomic<int> *flag) : prev(tl_cancel) { tl_cancel = flag; }
CancelScope::~CancelScope() { tl_cancel = prev; }
bool cancel_requested() { return tl_cancel && tl_cancel->load(std::memory_order_acquire) != 0; }
bool cancel_consume() { return tl_cancel && tl_cancel->exchange(0, std::memory_order_acq_rel) != 0; }
} // namespace avs

avs::PyramidView avs_ctx::view() const
{
    PyramidView P{};
    P.levels = desc.levels;
    P.n[0] = desc.nx;
    P.n[1] = desc.ny;
    P.n[2] = desc.nz;
    P.enhanced = desc.use_enhanced_gradients;
    P.f32 = desc.precision == AVS_PRECISION_F32;
    P.dx = desc.dx;
    P.dt = desc.dt;
    for (int l = 0; l < AVS_MAX_LEVELS; ++l) {
        P.labels[l] = labels[l].p;
        P.cidx[l] = cidx[l].p;
        for (int a = 0; a < 3; ++a) {
            P.vidx[l][a] = vidx[l][a].p;
            P.eidx[l][a] = eidx[l][a].p;
        }
    }
    auto fv = [](const Field &f) { return FieldView{f.buf.p, f.cval, f.is_const ? 1 : 0}; };
    P.centerw = fv(centerw);
    P.visc = fv(visc);
    P.dens = fv(dens);
    for (int a = 0; a < 3; ++a) {
        P.edgew[a] = fv(edgew[a]);
        P.facew[a] = fv(facew[a]);
        P.vel[a] = fv(vel[a]);
        P.solidvel[a] = fv(solidvel[a]);
    }
    return P;
}

extern "C" {

const char *avs_last_error(void) { return avs::g_err; }
const char *avs_version(void) { return "avs-mi355x 0.2 (gfx950)"; }
int32_t avs_abi_version(void) { return AVS_ABI_VERSION; }

avs_status avs_create(const avs_desc *d, avs_ctx **out)
{
    AVS_REQUIRE(d && out, AVS_EINVAL, "null argument");
    AVS_REQUIRE(is_pow2(d->nx) && is_pow2(d->ny) && is_pow2(d->nz), AVS_EINVAL,
                "base resolution must be a power of two per axis (got %d %d %d)", d->nx, d->ny, d->nz);
    AVS_REQUIRE(d->levels >= 1 && d->levels <= AVS_MAX_LEVELS, AVS_EINVAL, "levels must be in [1, %d]", AVS_MAX_LEVELS);
    AVS_REQUIRE((d->nx >> (d->levels - 1)) >= 1 && (d->ny >> (d->levels - 1)) >= 1 && (d->nz >> (d->levels - 1)) >= 1,
                AVS_EINVAL, "too many levels for this resolution");
    AVS_REQUIRE(d->dx > 0. && std::isfinite(d->dx) && std::isfinite(d->dt), AVS_EINVAL, "dx must be positive and finite");
    AVS_REQUIRE(d->field_nx >= 0 && d->field_nx <= d->nx && d->field_ny >= 0 && d->field_ny <= d->ny && d->field_nz >= 0 &&
                    d->field_nz <= d->nz,
                AVS_EINVAL, "field resolution %d %d %d must lie in [0, octree resolution]", d->field_nx, d->field_ny, d->field_nz);
    AVS_REQUIRE(d->precision == AVS_PRECISION_F64 || d->precision == AVS_PRECISION_F32, AVS_EINVAL, "precision must be AVS_PRECISION_F64 or AVS_PRECISION_F32");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    AVS_REQUIRE(e == hipSuccess && ndev > 0, AVS_EHIP, "no HIP device available (%s)", hipGetErrorString(e));
    AVS_REQUIRE(d->device >= 0 && d->device < ndev, AVS_EINVAL, "device %d out of range (%d devices)", d->device, ndev);
    AVS_HIP(hipSetDevice(d->device));
    avs_ctx *c = new (std::nothrow) avs_ctx();
    AVS_REQUIRE(c, AVS_ENOMEM, "out of host memory");
    c->desc = *d;
    c->opt = options_from_env(); // the environment is read here, once per context
    if (c->desc.field_nx == 0) c->desc.field_nx = d->nx;
    if (c->desc.field_ny == 0) c->desc.field_ny = d->ny;
    if (c->desc.field_nz == 0) c->desc.field_nz = d->nz;
    if (d->stream) c->stream = reinterpret_cast<hipStream_t>(d->stream);
    else {
        if (hipStreamCreate(&c->stream) != hipSuccess) { // blocking: ordered after work on the null stream
            delete c;
            set_error("hipStreamCreate failed");
            return AVS_EHIP;
        }
        c->own_stream = true;
    }
    // defaults = what an absent Houdini field means in the synthetic harness
    c->visc.cval = 1.f;
    c->dens.cval = 1.f;
    for (int a = 0; a < 3; ++a) c->facew[a].cval = 1.f;
    *out = c;
    return AVS_OK;
}

void avs_destroy(avs_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->desc.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    avs::dist_release(c);
    pcg_destroy(c->pcg);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static void invalidate(avs_ctx *c, bool tables)
{
    if (tables) c->tables_ready = false;
    c->stencils_ready = c->guess_ready = c->guess_partial = c->system_ready = c->solved = false;
}

avs_status avs_set_labels(avs_ctx *c, int32_t level, const int8_t *labels, avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && labels, AVS_EINVAL, "null argument");
    AVS_REQUIRE(level >= 0 && level < c->desc.levels, AVS_EINVAL, "level %d out of range", level);
    AVS_HIP(hipSetDevice(c->desc.device));
    int r[3];
    grid_res(c->desc, 2, level, 0, r);
    AVS_TRY(c->labels[level].alloc(vol3(r)));
    AVS_HIP(copy_in(c->labels[level].p, labels, vol3(r), where, c->stream));
    if (where == AVS_MEM_HOST) AVS_HIP(hipStreamSynchronize(c->stream));
    c->have_labels[level] = true;
    invalidate(c, false);
    return AVS_OK;
}

avs_status avs_set_index_field(avs_ctx *c, avs_index_kind kind, int32_t level, int32_t axis, const int32_t *idx,
                               avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && idx, AVS_EINVAL, "null argument");
    AVS_REQUIRE(level >= 0 && level < c->desc.levels, AVS_EINVAL, "level %d out of range", level);
    AVS_REQUIRE(kind == AVS_INDEX_CENTER || (axis >= 0 && axis < 3), AVS_EINVAL, "axis %d out of range", axis);
    AVS_HIP(hipSetDevice(c->desc.device));
    int r[3];
    LatBuf<int32_t> *buf;
    bool *have;
    switch (kind) {
    case AVS_INDEX_VELOCITY: grid_res(c->desc, 0, level, axis, r); buf = &c->vidx[level][axis]; have = &c->have_vidx[level][axis]; break;
    case AVS_INDEX_EDGE: grid_res(c->desc, 1, level, axis, r); buf = &c->eidx[level][axis]; have = &c->have_eidx[level][axis]; break;
    case AVS_INDEX_CENTER: grid_res(c->desc, 2, level, 0, r); buf = &c->cidx[level]; have = &c->have_cidx[level]; break;
    default: set_error("unknown index kind %d", (int)kind); return AVS_EINVAL;
    }
    AVS_TRY(buf->alloc(vol3(r)));
    AVS_HIP(copy_in(buf->p, idx, vol3(r) * sizeof(int32_t), where, c->stream));
    if (where == AVS_MEM_HOST) AVS_HIP(hipStreamSynchronize(c->stream));
    *have = true;
    invalidate(c, true);
    return AVS_OK;
}

avs_status avs_set_dof_counts(avs_ctx *c, int64_t nv, int64_t ne, int64_t nc)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(nv >= 0 && ne >= 0 && nc >= 0 && nv < INT32_MAX && ne < INT32_MAX && nc < INT32_MAX / 3, AVS_EINVAL,
                "DOF counts out of range");
    c->n_vel = nv;
    c->n_edge = ne;
    c->n_center = nc;
    invalidate(c, true);
    return AVS_OK;
}

} // extern "C"

namespace {
// lattice (sx, sy, sz) -> (rx, ry, rz) >= it: outside, the border value (replicate) or `fill`
template <typename T>
__global__ __launch_bounds__(256) void k_pad_lattice(const T *__restrict__ in, int sx, int sy, int sz, T *__restrict__ out, int rx, int ry, int rz,
                                                     int replicate, T fill)
{
    const size_t total = (size_t)rx * ry * rz;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int i = (int)(o % rx), j = (int)((o / rx) % ry), k = (int)(o / ((size_t)rx * ry));
        T v = fill;
        if (replicate || (i < sx && j < sy && k < sz)) {
            const int ci = i < sx ? i : sx - 1, cj = j < sy ? j : sy - 1, ck = k < sz ? k : sz - 1;
            v = in[(size_t)ci + (size_t)sx * ((size_t)cj + (size_t)sy * ck)];
        }
        out[o] = v;
    }
}
__global__ __launch_bounds__(256) void k_crop_lattice(const float *__restrict__ in, int rx, int ry, int rz, float *__restrict__ out, int sx, int sy,
                                                      int sz)
{
    const size_t total = (size_t)sx * sy * sz;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int i = (int)(o % sx), j = (int)((o / sx) % sy), k = (int)(o / ((size_t)sx * sy));
        out[o] = in[(size_t)i + (size_t)rx * ((size_t)j + (size_t)ry * k)];
    }
}
unsigned pad_grid(size_t n)
{
    const size_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b < 65536 ? b : 65536));
}
} // namespace

namespace avs {
avs_status pad_lattice_f32(const float *src, int sx, int sy, int sz, float *dst, int rx, int ry, int rz, bool replicate, float fill, hipStream_t st)
{
    hipLaunchKernelGGL(k_pad_lattice<float>, dim3(pad_grid((size_t)rx * ry * rz)), dim3(256), 0, st, src, sx, sy, sz, dst, rx, ry, rz,
                       replicate ? 1 : 0, fill);
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}
avs_status pad_lattice_i32(const int32_t *src, int sx, int sy, int sz, int32_t *dst, int rx, int ry, int rz, int32_t fill, hipStream_t st)
{
    hipLaunchKernelGGL(k_pad_lattice<int32_t>, dim3(pad_grid((size_t)rx * ry * rz)), dim3(256), 0, st, src, sx, sy, sz, dst, rx, ry, rz, 0, fill);
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}
avs_status crop_lattice_f32(const float *src, int rx, int ry, int rz, float *dst, int sx, int sy, int sz, hipStream_t st)
{
    hipLaunchKernelGGL(k_crop_lattice, dim3(pad_grid((size_t)sx * sy * sz)), dim3(256), 0, st, src, rx, ry, rz, dst, sx, sy, sz);
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}
} // namespace avs

extern "C" {

avs_status avs_set_scalar_field(avs_ctx *c, avs_field_kind kind, int32_t axis, const float *data, float constant,
                                avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    return avs::set_scalar_field_lattice(c, kind, axis, data, constant, where, false);
}

} // extern "C"

avs_status avs::set_scalar_field_lattice(avs_ctx *c, avs_field_kind kind, int32_t axis, const float *data, float constant, avs_memspace where,
                                         bool padded_lattice)
{
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_HIP(hipSetDevice(c->desc.device));
    int r[3];
    avs_ctx::Field *f = nullptr;
    const bool vec = (kind == AVS_FIELD_EDGE_WEIGHTS || kind == AVS_FIELD_FACE_WEIGHTS || kind == AVS_FIELD_VELOCITY ||
                      kind == AVS_FIELD_SOLID_VELOCITY);
    AVS_REQUIRE(!vec || (axis >= 0 && axis < 3), AVS_EINVAL, "axis %d out of range", axis);
    switch (kind) {
    case AVS_FIELD_CENTER_WEIGHTS: grid_res(c->desc, 2, 0, 0, r); f = &c->centerw; break;
    case AVS_FIELD_EDGE_WEIGHTS: grid_res(c->desc, 1, 0, axis, r); f = &c->edgew[axis]; break;
    case AVS_FIELD_FACE_WEIGHTS: grid_res(c->desc, 0, 0, axis, r); f = &c->facew[axis]; break;
    case AVS_FIELD_VISCOSITY: grid_res(c->desc, 2, 0, 0, r); f = &c->visc; break;
    case AVS_FIELD_DENSITY: grid_res(c->desc, 2, 0, 0, r); f = &c->dens; break;
    case AVS_FIELD_VELOCITY: grid_res(c->desc, 0, 0, axis, r); f = &c->vel[axis]; break;
    case AVS_FIELD_SOLID_VELOCITY: grid_res(c->desc, 0, 0, axis, r); f = &c->solidvel[axis]; break;
    default: set_error("unknown field kind %d", (int)kind); return AVS_EINVAL;
    }
    if (!data) {
        f->buf.release();
        f->is_const = true;
        f->cval = constant;
    } else {
        AVS_TRY(f->buf.alloc(vol3(r)));
        // the caller's array has the SIMULATION grid's lattice: r minus the padding HDK_OctreeGrid::init added
        const int s3[3] = {r[0] - (c->desc.nx - c->desc.field_nx), r[1] - (c->desc.ny - c->desc.field_ny),
                           r[2] - (c->desc.nz - c->desc.field_nz)};
        if (padded_lattice || (s3[0] == r[0] && s3[1] == r[1] && s3[2] == r[2])) {
            AVS_HIP(copy_in(f->buf.p, data, vol3(r) * sizeof(float), where, c->stream));
            if (where == AVS_MEM_HOST) AVS_HIP(hipStreamSynchronize(c->stream));
        } else {
            DevBuf<float> src;
            const float *sp = data;
            if (where == AVS_MEM_HOST) {
                AVS_TRY(src.alloc(vol3(s3)));
                AVS_HIP(copy_in(src.p, data, vol3(s3) * sizeof(float), where, c->stream));
                sp = src.p;
            }
            const bool replicate = (kind == AVS_FIELD_VISCOSITY || kind == AVS_FIELD_DENSITY);
            AVS_TRY(pad_lattice_f32(sp, s3[0], s3[1], s3[2], f->buf.p, r[0], r[1], r[2], replicate, 0.f, c->stream));
            AVS_HIP(hipStreamSynchronize(c->stream)); // src dies here
        }
        f->is_const = false;
    }
    invalidate(c, false);
    return AVS_OK;
}

uint64_t avs::next_buffer_id()
{
    static std::atomic<uint64_t> counter{0};
    return ++counter;
}

// avs_prepass_apply: the context takes references on the pre-pass's lattices (labels, index pyramids, weights on the padded octree
// lattices, regular-grid indices) -- what the avs_set_* entries do by copying, without the copies
avs_status avs::adopt_prepass_lattices(avs_ctx *c, const PrepassLoan &loan)
{
    AVS_REQUIRE(c && loan.levels == c->desc.levels, AVS_EINVAL, "loan does not match the context");
    AVS_HIP(hipSetDevice(c->desc.device));
    int r[3];
    auto fits = [&](size_t have, int kind, int level, int axis) {
        grid_res(c->desc, kind, level, axis, r);
        return have == vol3(r);
    };
    for (int l = 0; l < loan.levels; ++l) {
        AVS_REQUIRE(loan.labels[l] && loan.cidx[l] && fits(loan.labels[l]->n, 2, l, 0) && fits(loan.cidx[l]->n, 2, l, 0), AVS_EINVAL,
                    "lent cell lattices of level %d do not match the context", l);
        for (int a = 0; a < 3; ++a)
            AVS_REQUIRE(loan.vidx[l][a] && loan.eidx[l][a] && fits(loan.vidx[l][a]->n, 0, l, a) && fits(loan.eidx[l][a]->n, 1, l, a), AVS_EINVAL,
                        "lent index lattices of level %d axis %d do not match the context", l, a);
    }
    AVS_REQUIRE(loan.centerw && fits(loan.centerw->n, 2, 0, 0), AVS_EINVAL, "lent centre weights do not match the context");
    for (int a = 0; a < 3; ++a)
        AVS_REQUIRE(loan.edgew[a] && loan.facew[a] && fits(loan.edgew[a]->n, 1, 0, a) && fits(loan.facew[a]->n, 0, 0, a), AVS_EINVAL,
                    "lent weight lattices of axis %d do not match the context", a);
    AVS_REQUIRE(loan.counts[0] >= 0 && loan.counts[1] >= 0 && loan.counts[2] >= 0 && loan.counts[0] < INT32_MAX && loan.counts[1] < INT32_MAX &&
                    loan.counts[2] < INT32_MAX / 3, AVS_EINVAL, "DOF counts out of range");
    for (int l = 0; l < loan.levels; ++l) {
        c->labels[l].adopt(loan.labels[l]);
        c->have_labels[l] = true;
        c->cidx[l].adopt(loan.cidx[l]);
        c->have_cidx[l] = true;
        for (int a = 0; a < 3; ++a) {
            c->vidx[l][a].adopt(loan.vidx[l][a]);
            c->eidx[l][a].adopt(loan.eidx[l][a]);
            c->have_vidx[l][a] = c->have_eidx[l][a] = true;
        }
    }
    auto field = [](avs_ctx::Field &f, const std::shared_ptr<DevBuf<float>> &h) {
        f.buf.adopt(h);
        f.is_const = false;
    };
    field(c->centerw, loan.centerw);
    for (int a = 0; a < 3; ++a) {
        field(c->edgew[a], loan.edgew[a]);
        field(c->facew[a], loan.facew[a]);
    }
    c->n_vel = loan.counts[0];
    c->n_edge = loan.counts[1];
    c->n_center = loan.counts[2];
    invalidate(c, true);
    c->slab = loan.slab;
    for (int k = 0; k < 3; ++k) {
        c->wlist[k].adopt(loan.slab.on ? loan.wlist[k] : nullptr);
        c->n_window[k] = loan.slab.on ? loan.n_window[k] : 0;
    }
    if (loan.slab.on)
        AVS_REQUIRE(loan.dof[0] && loan.dof[1] && loan.dof[2] && loan.wlist[0] && loan.wlist[1] && loan.wlist[2], AVS_EINTERNAL,
                    "slab-local pre-pass without dof tables / window lists");
    if (loan.dof[0] && loan.dof[1] && loan.dof[2] && loan.dof[0]->n >= (size_t)c->n_vel * 4 && loan.dof[1]->n >= (size_t)c->n_edge * 4 &&
        loan.dof[2]->n >= (size_t)c->n_center * 4) { // the numbering pass wrote the dof tables: no sweep over the index lattices (build_dof_tables)
        c->vdof.adopt(loan.dof[0]);
        c->edof.adopt(loan.dof[1]);
        c->cdof.adopt(loan.dof[2]);
        c->tables_ready = true;
    }
    for (int a = 0; a < 3; ++a) AVS_TRY(adopt_regular_index_lattice(c, a, loan.ridx[a], loan.ridx_occ[a], loan.ridx_occ_tiles[a]));
    return AVS_OK;
}

extern "C" {

avs_status avs_build_stencils(avs_ctx *c)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_HIP(hipSetDevice(c->desc.device));
    return build_stencils(c);
}
avs_status avs_build_initial_guess(avs_ctx *c)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_HIP(hipSetDevice(c->desc.device));
    AVS_REQUIRE(!c->slab.on, AVS_ESTATE, "the context holds a slab-local pre-pass (this rank's window only): only avs_dist_assemble works on it");
    return build_initial_guess(c);
}
avs_status avs_build_system(avs_ctx *c)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_HIP(hipSetDevice(c->desc.device));
    AVS_REQUIRE(!c->slab.on, AVS_ESTATE, "the context holds a slab-local pre-pass (this rank's window only): only avs_dist_assemble works on it");
    return build_system(c);
}

avs_status avs_assemble(avs_ctx *c, avs_assembly_info *info)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_HIP(hipSetDevice(c->desc.device));
    AVS_REQUIRE(!c->slab.on, AVS_ESTATE, "the context holds a slab-local pre-pass (this rank's window only): only avs_dist_assemble works on it");
    Timer t(c->stream);
    t.start();
    AVS_TRY(build_stencils(c)); // includes the dof tables; scopes "Build Edge / Cell Stress Stencils" inside
    c->ainfo.stencil_ms = t.stop();
    t.start();
    {
        Scope sc("Interpolate Regular Grid Velocities at Octree Velocity Faces"); // cpp:516
        AVS_TRY(build_initial_guess(c));
    }
    c->ainfo.guess_ms = t.stop();
    {
        Scope sc("Build Octree Linear System"); // cpp:554 (+ setFromTriplets, cpp:613)
        t.start();
        AVS_TRY(build_system(c));
        c->ainfo.system_ms = t.stop();
        t.start();
        c->reordered = false;
        c->brick_shift = c->opt.brick_shift;
        if (c->brick_shift >= 0) AVS_TRY(build_reordered_system(c, c->brick_shift));
        c->ainfo.csr_ms = t.stop();
        if (c->opt.trace_phases) fprintf(stderr, "[avs assemble] stencils %.3f guess %.3f system %.3f csr %.3f ms\n", c->ainfo.stencil_ms, c->ainfo.guess_ms, c->ainfo.system_ms, c->ainfo.csr_ms);
    }
    c->ainfo.n_velocity = c->n_vel;
    c->ainfo.n_edge = c->n_edge;
    c->ainfo.n_center = c->n_center;
    c->ainfo.nnz = c->nnz;
    c->ainfo.raw_triplets = c->nraw;
    if (info) *info = c->ainfo;
    return AVS_OK;
}

int32_t avs_spmv_tile_rows(void) { return spmv_tile_rows(); }

avs_status avs_get_matrix_format(avs_ctx *c, avs_matrix_format *out)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && out, AVS_EINVAL, "null argument");
    // the struct has grown between ABI revisions: only the bytes the caller's header knows are written
    const int32_t size = out->struct_size;
    AVS_REQUIRE(size >= (int32_t)(offsetof(avs_matrix_format, reordered) + sizeof(int32_t)) && size <= 4096, AVS_EINVAL,
                "avs_matrix_format.struct_size must be set to sizeof(avs_matrix_format) before the call (got %d)", (int)size);
    avs_matrix_format f{};
    avs_matrix_format *fmt = &f;
    f.struct_size = (int32_t)sizeof(avs_matrix_format);
    auto hand_over = [&]() {
        memcpy(out, &f, (size_t)size < sizeof(f) ? (size_t)size : sizeof(f));
        out->struct_size = size;
    };
    if (!c->system_ready && dist_matrix_format(c, fmt)) { hand_over(); return AVS_OK; } // avs_dist_assemble: the rank's own rows
    AVS_REQUIRE(c->system_ready, AVS_ESTATE, "no system: call avs_assemble first");
    const bool vi = c->reordered && c->vi.table_size > 0;
    fmt->reordered = c->reordered ? 1 : 0;
    fmt->value_table_size = vi ? c->vi.table_size : 0;
    fmt->column_bits = vi ? c->vi.col_bits : 0;
    fmt->bytes_per_nonzero = vi ? c->vi.bytes_per_nonzero() : 12;
    fmt->tile_local_tables = vi && c->vi.tile_tables ? 1 : 0;
    fmt->column_windows = vi && c->vi.col_windows ? 1 : 0;
    const bool bk = c->reordered && c->brick.ready;
    fmt->brick_tiles = bk ? c->brick.ntiles : 0;
    fmt->brick_patterns = bk ? c->brick.patterns : 0;
    fmt->brick_pattern_rows = bk ? c->brick.regular_rows : 0;
    fmt->brick_bytes = bk ? c->brick.stored_bytes(c->n_vel) : 0;
    fmt->brick_walk = bk ? c->brick_view.walk : 0;
    fmt->brick_value_codes = bk && c->brick.vc ? 1 : 0;
    pcg_fused_state(c->pcg, &fmt->fused_vector_update, &fmt->fused_vector_faults);
    hand_over();
    return AVS_OK;
}

avs_status avs_get_assembly_info(avs_ctx *c, avs_assembly_info *info)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && info, AVS_EINVAL, "null argument");
    c->ainfo.n_velocity = c->n_vel;
    c->ainfo.n_edge = c->n_edge;
    c->ainfo.n_center = c->n_center;
    c->ainfo.nnz = c->nnz;
    c->ainfo.raw_triplets = c->nraw;
    *info = c->ainfo;
    return AVS_OK;
}

static CsrView csr_of(avs_ctx *c)
{
    CsrView A;
    A.n = c->n_vel;
    A.nnz = c->nnz;
    A.row_ptr = c->reordered ? c->p_row_ptr.p : c->row_ptr.p;
    A.col = c->reordered ? c->p_col.p : c->col.p;
    A.val = c->reordered ? c->p_val.p : c->val.p;
    if (c->reordered) c->vi.apply(A);
    A.no_precond = c->no_precond;
    A.brick = (c->reordered && c->brick.ready) ? &c->brick_view : nullptr;
    A.f32_vectors = c->desc.precision == AVS_PRECISION_F32 ? c->opt.f32_vectors : 0;
    return A;
}

__global__ __launch_bounds__(256) void k_narrow_f32(double *__restrict__ x, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = (double)(float)x[i];
}

avs_status avs_set_solver_option(avs_ctx *c, avs_solver_option option, int32_t value)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    switch (option) {
    case AVS_OPTION_PRECONDITIONER:
        AVS_REQUIRE(value == AVS_PRECONDITIONER_JACOBI || value == AVS_PRECONDITIONER_NONE, AVS_EINVAL, "preconditioner must be JACOBI or NONE");
        c->no_precond = value == AVS_PRECONDITIONER_NONE;
        return AVS_OK;
    case AVS_OPTION_RESIDENT_LOOP: c->opt.resident = value != 0; return AVS_OK;
    case AVS_OPTION_TRANSPORT:
        AVS_REQUIRE(value >= AVS_USE_TRANSPORT_AUTO && value <= AVS_USE_TRANSPORT_DIRECT, AVS_EINVAL, "transport must be AUTO, RCCL or DIRECT");
        c->opt.transport = value;
        return AVS_OK;
    case AVS_OPTION_PARANOID: c->opt.paranoid = value != 0; return AVS_OK;
    case AVS_OPTION_GRAPH_REPLAY: c->opt.graph = value != 0; return AVS_OK;
    case AVS_OPTION_BRICK_FORM:
        AVS_REQUIRE(value >= AVS_BRICK_AUTO && value <= AVS_BRICK_TUNE, AVS_EINVAL, "brick form must be AUTO, NEVER, ALWAYS or TUNE");
        c->opt.brick = value;
        c->brick_verdict_rows = 0; // decided again at the next assembly
        return AVS_OK;
    case AVS_OPTION_FUSED_SCALAR_STEPS: c->opt.fuse_beta = value != 0; return AVS_OK;
    case AVS_OPTION_RELOAD_ENVIRONMENT: c->opt = options_from_env(); c->brick_verdict_rows = 0; return AVS_OK;
    case AVS_OPTION_F32_VECTORS: c->opt.f32_vectors = value < 0 ? -1 : (value > 0 ? 1 : 0); return AVS_OK;
    case AVS_OPTION_FUSED_VECTOR_UPDATE: c->opt.fuse_vectors = value < 0 ? -1 : (value > 0 ? 1 : 0); return AVS_OK;
    }
    set_error("unknown solver option %d", (int)option);
    return AVS_EINVAL;
}

extern "C++" {
namespace avs {
void narrow_solution_if_f32(avs_ctx *c, double *x, int64_t n)
{
    if (c->desc.precision == AVS_PRECISION_F32 && n > 0) // viscositySolution is an Eigen::VectorXf there: what the transfer reads are float values
        hipLaunchKernelGGL(k_narrow_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, x, n);
}
} // namespace avs
} // extern "C++"

avs_status avs_cancel(avs_ctx *c)
{
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    // no HIP call, no lock: this runs on another thread than the solve it ends (the reference polls opInterrupt() the same way, cpp:2528)
    c->cancel.store(1, std::memory_order_release);
    return AVS_OK;
}

avs_status avs_cancel_clear(avs_ctx *c)
{
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    c->cancel.store(0, std::memory_order_release); // a request that arrived after the solve it was meant for must not end the next one
    return AVS_OK;
}

avs_status avs_solve(avs_ctx *c, double tol, int32_t max_iters, avs_solve_info *info)
{
    avs::OptScope opt_scope_(c);
    avs::CancelScope cancel_scope_(c ? &c->cancel : nullptr);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->system_ready, AVS_ESTATE, "avs_assemble must succeed before avs_solve");
    AVS_REQUIRE(tol >= 0. && max_iters >= 0, AVS_EINVAL, "tolerance / max_iterations out of range");
    AVS_HIP(hipSetDevice(c->desc.device));
    Scope scope("Solve Linear System"); // cpp:603
    const int64_t n = c->n_vel;
    if (c->pcg && pcg_rows(c->pcg) != n) { // the context was re-assembled with another DOF count (next frame)
        pcg_destroy(c->pcg);
        c->pcg = nullptr;
    }
    if (c->pcg == nullptr) AVS_TRY(pcg_create(&c->pcg, n, n, c->stream));
    AVS_TRY(c->x.alloc((size_t)n));
    avs_solve_info local{};
    // solveWithGuess(rhs, viscositySolution): warm start from the restricted velocity (cpp:627)
    if (c->reordered) {
        AVS_TRY(c->p_x.alloc((size_t)n));
        AVS_HIP(hipMemcpyAsync(c->p_x.p, c->p_x0.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        AVS_TRY(pcg_solve(c->pcg, csr_of(c), c->p_rhs.p, c->p_x.p, tol, max_iters, c->stream, &local, nullptr));
        AVS_TRY(unpermute(c, c->p_x.p, c->x.p)); // back to the reference's DOF numbering
    } else {
        AVS_HIP(hipMemcpyAsync(c->x.p, c->x0.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        AVS_TRY(pcg_solve(c->pcg, csr_of(c), c->rhs.p, c->x.p, tol, max_iters, c->stream, &local, nullptr));
    }
    narrow_solution_if_f32(c, c->x.p, n);
    if (info) *info = local;
    c->solved = true;
    return AVS_OK;
}

static avs_status get_vec(avs_ctx *c, const double *src, int64_t have, double *dst, int64_t n, avs_memspace where)
{
    AVS_REQUIRE(dst, AVS_EINVAL, "null argument");
    AVS_REQUIRE(n == have, AVS_EINVAL, "vector length mismatch: caller passed %lld, system has %lld", (long long)n, (long long)have);
    AVS_HIP(hipSetDevice(c->desc.device));
    AVS_HIP(copy_out(dst, src, (size_t)n * sizeof(double), where, c->stream));
    AVS_HIP(hipStreamSynchronize(c->stream));
    return AVS_OK;
}

avs_status avs_get_solution(avs_ctx *c, double *x, int64_t n, avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->solved, AVS_ESTATE, "no solution: call avs_solve first");
    return get_vec(c, c->x.p, c->n_vel, x, n, where);
}

avs_status avs_set_solution(avs_ctx *c, const double *x, int64_t n, avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && x, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->n_vel >= 0 && n == c->n_vel, AVS_EINVAL, "vector length %lld does not match the %lld velocity DOFs", (long long)n, (long long)c->n_vel);
    AVS_HIP(hipSetDevice(c->desc.device));
    AVS_TRY(c->x.alloc((size_t)(n > 0 ? n : 1)));
    AVS_HIP(copy_in(c->x.p, x, (size_t)n * sizeof(double), where, c->stream));
    AVS_HIP(hipStreamSynchronize(c->stream));
    narrow_solution_if_f32(c, c->x.p, n);
    c->solved = true;
    return AVS_OK;
}

avs_status avs_get_initial_guess(avs_ctx *c, double *x0, int64_t n, avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->guess_ready, AVS_ESTATE, "no initial guess: call avs_build_initial_guess / avs_assemble first");
    return get_vec(c, c->x0.p, c->n_vel, x0, n, where);
}

avs_status avs_get_csr(avs_ctx *c, int32_t *row_ptr, int32_t *col, double *val, double *rhs, avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->system_ready, AVS_ESTATE, "no system: call avs_assemble first");
    AVS_HIP(hipSetDevice(c->desc.device));
    hipStream_t s = c->stream;
    if (row_ptr) AVS_HIP(copy_out(row_ptr, c->row_ptr.p, ((size_t)c->n_vel + 1) * sizeof(int32_t), where, s));
    if (col) AVS_HIP(copy_out(col, c->col.p, (size_t)c->nnz * sizeof(int32_t), where, s));
    if (val) AVS_HIP(copy_out(val, c->val.p, (size_t)c->nnz * sizeof(double), where, s));
    if (rhs) AVS_HIP(copy_out(rhs, c->rhs.p, (size_t)c->n_vel * sizeof(double), where, s));
    AVS_HIP(hipStreamSynchronize(s));
    return AVS_OK;
}

static avs_status get_stencils(avs_ctx *c, bool edge, int32_t *cnt, int32_t *idx, double *coef, int32_t *bcnt,
                               double *bval, double *weight, avs_memspace where)
{
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->stencils_ready, AVS_ESTATE, "no stencils: call avs_build_stencils / avs_assemble first");
    AVS_HIP(hipSetDevice(c->desc.device));
    hipStream_t s = c->stream;
    const size_t ns = edge ? (size_t)c->n_edge : (size_t)c->n_center * 3;
    const size_t nw = edge ? (size_t)c->n_edge : (size_t)c->n_center;
    const size_t cap = edge ? AVS_EDGE_STENCIL_CAP : AVS_CENTER_STENCIL_CAP;
    const size_t bcap = edge ? AVS_EDGE_BOUNDARY_CAP : AVS_CENTER_BOUNDARY_CAP;
    AVS_TRY(pad_stencils(c, edge));
    if (cnt) AVS_HIP(copy_out(cnt, edge ? c->e_cnt.p : c->c_cnt.p, ns * sizeof(int32_t), where, s));
    if (idx) AVS_HIP(copy_out(idx, edge ? c->e_idx.p : c->c_idx.p, ns * cap * sizeof(int32_t), where, s));
    if (coef) AVS_HIP(copy_out(coef, edge ? c->e_coef.p : c->c_coef.p, ns * cap * sizeof(double), where, s));
    if (bcnt) AVS_HIP(copy_out(bcnt, edge ? c->e_bcnt.p : c->c_bcnt.p, ns * sizeof(int32_t), where, s));
    if (bval) AVS_HIP(copy_out(bval, edge ? c->e_bval.p : c->c_bval.p, ns * bcap * sizeof(double), where, s));
    if (weight) AVS_HIP(copy_out(weight, edge ? c->e_w.p : c->c_w.p, nw * sizeof(double), where, s));
    AVS_HIP(hipStreamSynchronize(s));
    return AVS_OK;
}

avs_status avs_get_edge_stencils(avs_ctx *c, int32_t *cnt, int32_t *idx, double *coef, int32_t *bcnt, double *bval,
                                 double *weight, avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    return get_stencils(c, true, cnt, idx, coef, bcnt, bval, weight, where);
}
avs_status avs_get_center_stencils(avs_ctx *c, int32_t *cnt, int32_t *idx, double *coef, int32_t *bcnt, double *bval,
                                   double *weight, avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    return get_stencils(c, false, cnt, idx, coef, bcnt, bval, weight, where);
}

// ---------------------------------------------------------------------------------------------
// seam A: solve only
// ---------------------------------------------------------------------------------------------
avs_status avs_pcg_csr(int64_t n, const int32_t *row_ptr, const int32_t *col, const double *val, const double *b,
                       double *x, double tol, int32_t max_iters, avs_memspace where, int32_t device, void *stream,
                       avs_solve_info *info)
{
    avs::OptScope opt_scope_(nullptr); // context-free entry: the AVS_* environment of this call
    AVS_REQUIRE(n >= 0 && row_ptr && b && x && (n == 0 || (col && val)), AVS_EINVAL, "null argument");
    AVS_REQUIRE(tol >= 0. && max_iters >= 0, AVS_EINVAL, "tolerance / max_iterations out of range");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    AVS_REQUIRE(e == hipSuccess && ndev > 0, AVS_EHIP, "no HIP device available (%s)", hipGetErrorString(e));
    AVS_REQUIRE(device >= 0 && device < ndev, AVS_EINVAL, "device %d out of range", device);
    AVS_HIP(hipSetDevice(device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    bool own = false;
    if (!s) {
        AVS_HIP(hipStreamCreate(&s));
        own = true;
    }
    avs_status rc = AVS_OK;
    PcgWork *w = nullptr;
    {
        DevBuf<int32_t> d_rp, d_col;
        DevBuf<double> d_val, d_b, d_x;
        ValueIndex vi;
        CsrView A;
        A.n = n;
        const double *bp = b;
        double *xp = x;
        int32_t nnz = 0;
        do {
            if (where == AVS_MEM_HOST) {
                nnz = row_ptr[n];
                if ((rc = d_rp.alloc((size_t)n + 1)) || (rc = d_col.alloc((size_t)nnz)) || (rc = d_val.alloc((size_t)nnz)) ||
                    (rc = d_b.alloc((size_t)n)) || (rc = d_x.alloc((size_t)n)))
                    break;
                if (hipMemcpyAsync(d_rp.p, row_ptr, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
                    hipMemcpyAsync(d_col.p, col, (size_t)nnz * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
                    hipMemcpyAsync(d_val.p, val, (size_t)nnz * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
                    hipMemcpyAsync(d_b.p, b, (size_t)n * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
                    hipMemcpyAsync(d_x.p, x, (size_t)n * 8, hipMemcpyHostToDevice, s) != hipSuccess) {
                    set_error("host -> device copy failed");
                    rc = AVS_EHIP;
                    break;
                }
                A.row_ptr = d_rp.p; A.col = d_col.p; A.val = d_val.p;
                bp = d_b.p; xp = d_x.p;
            } else {
                if (hipMemcpyAsync(&nnz, row_ptr + n, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess) {
                    set_error("device -> host copy failed");
                    rc = AVS_EHIP;
                    break;
                }
                A.row_ptr = row_ptr; A.col = col; A.val = val;
            }
            A.nnz = nnz;
            // the same lossless stream compression the assembled path gets (value dictionary, packed words); the
            // caller's numbering is kept, so only the x locality of the brick-major order is missing
            if (nnz > 0) {
                if ((rc = build_matrix_index(A.row_ptr, A.col, A.val, n, nnz, n, vi, s))) break;
                vi.apply(A);
            }
            if ((rc = pcg_create(&w, n, n, s))) break;
            if ((rc = pcg_solve(w, A, bp, xp, tol, max_iters, s, info, nullptr))) break;
            if (where == AVS_MEM_HOST) {
                if (hipMemcpyAsync(x, xp, (size_t)n * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess) {
                    set_error("device -> host copy failed");
                    rc = AVS_EHIP;
                }
            }
        } while (0);
        (void)hipStreamSynchronize(s);
    }
    pcg_destroy(w);
    if (own) (void)hipStreamDestroy(s);
    return rc;
}

#ifdef AVS_PROBES // measurement / test entries (include/avs_probe.h): compiled into libavs_probe.so only
avs_status avs_spmv_csr(int64_t n, const int32_t *row_ptr, const int32_t *col, const double *val, const double *x,
                        double *y, int32_t variant, int32_t repeats, void *stream)
{
    AVS_REQUIRE(n >= 0 && row_ptr && x && y, AVS_EINVAL, "null argument");
    CsrView A;
    A.n = n;
    A.nnz = 0;
    A.row_ptr = row_ptr;
    A.col = col;
    A.val = val;
    for (int i = 0; i < (repeats > 0 ? repeats : 1); ++i)
        AVS_TRY(spmv_launch(A, x, y, variant, reinterpret_cast<hipStream_t>(stream)));
    return AVS_OK;
}

namespace {
__global__ void k_count_bit_diff(int64_t n, const double *__restrict__ a, const double *__restrict__ b, unsigned long long *cnt)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && __double_as_longlong(a[i]) != __double_as_longlong(b[i])) atomicAdd(cnt, 1ull);
}
} // namespace

avs_status avs_bench_spmv(avs_ctx *c, int32_t variant, int32_t repeats, double *ms_per_launch)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && ms_per_launch, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->system_ready, AVS_ESTATE, "no system: call avs_assemble first");
    AVS_REQUIRE(repeats > 0, AVS_EINVAL, "repeats must be positive");
    AVS_HIP(hipSetDevice(c->desc.device));
    const int64_t n = c->n_vel;
    DevBuf<double> y;
    AVS_TRY(y.alloc((size_t)n));
    CsrView A = csr_of(c);
    Timer t(c->stream);
    const double *xin = c->reordered ? c->p_x0.p : c->x0.p;
    const bool dot = variant >= 100; // 100 + id: the fused-dot form the solver launches
    if (dot) variant -= 100;
    if (variant == 61) { A.brick = nullptr; variant = 0; } // the value-indexed stream kernel where the brick form would be used
    DevBuf<double> partial;
    if (dot) AVS_TRY(partial.alloc(spmv_partial_elems(n)));
    AVS_TRY(dot ? spmv_dot_launch(A, xin, y.p, partial.p, variant, c->stream) : spmv_launch(A, xin, y.p, variant, c->stream)); // warm-up
    const char *thrash = getenv("AVS_BENCH_THRASH_MB"); // > 0: that many MB are overwritten between the launches (a COLD launch, as
                                                         // inside the PCG loop, whose vector kernels move ~0.6 GB between two products)
    if (thrash && atoi(thrash) > 0) {
        DevBuf<char> junk;
        const size_t jb = (size_t)atoi(thrash) << 20;
        AVS_TRY(junk.alloc(jb));
        double acc = 0.;
        for (int i = 0; i < repeats; ++i) {
            AVS_HIP(hipMemsetAsync(junk.p, i & 0xff, jb, c->stream));
            t.start();
            AVS_TRY(dot ? spmv_dot_launch(A, xin, y.p, partial.p, variant, c->stream) : spmv_launch(A, xin, y.p, variant, c->stream));
            acc += t.stop();
        }
        *ms_per_launch = acc / repeats;
    } else {
        t.start();
        for (int i = 0; i < repeats; ++i)
            AVS_TRY(dot ? spmv_dot_launch(A, xin, y.p, partial.p, variant, c->stream) : spmv_launch(A, xin, y.p, variant, c->stream));
        *ms_per_launch = t.stop() / repeats;
    }
    // every variant must reproduce the plain 12-B CSR kernel bit for bit (same products, same left-to-right row sums)
    if (n > 0) {
        DevBuf<double> yref;
        DevBuf<unsigned long long> cnt;
        AVS_TRY(yref.alloc((size_t)n));
        AVS_TRY(cnt.alloc(1));
        AVS_HIP(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), c->stream));
        CsrView P = A;
        P.codes = nullptr;
        P.packed = nullptr;
        AVS_TRY(spmv_launch(P, xin, yref.p, 14, c->stream));
        hipLaunchKernelGGL(k_count_bit_diff, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, y.p, yref.p, cnt.p);
        unsigned long long h = 0;
        AVS_HIP(hipMemcpyAsync(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        AVS_HIP(hipStreamSynchronize(c->stream));
        // (phase switches of the brick kernel -- AVS_BRICK_DEBUG & 7, measurement builds only -- produce wrong rows on purpose)
        const char *dbg = getenv("AVS_BRICK_DEBUG");
        if (h && dbg && (atoi(dbg) & 7)) fprintf(stderr, "(AVS_BRICK_DEBUG=%s: %llu rows differ, as expected)\n", dbg, h);
        else if (h) { set_error("SpMV variant %d differs from the plain CSR kernel in %llu rows", variant, h); return AVS_EINTERNAL; }
    }
    return AVS_OK;
}


namespace {
__global__ void k_probe_gather(int64_t n, const double *__restrict__ src, const int32_t *__restrict__ idx, double *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_probe_scatter(int64_t n, const double *__restrict__ src, const int32_t *__restrict__ idx, double *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = src[i];
}
} // namespace

extern "C++" {
namespace avs {
// y = A x through the solver's form (+ the folded partial sums of x.y of the fused-dot instantiation)
avs_status probe_spmv_form(const CsrView &A, const double *x, double *y, bool fused, double *dot_out, hipStream_t st)
{
    if (A.f32_vectors) return spmv_f32_probe(A, x, y, fused, dot_out, st); // AVS_PRECISION_F32: the float kernels the solve launches
    if (!fused) return spmv_launch(A, x, y, 0, st);
    DevBuf<double> partial;
    size_t np = spmv_partial_elems(A.n);
    if (A.brick && A.brick->ntiles > 0 && (size_t)brick_partial_count(*A.brick, 8) > np) np = (size_t)brick_partial_count(*A.brick, 8);
    AVS_TRY(partial.alloc(np));
    AVS_HIP(hipMemsetAsync(partial.p, 0, np * sizeof(double), st));
    AVS_TRY(spmv_dot_launch(A, x, y, partial.p, 0, st));
    if (dot_out) {
        std::vector<double> h(np);
        AVS_HIP(hipMemcpyAsync(h.data(), partial.p, np * sizeof(double), hipMemcpyDeviceToHost, st));
        AVS_HIP(hipStreamSynchronize(st));
        double s = 0.;
        for (double v : h) s += v; // (slots the launch did not write are zero)
        *dot_out = s;
    }
    return AVS_OK;
}
} // namespace avs
}

// Measurement: how evenly the pattern rows of a tile load the eight waves of its workgroup.  Lane i of the execution order walks row i
// (rows sorted by pattern length): a wave's time is its LONGEST row, the tile's walk the slowest wave.  Host-side pass over the form.
avs_status avs_brick_wave_stats(avs_ctx *c, double *out6)
{
    AVS_REQUIRE(c && c->system_ready, AVS_ESTATE, "no system");
    const avs::BrickView &B = c->brick_view;
    AVS_REQUIRE(B.ntiles > 0, AVS_ESTATE, "no brick form");
    AVS_HIP(hipSetDevice(c->desc.device));
    std::vector<uint2> tb((size_t)B.ntiles);
    AVS_HIP(hipMemcpy(tb.data(), B.tile_blk, tb.size() * sizeof(uint2), hipMemcpyDeviceToHost));
    size_t units = 0;
    for (const uint2 &t : tb) units = std::max(units, (size_t)t.x + t.y);
    std::vector<uint32_t> blocks(units * 4);
    AVS_HIP(hipMemcpy(blocks.data(), B.blocks, blocks.size() * 4, hipMemcpyDeviceToHost));
    size_t nrd = 0;
    for (const uint2 &t : tb) { const uint32_t *bw = &blocks[(size_t)t.x * 4]; nrd = std::max(nrd, (size_t)bw[10] + bw[5]); }
    std::vector<uint2> rdesc(nrd);
    AVS_HIP(hipMemcpy(rdesc.data(), B.rdesc, nrd * sizeof(uint2), hipMemcpyDeviceToHost));
    double sum_all = 0., sum_max = 0., sum_mean = 0., sum_lanes = 0., sum_rows = 0., sum_quads = 0.;
    long run_hist[17] = {}, runs_total = 0, groups4 = 0, groups8 = 0, tiles_all = 0;
    double sum_nrows = 0.;
    for (const uint2 &t : tb) { // fill runs of every tile: lengths, and how many 4- / 8-lane groups would cover them
        const uint32_t *bw = &blocks[(size_t)t.x * 4];
        const int nruns = (int)bw[3];
        ++tiles_all;
        sum_nrows += bw[1];
        for (int q = 0; q < nruns; ++q) {
            const int len = (int)(bw[avs::kBlkHdrWords + 2 * q + 1] & 15u) + 1;
            run_hist[len]++;
            ++runs_total;
            groups4 += (len + 3) / 4;
            groups8 += (len + 7) / 8;
        }
    }
    fprintf(stderr, "brick fill runs: %ld tiles, %.1f rows per tile, %.1f runs per tile (16-lane groups), %.1f 8-lane groups, %.1f 4-lane groups per tile; lengths:", tiles_all,
            sum_nrows / tiles_all, (double)runs_total / tiles_all, (double)groups8 / tiles_all, (double)groups4 / tiles_all);
    for (int l = 1; l <= 16; ++l) fprintf(stderr, " %d:%.1f%%", l, 100. * run_hist[l] / (double)(runs_total ? runs_total : 1));
    fprintf(stderr, "\n");
    long gt = 0;
    std::vector<long> hist(40, 0);
    for (const uint2 &t : tb) {
        const uint32_t *bw = &blocks[(size_t)t.x * 4];
        const int npat = (int)bw[2], nruns = (int)bw[3], npq = (int)bw[4], nprow = (int)bw[5], rd0 = (int)bw[10];
        if (npat == 0 || nprow == 0) continue;
        const uint32_t *pinfo = bw + avs::kBlkHdrWords + 2 * nruns + npq;
        int wmax[8] = {};   // quads a wave walks: sum over its (<= 2) row chunks of the chunk's longest row
        for (int k = 0; k * 512 < nprow; ++k)
            for (int w = 0; w < 8; ++w) {
                int m = 0;
                for (int l = 0; l < 64; ++l) {
                    const int i = k * 512 + w * 64 + l;
                    if (i >= nprow) break;
                    const int nq = (int)((pinfo[rdesc[(size_t)rd0 + i].x >> 20] >> 16) & 0x7fffu);
                    m = std::max(m, nq);
                    sum_quads += nq;
                }
                wmax[w] += m;
            }
        int mx = 0, tot = 0;
        for (int w = 0; w < 8; ++w) { mx = std::max(mx, wmax[w]); tot += wmax[w]; }
        sum_max += mx;
        sum_mean += tot / 8.;
        sum_rows += nprow;
        ++gt;
        hist[std::min(39, mx)]++;
    }
    fprintf(stderr, "brick wave stats: %ld G tiles, rows per tile %.1f, quads per row %.2f, per tile: slowest wave %.2f quads, mean wave %.2f quads (ratio %.2f)\n",
            gt, sum_rows / gt, sum_quads / sum_rows, sum_max / gt, sum_mean / gt, sum_max / sum_mean);
    fprintf(stderr, "  slowest-wave quads histogram:");
    for (int i = 0; i < 40; ++i) if (hist[i]) fprintf(stderr, " %d:%ld", i, hist[i]);
    fprintf(stderr, "\n");
    if (out6) { out6[0] = (double)gt; out6[1] = sum_rows / gt; out6[2] = sum_quads / sum_rows; out6[3] = sum_max / gt; out6[4] = sum_mean / gt; out6[5] = sum_all + sum_lanes; }
    return AVS_OK;
}

avs_status avs_spmv_solver_form(avs_ctx *c, const double *x, double *y, int32_t fused_dot, double *dot_out)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && x && y, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->system_ready, AVS_ESTATE, "no system: call avs_assemble first");
    AVS_HIP(hipSetDevice(c->desc.device));
    const int64_t n = c->n_vel;
    if (n == 0) return AVS_OK;
    CsrView A = csr_of(c);
    const unsigned g = (unsigned)((n + 255) / 256);
    if (!c->reordered) {
        AVS_TRY(probe_spmv_form(A, x, y, fused_dot != 0, dot_out, c->stream));
    } else {
        DevBuf<double> xp, yp;
        AVS_TRY(xp.alloc((size_t)n));
        AVS_TRY(yp.alloc((size_t)n));
        hipLaunchKernelGGL(k_probe_gather, dim3(g), dim3(256), 0, c->stream, n, x, c->perm.p, xp.p);       // xp[new] = x[perm[new]]
        AVS_TRY(probe_spmv_form(A, xp.p, yp.p, fused_dot != 0, dot_out, c->stream));
        hipLaunchKernelGGL(k_probe_scatter, dim3(g), dim3(256), 0, c->stream, n, yp.p, c->perm.p, y);      // y[perm[new]] = yp[new]
        AVS_HIP(hipGetLastError());
        AVS_HIP(hipStreamSynchronize(c->stream));
    }
    AVS_HIP(hipStreamSynchronize(c->stream));
    return AVS_OK;
}

// Measured stream ceilings (mode 0: read-only 16 B/lane, 1: read-only non-temporal, 2: copy) on
// `bytes` of HBM; returns GB/s of bytes MOVED (copy counts read + write).
avs_status avs_bench_stream(int32_t mode, int64_t bytes, int32_t repeats, int32_t device, double *gbps)
{
    AVS_REQUIRE(gbps && bytes >= 1024 && repeats > 0, AVS_EINVAL, "bad argument");
    AVS_HIP(hipSetDevice(device));
    const int64_t n = bytes / 8;
    DevBuf<double> a, b, sink;
    AVS_TRY(a.alloc((size_t)n));
    AVS_TRY(sink.alloc(8));
    if (mode == 2) AVS_TRY(b.alloc((size_t)n));
    hipStream_t st;
    AVS_HIP(hipStreamCreate(&st));
    AVS_HIP(hipMemsetAsync(a.p, 0x11, (size_t)n * 8, st));
    Timer t(st);
    const int grid = 256 * 8;
    avs_status rc = stream_probe(mode, a.p, b.p, n, sink.p, grid, st);
    t.start();
    for (int i = 0; i < repeats && rc == AVS_OK; ++i) rc = stream_probe(mode, a.p, b.p, n, sink.p, grid, st);
    const double ms = t.stop() / repeats;
    (void)hipStreamDestroy(st);
    if (rc != AVS_OK) return rc;
    *gbps = (mode == 2 ? 2.0 : 1.0) * (double)n * 8.0 / (ms * 1e-3) / 1e9;
    return AVS_OK;
}
#endif // AVS_PROBES

} // extern "C"
