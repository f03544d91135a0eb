// HDK_AdaptiveViscosity_avs.cpp -- Houdini-side shim: the GAS micro-solver of rgoldade/AdaptiveViscositySolver with its
// whole numerical body (reference Source/HDK_AdaptiveViscosity.cpp:233-707, "cpp:" below) replaced by calls into libavs_hip.so
// (include/avs.h).  SOURCE ONLY: it needs the proprietary HDK (UT_*, SIM_*, GAS_* headers), which exists neither in this
// repository's build container nor on the GPU test boxes, so it has never been compiled or run -- SURVEY.md 8(f) #3.
//
// How a maintainer uses it: keep the reference's HDK_AdaptiveViscosity.h (class declaration, parameter getters h:28-41) and the
// parameter template / getDopDescription() part of the reference's .cpp (cpp:36-124); drop the rest of that .cpp and compile
// this file instead; link with -lavs_hip.  What stays of the reference per frame: fetching the named fields.  What moves to
// the GPU: integration weights, refinement mask, octree, classification + numbering (avs_prepass_*), stencils, assembly,
// Jacobi-PCG (avs_assemble / avs_solve), octree -> regular grid transfer (avs_transfer_to_regular_grid).
//
// Not carried over (out of scope): outputOctreeGeometry / "doPrintOctree", the debug unit tests of the octree.
#include "HDK_AdaptiveViscosity.h" // the reference's own header, unchanged

#include <GAS/GAS_SubSolver.h>
#include <SIM/SIM_Object.h>
#include <SIM/SIM_RawField.h>
#include <SIM/SIM_ScalarField.h>
#include <SIM/SIM_VectorField.h>
#include <UT/UT_DSOVersion.h>
#include <UT/UT_Interrupt.h>
#include <UT/UT_ParallelUtil.h>
#include <UT/UT_PerfMonAutoEvent.h>
#include <UT/UT_VoxelArray.h>
#include <UT/UT_WorkBuffer.h>

#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "avs.h"

// plugin registration: the standard HDK entry point (same role as cpp:20-24)
void initializeSIM(void *) { IMPLEMENT_DATAFACTORY(HDK_AdaptiveViscosity); }

HDK_AdaptiveViscosity::HDK_AdaptiveViscosity(const SIM_DataFactory *factory) : BaseClass(factory) {}
static void releaseHandlesOf(const void *node);
// a deleted solver node gives back its pre-pass and context (device lattices, PCG workspace: tens of GB at 1024^3) -- and a new node
// that happens to be allocated at the same address never inherits them
HDK_AdaptiveViscosity::~HDK_AdaptiveViscosity() { releaseHandlesOf(this); }

namespace {

// One dense x-fastest fp32 array per field: the layout of include/avs.h ("i + rx (j + ry k)").  A constant field (every
// tile constant with the same value) is handed over as a scalar -- the library's constant-field fast path (cpp:2090, 2248).
struct Flat {
    std::vector<float> data;
    float constant = 0.f;
    bool is_constant = false;
    int res[3] = {0, 0, 0};
    const float *ptr() const { return is_constant ? nullptr : data.data(); }
};

Flat flatten(const SIM_RawField &field)
{
    Flat f;
    const UT_VoxelArrayF *voxels = field.field();
    f.res[0] = voxels->getXRes();
    f.res[1] = voxels->getYRes();
    f.res[2] = voxels->getZRes();
    float c = 0.f;
    if (voxels->isConstant(&c)) {
        f.is_constant = true;
        f.constant = c;
        return f;
    }
    f.data.resize((size_t)f.res[0] * f.res[1] * f.res[2]);
    UTparallelFor(UT_BlockedRange<int>(0, f.res[2]), [&](const UT_BlockedRange<int> &range) {
        for (int k = range.begin(); k != range.end(); ++k)
            for (int j = 0; j < f.res[1]; ++j) {
                float *row = f.data.data() + (size_t)f.res[0] * ((size_t)j + (size_t)f.res[1] * k);
                for (int i = 0; i < f.res[0]; ++i) row[i] = voxels->getValue(i, j, k);
            }
    });
    return f;
}

void unflatten(const std::vector<float> &src, SIM_RawField &field)
{
    UT_VoxelArrayF *voxels = field.fieldNC();
    const int rx = voxels->getXRes(), ry = voxels->getYRes(), rz = voxels->getZRes();
    UTparallelFor(UT_BlockedRange<int>(0, rz), [&](const UT_BlockedRange<int> &range) {
        for (int k = range.begin(); k != range.end(); ++k)
            for (int j = 0; j < ry; ++j) {
                const float *row = src.data() + (size_t)rx * ((size_t)j + (size_t)ry * k);
                for (int i = 0; i < rx; ++i) voxels->setValue(i, j, k, row[i]);
            }
    });
}

int next_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// The two library handles, KEPT from substep to substep (round 5): avs_prepass_apply lends the pre-pass's lattices to the context
// instead of copying them, and an avs_prepass object that lives across frames skips what its allocations already hold from the
// previous frame (weight bricks far from the surface, index tiles outside the last occupancy) -- both only pay off when the objects
// survive the call.  One set per solver node (keyed by the node's address; released when the geometry of the grids changes, on any
// error return, and by the node's destructor).  A fresh pair per call, as before, is still correct: only slower.
struct Handles {
    avs_prepass *pp = nullptr;
    avs_ctx *ctx = nullptr;
    int oct[3] = {0, 0, 0}, sim[3] = {0, 0, 0}, want_levels = 0, ctx_levels = 0, n_super = 0, enhanced = -1;
    double dx = 0., extrapolation = 0., dt = 0.;
    void releaseCtx() { if (ctx) avs_destroy(ctx); ctx = nullptr; ctx_levels = 0; }
    void release() { releaseCtx(); if (pp) avs_prepass_destroy(pp); pp = nullptr; }
    ~Handles() { release(); }
};
// The map lives behind an intentionally leaked pointer: a static object would be destroyed at exit / dlclose, i.e. possibly AFTER the HIP
// runtime has been torn down, and avs_destroy / hipFree from there can crash or hang the shutdown.  Nodes release their own entry
// (destructor above); what a host never deletes is reclaimed with the process.
static std::mutex theHandlesLock;
static std::map<const void *, std::unique_ptr<Handles>> &handlesMap()
{
    static auto *m = new std::map<const void *, std::unique_ptr<Handles>>();
    return *m;
}
static Handles &handlesOf(const void *node)
{
    std::lock_guard<std::mutex> lk(theHandlesLock);
    std::unique_ptr<Handles> &h = handlesMap()[node];
    if (!h) h.reset(new Handles());
    return *h;
}
} // namespace
static void releaseHandlesOf(const void *node)
{
    std::lock_guard<std::mutex> lk(theHandlesLock);
    handlesMap().erase(node); // ~Handles: avs_destroy + avs_prepass_destroy
}
namespace {

// Optional frame export for the offline harness (adaptiveviscositysolver_amd/dump.py, examples/hotpath_from_dump.cpp):
// "AVSDUMP2", nx ny nz levels enhanced field_nx field_ny field_nz (int32), dx dt (f64), n_vel n_edge n_center (int64); per level
// labels int8, vidx[3], eidx[3], cidx int32 on the padded OCTREE lattices; then centre weights, edge weights[3], face weights[3],
// viscosity, density, velocity[3], solid velocity[3], each as int32 is_const + one float or the dense array on the SIMULATION grid's
// lattices (field_n*: what avs_set_scalar_field takes) -- so any frame can be replayed, not only power-of-two grids (round 2's
// AVSDUMP1 writer refused everything else, i.e. every real frame).
bool write_dump(const char *path, avs_prepass *pp, const avs_prepass_info &info, const int n[3], const int sim[3], double dx, double dt,
                bool enhanced, const Flat &face_w0, const Flat &face_w1, const Flat &face_w2, const Flat &visc, const Flat &dens,
                const Flat vel[3], const Flat solidvel[3])
{
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    auto put = [&](const void *p, size_t bytes) { return std::fwrite(p, 1, bytes, f) == bytes; };
    bool ok = put("AVSDUMP2", 8);
    const int32_t head[8] = {n[0], n[1], n[2], info.levels, enhanced ? 1 : 0, sim[0], sim[1], sim[2]};
    const double scal[2] = {dx, dt};
    const int64_t counts[3] = {info.n_velocity, info.n_edge, info.n_center};
    ok = ok && put(head, sizeof(head)) && put(scal, sizeof(scal)) && put(counts, sizeof(counts));
    auto lattice = [&](int kind, int level, int axis) {
        size_t r[3] = {(size_t)(n[0] >> level), (size_t)(n[1] >> level), (size_t)(n[2] >> level)};
        if (kind == 0) r[axis] += 1;
        if (kind == 1)
            for (int b = 0; b < 3; ++b) r[b] += (b != axis);
        return r[0] * r[1] * r[2];
    };
    std::vector<int8_t> lab;
    std::vector<int32_t> idx;
    for (int l = 0; l < info.levels && ok; ++l) {
        lab.resize(lattice(2, l, 0));
        ok = avs_prepass_get_labels(pp, l, lab.data(), AVS_MEM_HOST) == AVS_OK && put(lab.data(), lab.size());
        for (int kind = 0; kind < 2; ++kind)
            for (int a = 0; a < 3 && ok; ++a) {
                idx.resize(lattice(kind, l, a));
                ok = avs_prepass_get_index(pp, kind == 0 ? AVS_INDEX_VELOCITY : AVS_INDEX_EDGE, l, a, idx.data(), AVS_MEM_HOST) == AVS_OK &&
                     put(idx.data(), idx.size() * 4);
            }
        idx.resize(lattice(2, l, 0));
        ok = ok && avs_prepass_get_index(pp, AVS_INDEX_CENTER, l, 0, idx.data(), AVS_MEM_HOST) == AVS_OK && put(idx.data(), idx.size() * 4);
    }
    auto put_dense = [&](const std::vector<float> &v) {
        const int32_t zero = 0;
        return put(&zero, 4) && put(v.data(), v.size() * 4);
    };
    auto put_flat = [&](const Flat &v) {
        if (!v.is_constant) return put_dense(v.data);
        const int32_t one = 1;
        return put(&one, 4) && put(&v.constant, 4);
    };
    // the pre-pass keeps its weights on the octree lattice: crop to the simulation grid's centre / edge lattices
    auto put_cropped = [&](const std::vector<float> &full, int kind, int axis) {
        size_t ro[3] = {(size_t)n[0], (size_t)n[1], (size_t)n[2]}, rs[3] = {(size_t)sim[0], (size_t)sim[1], (size_t)sim[2]};
        if (kind == 1)
            for (int b = 0; b < 3; ++b) { ro[b] += (b != axis); rs[b] += (b != axis); }
        std::vector<float> c(rs[0] * rs[1] * rs[2]);
        for (size_t k = 0; k < rs[2]; ++k)
            for (size_t j = 0; j < rs[1]; ++j)
                std::memcpy(&c[(k * rs[1] + j) * rs[0]], &full[(k * ro[1] + j) * ro[0]], rs[0] * sizeof(float));
        return put_dense(c);
    };
    std::vector<float> w;
    w.resize(lattice(2, 0, 0));
    ok = ok && avs_prepass_get_weights(pp, AVS_FIELD_CENTER_WEIGHTS, 0, w.data(), AVS_MEM_HOST) == AVS_OK && put_cropped(w, 2, 0);
    for (int a = 0; a < 3 && ok; ++a) {
        w.resize(lattice(1, 0, a));
        ok = avs_prepass_get_weights(pp, AVS_FIELD_EDGE_WEIGHTS, a, w.data(), AVS_MEM_HOST) == AVS_OK && put_cropped(w, 1, a);
    }
    ok = ok && put_flat(face_w0) && put_flat(face_w1) && put_flat(face_w2) && put_flat(visc) && put_flat(dens);
    for (int a = 0; a < 3 && ok; ++a) ok = put_flat(vel[a]);
    for (int a = 0; a < 3 && ok; ++a) ok = put_flat(solidvel[a]);
    return std::fclose(f) == 0 && ok;
}

} // namespace

bool HDK_AdaptiveViscosity::solveGasSubclass(SIM_Engine &engine, SIM_Object *obj, SIM_Time time, SIM_Time timestep)
{
    // ---- the fields of the DOP network: same names as the reference (cpp:138-231) ----------------------------------------
    const SIM_ScalarField *surface = getConstScalarField(obj, GAS_NAME_SURFACE);
    SIM_VectorField *velocity = getVectorField(obj, GAS_NAME_VELOCITY);
    const SIM_ScalarField *collision = getConstScalarField(obj, GAS_NAME_COLLISION);
    const SIM_VectorField *collision_vel = getConstVectorField(obj, GAS_NAME_COLLISIONVELOCITY);
    const SIM_VectorField *face_weights = getConstVectorField(obj, "faceWeights");
    const SIM_ScalarField *viscosity = getConstScalarField(obj, "viscosity");
    const SIM_ScalarField *density = getConstScalarField(obj, GAS_NAME_DENSITY);
    auto fail = [&](const char *what) {
        addError(obj, SIM_MESSAGE, what, UT_ERROR_WARNING);
        return false;
    };
    if (!velocity || !velocity->isFaceSampled()) return fail("avs: the liquid velocity must be a face-sampled (staggered) vector field");
    if (!surface) return fail("avs: liquid surface field not found");
    if (!collision || !collision_vel) return fail("avs: collision surface / collision velocity field not found");
    if (!face_weights || !face_weights->isAligned(velocity)) return fail("avs: face weights must exist and be aligned with the velocity");
    if (!viscosity || !viscosity->getField()->isAligned(surface->getField())) return fail("avs: viscosity must be aligned with the surface field");
    if (!density || !density->getField()->isAligned(surface->getField())) return fail("avs: density must be aligned with the surface field");
    if (!collision->getField()->isAligned(surface->getField())) return fail("avs: collision surface must be aligned with the surface field");

    const SIM_RawField &liquid = *surface->getField();
    int sim[3];
    liquid.getVoxelRes(sim[0], sim[1], sim[2]);
    const int oct[3] = {next_pow2(sim[0]), next_pow2(sim[1]), next_pow2(sim[2])}; // HDK_OctreeGrid::init, oct.cpp:10-24
    const double dx = liquid.getVoxelSize().maxComponent();                        // cpp:242

    Handles &h = handlesOf(this);
    auto check = [&](avs_status s) {
        if (s == AVS_OK) return true;
        UT_WorkBuffer msg;
        msg.sprintf("avs: %s", avs_last_error());
        addError(obj, SIM_MESSAGE, msg.buffer(), UT_ERROR_WARNING);
        h.release(); // never leave a half-configured pre-pass / context for the next substep: it starts from fresh objects
        return false;
    };

    // ---- everything before the linear system: weights, mask, octree, classification, numbering (cpp:233-416) -------------
    avs_prepass_info pinfo;
    {
        UT_PerfMonAutoSolveEvent event(this, "Build Octree and Labels (GPU)");
        avs_prepass_desc pd;
        std::memset(&pd, 0, sizeof(pd));
        pd.nx = oct[0]; pd.ny = oct[1]; pd.nz = oct[2];
        pd.field_nx = sim[0]; pd.field_ny = sim[1]; pd.field_nz = sim[2];
        pd.dx = dx;
        pd.desired_levels = getOctreeLevels();
        pd.n_super = getNumberSuperSamples();
        pd.extrapolation_scale = getExtrapolation();
        pd.device = 0;
        const bool same_grid = h.pp && h.oct[0] == oct[0] && h.oct[1] == oct[1] && h.oct[2] == oct[2] && h.sim[0] == sim[0] && h.sim[1] == sim[1] &&
                               h.sim[2] == sim[2] && h.dx == dx && h.want_levels == pd.desired_levels && h.n_super == pd.n_super &&
                               h.extrapolation == pd.extrapolation_scale;
        if (!same_grid) { // a new grid (first substep, resized simulation box, changed parameters): new objects
            h.release();
            if (!check(avs_prepass_create(&pd, &h.pp))) return false;
            for (int a = 0; a < 3; ++a) { h.oct[a] = oct[a]; h.sim[a] = sim[a]; }
            h.dx = dx; h.want_levels = pd.desired_levels; h.n_super = pd.n_super; h.extrapolation = pd.extrapolation_scale;
        }
        const Flat liquid_sdf = flatten(liquid), solid_sdf = flatten(*collision->getField());
        std::vector<float> liquid_dense, solid_dense; // the pre-pass wants dense SDFs
        auto dense = [&](const Flat &f, std::vector<float> &store) -> const float * {
            if (!f.is_constant) return f.data.data();
            store.assign((size_t)sim[0] * sim[1] * sim[2], f.constant);
            return store.data();
        };
        if (!check(avs_prepass_run(h.pp, dense(liquid_sdf, liquid_dense), dense(solid_sdf, solid_dense), AVS_MEM_HOST))) return false;
        if (!check(avs_prepass_get_info(h.pp, &pinfo))) return false;
        if (pinfo.levels == 0) return true; // no liquid in the refinement band: nothing to do this step
    }

    // ---- context + the scalar fields of this frame -----------------------------------------------------------------------
    avs_desc d;
    std::memset(&d, 0, sizeof(d));
    d.nx = oct[0]; d.ny = oct[1]; d.nz = oct[2];
    d.field_nx = sim[0]; d.field_ny = sim[1]; d.field_nz = sim[2];
    d.dx = dx;
    d.dt = timestep;                                   // cpp:130
    d.levels = pinfo.levels;
    d.use_enhanced_gradients = getUseEnhancedGradients() ? 1 : 0;
#ifdef USESINGLEPRECISION // HDK_Utilities.h:25-37: SolveType = fpreal32
    d.precision = AVS_PRECISION_F32;
#endif
    d.device = 0;
    // a context is created for one octree depth and one timestep (avs_desc): another depth this frame, or another substep length, re-creates it
    if (h.ctx && (h.ctx_levels != pinfo.levels || h.enhanced != d.use_enhanced_gradients || h.dt != d.dt)) h.releaseCtx();
    if (!h.ctx) {
        if (!check(avs_create(&d, &h.ctx))) return false;
        h.ctx_levels = pinfo.levels;
        h.enhanced = d.use_enhanced_gradients;
        h.dt = d.dt;
    }
    if (!check(avs_prepass_apply(h.pp, h.ctx))) return false;  // labels, index pyramids, counts, centre / edge weights, regular indices
    const Flat visc = flatten(*viscosity->getField()), dens = flatten(*density->getField());
    Flat fw[3], vel[3], svel[3];
    for (int a = 0; a < 3; ++a) {
        fw[a] = flatten(*face_weights->getField(a));   // Houdini's "surfaceweights" stay the face integration weights (cpp:144)
        vel[a] = flatten(*velocity->getField(a));
        svel[a] = flatten(*collision_vel->getField(a));
    }
    auto put = [&](avs_field_kind kind, int axis, const Flat &f) { return check(avs_set_scalar_field(h.ctx, kind, axis, f.ptr(), f.constant, AVS_MEM_HOST)); };
    if (!put(AVS_FIELD_VISCOSITY, 0, visc) || !put(AVS_FIELD_DENSITY, 0, dens)) return false;
    for (int a = 0; a < 3; ++a)
        if (!put(AVS_FIELD_FACE_WEIGHTS, a, fw[a]) || !put(AVS_FIELD_VELOCITY, a, vel[a]) || !put(AVS_FIELD_SOLID_VELOCITY, a, svel[a])) return false;

    // optional: export the frame for the offline harness (any grid: AVSDUMP2 carries the simulation grid, see write_dump)
    if (const char *dump = std::getenv("AVS_DUMP_PATH"))
        (void)write_dump(dump, h.pp, pinfo, oct, sim, dx, timestep, getUseEnhancedGradients(), fw[0], fw[1], fw[2], visc, dens, vel, svel);

    // ---- the hot path: cpp:418-653 -----------------------------------------------------------------------------------------
    avs_assembly_info ainfo;
    avs_solve_info sinfo;
    {
        UT_PerfMonAutoSolveEvent event(this, "Build Octree Linear System (GPU)");
        if (!check(avs_assemble(h.ctx, &ainfo))) return false;
    }
    {
        UT_PerfMonAutoSolveEvent event(this, "Solve Linear System (GPU)");
#ifndef USEEIGEN // the non-Eigen build passes no preconditioner to UT_SparseMatrixRowT::solveConjugateGradient (cpp:638-642)
        if (!check(avs_set_solver_option(h.ctx, AVS_OPTION_PRECONDITIONER, AVS_PRECONDITIONER_NONE))) return false;
#endif
        // The reference polls UT_Interrupt::opInterrupt() inside its loops (cpp:2528; HDK_OctreeGrid.cpp:584-588).  The solve runs on the
        // device; a watcher thread polls the interrupt server while this thread is inside avs_solve and forwards a user break as avs_cancel
        // (any thread may call it): the loop ends at its next poll of the device state, sinfo.cancelled == 1.
        UT_Interrupt *boss = UTgetInterrupt();
        std::atomic<bool> solving{true};
        std::thread watcher([&] {
            while (solving.load(std::memory_order_acquire)) {
                if (boss->opInterrupt()) { (void)avs_cancel(h.ctx); break; }
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        });
        const bool solved = check(avs_solve(h.ctx, getSolverTolerance(), getMaxIterations(), &sinfo)); // non-convergence is not an error (cpp:645-652)
        solving.store(false, std::memory_order_release);
        watcher.join();
        // the watcher may have called avs_cancel just after the loop ended: a stale request must not end the NEXT substep's solve
        if (h.ctx) (void)avs_cancel_clear(h.ctx);
        if (!solved || sinfo.cancelled) return false;
        UT_WorkBuffer extra;
        extra.sprintf("iterations=%d, error=%.6f, octree DOFS=%d, regular DOFs=%d", (int)sinfo.iterations, sinfo.error, (int)sinfo.n,
                      (int)pinfo.n_regular);
        event.setExtraInfo(extra.buffer());
    }

    // ---- octree solution -> regular MAC grid, written back into "vel" (cpp:655-707) ------------------------------------------
    {
        UT_PerfMonAutoSolveEvent event(this, "Apply Octree Solution to Regular Grid (GPU)");
        std::vector<float> out[3];
        for (int a = 0; a < 3; ++a) {
            int r[3] = {sim[0], sim[1], sim[2]};
            r[a] += 1;
            out[a].resize((size_t)r[0] * r[1] * r[2]);
        }
        if (!check(avs_transfer_to_regular_grid(h.ctx, out[0].data(), out[1].data(), out[2].data(), AVS_MEM_HOST))) return false;
        for (int a = 0; a < 3; ++a) unflatten(out[a], *velocity->getField(a));
        velocity->pubHandleModification();
    }
    return true;
}
