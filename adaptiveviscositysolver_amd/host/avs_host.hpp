// avs_host.hpp -- C++ host side above the C ABI (include/avs.h).
//
// Mirrors the hot-path slice of HDK_AdaptiveViscosity::solveGasSubclass (HDK_AdaptiveViscosity.cpp,
// "cpp:") with the reference's own phase names, on plain dense arrays instead of SIM_Raw*Field:
//
//   reference (cpp)                                   here
//   ------------------------------------------------  ------------------------------------------
//   octreeLabels.getGridLabels(level)       oct.h:144  setOctreeLabels(level, ...)
//   octreeVelocityIndices / edgeStressIndices /       setOctreeVelocityIndices / setEdgeStressIndices /
//   centerStressIndices                  cpp:337-393  setCenterStressIndices + setDOFCounts
//   buildEdgeStressStencils, buildCenterStress*       buildStressStencils()            cpp:443-498
//   buildVelocityMapping                 cpp:518-528  buildVelocityMapping()
//   buildOctreeSystemFromStencils + setFromTriplets   buildOctreeSystemFromStencils()  cpp:577-593, 613-614
//   ConjugateGradient::solveWithGuess    cpp:618-630  solveConjugateGradient(tol, maxIterations)
//   regularVelocityIndices               cpp:303-329  setRegularVelocityIndices(axis, ...)
//   setOctreeVelocity + interpolator +
//   applyVelocitiesToRegularGrid         cpp:661-707  applyVelocitiesToRegularGrid(u, v, w)
//
// Header-only; needs only include/avs.h and -lavs_hip.  Errors of the C ABI become std::runtime_error
// (the reference returns false after addError; Houdini-side glue is in INTEGRATION.md).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "avs.h"

namespace avs_host {

struct SolveResult {
    int iterations = 0;
    bool converged = false;
    double error = 0.;
    double solveMs = 0., spmvMs = 0.;
};

class AdaptiveViscosity {
public:
    // nx, ny, nz: padded octree resolution; fieldRes: the simulation grid the scalar fields live on (oct.cpp:13-24),
    // nullptr = the same
    AdaptiveViscosity(int nx, int ny, int nz, double dx, double dt, int octreeLevels, bool useEnhancedGradients = true,
                      int device = 0, const int *fieldRes = nullptr, int solveType = AVS_PRECISION_F64)
    {
        avs_desc d{};
        d.nx = nx; d.ny = ny; d.nz = nz;
        if (fieldRes) { d.field_nx = fieldRes[0]; d.field_ny = fieldRes[1]; d.field_nz = fieldRes[2]; }
        d.dx = dx; d.dt = dt;
        d.levels = octreeLevels;
        d.use_enhanced_gradients = useEnhancedGradients ? 1 : 0;
        d.device = device;
        d.stream = nullptr;
        d.precision = solveType; // SolveType of the reference build (util.h:25-37): fpreal64, or fpreal32 under USESINGLEPRECISION
        check(avs_create(&d, &myCtx), "avs_create");
        myLevels = octreeLevels;
    }
    ~AdaptiveViscosity() { avs_destroy(myCtx); }
    AdaptiveViscosity(const AdaptiveViscosity &) = delete;
    AdaptiveViscosity &operator=(const AdaptiveViscosity &) = delete;

    // ---- inputs produced by cpp:233-416 -----------------------------------------------------------
    void setOctreeLabels(int level, const int8_t *labels) { check(avs_set_labels(myCtx, level, labels, AVS_MEM_HOST), "labels"); }
    void setOctreeVelocityIndices(int level, int axis, const int32_t *idx) { setIndex(AVS_INDEX_VELOCITY, level, axis, idx); }
    void setEdgeStressIndices(int level, int axis, const int32_t *idx) { setIndex(AVS_INDEX_EDGE, level, axis, idx); }
    void setCenterStressIndices(int level, const int32_t *idx) { setIndex(AVS_INDEX_CENTER, level, 0, idx); }
    void setDOFCounts(int64_t octreeVelocityDOFCount, int64_t edgeStressDOFCount, int64_t centerStressDOFCount)
    {
        check(avs_set_dof_counts(myCtx, octreeVelocityDOFCount, edgeStressDOFCount, centerStressDOFCount), "dof counts");
    }
    // data == nullptr: constant field (HDK field()->isConstant fast path)
    void setField(avs_field_kind kind, int axis, const float *data, float constant = 0.f)
    {
        check(avs_set_scalar_field(myCtx, kind, axis, data, constant, AVS_MEM_HOST), "field");
    }

    // solver switches (avs_set_solver_option): e.g. (AVS_OPTION_PRECONDITIONER, AVS_PRECONDITIONER_NONE) = the build without USEEIGEN
    // (cpp:633-642); (AVS_OPTION_RESIDENT_LOOP, 0) on a GPU shared with a viewport; (AVS_OPTION_BRICK_FORM, AVS_BRICK_NEVER)
    void setSolverOption(avs_solver_option option, int value) { check(avs_set_solver_option(myCtx, option, value), "avs_set_solver_option"); }

    // ---- hot path ---------------------------------------------------------------------------------
    void buildStressStencils() { check(avs_build_stencils(myCtx), "buildStressStencils"); }
    void buildVelocityMapping() { check(avs_build_initial_guess(myCtx), "buildVelocityMapping"); }
    void buildOctreeSystemFromStencils() { check(avs_build_system(myCtx), "buildOctreeSystemFromStencils"); }
    avs_assembly_info buildLinearSystem()
    {
        avs_assembly_info info{};
        check(avs_assemble(myCtx, &info), "avs_assemble");
        return info;
    }
    SolveResult solveConjugateGradient(double solverTolerance = 1e-3, int maxSolverIterations = 2500)
    {
        avs_solve_info s{};
        check(avs_solve(myCtx, solverTolerance, maxSolverIterations, &s), "avs_solve");
        SolveResult r;
        r.iterations = s.iterations;
        r.converged = s.converged != 0;
        r.error = s.error;
        r.solveMs = s.solve_ms;
        r.spmvMs = s.spmv_ms;
        myDofs = s.n;
        return r;
    }
    std::vector<double> viscositySolution() const
    {
        std::vector<double> x((size_t)myDofs);
        check(avs_get_solution(myCtx, x.data(), myDofs, AVS_MEM_HOST), "avs_get_solution");
        return x;
    }
    // ---- post-solve transfer, cpp:655-707 ----------------------------------------------------------
    void setRegularVelocityIndices(int axis, const int32_t *idx)
    {
        check(avs_set_regular_index_field(myCtx, axis, idx, AVS_MEM_HOST), "regular index field");
    }
    // u: (nx+1)*ny*nz, v: nx*(ny+1)*nz, w: nx*ny*(nz+1) floats, x fastest
    void applyVelocitiesToRegularGrid(float *u, float *v, float *w)
    {
        check(avs_transfer_to_regular_grid(myCtx, u, v, w, AVS_MEM_HOST), "applyVelocitiesToRegularGrid");
    }
    // the same as an in-place update of a DEVICE-resident velocity field (what the reference does to `vel`, cpp:655-707): the arrays already
    // hold what was given as AVS_FIELD_VELOCITY; only the faces the transfer changes are written
    void applyVelocitiesToRegularGridInPlace(float *d_u, float *d_v, float *d_w)
    {
        check(avs_transfer_to_regular_grid_in_place(myCtx, d_u, d_v, d_w), "applyVelocitiesToRegularGridInPlace");
    }
    int octreeLevels() const { return myLevels; }
    avs_ctx *handle() const { return myCtx; }

private:
    void setIndex(avs_index_kind kind, int level, int axis, const int32_t *idx)
    {
        check(avs_set_index_field(myCtx, kind, level, axis, idx, AVS_MEM_HOST), "index field");
    }
    static void check(avs_status s, const char *what)
    {
        if (s != AVS_OK) throw std::runtime_error(std::string(what) + ": " + avs_last_error());
    }
    avs_ctx *myCtx = nullptr;
    int myLevels = 0;
    int64_t myDofs = 0;
};

} // namespace avs_host
