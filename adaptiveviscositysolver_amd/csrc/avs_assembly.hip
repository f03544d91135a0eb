// avs_assembly.hip -- on-device assembly of the variational viscosity system over the octree.
//
// Replaces cpp:418-594 + Eigen's setFromTriplets (cpp:613-614) of the reference
// (HDK_AdaptiveViscosity.cpp).  The reference sweeps every voxel of every (level, axis) grid and
// chases nested heap lists; here every kernel is launched per DOF / per stress through
// "dof -> (level, axis, i, j, k)" tables, so lanes are never idle on empty voxels and global ids
// keep the reference's tile-major order (neighbouring lanes touch neighbouring voxels).
//
//   K0  k_dof_table        index grids -> dof tables (+ max id validation)
//   K1  k_edge_stencils    getEdgeStressFaces + edgeOctreeVolumes + weights   cpp:1717-1908, 2004-2160
//   K2  k_center_stencils  getCenterStressFaces + weights                    cpp:1910-1963, 2162-2289
//   K3  k_initial_guess (+ _coarse)  buildVelocityMappingPartial; rows of level >= 3 by 4 lanes each   cpp:2291-2402
//   K4  k_rows<false>      dry run of the row sweep: raw triplets per row    cpp:2459-2777
//   K5  exclusive scan     wave64 shuffle scan -> raw offsets / row pointers
//   K6  k_rows<true>       row sweep, raw triplets in emission order, wave-transposed     cpp:2404-2457
//   K6b k_unique_rows      first occurrences of every row's columns -> unique counts -> row pointers
//   K7  k_merge_rows       per row, in registers: rank of the first occurrences + left fold of the duplicates in
//                          emission order = setFromTriplets for one row, straight into the final CSR   cpp:613-614
//
// Rows are independent (a gather): no atomics on the data path (one per wave to list the coarse rows).  All arithmetic is done in the reference's
// order with one rounding per operation (compile with -ffp-contract=off), so the CSR values, the
// right-hand side and the initial guess are bit-identical to the CPU oracle.
#include "avs_device_common.hpp"

namespace avs {

static constexpr int kBlock = 256;

__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

// ---------------------------------------------------------------------------------------------
// K0: dof tables
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_dof_table(const int32_t *__restrict__ grid, I3 r, int level, int axis,
                                                      int32_t *__restrict__ table, int64_t ndof, int *err)
{
    const size_t total = (size_t)r[0] * r[1] * r[2];
    for (size_t o = (size_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += (size_t)gridDim.x * kBlock) {
        const int32_t id = grid[o];
        if (id >= 0) {
            if ((int64_t)id >= ndof) { *err = 1; continue; }
            const int i = (int)(o % r[0]);
            const size_t q = o / r[0];
            const int j = (int)(q % r[1]);
            const int k = (int)(q / r[1]);
            int4 rec = make_int4(level | (axis << 8), i, j, k);
            reinterpret_cast<int4 *>(table)[id] = rec;
        } else if (id < AVS_OUTSIDE) *err = 2;
    }
}

// every dof id in [0, n) must have been written exactly once: table is pre-filled with -1
__global__ __launch_bounds__(kBlock) void k_check_table(const int32_t *__restrict__ table, int64_t ndof, int *err)
{
    for (int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x; d < ndof; d += (int64_t)gridDim.x * kBlock)
        if (table[4 * d] < 0) *err = 3;
}

// x as the fp32 build of the reference would hold it (a float value in a double)
__device__ __forceinline__ double solve_type(double x, int f32) { return f32 ? (double)(float)x : x; }

// ---------------------------------------------------------------------------------------------
// K1: edge stress stencils
// ---------------------------------------------------------------------------------------------
struct StencilWriter {
    int32_t *idx;
    double *coef;
    int64_t stride; // = number of stencils
    int64_t s;      // stencil id
    int cap, cnt, overflow;
    __device__ __forceinline__ void push(int32_t id, double c)
    {
        if (cnt < cap) {
            idx[(size_t)cnt * stride + s] = id;
            coef[(size_t)cnt * stride + s] = c;
            ++cnt;
        } else overflow = 1;
    }
};

// Slab-local assembly (round 6): a rank builds the stencils of the stresses near its slab only.  `list` names the stresses inside the
// rank's window (null: all, in order); of those, the ones whose position along the cut axis lies within [lo[level], hi[level]) of their
// level's lattice get a stencil, the others keep cnt = 0, which the row sweep reports (a row never finds itself in an empty stencil).
struct StencilCore {
    const int32_t *list;
    int64_t m;   // entries of list
    int axis;    // cut axis
    int lo[AVS_MAX_LEVELS], hi[AVS_MAX_LEVELS];
};

__global__ __launch_bounds__(kBlock) void k_edge_stencils(PyramidView P, const int32_t *__restrict__ edof,
                                                          StencilView S, int *err, StencilCore core)
{
    const int64_t slot = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (slot >= (core.list ? core.m : S.count)) return;
    const int64_t id = core.list ? core.list[slot] : slot;
    const int4 rec = reinterpret_cast<const int4 *>(edof)[id];
    const int level = rec.x & 0xff, axis = rec.x >> 8;
    const I3 edge{{rec.y, rec.z, rec.w}};
    if (core.list) {
        const int c = core.axis == 0 ? rec.y : (core.axis == 1 ? rec.z : rec.w);
        if (c < core.lo[level] || c >= core.hi[level]) return;
    }
    const double dx = P.dx * (double)(1 << level); // cpp:1733
    const double vdx0 = (double)(1 << level);      // cpp:2014 (fine-voxel units)

    // pass 1 (cpp:1740-1787) fused with edgeOctreeVolumes (cpp:2019-2054): both classify the same
    // four faces.  UT_Vector3 accumulators are fp32 in the reference.
    bool atTransition[3] = {false, false, false}, faceOutside[3] = {false, false, false};
    float gdx[3] = {0.f, 0.f, 0.f};
    float vdx[3] = {0.f, 0.f, 0.f};
    vdx[axis] = (float)vdx0;
    int32_t fidx[3][2]; // velocity index of the 4 slots, [faceAxis][direction]; INT32_MIN = out of bounds
#pragma unroll
    for (int fa = 0; fa < 3; ++fa) {
        if (fa == axis) continue;
        const int ga = 3 - fa - axis;
        const I3 fr = face_res(P, level, fa);
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            I3 face = edge;
            if (dir == 0) --face[ga]; // HDKedgeToFace, util.h:151-167
            int32_t vi;
            if (face[ga] < 0 || face[ga] >= fr[ga]) {
                vi = INT32_MIN;
                gdx[ga] = (float)((double)gdx[ga] + .5 * dx);
                vdx[ga] = (float)((double)vdx[ga] + .5 * vdx0);
                faceOutside[ga] = true;
            } else {
                vi = P.vidx[level][fa][lin(fr, face)];
                if (vi >= 0) {
                    gdx[ga] = (float)((double)gdx[ga] + .5 * dx);
                    vdx[ga] = (float)((double)vdx[ga] + .5 * vdx0);
                } else if (vi == AVS_OUTSIDE || vi == AVS_SOLIDBOUNDARY) {
                    gdx[ga] = (float)((double)gdx[ga] + .5 * dx);
                    vdx[ga] = (float)((double)vdx[ga] + .5 * vdx0);
                    faceOutside[ga] = true;
                } else {
                    gdx[ga] = (float)((double)gdx[ga] + dx);
                    vdx[ga] = (float)((double)vdx[ga] + vdx0);
                    if (P.enhanced) atTransition[ga] = true;
                }
            }
            fidx[fa][dir] = vi;
        }
    }

    StencilWriter w{S.idx, S.coef, S.count, id, AVS_EDGE_STENCIL_CAP, 0, 0};
    int bcnt = 0;
    int bad = 0;
    // pass 2 (cpp:1789-1907)
#pragma unroll
    for (int fa = 0; fa < 3; ++fa) {
        if (fa == axis) continue;
        const int ga = 3 - fa - axis;
        const I3 fr = face_res(P, level, fa);
        const double g = (double)gdx[ga];
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            const int32_t vi = fidx[fa][dir];
            if (vi == INT32_MIN) continue;
            I3 face = edge;
            if (dir == 0) --face[ga];
            const double sign = (dir == 0) ? -1. : 1.;
            if (vi >= 0) {
                if (atTransition[ga] && !faceOutside[ga]) { // cpp:1814-1824
                    I3 sib = face;
                    sib[axis] += (edge[axis] % 2 == 0) ? 1 : -1;
                    const int32_t si = P.vidx[level][fa][lin(fr, sib)];
                    if (si < 0) bad = 1; // assert cpp:1820
                    w.push(si, .25 * sign / g);
                    w.push(vi, .25 * sign / g);
                } else w.push(vi, .5 * sign / g); // cpp:1827
            } else if (vi == AVS_UNASSIGNED) {
                if (level + 1 >= P.levels) { bad = 1; continue; }
                if (edge[fa] % 2 != 0) { // dangling edge inside a coarse cell, cpp:1835-1884
#pragma unroll
                    for (int oi = 0; oi < 2; ++oi) {
                        I3 of = face;
                        of[fa] += oi == 0 ? -1 : 1;
                        const I3 pf = half3(of);
                        const int32_t pi = vidx_at(P, level + 1, fa, pf);
                        if (pi >= 0) w.push(pi, .25 * sign / g);
                        else if (pi == AVS_UNASSIGNED) {
                            for (int ci = 0; ci < 4; ++ci) {
                                const I3 cf = child_face(pf, fa, ci);
                                const int32_t cvi = P.vidx[level][fa][lin(fr, cf)];
                                if (cvi >= 0) w.push(cvi, .0625 * sign / g);
                                else bad = 1; // assert(false) cpp:1878
                            }
                        }
                    }
                } else { // cpp:1886-1894
                    const int32_t pi = vidx_at(P, level + 1, fa, half3(face));
                    if (pi < 0) bad = 1;
                    w.push(pi, .5 * sign / g);
                }
            } else if (vi == AVS_SOLIDBOUNDARY) { // cpp:1896-1905 (component = EDGE axis: reference quirk)
                const double lv = (double)sample_f32(P.solidvel[axis], face_res(P, 0, axis), off_face(axis),
                                                     pos2_face(level, fa, face));
                if (bcnt < AVS_EDGE_BOUNDARY_CAP) S.bval[(size_t)bcnt++ * S.count + id] = .5 * sign * lv / g;
            }
        }
    }
    S.cnt[id] = w.cnt;
    S.bcnt[id] = bcnt;

    // weight, cpp:2124-2155
    float vol = vdx[0] * vdx[1];
    vol = vol * vdx[2];
    double wgt;
    if (level == 0) {
        wgt = (double)field_at(P.edgew[axis], edge_res(P, 0, axis), edge);
        if (wgt == 1.) wgt = (double)vol;
    } else wgt = (double)vol;
    if (P.visc.is_const) wgt *= (double)P.visc.cval;
    else wgt *= (double)sample_f32(P.visc, cell_res(P, 0), I3{{1, 1, 1}}, pos2_edge(level, axis, edge));
    S.weight[id] = 4. * P.dt * wgt;
    if (w.overflow || bad) *err = w.overflow ? 4 : 5;
}

// ---------------------------------------------------------------------------------------------
// K2: centre stress stencils (3 lists per active cell) + weight
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_center_stencils(PyramidView P, const int32_t *__restrict__ cdof,
                                                            StencilView S, int *err, StencilCore core)
{
    const int64_t nc = S.count / 3;
    const int64_t slot = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (slot >= (core.list ? core.m : nc)) return;
    const int64_t id = core.list ? core.list[slot] : slot;
    const int4 rec = reinterpret_cast<const int4 *>(cdof)[id];
    const int level = rec.x & 0xff;
    const I3 cell{{rec.y, rec.z, rec.w}};
    if (core.list) {
        const int c = core.axis == 0 ? rec.y : (core.axis == 1 ? rec.z : rec.w);
        if (c < core.lo[level] || c >= core.hi[level]) return;
    }
    const double dx = P.dx * (double)(1 << level); // cpp:1923
    int bad = 0;
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
        const int64_t sid = id + nc * axis; // cpp:2186, 2207
        StencilWriter w{S.idx, S.coef, S.count, sid, AVS_CENTER_STENCIL_CAP, 0, 0};
        int bcnt = 0;
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            I3 face = cell;
            if (dir == 1) ++face[axis];
            const double sign = (dir == 0) ? -1. : 1.;
            const int32_t vi = vidx_at(P, level, axis, face);
            if (vi >= 0) w.push(vi, sign / dx);
            else if (vi == AVS_UNASSIGNED) { // cpp:1937-1951
                if (level == 0) { bad = 1; continue; }
                for (int ci = 0; ci < 4; ++ci) {
                    const int32_t cvi = vidx_at(P, level - 1, axis, child_face(face, axis, ci));
                    if (cvi < 0) bad = 1;
                    w.push(cvi, .25 * sign / dx);
                }
            } else if (vi == AVS_SOLIDBOUNDARY) { // cpp:1952-1961
                const double lv = (double)sample_f32(P.solidvel[axis], face_res(P, 0, axis), off_face(axis),
                                                     pos2_face(level, axis, face));
                if (bcnt < AVS_CENTER_BOUNDARY_CAP) S.bval[(size_t)bcnt++ * S.count + sid] = sign * lv / dx;
            }
        }
        S.cnt[sid] = w.cnt;
        S.bcnt[sid] = bcnt;
        if (w.overflow) bad = 1;
    }
    double wgt;
    if (level == 0) wgt = (double)field_at(P.centerw, cell_res(P, 0), cell); // cpp:2271-2272
    else {
        const double d = (double)(1 << level);
        wgt = d * d * d;
    }
    if (P.visc.is_const) wgt *= (double)P.visc.cval;
    else wgt *= (double)sample_f32(P.visc, cell_res(P, 0), I3{{1, 1, 1}}, pos2_center(level, cell));
    S.weight[id] = 2. * P.dt * wgt; // cpp:2284
    if (bad) *err = 6;
}

// ---------------------------------------------------------------------------------------------
// K3: initial guess = restriction of the regular-grid face velocity (cpp:2291-2402).
// The reference pops a FIFO queue of (face, weight, level); the leaves come out in lexicographic
// (child, in-axis offset) order per level, which is the order of the mixed-radix counter below.
// ---------------------------------------------------------------------------------------------
// the 12^ND leaves under face `f` (ND levels up): all gathers first ...
__device__ __forceinline__ I3 child_face(const I3 &f, int digit, int axis, int a1, int a2) // digit = 3 * child + (offset + 1), cpp:2323
{
    const int ci = digit / 3, off = digit % 3 - 1;
    I3 h{{2 * f[0], 2 * f[1], 2 * f[2]}};
    if (ci & 1) ++h[a1];
    if (ci & 2) ++h[a2];
    h[axis] += off;
    return h;
}

__device__ __forceinline__ float leaf_at(const FieldView &V, const I3 &vr, const I3 &h)
{
    const I3 fc{{clampi(h[0], 0, vr[0] - 1), clampi(h[1], 0, vr[1] - 1), clampi(h[2], 0, vr[2] - 1)}};
    return field_at(V, vr, fc);
}

__device__ __forceinline__ void gather_12(const FieldView &V, const I3 &vr, const I3 &f, int axis, int a1, int a2, float (&leaf)[12])
{
#pragma clang loop unroll(full)
    for (int d = 0; d < 12; ++d) leaf[d] = leaf_at(V, vr, child_face(f, d, axis, a1, a2));
}

__device__ __forceinline__ void gather_144(const FieldView &V, const I3 &vr, const I3 &f, int axis, int a1, int a2, float (&leaf)[144])
{
#pragma clang loop unroll(full)
    for (int d1 = 0; d1 < 12; ++d1) {
        const I3 g = child_face(f, d1, axis, a1, a2);
#pragma clang loop unroll(full)
        for (int d0 = 0; d0 < 12; ++d0) leaf[d1 * 12 + d0] = leaf_at(V, vr, child_face(g, d0, axis, a1, a2));
    }
}

// ... then the fold in the FIFO's order (weight `wgt` so far; fpreal32 myWeight, cpp:2318)
__device__ __forceinline__ float child_weight(float w, int digit)
{
    const double rw = (digit % 3 == 1) ? (1. / 8.) : (1. / 16.); // cpp:2323
    return (float)(rw * (double)w);
}

__device__ __forceinline__ void fold_12(float wgt, const float (&leaf)[12], double &acc)
{
#pragma clang loop unroll(full)
    for (int d = 0; d < 12; ++d) acc += (double)child_weight(wgt, d) * (double)leaf[d];
}

__device__ __forceinline__ void fold_144(float wgt, const float (&leaf)[144], double &acc)
{
#pragma clang loop unroll(full)
    for (int d1 = 0; d1 < 12; ++d1) {
        const float w1 = child_weight(wgt, d1);
#pragma clang loop unroll(full)
        for (int d0 = 0; d0 < 12; ++d0) acc += (double)child_weight(w1, d0) * (double)leaf[d1 * 12 + d0];
    }
}

// A level-L row walks 12^L leaves (1728 at level 3) and the sum has to be taken in the FIFO's order.  One thread per row with
// one dependent gather per leaf made this kernel as slow as its coarsest row (2.5 ms at 512^3, the chip idle: 106 k level-3
// rows are 1.6 waves per SIMD).  Now levels 1 and 2 gather all their 12 / 144 leaves before folding, and a row of level >= 3
// is spread over 4 lanes: lane t takes the t-th 144-leaf unit of the current four (144 gathers in flight in every lane), then
// the four lanes fold one after the other, handing the running sum on -- the order of the additions is unchanged.  (The fold
// is the serial part: with G lanes per row a wave spends G x the issue slots on it, so G stays small; 12 lanes: 0.90 ms.)
__global__ __launch_bounds__(kBlock) void k_initial_guess(PyramidView P, const int32_t *__restrict__ vdof, int64_t n,
                                                          double *__restrict__ x0, const int32_t *__restrict__ ids,
                                                          int32_t *__restrict__ coarse_list /* [0]: count, then DOF ids */)
{
    const int64_t slot = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = slot < n;
    const int id = valid ? (ids ? ids[slot] : (int)slot) : 0; // multi-GPU: only the DOFs this rank owns (x0 stays indexed by DOF)
    int4 rec = make_int4(0, 0, 0, 0);
    if (valid) rec = reinterpret_cast<const int4 *>(vdof)[id];
    const int level = rec.x & 0xff, axis = rec.x >> 8;
    { // rows of level >= 3 go on a list for k_initial_guess_coarse (one atomic per wave; the order of the list does not matter)
        const bool coarse = valid && level > 2;
        const unsigned long long cm = __ballot(coarse);
        if (cm) {
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == __ffsll((long long)cm) - 1) base = atomicAdd(coarse_list, __popcll(cm));
            base = __shfl(base, __ffsll((long long)cm) - 1, 64);
            if (coarse) coarse_list[1 + base + __popcll(cm & ((1ull << lane) - 1ull))] = id;
        }
    }
    if (!valid || level > 2) return;
    const I3 face{{rec.y, rec.z, rec.w}};
    const I3 vr = face_res(P, 0, axis);
    const FieldView &V = P.vel[axis];
    const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
    if (level == 0) {
        x0[id] = 1.0 * (double)field_at(V, vr, face); // weight 1 (cpp:2347, 2373) (an fp32 value: nothing to narrow)
    } else if (level == 1) {
        float leaf[12];
        double acc = 0.;
        gather_12(V, vr, face, axis, a1, a2, leaf);
        fold_12(1.f, leaf, acc);
        x0[id] = solve_type(acc, P.f32); // initialGuess(octreeFaceIndex) = restrictedVelocity, cpp:2371
    } else {
        float leaf[144];
        double acc = 0.;
        gather_144(V, vr, face, axis, a1, a2, leaf);
        fold_144(1.f, leaf, acc);
        x0[id] = solve_type(acc, P.f32); // initialGuess(octreeFaceIndex) = restrictedVelocity, cpp:2371
    }
}

static constexpr int kCoarseGrid = 256;  // persistent workgroups of k_initial_guess_coarse (256 VGPRs: one wave per SIMD = one workgroup per CU)

__global__ __launch_bounds__(kBlock) void k_initial_guess_coarse(PyramidView P, const int32_t *__restrict__ vdof, double *__restrict__ x0,
                                                                 const int32_t *__restrict__ coarse_list)
{
    const int lane = threadIdx.x & 63;
    constexpr int G = 4;                  // lanes per row: sixteen rows at a time
    const int s = lane / G, t = lane % G;
    const int count = coarse_list[0];
    const int nwaves = gridDim.x * (kBlock / 64);
    // persistent waves over groups of sixteen listed rows (the rows of one level sit in one id range, or -- multi-GPU, rows
    // in brick order -- are scattered: either way every wave gets full groups)
    for (int g0 = (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * (64 / G); g0 < count; g0 += nwaves * (64 / G)) {
        const bool on = g0 + s < count;
        const int rid = on ? coarse_list[1 + g0 + s] : 0;
        int4 rec = make_int4(0, 0, 0, 0);
        if (on) rec = reinterpret_cast<const int4 *>(vdof)[rid];
        const int rlevel = rec.x & 0xff, raxis = rec.x >> 8;
        const I3 face{{rec.y, rec.z, rec.w}};
        const I3 vr = face_res(P, 0, raxis);
        const FieldView &V = P.vel[raxis];
        const int a1 = (raxis + 1) % 3, a2 = (raxis + 2) % 3;
        unsigned units = 0; // 12^(level-2) units of 144 leaves
        if (on) {
            units = 12;
            for (int l = 3; l < rlevel; ++l) units *= 12u;
        }
        const unsigned umax = (unsigned)wave_max_i32((int)units);
        double carry = 0.;
        for (unsigned u0 = 0; u0 < umax; u0 += G) { // G consecutive units of the row, one per lane
            const unsigned unit = u0 + (unsigned)t;
            const bool act = on && unit < units;
            float leaf[144];
            float wgt = 1.f;
            I3 f = face;
            if (act) {
                // the leading level-2 digits of this lane's unit, most significant first: (child, offset) of every step down
                unsigned pw = units / 12u; // weight of the most significant digit
                for (int l = 0; l + 2 < rlevel; ++l) {
                    const int digit = (int)((unit / pw) % 12u);
                    pw = pw >= 12u ? pw / 12u : 1u;
                    f = child_face(f, digit, raxis, a1, a2);
                    wgt = child_weight(wgt, digit);
                }
            }
            gather_144(V, vr, f, raxis, a1, a2, leaf); // idle lanes gather too (clamped, harmless): the array stays in registers
            for (int r = 0; r < G; ++r) {              // the G units, in order: the running sum is handed from lane to lane
                double mine = carry;
                if (act && t == r) fold_144(wgt, leaf, mine);
                carry = __shfl(mine, s * G + r, 64);
            }
        }
        if (on && t == 0) x0[rid] = solve_type(carry, P.f32);
    }
}

// ---------------------------------------------------------------------------------------------
// K4 / K6: the row sweep
// ---------------------------------------------------------------------------------------------
static constexpr int kRawStride = 64; // raw triplets are stored transposed inside every wave of 64 rows (see k_rows / k_merge_rows)

struct RowAcc {
    int n;          // raw entries so far
    int32_t *col;   // row storage in the raw arrays (emission order)
    double *val;
    double diag, rhs;
    int bad;
    int f32; // SolveType = fpreal32 (PyramidView::f32): triplets narrowed where Eigen::Triplet<SolveType> is built, rhs updated in float steps
};


template <bool EMIT>
__device__ __forceinline__ void row_push(RowAcc &ra, int32_t c, double v)
{
    if (EMIT) { // raw triplet, emission order (what the reference push_backs, cpp:2447); wave-transposed: entry k of the row
        ra.col[(size_t)ra.n * kRawStride] = c; // of lane l lives at base + 64 k + l, so the k-th push of a wave is ONE coalesced store
        ra.val[(size_t)ra.n * kRawStride] = solve_type(v, ra.f32); // Eigen::Triplet<SolveType>(row, col, element), cpp:2447 / 2768
    }
    ++ra.n;
}


// applyToMatrix, cpp:2404-2457
template <bool EMIT>
__device__ void apply_stencil(RowAcc &ra, double coefficient, int32_t vi, int cnt, const int32_t *__restrict__ idx,
                              const double *__restrict__ coef, int64_t stride, int bcnt,
                              const double *__restrict__ bval)
{
    // The reference searches the stencil for the row's own entry (first match), then walks it again.  Done literally that is
    // up to 2 cnt DEPENDENT loads per stencil (the search stops at the match, the walk stores between its loads), six
    // stencils per row.  The first kStencilBatch entries -- nearly every stencil has no more -- are requested at once.
    constexpr int kStencilBatch = EMIT ? 8 : 6; // measured: 4 / 6 / 8 -> dry run 0.63 / 0.59 / 0.66 ms, emit 1.62 / 1.50 / 1.50 ms
    int32_t ji[kStencilBatch];
    double cf[kStencilBatch];
#pragma unroll
    for (int k = 0; k < kStencilBatch; ++k) {
        ji[k] = k < cnt ? idx[(size_t)k * stride] : -1;
        cf[k] = (EMIT && k < cnt) ? coef[(size_t)k * stride] : 0.;
    }
    bool found = false;
#pragma unroll
    for (int k = 0; k < kStencilBatch; ++k)
        if (!found && k < cnt && ji[k] == vi) {
            coefficient *= cf[k];
            found = true;
        }
    for (int i = kStencilBatch; !found && i < cnt; ++i)
        if (idx[(size_t)i * stride] == vi) {
            coefficient *= coef[(size_t)i * stride];
            found = true;
        }
    if (!found) ra.bad = 1; // assert(foundSelf) cpp:2436
#pragma unroll
    for (int k = 0; k < kStencilBatch; ++k)
        if (k < cnt) {
            const int32_t j = ji[k];
            if (EMIT) {
                const double element = coefficient * cf[k];
                if (j == vi) ra.diag += element;
                else row_push<true>(ra, j, element);
            } else if (j != vi) ++ra.n;
        }
    for (int i = kStencilBatch; i < cnt; ++i) {
        const int32_t j = idx[(size_t)i * stride];
        if (EMIT) {
            const double element = coefficient * coef[(size_t)i * stride];
            if (j == vi) ra.diag += element;
            else row_push<true>(ra, j, element);
        } else if (j != vi) ++ra.n;
    }
    if (EMIT)
        for (int i = 0; i < bcnt; ++i) ra.rhs = solve_type(ra.rhs - coefficient * bval[(size_t)i * stride], ra.f32); // rhs(row) -= ..., cpp:2456
}

template <bool EMIT>
__device__ __forceinline__ void apply_edge(RowAcc &ra, int32_t vi, int32_t eid, const StencilView &E)
{
    apply_stencil<EMIT>(ra, EMIT ? E.weight[eid] : 0., vi, E.cnt[eid], E.idx + eid, E.coef + eid, E.count,
                        E.bcnt[eid], E.bval + eid);
}
template <bool EMIT>
__device__ __forceinline__ void apply_center(RowAcc &ra, int32_t vi, int32_t cid, int axis, const StencilView &C)
{
    const int64_t nc = C.count / 3;
    const int64_t sid = cid + nc * axis;
    apply_stencil<EMIT>(ra, EMIT ? C.weight[cid] : 0., vi, C.cnt[sid], C.idx + sid, C.coef + sid, C.count,
                        C.bcnt[sid], C.bval + sid);
}

// faceOctreeVolumes, cpp:1965-2002
__device__ double face_octree_volume(const PyramidView &P, int level, int axis, const I3 &face, int &bad)
{
    const I3 cr = cell_res(P, level);
    const double dx = (double)(1 << level);
    double g = 0.;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        I3 cell = face;
        if (dir == 0) --cell[axis];
        if (cell[axis] < 0 || cell[axis] >= cr[axis]) g += .5 * dx;
        else {
            const int lb = P.labels[level][lin(cr, cell)];
            if (lb == AVS_ACTIVE || lb == AVS_INACTIVE) g += .5 * dx;
            else if (level + 1 < P.levels && label_at(P, level + 1, half3(cell)) == AVS_ACTIVE) g += dx;
            else bad = 1; // assert(false) cpp:1996
        }
    }
    return dx * dx * g;
}

template <bool EMIT>
__global__ __launch_bounds__(kBlock) void k_rows(PyramidView P, const int32_t *__restrict__ vdof, int64_t n,
                                                 StencilView E, StencilView C, const double *__restrict__ x0,
                                                 const int32_t *__restrict__ rawptr, int32_t *__restrict__ raw_col,
                                                 double *__restrict__ raw_val, int32_t *__restrict__ row_count,
                                                 double *__restrict__ rhs, int *err, const int32_t *__restrict__ ids)
{
    // `ids` (multi-GPU: the rows this rank owns) maps output row -> velocity DOF; nullptr = all DOFs in order
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= n) return;
    const int32_t vi = ids ? ids[row] : (int32_t)row;
    const int4 rec = reinterpret_cast<const int4 *>(vdof)[vi];
    const int level = rec.x & 0xff, axis = rec.x >> 8;
    const I3 face{{rec.y, rec.z, rec.w}};
    const I3 cr = cell_res(P, level);
    const I3 fr = face_res(P, level, axis);
    RowAcc ra{0, nullptr, nullptr, 0., 0., 0, P.f32};
    if (EMIT) { // rawptr: first slot of the row's WAVE (64 consecutive rows) in the transposed raw arrays
        const size_t base = (size_t)rawptr[row >> 6] + (size_t)(row & 63);
        ra.col = raw_col + base;
        ra.val = raw_val + base;
    }

    // centre stresses + inset T-junction edge stresses of the two axial cells, cpp:2547-2650
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        I3 cell = face;
        if (dir == 0) --cell[axis];
        if (cell[axis] < 0 || cell[axis] >= cr[axis]) continue;
        I3 sc = cell;
        int sl = level;
        if (P.labels[level][lin(cr, cell)] != AVS_ACTIVE) { // face grading: the parent is the leaf
            sc = half3(cell);
            sl = level + 1;
            if (sl >= P.levels) { ra.bad = 1; continue; }
        }
        const int32_t ci = P.cidx[sl][lin(cell_res(P, sl), sc)];
        if (ci >= 0) apply_center<EMIT>(ra, vi, ci, axis, C);
#pragma unroll
        for (int fa = 0; fa < 3; ++fa) {
            if (fa == axis) continue;
            const int ea = 3 - fa - axis;
#pragma unroll
            for (int fd = 0; fd < 2; ++fd) {
                I3 af = sc;
                if (fd == 1) ++af[fa];
                if (vidx_at(P, sl, fa, af) != AVS_UNASSIGNED) continue;
                if (sl == 0) { ra.bad = 1; continue; }
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) { // getChildEdgeInFace, oct.h:126-142
                    I3 e{{2 * af[0], 2 * af[1], 2 * af[2]}};
                    if (ii == 1) ++e[ea];
                    ++e[axis]; // 3 - faceAxis - edgeAxis == axis
                    const int32_t ei = eidx_at(P, sl - 1, ea, e);
                    if (ei >= 0) apply_edge<EMIT>(ra, vi, ei, E);
                }
            }
        }
    }
    // the four edges around the face, cpp:2652-2745
#pragma unroll
    for (int ea = 0; ea < 3; ++ea) {
        if (ea == axis) continue;
        const int ta = 3 - ea - axis;
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            I3 e = face;
            if (dir == 1) ++e[ta]; // HDKfaceToEdge, util.h:115-131
            const int32_t ei = eidx_at(P, level, ea, e);
            if (ei >= 0) {
                if (P.enhanced) { // cpp:2664-2697
                    I3 af = face;
                    af[ta] += (dir == 0) ? -1 : 1;
                    if (af[ta] >= 0 && af[ta] < fr[ta] && P.vidx[level][axis][lin(fr, af)] == AVS_UNASSIGNED) {
                        I3 se = e;
                        se[ea] += (e[ea] % 2 == 0) ? 1 : -1;
                        const int32_t sei = eidx_at(P, level, ea, se);
                        if (sei < 0) ra.bad = 1; // assert cpp:2680
                        else apply_edge<EMIT>(ra, vi, sei, E);
                    }
                }
                apply_edge<EMIT>(ra, vi, ei, E);
            } else if (ei == AVS_UNASSIGNED && level > 0) { // cpp:2714-2742
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    I3 ce{{2 * e[0], 2 * e[1], 2 * e[2]}};
                    if (ci > 0) ++ce[ea];
                    const int32_t cei = eidx_at(P, level - 1, ea, ce);
                    if (cei >= 0) apply_edge<EMIT>(ra, vi, cei, E);
                }
            }
        }
    }
    // mass term and right-hand side, cpp:2748-2772
    if (EMIT) {
        double fw;
        int bad = 0;
        if (level == 0) {
            fw = (double)field_at(P.facew[axis], face_res(P, 0, axis), face);
            if (fw == 1.) fw = face_octree_volume(P, level, axis, face, bad);
        } else fw = face_octree_volume(P, level, axis, face, bad);
        if (bad) ra.bad = 1;
        if (P.dens.is_const) fw *= (double)P.dens.cval;
        else fw *= (double)sample_f32(P.dens, cell_res(P, 0), I3{{1, 1, 1}}, pos2_face(level, axis, face));
        row_push<true>(ra, vi, fw + ra.diag);
        ra.rhs = solve_type(ra.rhs + fw * x0[vi], ra.f32); // x0 is indexed by DOF (cpp:2772; x0 already holds SolveType values)
        rhs[row] = ra.rhs;
    } else ++ra.n;
    if (EMIT) {
        if (ra.n != row_count[row]) *err = 8; // the dry run and the emit pass must agree
    } else row_count[row] = ra.n;
    if (ra.bad) *err = 7;
}

// ---------------------------------------------------------------------------------------------
// K6b + K7: Eigen::SparseMatrix::setFromTriplets (cpp:613-614) per row: stable sort by column, duplicates summed left to right in
// emission order -- straight into the final CSR.
//
// Round 1 emitted the raw triplets row-contiguously (17 dependent 12-B stores per thread: 4x the bytes in partial-line
// traffic), re-read them with a half-wave per row, sorted in registers, wrote them back and compacted: 9.7 ms at 512^3.  Now
// the raw arrays are wave-transposed (entry k of lane l at wave_base + 64 k + l): the k-th push of a wave is one coalesced
// store, and ONE thread per row can walk its entries with coalesced, cache-friendly loads.  A row has ~17 raw entries, so
// the O(R^2) ranking below is a few hundred L1 hits per row:
//   k_wave_slots   slots of a wave = 64 x its longest row
//   k_unique_rows  first-occurrence mask of every row -> unique counts (row pointers by scan)
//   k_merge_rows   rank among the first occurrences + left-to-right fold of the later duplicates; the wave's 64 merged rows are
//                  contiguous in the CSR, so they are staged in LDS and written out with coalesced stores
// Rows of <= 32 raw entries (all but a few transition rows) live in registers: independent coalesced loads, unrolled compares.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_wave_slots(int64_t n, const int32_t *__restrict__ row_count, int32_t *__restrict__ wave_slots)
{
    const int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t nw = (n + 63) >> 6;
    if (w > nw) return;
    if (w == nw) { wave_slots[w] = 0; return; }
    int m = 0;
    for (int l = 0; l < 64; ++l) {
        const int64_t r = w * 64 + l;
        if (r < n) m = max(m, row_count[r]);
    }
    wave_slots[w] = m * kRawStride;
}

// The common row (<= kFast raw entries) is handled entirely in registers by its own thread: all of its columns and values are
// requested with independent, coalesced loads, then ranked / folded with fully unrolled compares -- no dependent load in any
// loop.  A row of 33..64 raw entries (transition rows) is handled by its WHOLE WAVE: lane e holds raw entry e, ranks are
// counted with v_readlane broadcasts (the algorithm of round 1's k_sort_rows, 64 lanes wide).  Rows of more than 64 raw
// entries (a coarse face surrounded by finer ones: a handful per scene, but a serial O(R^3) path for them took 15 ms) are
// walked by the wave too, 64 entries at a time (k_unique_rows marks their later duplicates with the sign bit).
// (A single-pass variant -- unique counts, row pointers by decoupled look-back and the merge in one kernel -- was measured
// SLOWER, 3.5-3.7 ms against 0.86 + 1.95 ms: the waves of transition rows publish their counts late and every later
// workgroup waits for them.)
static constexpr int kFast = 32;

__device__ __forceinline__ void load_cols(const int32_t *__restrict__ rc, int R, int32_t (&c)[kFast])
{
#pragma unroll
    for (int k = 0; k < kFast; ++k) c[k] = k < R ? rc[(size_t)k * kRawStride] : INT32_MAX;
}

__device__ __forceinline__ unsigned first_mask_fast(const int32_t (&c)[kFast], int R, int wmax /* wave-uniform, >= R */)
{
    unsigned mask = 0u;
#pragma unroll
    for (int k = 0; k < kFast; ++k) {
        if (k < wmax) {
            bool dup = false;
#pragma unroll
            for (int j = 0; j < k; ++j) dup |= c[j] == c[k];
            if (k < R && !dup) mask |= 1u << k;
        }
    }
    return mask;
}

__device__ __forceinline__ double lane_bcast(double v, int q) // q wave-uniform
{
    const int l = __builtin_amdgcn_readlane(__double2loint(v), q);
    const int h = __builtin_amdgcn_readlane(__double2hiint(v), q);
    return __hiloint2double(h, l);
}

// wave-cooperative: is raw entry `lane` of the row (c_e; INT32_MAX beyond R) the first occurrence of its column?
__device__ __forceinline__ bool coop_first(int32_t c_e, int R, int lane)
{
    bool first = lane < R;
    for (int q = 0; q < R; ++q) {
        const int32_t cq = __builtin_amdgcn_readlane(c_e, q);
        if (q < lane && cq == c_e) first = false;
    }
    return first;
}

static constexpr int32_t kDupBit = (int32_t)0x80000000u; // rows of > 64 raw entries: k_unique_rows marks the later duplicates
static constexpr int32_t kColMask = 0x7fffffff;

__global__ __launch_bounds__(kBlock) void k_unique_rows(int64_t n, const int32_t *__restrict__ rawptr, int32_t *__restrict__ raw_col,
                                                        const int32_t *__restrict__ row_count, int32_t *__restrict__ ucount,
                                                        int32_t *__restrict__ long_rows /* [0]: count (zeroed by the caller), then rows of > 64 raw entries */)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (row == n) ucount[n] = 0;
    const int R = row < n ? row_count[row] : 0;
    const int64_t wrow0 = row - lane;
    const size_t wbase = wrow0 < n ? (size_t)rawptr[wrow0 >> 6] : 0; // first raw slot of this wave (uniform)
    const int wmax = wave_max_i32((row < n && R <= kFast) ? R : 0);
    if (row < n && R <= kFast) {
        int32_t c[kFast];
        load_cols(raw_col + wbase + lane, R, c);
        ucount[row] = __popc(first_mask_fast(c, R, wmax));
    }
    // longer rows (a coarse face surrounded by finer ones) go on a list for k_unique_long: they are consecutive DOFs of the coarsest level,
    // i.e. they all sat in the LAST waves of this launch, which walked them one after the other (R^2 steps each) while the rest of the
    // GPU was done -- viscousBeam equivalent: 325 us for 469 k rows, a third of what 7.4 M rows take
    if (row < n && R > 64) long_rows[1 + atomicAdd(long_rows, 1)] = (int32_t)row; // (the order of the list does not matter)
    unsigned long long todo = __ballot(row < n && R > kFast && R <= 64);
    while (todo) { // wave-uniform loop over the rows that need the whole wave
        const int which = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int Rr = __builtin_amdgcn_readlane(R, which);
        const int32_t c_e = lane < Rr ? raw_col[wbase + (size_t)lane * kRawStride + which] : INT32_MAX;
        const unsigned long long firsts = __ballot(coop_first(c_e, Rr, lane));
        if (lane == which) ucount[row] = __popcll(firsts);
    }
}

// one WAVE per listed row of more than 64 raw entries: 64 entries at a time against the earlier ones; every later duplicate gets the sign
// bit, so that k_merge_long does not have to search for first occurrences again
__global__ __launch_bounds__(kBlock) void k_unique_long(const int32_t *__restrict__ rawptr, int32_t *__restrict__ raw_col, const int32_t *__restrict__ row_count,
                                                        int32_t *__restrict__ ucount, const int32_t *__restrict__ long_rows)
{
    const int lane = threadIdx.x & 63;
    const int nw = (int)gridDim.x * (kBlock / 64);
    const int n_list = long_rows[0];
    for (int li = (int)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); li < n_list; li += nw) {
        const int64_t row = long_rows[1 + li];
        const int which = (int)(row & 63);
        const size_t wbase = (size_t)rawptr[row >> 6];
        const int Rr = row_count[row];
        int count = 0;
        for (int a0 = 0; a0 < Rr; a0 += 64) {
            const int e = a0 + lane;
            const size_t at = wbase + (size_t)e * kRawStride + which;
            const int32_t c_e = e < Rr ? raw_col[at] : INT32_MAX;
            bool first = e < Rr;
            for (int b0 = 0; b0 <= a0; b0 += 64) {
                const int32_t cb = b0 == a0 ? c_e : raw_col[wbase + (size_t)(b0 + lane) * kRawStride + which]; // earlier chunks are full
                const int nb = Rr - b0 < 64 ? Rr - b0 : 64;
                for (int q = 0; q < nb; ++q) {
                    const int32_t cq = __builtin_amdgcn_readlane(cb, q);
                    if (b0 + q < e && (cq & kColMask) == c_e) first = false; // (an earlier chunk may already carry its marks)
                }
            }
            if (e < Rr && !first) raw_col[at] = c_e | kDupBit;
            count += __popcll(__ballot(first));
        }
        if (lane == 0) ucount[row] = count;
    }
}

static constexpr unsigned kLongGrid = 512; // workgroups (four waves each) walking the list of long rows
static constexpr int kMergeLds = 1088; // merged entries of one wave staged in LDS (64 rows x 17; interior rows have 15): 12.75 KiB per wave, 3 workgroups per CU

// F32: the duplicates of SolveType = fpreal32 triplets are summed in float (Eigen::SparseMatrix<float>::setFromTriplets); the raw values
// are float values already (row_push)
template <bool F32> __device__ __forceinline__ double merge_add(double a, double b)
{
    return F32 ? (double)((float)a + (float)b) : a + b;
}
template <bool F32>
__global__ __launch_bounds__(kBlock) void k_merge_rows(int64_t n, const int32_t *__restrict__ rawptr, const int32_t *__restrict__ raw_col,
                                                       const double *__restrict__ raw_val, const int32_t *__restrict__ row_count,
                                                       const int32_t *__restrict__ row_ptr, int32_t *__restrict__ col, double *__restrict__ val)
{
    __shared__ int32_t lcol[kBlock / 64][kMergeLds];
    __shared__ double lval[kBlock / 64][kMergeLds];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t wrow0 = row - lane;                                   // first row of this wave
    const int64_t wlast = (wrow0 + 64 < n) ? wrow0 + 64 : n;
    const int seg0 = wrow0 < n ? row_ptr[wrow0] : 0;
    const int seglen = wrow0 < n ? row_ptr[wlast] - seg0 : 0;
    const bool staged = seglen <= kMergeLds;                            // wave-uniform
    const size_t wbase = wrow0 < n ? (size_t)rawptr[wrow0 >> 6] : 0;    // first raw slot of this wave
    const int R = row < n ? row_count[row] : 0;
    const int dst0 = row < n ? row_ptr[row] : 0;
    // output addressed by CSR position minus `ooff` (never form an LDS pointer below its array: LDS pointers are 32-bit)
    int32_t *ocw = staged ? &lcol[wv][0] : col;
    double *ovw = staged ? &lval[wv][0] : val;
    const int ooff = staged ? seg0 : 0;
    bool done = row >= n;
    const int wmax = wave_max_i32((row < n && R <= kFast) ? R : 0); // longest register-path row of this wave (uniform)
    if (row < n && R <= kFast) {
        int32_t c[kFast];
        double v[kFast];
        load_cols(raw_col + wbase + lane, R, c);
        const double *rv = raw_val + wbase + lane;
#pragma unroll
        for (int k = 0; k < kFast; ++k) v[k] = k < R ? rv[(size_t)k * kRawStride] : 0.;
        // A neighbour face usually appears in several of the row's stress stencils: most rows DO carry duplicate columns.
        // One triangular pass finds the first occurrences AND folds: every later duplicate k is added to ALL earlier entries j
        // of the same column, in ascending k = emission order (v[k] is still the raw value then: it is only ever modified by
        // pairs (k, k' > k)); the sums of non-first entries are simply not used.  Everything is indexed statically; the
        // wave-uniform bound keeps the work at the wave's longest row instead of 32.
        unsigned m = 0u;
#pragma unroll
        for (int k = 0; k < kFast; ++k) {
            if (k < wmax) {
                bool dup = false;
#pragma unroll
                for (int j = 0; j < k; ++j) {
                    const bool eq = c[j] == c[k];
                    dup |= eq;
                    v[j] = eq ? merge_add<F32>(v[j], v[k]) : v[j];
                }
                if (k < R && !dup) m |= 1u << k;
            }
        }
#pragma unroll
        for (int k = 0; k < kFast; ++k) c[k] = ((m >> k) & 1u) ? c[k] : INT32_MAX; // only first occurrences take part in the ranking
#pragma unroll
        for (int k = 0; k < kFast; ++k) {
            if (k < wmax) {
                int urank = 0;
#pragma unroll
                for (int j0 = 0; j0 < kFast; j0 += 8) {
                    if (j0 < wmax) {
#pragma unroll
                        for (int j = j0; j < j0 + 8; ++j) urank += (int)(c[j] < c[k]);
                    }
                }
                if ((m >> k) & 1u) {
                    ocw[dst0 - ooff + urank] = c[k];
                    ovw[dst0 - ooff + urank] = v[k];
                }
            }
        }
        done = true;
    }
    if (R > 64) done = true; // (k_merge_long, one wave per such row, behind this launch)
    unsigned long long todo = __ballot(!done);
    while (todo) { // rows of 33..64 raw entries: the whole wave, lane e = raw entry e
        const int which = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int Rr = __builtin_amdgcn_readlane(R, which);
        const int dr = __builtin_amdgcn_readlane(dst0, which);
        const size_t at = wbase + (size_t)lane * kRawStride + which;
        const int32_t c_e = lane < Rr ? raw_col[at] : INT32_MAX;
        const double v_e = lane < Rr ? raw_val[at] : 0.;
        const bool first = coop_first(c_e, Rr, lane);
        const unsigned long long firsts = __ballot(first);
        int urank = 0;
        double sum = v_e;
        for (int q = 0; q < Rr; ++q) {
            const int32_t cq = __builtin_amdgcn_readlane(c_e, q);
            const double vq = lane_bcast(v_e, q);
            if (((firsts >> q) & 1ull) && cq < c_e) ++urank;
            if (first && q > lane && cq == c_e) sum = merge_add<F32>(sum, vq); // left fold in emission order
        }
        if (first) {
            ocw[dr - ooff + urank] = c_e;
            ovw[dr - ooff + urank] = sum;
        }
    }
    __syncthreads();
    if (staged)
        for (int e = lane; e < seglen; e += 64) { // the wave's 64 merged rows are contiguous in the CSR: coalesced stores
            col[seg0 + e] = lcol[wv][e];
            val[seg0 + e] = lval[wv][e];
        }
}

// rows of more than 64 raw entries, any length: ONE WAVE per listed row, 64 entries at a time against 64 at a time, straight into the CSR
// (launched behind k_merge_rows: where that kernel staged the wave's segment in LDS it copied these rows' slots unset)
template <bool F32>
__global__ __launch_bounds__(kBlock) void k_merge_long(const int32_t *__restrict__ rawptr, const int32_t *__restrict__ raw_col, const double *__restrict__ raw_val,
                                                       const int32_t *__restrict__ row_count, const int32_t *__restrict__ row_ptr, int32_t *__restrict__ col,
                                                       double *__restrict__ val, const int32_t *__restrict__ long_rows)
{
    const int lane = threadIdx.x & 63;
    const int nw = (int)gridDim.x * (kBlock / 64);
    const int n_list = long_rows[0];
    for (int li = (int)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); li < n_list; li += nw) {
        const int64_t row = long_rows[1 + li];
        const int which = (int)(row & 63);
        const size_t wbase = (size_t)rawptr[row >> 6];
        const int Rr = row_count[row];
        const int dr = row_ptr[row];
        for (int a0 = 0; a0 < Rr; a0 += 64) {
            const int e = a0 + lane;
            const size_t at = wbase + (size_t)e * kRawStride + which;
            const int32_t craw = e < Rr ? raw_col[at] : INT32_MAX;   // sign bit: a later duplicate (k_unique_long)
            const bool first = e < Rr && craw >= 0;
            const int32_t c_e = craw & kColMask;
            double sum = e < Rr ? raw_val[at] : 0.;
            int urank = 0;
            for (int b0 = 0; b0 < Rr; b0 += 64) {
                const int eb = b0 + lane;
                const size_t bt = wbase + (size_t)eb * kRawStride + which;
                const int32_t cb = eb < Rr ? raw_col[bt] : INT32_MAX;
                const double vb = eb < Rr ? raw_val[bt] : 0.;
                const int nb = Rr - b0 < 64 ? Rr - b0 : 64;
                for (int q = 0; q < nb; ++q) {
                    const int32_t cq = __builtin_amdgcn_readlane(cb, q);
                    const double vq = lane_bcast(vb, q);
                    if (cq >= 0 && cq < c_e) ++urank;                                   // first occurrences only
                    if (first && b0 + q > e && (cq & kColMask) == c_e) sum = merge_add<F32>(sum, vq); // left fold in emission order
                }
            }
            if (first) {
                col[dr + urank] = c_e;
                val[dr + urank] = sum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K5: exclusive scan (int32), three phases, wave64 shuffles inside the block
// ---------------------------------------------------------------------------------------------
static constexpr int kScanItems = 8;
static constexpr int kScanTile = kBlock * kScanItems;

__device__ __forceinline__ int block_exclusive_scan(int v, int *lds_wave /*[4]*/, int &block_total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    __syncthreads();
    if (lane == 63) lds_wave[wave] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) base += lds_wave[w];
    block_total = lds_wave[0] + lds_wave[1] + lds_wave[2] + lds_wave[3];
    return base + inc - v;
}

__global__ __launch_bounds__(kBlock) void k_scan_sums(const int32_t *__restrict__ in, int64_t n, int32_t *__restrict__ sums)
{
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
        if (base + i < n) s += in[base + i];
    int total;
    (void)block_exclusive_scan(s, lds, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// single block: exclusive scan of the block sums in place; sums[nb] = grand total
__global__ __launch_bounds__(kBlock) void k_scan_top(int32_t *__restrict__ sums, int64_t nb)
{
    __shared__ int lds[4];
    __shared__ long long carry; // 64-bit: a total beyond int32 is reported as -1 instead of wrapping (any number of times)
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t start = 0; start < nb; start += kBlock) {
        const int64_t i = start + threadIdx.x;
        const int v = (i < nb) ? sums[i] : 0; // one tile's sum: <= kScanTile small counts, cannot wrap
        int total;
        const int ex = block_exclusive_scan(v, lds, total);
        const long long c = carry;
        if (i < nb) sums[i] = (int)(c + ex);
        __syncthreads();
        if (threadIdx.x == 0) carry = c + (long long)(unsigned)total;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[nb] = carry > 2147483647ll ? -1 : (int)carry;
}

__global__ __launch_bounds__(kBlock) void k_scan_apply(const int32_t *__restrict__ in, int32_t *__restrict__ out, int64_t n,
                                                       const int32_t *__restrict__ sums, int64_t nb)
{
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int v[kScanItems];
    int s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        s += v[i];
    }
    int total;
    int ex = block_exclusive_scan(s, lds, total) + sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = sums[nb];
}

size_t scan_tmp_elems(int64_t n) { return (size_t)((n + kScanTile - 1) / kScanTile) + 2; }

avs_status exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, int32_t *tmp, size_t tmp_elems,
                              hipStream_t stream)
{
    const int64_t nb = (n + kScanTile - 1) / kScanTile;
    AVS_REQUIRE(tmp_elems >= (size_t)nb + 1, AVS_EINTERNAL, "scan scratch too small");
    if (n == 0) {
        AVS_HIP(hipMemsetAsync(out, 0, sizeof(int32_t), stream));
        return AVS_OK;
    }
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(kBlock), 0, stream, in, n, tmp);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kBlock), 0, stream, tmp, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(kBlock), 0, stream, in, out, n, tmp, nb);
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

// ---------------------------------------------------------------------------------------------
// host-side phase drivers (called from avs_api.hip)
// ---------------------------------------------------------------------------------------------
static inline unsigned grid_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock > 0 ? (n + kBlock - 1) / kBlock : 1); }

static avs_status read_err(int *d_err, hipStream_t stream, int *out)
{
    AVS_HIP(hipMemcpyAsync(out, d_err, sizeof(int), hipMemcpyDeviceToHost, stream));
    AVS_HIP(hipStreamSynchronize(stream));
    return AVS_OK;
}

avs_status build_dof_tables(avs_ctx *c)
{
    hipStream_t st = c->stream;
    const int L = c->desc.levels;
    AVS_REQUIRE(c->n_vel >= 0 && c->n_edge >= 0 && c->n_center >= 0, AVS_ESTATE, "avs_set_dof_counts was not called");
    for (int l = 0; l < L; ++l) {
        AVS_REQUIRE(c->have_labels[l] && c->have_cidx[l], AVS_ESTATE, "labels / centre indices of level %d missing", l);
        for (int a = 0; a < 3; ++a)
            AVS_REQUIRE(c->have_vidx[l][a] && c->have_eidx[l][a], AVS_ESTATE, "index grids of level %d axis %d missing", l, a);
    }
    AVS_TRY(c->vdof.alloc((size_t)c->n_vel * 4));
    AVS_TRY(c->edof.alloc((size_t)c->n_edge * 4));
    AVS_TRY(c->cdof.alloc((size_t)c->n_center * 4));
    AVS_HIP(hipMemsetAsync(c->vdof.p, 0xff, c->vdof.n * sizeof(int32_t), st));
    AVS_HIP(hipMemsetAsync(c->edof.p, 0xff, c->edof.n * sizeof(int32_t), st));
    AVS_HIP(hipMemsetAsync(c->cdof.p, 0xff, c->cdof.n * sizeof(int32_t), st));
    DevBuf<int> err;
    AVS_TRY(err.alloc(1));
    AVS_HIP(hipMemsetAsync(err.p, 0, sizeof(int), st));
    PyramidView P = c->view();
    for (int l = 0; l < L; ++l) {
        const int cr[3] = {P.n[0] >> l, P.n[1] >> l, P.n[2] >> l};
        for (int a = 0; a < 3; ++a) {
            I3 fr{{cr[0], cr[1], cr[2]}};
            fr.v[a] += 1;
            I3 er{{cr[0] + (a != 0), cr[1] + (a != 1), cr[2] + (a != 2)}};
            const size_t nf = (size_t)fr.v[0] * fr.v[1] * fr.v[2], ne = (size_t)er.v[0] * er.v[1] * er.v[2];
            hipLaunchKernelGGL(k_dof_table, dim3(grid_for((int64_t)nf) < 8192 ? grid_for((int64_t)nf) : 8192), dim3(kBlock), 0, st,
                               c->vidx[l][a].p, fr, l, a, c->vdof.p, c->n_vel, err.p);
            hipLaunchKernelGGL(k_dof_table, dim3(grid_for((int64_t)ne) < 8192 ? grid_for((int64_t)ne) : 8192), dim3(kBlock), 0, st,
                               c->eidx[l][a].p, er, l, a, c->edof.p, c->n_edge, err.p);
        }
        I3 r{{cr[0], cr[1], cr[2]}};
        const size_t ncell = (size_t)cr[0] * cr[1] * cr[2];
        hipLaunchKernelGGL(k_dof_table, dim3(grid_for((int64_t)ncell) < 8192 ? grid_for((int64_t)ncell) : 8192), dim3(kBlock), 0, st,
                           c->cidx[l].p, r, l, 0, c->cdof.p, c->n_center, err.p);
    }
    if (c->n_vel) hipLaunchKernelGGL(k_check_table, dim3(grid_for(c->n_vel) < 8192 ? grid_for(c->n_vel) : 8192), dim3(kBlock), 0, st, c->vdof.p, c->n_vel, err.p);
    if (c->n_edge) hipLaunchKernelGGL(k_check_table, dim3(grid_for(c->n_edge) < 8192 ? grid_for(c->n_edge) : 8192), dim3(kBlock), 0, st, c->edof.p, c->n_edge, err.p);
    if (c->n_center) hipLaunchKernelGGL(k_check_table, dim3(grid_for(c->n_center) < 8192 ? grid_for(c->n_center) : 8192), dim3(kBlock), 0, st, c->cdof.p, c->n_center, err.p);
    AVS_HIP(hipGetLastError());
    int e = 0;
    AVS_TRY(read_err(err.p, st, &e));
    AVS_REQUIRE(e == 0, AVS_EINVAL,
                e == 1 ? "an index grid holds an id >= the declared DOF count"
                       : (e == 2 ? "an index grid holds a value below AVS_OUTSIDE"
                                 : "declared DOF counts exceed the ids present in the index grids"));
    c->tables_ready = true;
    return AVS_OK;
}

static StencilView edge_view(avs_ctx *c)
{
    return StencilView{c->n_edge, c->e_cnt.p, c->e_idx.p, c->e_bcnt.p, c->e_coef.p, c->e_bval.p, c->e_w.p};
}
static StencilView center_view(avs_ctx *c)
{
    return StencilView{c->n_center * 3, c->c_cnt.p, c->c_idx.p, c->c_bcnt.p, c->c_coef.p, c->c_bval.p, c->c_w.p};
}

avs_status build_stencils(avs_ctx *c)
{
    hipStream_t st = c->stream;
    if (!c->tables_ready) AVS_TRY(build_dof_tables(c));
    const size_t ne = (size_t)c->n_edge, n3 = (size_t)c->n_center * 3;
    AVS_TRY(c->e_cnt.alloc(ne));
    AVS_TRY(c->e_bcnt.alloc(ne));
    AVS_TRY(c->e_w.alloc(ne));
    AVS_TRY(c->e_idx.alloc(ne * AVS_EDGE_STENCIL_CAP));
    AVS_TRY(c->e_coef.alloc(ne * AVS_EDGE_STENCIL_CAP));
    AVS_TRY(c->e_bval.alloc(ne * AVS_EDGE_BOUNDARY_CAP));
    AVS_TRY(c->c_cnt.alloc(n3));
    AVS_TRY(c->c_bcnt.alloc(n3));
    AVS_TRY(c->c_w.alloc((size_t)c->n_center));
    AVS_TRY(c->c_idx.alloc(n3 * AVS_CENTER_STENCIL_CAP));
    AVS_TRY(c->c_coef.alloc(n3 * AVS_CENTER_STENCIL_CAP));
    AVS_TRY(c->c_bval.alloc(n3 * AVS_CENTER_BOUNDARY_CAP));
    // the slots beyond a stencil's count are never read by the row sweep; pad_stencils() fills them for a read-back
    DevBuf<int> err;
    AVS_TRY(err.alloc(1));
    AVS_HIP(hipMemsetAsync(err.p, 0, sizeof(int), st));
    PyramidView P = c->view();
    StencilCore ce{}, cc{};
    int64_t me = c->n_edge, mc = c->n_center;
    if (c->slab.on) { // the stresses near this rank's slab only; everybody else's count stays 0
        AVS_REQUIRE(c->wlist[1].p && c->wlist[2].p, AVS_ESTATE, "slab-local context without window lists");
        AVS_HIP(hipMemsetAsync(c->e_cnt.p, 0, ne * sizeof(int32_t), st));
        AVS_HIP(hipMemsetAsync(c->c_cnt.p, 0, n3 * sizeof(int32_t), st));
        ce.list = c->wlist[1].p; ce.m = me = c->n_window[1];
        cc.list = c->wlist[2].p; cc.m = mc = c->n_window[2];
        ce.axis = cc.axis = c->slab.axis;
        for (int l = 0; l < c->desc.levels; ++l) {
            const int s_lo = c->slab.cuts[c->slab.rank] >> l, s_hi = (c->slab.cuts[c->slab.rank + 1] + (1 << l) - 1) >> l;
            ce.lo[l] = cc.lo[l] = s_lo - kSlabStencilMargin;
            ce.hi[l] = cc.hi[l] = s_hi + kSlabStencilMargin + 1; // (+ 1: the edge lattice's extra entry)
        }
    }
    if (me) {
        Scope sc("Build Edge Stress Stencils"); // cpp:441
        hipLaunchKernelGGL(k_edge_stencils, dim3(grid_for(me)), dim3(kBlock), 0, st, P, c->edof.p, edge_view(c), err.p, ce);
    }
    if (mc) {
        Scope sc("Build Cell Stress Stencils"); // cpp:473
        hipLaunchKernelGGL(k_center_stencils, dim3(grid_for(mc)), dim3(kBlock), 0, st, P, c->cdof.p, center_view(c), err.p, cc);
    }
    AVS_HIP(hipGetLastError());
    int e = 0;
    AVS_TRY(read_err(err.p, st, &e));
    AVS_REQUIRE(e == 0, AVS_EINTERNAL, "stencil construction hit a reference assert (code %d): the index pyramids are inconsistent", e);
    c->stencils_ready = true;
    return AVS_OK;
}

// unused slots read back as (-1, 0.0) like the oracle's: done when somebody asks for the stencils (avs_get_*_stencils), not in
// every assembly (six memsets, 0.56 ms at 512^3)
__global__ __launch_bounds__(kBlock) void k_pad_stencils(StencilView S, int cap, int bcap, int32_t *__restrict__ idx, double *__restrict__ coef,
                                                         double *__restrict__ bval)
{
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= S.count) return;
    for (int i = S.cnt[s]; i < cap; ++i) {
        idx[(size_t)i * S.count + s] = -1;
        coef[(size_t)i * S.count + s] = 0.;
    }
    for (int i = S.bcnt[s]; i < bcap; ++i) bval[(size_t)i * S.count + s] = 0.;
}

avs_status pad_stencils(avs_ctx *c, bool edge)
{
    StencilView S = edge ? edge_view(c) : center_view(c);
    if (S.count == 0) return AVS_OK;
    hipLaunchKernelGGL(k_pad_stencils, dim3(grid_for(S.count)), dim3(kBlock), 0, c->stream, S, edge ? AVS_EDGE_STENCIL_CAP : AVS_CENTER_STENCIL_CAP,
                       edge ? AVS_EDGE_BOUNDARY_CAP : AVS_CENTER_BOUNDARY_CAP, edge ? c->e_idx.p : c->c_idx.p, edge ? c->e_coef.p : c->c_coef.p,
                       edge ? c->e_bval.p : c->c_bval.p);
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

avs_status build_initial_guess(avs_ctx *c)
{
    if (!c->tables_ready) AVS_TRY(build_dof_tables(c));
    AVS_TRY(c->x0.alloc((size_t)c->n_vel));
    if (c->n_vel) {
        AVS_TRY(c->scratch.coarse_list.reserve((size_t)c->n_vel + 1));
        AVS_HIP(hipMemsetAsync(c->scratch.coarse_list.p, 0, sizeof(int32_t), c->stream));
        hipLaunchKernelGGL(k_initial_guess, dim3(grid_for(c->n_vel)), dim3(kBlock), 0, c->stream, c->view(), c->vdof.p, c->n_vel, c->x0.p,
                           (const int32_t *)nullptr, c->scratch.coarse_list.p);
        if (c->desc.levels > 3)
            hipLaunchKernelGGL(k_initial_guess_coarse, dim3(kCoarseGrid), dim3(kBlock), 0, c->stream, c->view(), c->vdof.p, c->x0.p,
                               (const int32_t *)c->scratch.coarse_list.p);
    }
    AVS_HIP(hipGetLastError());
    c->guess_ready = true;
    c->guess_partial = false;
    return AVS_OK;
}

// restriction for the DOFs `ids[0..m)` only; the other entries of x0 are zero and must not be read
avs_status build_initial_guess_rows(avs_ctx *c, const int32_t *ids, int64_t m)
{
    if (!c->tables_ready) AVS_TRY(build_dof_tables(c));
    AVS_TRY(c->x0.alloc((size_t)c->n_vel));
    AVS_HIP(hipMemsetAsync(c->x0.p, 0, (size_t)c->n_vel * sizeof(double), c->stream));
    if (m) {
        AVS_TRY(c->scratch.coarse_list.reserve((size_t)m + 1));
        AVS_HIP(hipMemsetAsync(c->scratch.coarse_list.p, 0, sizeof(int32_t), c->stream));
        hipLaunchKernelGGL(k_initial_guess, dim3(grid_for(m)), dim3(kBlock), 0, c->stream, c->view(), c->vdof.p, m, c->x0.p, ids, c->scratch.coarse_list.p);
        if (c->desc.levels > 3)
            hipLaunchKernelGGL(k_initial_guess_coarse, dim3(kCoarseGrid), dim3(kBlock), 0, c->stream, c->view(), c->vdof.p, c->x0.p,
                               (const int32_t *)c->scratch.coarse_list.p);
    }
    AVS_HIP(hipGetLastError());
    c->guess_ready = false; // avs_get_initial_guess / avs_build_system must not see a mostly-zero vector
    c->guess_partial = true;
    return AVS_OK;
}

// rows `ids[0..m)` (velocity DOFs; nullptr = all, in order) -> CSR with reference-numbered columns + rhs.
// Row-local work: a rank of a multi-GPU solve assembles only the rows it owns.
avs_status assemble_rows(avs_ctx *c, const int32_t *ids, int64_t m, DevBuf<int32_t> &row_ptr, DevBuf<int32_t> &col, DevBuf<double> &val,
                         DevBuf<double> &rhs, int64_t *nnz_out, int64_t *nraw_out)
{
    hipStream_t st = c->stream;
    AVS_REQUIRE(c->stencils_ready && (c->guess_ready || (ids && c->guess_partial)), AVS_ESTATE,
                "build the stencils and the initial guess first");
    const int64_t n = m;
    AVS_REQUIRE(c->n_vel < (int64_t)INT32_MAX, AVS_EINVAL, "too many DOFs for int32 columns");
    DevBuf<int32_t> &row_count = c->scratch.row_count, &rawptr = c->scratch.rawptr, &scan_tmp = c->scratch.scan_tmp, &raw_col = c->scratch.raw_col;
    DevBuf<double> &raw_val = c->scratch.raw_val;
    DevBuf<int> &err = c->scratch.err;
    AVS_TRY(row_count.reserve((size_t)n + 1));
    AVS_TRY(rawptr.reserve((size_t)n + 1));
    AVS_TRY(scan_tmp.reserve(scan_tmp_elems(n)));
    AVS_TRY(err.reserve(1));
    // (reserve, not alloc: the sizes change from frame to frame, and a hipFree + hipMalloc of the 100-MB arrays costs more than the rows)
    AVS_TRY(rhs.reserve((size_t)n));
    AVS_TRY(row_ptr.reserve((size_t)n + 1));
    AVS_HIP(hipMemsetAsync(err.p, 0, sizeof(int), st));
    PyramidView P = c->view();
    StencilView E = edge_view(c), C = center_view(c);
    PhaseTrace tr(st, "rows", cur_opt().trace_phases != 0);
    // K4 dry run -> raw triplet counts
    if (n) hipLaunchKernelGGL((k_rows<false>), dim3(grid_for(n)), dim3(kBlock), 0, st, P, c->vdof.p, n, E, C, c->x0.p,
                              (const int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr, row_count.p, (double *)nullptr, err.p, ids);
    // raw total (reported) and the wave-transposed layout: a wave of 64 rows gets 64 x its longest row
    AVS_TRY(exclusive_scan_i32(row_count.p, rawptr.p, n, scan_tmp.p, scan_tmp.n, st));
    int32_t nraw = 0;
    AVS_HIP(hipMemcpyAsync(&nraw, rawptr.p + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    const int64_t nwaves = (n + 63) >> 6;
    DevBuf<int32_t> &wslots = c->scratch.wave_slots;
    AVS_TRY(wslots.reserve((size_t)nwaves + 1));
    hipLaunchKernelGGL(k_wave_slots, dim3(grid_for(nwaves + 1)), dim3(kBlock), 0, st, n, (const int32_t *)row_count.p, wslots.p);
    AVS_TRY(exclusive_scan_i32(wslots.p, rawptr.p, nwaves, scan_tmp.p, scan_tmp.n, st)); // rawptr now: first slot of every wave
    int32_t nslots = 0;
    AVS_HIP(hipMemcpyAsync(&nslots, rawptr.p + nwaves, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    tr.mark("dry run, scans");
    AVS_REQUIRE(nraw >= 0 && nslots >= 0, AVS_EINVAL, "raw triplet count exceeds int32 (the scan reports -1 for any total above INT32_MAX)");
    if (nraw_out) *nraw_out = nraw;
    AVS_TRY(raw_col.reserve((size_t)nslots));
    AVS_TRY(raw_val.reserve((size_t)nslots));
    // K6 emit (wave-transposed raw triplets), K6b first-occurrence masks -> unique counts
    DevBuf<int32_t> &ucount = c->scratch.ucount;
    AVS_TRY(ucount.reserve((size_t)n + 1));
    if (n) {
        hipLaunchKernelGGL((k_rows<true>), dim3(grid_for(n)), dim3(kBlock), 0, st, P, c->vdof.p, n, E, C, c->x0.p,
                           (const int32_t *)rawptr.p, raw_col.p, raw_val.p, row_count.p, rhs.p, err.p, ids);
        DevBuf<int32_t> &long_rows = c->scratch.long_rows;
        AVS_TRY(long_rows.reserve((size_t)n + 1));
        AVS_HIP(hipMemsetAsync(long_rows.p, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(k_unique_rows, dim3(grid_for(n + 1)), dim3(kBlock), 0, st, n, (const int32_t *)rawptr.p, (int32_t *)raw_col.p,
                           (const int32_t *)row_count.p, ucount.p, long_rows.p);
        hipLaunchKernelGGL(k_unique_long, dim3(kLongGrid), dim3(kBlock), 0, st, (const int32_t *)rawptr.p, (int32_t *)raw_col.p, (const int32_t *)row_count.p,
                           ucount.p, (const int32_t *)long_rows.p);
    }
    AVS_TRY(exclusive_scan_i32(ucount.p, row_ptr.p, n, scan_tmp.p, scan_tmp.n, st));
    int32_t nnz = 0;
    AVS_HIP(hipMemcpyAsync(&nnz, row_ptr.p + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    int e = 0;
    AVS_TRY(read_err(err.p, st, &e));
    tr.mark("emit, unique, scan");
    AVS_REQUIRE(e == 0, AVS_EINTERNAL, "row assembly hit a reference assert (code %d): stencils and index pyramids disagree", e);
    AVS_REQUIRE(nnz >= 0, AVS_EINVAL, "non-zero count exceeds int32");
    if (nnz_out) *nnz_out = nnz;
    AVS_TRY(col.reserve((size_t)nnz));
    AVS_TRY(val.reserve((size_t)nnz));
    // K6b + K7: rank, fold, write the final CSR
    if (n && c->desc.precision == AVS_PRECISION_F32)
        hipLaunchKernelGGL(k_merge_rows<true>, dim3(grid_for(n)), dim3(kBlock), 0, st, n, (const int32_t *)rawptr.p, (const int32_t *)raw_col.p,
                           (const double *)raw_val.p, (const int32_t *)row_count.p, (const int32_t *)row_ptr.p, col.p, val.p);
    else if (n)
        hipLaunchKernelGGL(k_merge_rows<false>, dim3(grid_for(n)), dim3(kBlock), 0, st, n, (const int32_t *)rawptr.p, (const int32_t *)raw_col.p,
                           (const double *)raw_val.p, (const int32_t *)row_count.p, (const int32_t *)row_ptr.p, col.p, val.p);
    if (n && c->desc.precision == AVS_PRECISION_F32)
        hipLaunchKernelGGL(k_merge_long<true>, dim3(kLongGrid), dim3(kBlock), 0, st, (const int32_t *)rawptr.p, (const int32_t *)raw_col.p, (const double *)raw_val.p,
                           (const int32_t *)row_count.p, (const int32_t *)row_ptr.p, col.p, val.p, (const int32_t *)c->scratch.long_rows.p);
    else if (n)
        hipLaunchKernelGGL(k_merge_long<false>, dim3(kLongGrid), dim3(kBlock), 0, st, (const int32_t *)rawptr.p, (const int32_t *)raw_col.p, (const double *)raw_val.p,
                           (const int32_t *)row_count.p, (const int32_t *)row_ptr.p, col.p, val.p, (const int32_t *)c->scratch.long_rows.p);
    AVS_HIP(hipGetLastError());
    AVS_HIP(hipStreamSynchronize(st)); // the caller may read nnz-sized results right away; the raw buffers stay in the context
    tr.mark("merge");
    return AVS_OK;
}

// raw (pre-merge) triplet count of every row: the cost weight the multi-GPU slab cuts are balanced with
avs_status count_raw_rows(avs_ctx *c, DevBuf<int32_t> &counts)
{
    hipStream_t st = c->stream;
    AVS_REQUIRE(c->stencils_ready, AVS_ESTATE, "build the stencils first");
    const int64_t n = c->n_vel;
    DevBuf<int> &err = c->scratch.err;
    AVS_TRY(err.reserve(1));
    AVS_TRY(counts.reserve((size_t)n + 1));
    AVS_HIP(hipMemsetAsync(err.p, 0, sizeof(int), st));
    if (n) hipLaunchKernelGGL((k_rows<false>), dim3(grid_for(n)), dim3(kBlock), 0, st, c->view(), c->vdof.p, n, edge_view(c), center_view(c),
                              (const double *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr, counts.p,
                              (double *)nullptr, err.p, (const int32_t *)nullptr);
    int e = 0;
    AVS_TRY(read_err(err.p, st, &e));
    AVS_REQUIRE(e == 0, AVS_EINTERNAL, "row assembly hit a reference assert (code %d)", e);
    return AVS_OK;
}

avs_status build_system(avs_ctx *c)
{
    int64_t nnz = 0, nraw = 0;
    AVS_TRY(assemble_rows(c, nullptr, c->n_vel, c->row_ptr, c->col, c->val, c->rhs, &nnz, &nraw));
    c->nnz = nnz;
    c->nraw = nraw;
    c->system_ready = true;
    c->solved = false;
    return AVS_OK;
}

} // namespace avs
