// avs_post.hip -- post-solve transfer of the octree solution to the regular MAC grid, on the device
// (SURVEY.md 8(f) "next #2"; reference cpp:655-707 = setOctreeVelocity cpp:2779-2813,
// HDK_OctreeVectorFieldInterpolator interp.h:30-138 / interp.cpp:118-845, applyVelocitiesToRegularGrid
// cpp:2815-2894; "interp.cpp:" = Source/HDK_OctreeVectorFieldInterpolator.cpp).
//
//   T1 k_scatter_velocity   solution -> per-level fp32 face fields (one thread per DOF)
//   T2 k_nodes_sample       setActiveNodes + sampleActiveNodes fused (one thread per node)
//   T3 k_nodes_bubble       even nodes hand their sums to the co-located parent node (level by level)
//   T4 k_nodes_finish       T-junction / split-edge nodes: ghost faces from the coarse side
//   T5 k_nodes_normalize    value / weight
//   T6 k_nodes_distribute   dependent nodes copy the parent's value (top-down)
//   T7 k_apply_regular_tiled  per regular face: direct copy / solid velocity / interpSPGrid; untouched faces keep the input velocity
//
// Node values and weights are fp32 fields in the reference (SIM_RawField) with fpreal (double)
// arithmetic in between; the same conversions are made at the same places so results are bit-identical
// to the oracle.  Sample positions are lattice points: index-space arithmetic is exact.
#include <cmath>
#include <cstring>

#include "avs_device_common.hpp"

namespace avs {

static constexpr int kBlock = 256;

struct PostView {
    int levels;
    float *vel[AVS_MAX_LEVELS][3];  // octreeVelocity[level][axis]
    float *nval[AVS_MAX_LEVELS][3]; // myNodeValues
    float *nw[AVS_MAX_LEVELS][3];   // nodeWeights
    int32_t *nf[AVS_MAX_LEVELS];    // nodeFlags
    int8_t *nlab[AVS_MAX_LEVELS];   // myNodeLabels: 0 inactive, 1 active, 2 dependent
};

__device__ __forceinline__ I3 node_res(const PyramidView &P, int l)
{
    return I3{{(P.n[0] >> l) + 1, (P.n[1] >> l) + 1, (P.n[2] >> l) + 1}};
}
__device__ __forceinline__ I3 unlin(const I3 &r, size_t o)
{
    I3 p;
    p[0] = (int)(o % r[0]);
    const size_t q = o / r[0];
    p[1] = (int)(q % r[1]);
    p[2] = (int)(q / r[1]);
    return p;
}
// HDKnodeToFace, util.h:187-203
__device__ __forceinline__ I3 node_to_face(const I3 &n, int fa, int fi)
{
    I3 f = n;
    if (!(fi & 1)) --f[(fa + 1) % 3];
    if (!(fi & 2)) --f[(fa + 2) % 3];
    return f;
}
__device__ __forceinline__ I3 clamp3(const I3 &p, const I3 &r)
{
    return I3{{clampi(p[0], 0, r[0] - 1), clampi(p[1], 0, r[1] - 1), clampi(p[2], 0, r[2] - 1)}};
}
__device__ __forceinline__ int32_t vidx_clamped(const PyramidView &P, int l, int a, const I3 &f)
{
    const I3 r = face_res(P, l, a);
    return P.vidx[l][a][lin(r, clamp3(f, r))]; // the reference reads out of bounds next to the domain border
}
__device__ __forceinline__ float vel_clamped(const PyramidView &P, const PostView &W, int l, int a, const I3 &f)
{
    const I3 r = face_res(P, l, a);
    return W.vel[l][a][lin(r, clamp3(f, r))];
}

// T1 -----------------------------------------------------------------------------------------
// (`ids`, here and below: the DOFs of a slab-local context's window -- the only ones its dof table describes -- instead of 0 .. n)
__global__ __launch_bounds__(kBlock) void k_scatter_velocity(PyramidView P, PostView W, const int32_t *__restrict__ vdof, int64_t n,
                                                            const double *__restrict__ x, const int32_t *__restrict__ ids = nullptr)
{
    const int64_t slot = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (slot >= n) return;
    const int64_t id = ids ? ids[slot] : slot;
    const int4 rec = reinterpret_cast<const int4 *>(vdof)[id];
    const int level = rec.x & 0xff, axis = rec.x >> 8;
    W.vel[level][axis][lin(face_res(P, level, axis), I3{{rec.y, rec.z, rec.w}})] = (float)x[id]; // cpp:2808
}

// end of a transfer: the per-level face fields are all zero again (next transfer: no 4-B-per-face fill of every level)
__global__ __launch_bounds__(kBlock) void k_unscatter_velocity(PyramidView P, PostView W, const int32_t *__restrict__ vdof, int64_t n,
                                                              const int32_t *__restrict__ ids = nullptr)
{
    const int64_t slot = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (slot >= n) return;
    const int64_t id = ids ? ids[slot] : slot;
    const int4 rec = reinterpret_cast<const int4 *>(vdof)[id];
    const int level = rec.x & 0xff, axis = rec.x >> 8;
    W.vel[level][axis][lin(face_res(P, level, axis), I3{{rec.y, rec.z, rec.w}})] = 0.f;
}
// start of a transfer on node grids the previous transfer left behind: values are non-zero only where the label is (k_nodes_sample
// writes both for the nodes it activates, the later passes touch labelled nodes only) -- one 1-byte read per node instead of 13 B of fills
// (`list` != null: the nodes the previous transfer labelled, k_nodes_sample_dofs; else a sweep over the level's nodes)
__global__ __launch_bounds__(kBlock) void k_nodes_clear(PostView W, int l, size_t total, const uint32_t *__restrict__ list = nullptr)
{
    for (size_t it = (size_t)blockIdx.x * kBlock + threadIdx.x; it < total; it += (size_t)gridDim.x * kBlock) {
        const size_t o = list ? (size_t)list[it] : it;
        if (W.nlab[l][o] == 0) continue;
        W.nlab[l][o] = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) W.nval[l][a][o] = 0.f;
    }
}

// T2: interp.cpp:118-188 + 190-286 ------------------------------------------------------------
// Per-level lists of the nodes T2 labels: the later passes (bubble, finish, normalise, distribute, and the next transfer's clear) walk
// them instead of sweeping the node lattices.  cap: list capacity per level (an overflowing level keeps count > cap: the caller falls
// back to sweeps).  One atomic per wave (a counter hit by every labelled node serialises: 18 M same-address atomics took 65 ms).
struct NodeLists {
    uint32_t *list[AVS_MAX_LEVELS];
    unsigned *count; // [levels]
    unsigned cap[AVS_MAX_LEVELS];
};
__device__ __forceinline__ void node_list_append(const NodeLists &NL, int l, size_t o) // called by the lanes that labelled a node of level l (l: wave-uniform)
{
    const unsigned long long m = __ballot(1);
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned base = 0u;
    if (lane == leader) base = atomicAdd(NL.count + l, (unsigned)__popcll(m));
    base = __shfl(base, leader, 64);
    const unsigned at = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    if (at < NL.cap[l]) NL.list[l][at] = (uint32_t)o;
}

// One node: the twelve faces around it, all indices requested at once (the reference's loops stop at the first SOLIDBOUNDARY / OUTSIDE
// face, but "active and not inactive" does not depend on the order of the tests).  `my_slot` >= 0 (k_nodes_sample_dofs): the caller came
// from the DOF face in slot fa * 4 + fi and goes on only when that is the FIRST DOF face of the node -- so that exactly one of the up to
// twelve threads that reach a node labels it.  Returns true when the node was labelled (active).
__device__ __forceinline__ bool sample_node(const PyramidView &P, const PostView &W, int l, const I3 &node, size_t o, double weight, int my_slot)
{
    bool active = false, inactive = false;
    int32_t vi[12];
    size_t fo[12];
    int first = -1;
#pragma unroll
    for (int fa = 0; fa < 3; ++fa) {
        const I3 fr = face_res(P, l, fa);
        const int b1 = (fa + 1) % 3, b2 = (fa + 2) % 3;
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            const I3 f = node_to_face(node, fa, fi);
            const bool oob = f[b1] < 0 || f[b2] < 0 || f[b1] >= fr[b1] || f[b2] >= fr[b2];
            inactive |= oob;
            fo[fa * 4 + fi] = lin(fr, clamp3(f, fr));
            vi[fa * 4 + fi] = P.vidx[l][fa][fo[fa * 4 + fi]];
            if (!oob) {
                if (vi[fa * 4 + fi] >= 0 && first < 0) first = fa * 4 + fi;
                active |= vi[fa * 4 + fi] >= 0;
                inactive |= vi[fa * 4 + fi] == AVS_SOLIDBOUNDARY || vi[fa * 4 + fi] == AVS_OUTSIDE;
            }
        }
    }
    if (my_slot >= 0 && first != my_slot) return false; // another thread's node
    if (!(active && !inactive)) return false; // labels / values / weights / flags were zero-filled
    W.nlab[l][o] = 1;
    int32_t flag = 0;
#pragma unroll
    for (int fa = 0; fa < 3; ++fa) {
        double av = 0., aw = 0.;
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) { // active nodes have all 12 faces in bounds
            const int32_t v = vi[fa * 4 + fi];
            if (v >= 0) {
                av += weight * (double)W.vel[l][fa][fo[fa * 4 + fi]];
                aw += weight;
                flag += 1 << (fa * 4 + fi);
            } else if (v != AVS_UNASSIGNED) {
                aw += weight;
                flag += 1 << (fa * 4 + fi);
            }
        }
        W.nval[l][fa][o] = (float)av;
        W.nw[l][fa][o] = (float)aw;
    }
    W.nf[l][o] = flag;
    return true;
}

// (`NL.count` != null: the labelled nodes are listed)
__global__ __launch_bounds__(kBlock) void k_nodes_sample(PyramidView P, PostView W, int l, NodeLists NL)
{
    const I3 nr = node_res(P, l);
    const size_t total = (size_t)nr[0] * nr[1] * nr[2];
    const double weight = (double)(1 << (P.levels - l - 1));
    for (size_t o = (size_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += (size_t)gridDim.x * kBlock) {
        const I3 node = unlin(nr, o);
        { // A face DOF has an ACTIVE cell of this level on one side (cpp:1232-1319), and the twelve faces around a node separate
          // the eight cells around it: no ACTIVE cell among them => no DOF face => the node stays inactive.  Eight 1-byte
          // reads instead of twelve 4-byte ones for the bulk of the lattice (air, and the inside of coarser cells).
            const I3 cr = cell_res(P, l);
            bool any_active = false;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                const I3 c{{node[0] - 1 + (ci & 1), node[1] - 1 + ((ci >> 1) & 1), node[2] - 1 + ((ci >> 2) & 1)}};
                const bool in = c[0] >= 0 && c[1] >= 0 && c[2] >= 0 && c[0] < cr[0] && c[1] < cr[1] && c[2] < cr[2];
                const int8_t lb = P.labels[l][lin(cr, clamp3(c, cr))];
                any_active |= in && lb == AVS_ACTIVE;
            }
            if (!any_active) continue; // labels / values / weights / flags were zero-filled
        }
        if (sample_node(P, W, l, node, o, weight, -1) && NL.count) node_list_append(NL, l, o);
    }
}

// The same pass driven by the velocity DOFs (round 5), for VERY sparse scenes: a node can only be active when one of its twelve faces is
// a DOF, i.e. when it is a corner of a DOF face of its level -- so one thread per (DOF, corner) reaches every candidate, instead of one
// thread per node of every level.  Of the threads that reach a node the one that came from its first DOF face labels it (no atomics on
// the data path).  Same arithmetic per node, nodes are independent: same node grids, bit for bit.  A visit costs twelve index gathers, a
// node the sweep rejects eight 1-byte label reads: measured, the two are equal at 18 nodes per DOF (512^3 beam: transfer 7.6 ms either
// way) and the DOF-driven pass wins at 58 (1024^3 sheet with 18 M DOFs: 21.4 against 22.9 ms, 12.0 against 13.5 in place): it is
// taken from 32 nodes per DOF on (AVS_POST_DOF_SAMPLE=0 / 1 forces either).
__global__ __launch_bounds__(kBlock) void k_nodes_sample_dofs(PyramidView P, PostView W, const int32_t *__restrict__ vdof, int64_t n, NodeLists NL,
                                                             const int32_t *__restrict__ ids = nullptr)
{
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool won = false;
    int l = 0;
    size_t o = 0;
    if (t < 4 * n) {
        const int4 rec = reinterpret_cast<const int4 *>(vdof)[ids ? (int64_t)ids[t >> 2] : (t >> 2)];
        const int c = (int)(t & 3);
        const int fa = rec.x >> 8;
        l = rec.x & 0xff;
        const int b1 = (fa + 1) % 3, b2 = (fa + 2) % 3;
        I3 node{{rec.y, rec.z, rec.w}};
        node[b1] += c & 1;
        node[b2] += c >> 1;
        o = lin(node_res(P, l), node);
        const double weight = (double)(1 << (P.levels - l - 1));
        won = sample_node(P, W, l, node, o, weight, fa * 4 + (3 - c)); // (node_to_face: slot fi of the node is the face at corner 3 - fi)
    }
    for (int q = 0; q < P.levels; ++q) // (a wave's lanes may belong to different levels: one round per level present)
        if (won && l == q) node_list_append(NL, q, o);
}

// T3: interp.cpp:288-355; one thread per node of level l+1 (its co-located child is node 2*p) ------
// (`list` / `count`: the labelled nodes of level l + 1 instead of a sweep -- here and in the passes below)
__global__ __launch_bounds__(kBlock) void k_nodes_bubble(PyramidView P, PostView W, int l, const uint32_t *__restrict__ list = nullptr, size_t count = 0)
{
    const I3 nr = node_res(P, l), pr = node_res(P, l + 1);
    const size_t total = list ? count : (size_t)pr[0] * pr[1] * pr[2];
    for (size_t it = (size_t)blockIdx.x * kBlock + threadIdx.x; it < total; it += (size_t)gridDim.x * kBlock) {
        const size_t po = list ? (size_t)list[it] : it;
        if (W.nlab[l + 1][po] != 1) continue;
        const I3 par = unlin(pr, po);
        const I3 node{{2 * par[0], 2 * par[1], 2 * par[2]}};
        const size_t no = lin(nr, node);
        if (W.nlab[l][no] != 1) continue;
        W.nf[l + 1][po] = W.nf[l][no] + W.nf[l + 1][po];
        for (int a = 0; a < 3; ++a) {
            W.nw[l + 1][a][po] = (float)((double)W.nw[l][a][no] + (double)W.nw[l + 1][a][po]);
            W.nval[l + 1][a][po] = (float)((double)W.nval[l][a][no] + (double)W.nval[l + 1][a][po]);
        }
        W.nlab[l][no] = 2; // DEPENDENTNODE
    }
}

// T4: interp.cpp:357-567 ------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_nodes_finish(PyramidView P, PostView W, int l, const uint32_t *__restrict__ list = nullptr, size_t count = 0)
{
    const I3 nr = node_res(P, l);
    const size_t total = list ? count : (size_t)nr[0] * nr[1] * nr[2];
    const int L = P.levels;
    const double weight = (double)(1 << (L - l - 1));
    for (size_t it = (size_t)blockIdx.x * kBlock + threadIdx.x; it < total; it += (size_t)gridDim.x * kBlock) {
        const size_t no = list ? (size_t)list[it] : it;
        if (W.nlab[l][no] != 1) continue;
        int32_t flag = W.nf[l][no];
        if (flag == 0xFFF) continue;
        const I3 node = unlin(nr, no);
        int32_t temp = flag;
        for (int bit = 0; flag != 0xFFF && bit < 12; ++bit, temp >>= 1) {
            if (temp & 1) continue;
            const int fa = bit / 4, fi = bit % 4;
            const I3 face = node_to_face(node, fa, fi);
            bool found = false;
            if (node[fa] % 2 == 0) { // a face one level up may sit exactly here, interp.cpp:441-467
                const I3 pf = half3(face);
                const size_t pfo = lin(face_res(P, l + 1, fa), pf);
                if (P.vidx[l + 1][fa][pfo] >= 0) {
                    const double ghost = (double)W.vel[l + 1][fa][pfo];
                    double v = (double)W.nval[l][fa][no];
                    v += weight * ghost;
                    W.nval[l][fa][no] = (float)v;
                    double ww = (double)W.nw[l][fa][no];
                    ww += weight;
                    W.nw[l][fa][no] = (float)ww;
                    flag += 1 << bit;
                    found = true;
                }
            }
            if (!found) { // interpolate a ghost face inside the active coarse cell, interp.cpp:469-552
                I3 cell = face; // HDKfaceToCell(face, faceAxis, 1)
                int sl = l;
                for (;;) {
                    const I3 cr = cell_res(P, sl);
                    if (P.labels[sl][lin(cr, clamp3(cell, cr))] == AVS_ACTIVE || sl + 1 >= L) break;
                    cell = half3(cell);
                    ++sl;
                }
                const double ip = (double)((long long)face[fa] << l) / (double)(1 << sl);
                const double iw = ip - floor(ip);
                double ghost = 0.;
                for (int dir = 0; dir < 2; ++dir) {
                    I3 of = cell;
                    if (dir == 1) ++of[fa];
                    const double lw = dir == 0 ? 1. - iw : iw;
                    const int32_t oi = vidx_clamped(P, sl, fa, of);
                    if (oi >= 0) ghost += lw * (double)vel_clamped(P, W, sl, fa, of);
                    else if (oi == AVS_UNASSIGNED && sl > 0) {
                        for (int ci = 0; ci < 4; ++ci) {
                            const I3 cf = child_face(of, fa, ci);
                            if (vidx_clamped(P, sl - 1, fa, cf) >= 0) ghost += .25 * lw * (double)vel_clamped(P, W, sl - 1, fa, cf);
                        }
                    }
                }
                double v = (double)W.nval[l][fa][no];
                v += weight * ghost;
                W.nval[l][fa][no] = (float)v;
                double ww = (double)W.nw[l][fa][no];
                ww += weight;
                W.nw[l][fa][no] = (float)ww;
                flag += 1 << bit;
            }
        }
        W.nf[l][no] = flag;
    }
}

// T5: interp.cpp:569-613 ------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_nodes_normalize(PyramidView P, PostView W, int l, const uint32_t *__restrict__ list = nullptr, size_t count = 0)
{
    const I3 nr = node_res(P, l);
    const size_t total = list ? count : (size_t)nr[0] * nr[1] * nr[2];
    for (size_t it = (size_t)blockIdx.x * kBlock + threadIdx.x; it < total; it += (size_t)gridDim.x * kBlock) {
        const size_t o = list ? (size_t)list[it] : it;
        if (W.nlab[l][o] != 1) continue;
        for (int a = 0; a < 3; ++a) W.nval[l][a][o] = (float)((double)W.nval[l][a][o] / (double)W.nw[l][a][o]);
    }
}

// T6: interp.cpp:615-658 ------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_nodes_distribute(PyramidView P, PostView W, int l, const uint32_t *__restrict__ list = nullptr, size_t count = 0)
{
    const I3 nr = node_res(P, l), pr = node_res(P, l + 1);
    const size_t total = list ? count : (size_t)nr[0] * nr[1] * nr[2];
    for (size_t it = (size_t)blockIdx.x * kBlock + threadIdx.x; it < total; it += (size_t)gridDim.x * kBlock) {
        const size_t o = list ? (size_t)list[it] : it;
        if (W.nlab[l][o] != 2) continue;
        const size_t po = lin(pr, half3(unlin(nr, o)));
        for (int a = 0; a < 3; ++a) W.nval[l][a][o] = W.nval[l + 1][a][po];
        W.nlab[l][o] = 1;
    }
}

// interpSPGrid, interp.cpp:660-845; P2 = sample position in half fine cells -----------------------
__device__ double interp_sp_grid(const PyramidView &P, const PostView &W, const I3 &P2, int axis)
{
    const int L = P.levels;
    // The reference walks up from level 0 until it meets an ACTIVE cell: up to `levels` DEPENDENT reads (all of them for a
    // face deep inside a coarse cell -- most faces that come here).  All candidate labels are requested at once instead and
    // the lowest ACTIVE level is picked: same answer, one memory round trip.
    int found = -1;
    {
        I3 c{{P2[0] >> 1, P2[1] >> 1, P2[2] >> 1}}; // floor(indexPoint) on the level-0 node lattice
        int8_t lab[AVS_MAX_LEVELS];
#pragma unroll
        for (int level = 0; level < AVS_MAX_LEVELS; ++level) {
            lab[level] = 0;
            if (level < L) {
                const I3 cr = cell_res(P, level);
                lab[level] = P.labels[level][lin(cr, clamp3(c, cr))];
                c = half3(c);
            }
        }
#pragma unroll
        for (int level = AVS_MAX_LEVELS - 1; level >= 0; --level)
            if (level < L && lab[level] == AVS_ACTIVE) found = level;
    }
    if (found < 0) return 0.; // reference: assert(false)
    I3 cell{{P2[0] >> 1, P2[1] >> 1, P2[2] >> 1}};
    for (int l = 0; l < found; ++l) cell = half3(cell);
    {
        const int level = found;
        {
            const double scale = (double)(1 << (level + 1)); // half fine cells per cell of this level
            double ifp[3];
            I3 face;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                ifp[a] = (double)P2[a] / scale - (a == axis ? 0. : .5); // posToIndex on the face lattice
                face[a] = (int)floor(ifp[a]);
            }
            // HDKcellToNode(face, fi), interp.cpp:683-698: the eight faces around the sample.  Their indices (transition test) and
            // their velocities (trilinear branch) are requested together, no early exit: one round trip instead of up to nine
            bool transition = false;
            float fv[8];
#pragma unroll
            for (int fi = 0; fi < 8; ++fi) {
                const I3 nf{{face[0] + (fi & 1), face[1] + ((fi >> 1) & 1), face[2] + ((fi >> 2) & 1)}};
                transition |= vidx_clamped(P, level, axis, nf) == AVS_UNASSIGNED;
                fv[fi] = vel_clamped(P, W, level, axis, nf);
            }
            if (!transition) { // trilinear over the 8 faces, interp.cpp:700-728
                double iw[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    iw[a] = ifp[a] - (double)face[a];
                    iw[a] = iw[a] < 0. ? 0. : (iw[a] > 1. ? 1. : iw[a]);
                }
                double v = 0.;
#pragma unroll
                for (int fi = 0; fi < 8; ++fi) {
                    const I3 nf{{face[0] + (fi & 1), face[1] + ((fi >> 1) & 1), face[2] + ((fi >> 2) & 1)}};
                    double wt = 1.;
#pragma unroll
                    for (int a = 0; a < 3; ++a) wt *= (nf[a] - face[a] == 0) ? (1. - iw[a]) : iw[a];
                    v += wt * (double)fv[fi];
                }
                return v;
            }
            // node-based interpolation with the bubble correction, interp.cpp:730-836
            double ciw = (double)P2[axis] / scale - (double)cell[axis];
            ciw = ciw < 0. ? 0. : (ciw > 1. ? 1. : ciw);
            const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
            double fiv[2] = {0., 0.};
            for (int dir = 0; dir < 2; ++dir) {
                I3 af = cell;
                if (dir == 1) ++af[axis];
                int fl = level;
                if (vidx_clamped(P, level, axis, af) == AVS_UNASSIGNED && level > 0) { // project onto the child face
                    const double cs = (double)(1 << level);
                    const double cip1 = (double)P2[a1] / cs, cip2 = (double)P2[a2] / cs;
                    for (int ci = 0; ci < 4; ++ci) {
                        const I3 cf = child_face(af, axis, ci);
                        if ((double)cf[a1] <= cip1 && (double)cf[a2] <= cip2 && (double)(cf[a1] + 1) >= cip1 && (double)(cf[a2] + 1) >= cip2) {
                            fl = level - 1;
                            af = cf;
                            break;
                        }
                    }
                }
                const double ns = (double)(1 << (fl + 1));
                const double inp1 = (double)P2[a1] / ns, inp2 = (double)P2[a2] / ns;
                const double fw0 = inp1 - floor(inp1), fw1 = inp2 - floor(inp2);
                const double fvel = (double)vel_clamped(P, W, fl, axis, af);
                const I3 nr = node_res(P, fl);
                double avg = 0.;
                for (int ni = 0; ni < 4; ++ni) { // HDKfaceToNode, util.h:133-149
                    I3 nd = af;
                    if (ni & 1) ++nd[a1];
                    if (ni & 2) ++nd[a2];
                    double wt = 1.;
                    wt *= (nd[a1] - af[a1] == 0) ? (1. - fw0) : fw0;
                    wt *= (nd[a2] - af[a2] == 0) ? (1. - fw1) : fw1;
                    const double nv = (double)W.nval[fl][axis][lin(nr, clamp3(nd, nr))];
                    avg += nv;
                    fiv[dir] += nv * wt;
                }
                const double m3 = 1. - fw0, m4 = 1. - fw1;
                double mm = m3 < m4 ? m3 : m4;
                mm = fw1 < mm ? fw1 : mm;
                mm = fw0 < mm ? fw0 : mm;
                fiv[dir] += 2. * (fvel - .25 * avg) * mm;
            }
            return (1. - ciw) * fiv[0] + ciw * fiv[1];
        }
    }
}

// T7: cpp:2815-2894 -----------------------------------------------------------------------------
// T7, tiled (round 4): the regular-grid faces a solve touches (regular DOF or solid boundary) lie in a band around the liquid; the
// lattice is cut into tiles of 64 x 8 x 8 faces (256-B rows) and a tile without any such face is a plain copy of the input velocity.
// One launch per axis does what the copy of the input field + k_apply_regular did (one pass over the lattice instead of two and a half;
// 1024^3 thin sheet: 7.0 -> ~2 ms per axis), bit for bit.
constexpr int kRtX = 64, kRtY = 8, kRtZ = 8;
struct RTileGrid {
    int t[3];
    __host__ __device__ size_t vol() const { return (size_t)t[0] * t[1] * t[2]; }
};
static inline RTileGrid rtile_grid(const int fr[3]) { return RTileGrid{{(fr[0] + kRtX - 1) / kRtX, (fr[1] + kRtY - 1) / kRtY, (fr[2] + kRtZ - 1) / kRtZ}}; }

// flags[tile] = 1 when the tile holds a face the transfer writes; COPY: dst = src on the way (the lattice is being stored anyway)
// occ16 (optional; avs_prepass_apply): the 16^3-tile occupancy the lattice was classified with -- outside those tiles it is AVS_UNASSIGNED,
// so a 64 x 8 x 8 tile whose four 16^3 tiles are unoccupied is flagged 0 without being read (occ16_t: tiles per axis of that grid)
template <bool COPY>
__global__ __launch_bounds__(kBlock) void k_ridx_tile_flags(const int32_t *__restrict__ src, int32_t *__restrict__ dst, I3 fr, RTileGrid tg,
                                                            uint8_t *__restrict__ flags, const uint8_t *__restrict__ occ16 = nullptr, I3 occ16_t = I3{{0, 0, 0}})
{
    const int t = blockIdx.x;
    const int tx = t % tg.t[0], ty = (t / tg.t[0]) % tg.t[1], tz = t / (tg.t[0] * tg.t[1]);
    if (!COPY && occ16) {
        static_assert(kRtX % 16 == 0 && 16 % kRtY == 0 && 16 % kRtZ == 0, "a regular tile lies inside one row of 16^3 tiles");
        const int oy = ty * kRtY / 16, oz = tz * kRtZ / 16;
        bool any16 = false;
        for (int u = 0; u < kRtX / 16; ++u) {
            const int ox = tx * (kRtX / 16) + u;
            if (ox < occ16_t[0] && oy < occ16_t[1] && oz < occ16_t[2]) any16 |= occ16[ox + (size_t)occ16_t[0] * (oy + (size_t)occ16_t[1] * oz)] != 0;
        }
        if (!any16) {
            if (threadIdx.x == 0) flags[t] = 0;
            return;
        }
    }
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = tx * kRtX + lx;
    int any = 0;
    if (x < fr[0])
        for (int k = 0; k < kRtZ; ++k)
            for (int j = ly; j < kRtY; j += kBlock / 64) {
                const int y = ty * kRtY + j, z = tz * kRtZ + k;
                if (y >= fr[1] || z >= fr[2]) continue;
                const size_t o = ((size_t)z * fr[1] + y) * fr[0] + x;
                const int32_t ri = src[o];
                if (COPY) dst[o] = ri;
                any |= (ri >= 0 || ri == AVS_SOLIDBOUNDARY) ? 1 : 0;
            }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) flags[t] = any ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void k_apply_regular_tiled(PyramidView P, PostView W, int axis, const int32_t *__restrict__ ridx,
                                                                const uint8_t *__restrict__ flags, RTileGrid tg, const double *__restrict__ x,
                                                                const float *__restrict__ vel_in, float vel_const, float *__restrict__ out, int in_place,
                                                                int s_axis = -1, int s_lo = 0, int s_hi = 0, int ntx_launch = 0)
{
    // s_axis >= 0 (slab-local context): only the faces whose coordinate along s_axis lies in [s_lo, s_hi) are this rank's to write.
    // ntx_launch > 0 (a slab along x, in-place form): the workgroups' 64-face rows START at s_lo -- with tiles on the lattice's own
    // 64-face grid a 72-face slab touched three tile columns and two of them ran their waves for four live lanes (2.25 of a 3.1-ms transfer)
    const I3 fr = face_res(P, 0, axis);
    const int t = blockIdx.x;
    const int ntx = ntx_launch > 0 ? ntx_launch : tg.t[0];
    const int tx = t % ntx, ty = (t / ntx) % tg.t[1], tz = t / (ntx * tg.t[1]);
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int xx = (ntx_launch > 0 ? s_lo : 0) + tx * kRtX + lx;
    if (xx >= fr[0]) return;
    bool occupied = flags[(size_t)(xx / kRtX) + (size_t)tg.t[0] * ((size_t)ty + (size_t)tg.t[1] * (size_t)tz)] != 0;
    if (s_axis == 0 && (xx < s_lo || xx >= s_hi)) occupied = false;
    const bool same = in_place != 0; // in-place update of the caller's field (what the reference does to `vel`): untouched faces need no store
    for (int k = 0; k < kRtZ; ++k)
        for (int j = ly; j < kRtY; j += kBlock / 64) {
            const int y = ty * kRtY + j, z = tz * kRtZ + k;
            if (y >= fr[1] || z >= fr[2]) continue;
            const size_t o = ((size_t)z * fr[1] + y) * fr[0] + xx;
            if (!occupied || (s_axis == 1 && (y < s_lo || y >= s_hi)) || (s_axis == 2 && (z < s_lo || z >= s_hi))) {
                if (!same) out[o] = vel_in ? vel_in[o] : vel_const;
                continue;
            }
            const int32_t ri = ridx[o];
            if (same && !(ri >= 0 || ri == AVS_SOLIDBOUNDARY)) continue; // (the caller's array already holds the input velocity)
            const int32_t oi = P.vidx[0][axis][o]; // (requested with ridx: both lattices are the level-0 face lattice)
            float v = vel_in ? vel_in[o] : vel_const;
            if (ri >= 0) {
                if (oi >= 0) v = (float)x[oi];
                else if (oi == AVS_SOLIDBOUNDARY) v = sample_f32(P.solidvel[axis], fr, off_face(axis), pos2_face(0, axis, unlin(fr, o)));
                else if (oi == AVS_UNASSIGNED) v = (float)interp_sp_grid(P, W, pos2_face(0, axis, unlin(fr, o)), axis);
            } else if (ri == AVS_SOLIDBOUNDARY)
                v = sample_f32(P.solidvel[axis], fr, off_face(axis), pos2_face(0, axis, unlin(fr, o)));
            out[o] = v;
        }
}

static inline unsigned grid_for(size_t n, unsigned cap = 1u << 20)
{
    size_t b = (n + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

} // namespace avs

using namespace avs;

extern "C" {

// regularVelocityIndices[axis], cpp:303-329: >= 0 regular DOF, AVS_SOLIDBOUNDARY, else untouched
avs_status avs_set_regular_index_field(avs_ctx *c, int32_t axis, const int32_t *idx, avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    return avs::set_regular_index_lattice(c, axis, idx, where, false);
}

} // extern "C"

avs_status avs::set_regular_index_lattice(avs_ctx *c, int32_t axis, const int32_t *idx, avs_memspace where, bool padded_lattice)
{
    AVS_REQUIRE(c && idx, AVS_EINVAL, "null argument");
    AVS_REQUIRE(axis >= 0 && axis < 3, AVS_EINVAL, "axis out of range");
    AVS_HIP(hipSetDevice(c->desc.device));
    int r[3] = {c->desc.nx, c->desc.ny, c->desc.nz};
    int s3[3] = {c->desc.field_nx, c->desc.field_ny, c->desc.field_nz};
    r[axis] += 1;
    s3[axis] += 1;
    const size_t n = (size_t)r[0] * r[1] * r[2], ns = (size_t)s3[0] * s3[1] * s3[2];
    AVS_TRY(c->ridx[axis].alloc(n));
    const RTileGrid tg = rtile_grid(r);
    const I3 fr3{{r[0], r[1], r[2]}};
    AVS_TRY(c->ridx_tiles[axis].alloc(tg.vol()));
    bool flagged = false;
    if ((padded_lattice || n == ns) && where == AVS_MEM_DEVICE) { // stored and flagged in one pass
        hipLaunchKernelGGL(k_ridx_tile_flags<true>, dim3((unsigned)tg.vol()), dim3(kBlock), 0, c->stream, idx, c->ridx[axis].p, fr3, tg,
                           c->ridx_tiles[axis].p);
        AVS_HIP(hipGetLastError());
        flagged = true;
    } else if (padded_lattice || n == ns) {
        AVS_HIP(copy_in(c->ridx[axis].p, idx, n * sizeof(int32_t), where, c->stream));
        if (where == AVS_MEM_HOST) AVS_HIP(hipStreamSynchronize(c->stream));
    } else { // the regular grid is the simulation grid: faces of the padding are no regular DOFs (AVS_UNASSIGNED = untouched)
        DevBuf<int32_t> src;
        const int32_t *sp = idx;
        if (where == AVS_MEM_HOST) {
            AVS_TRY(src.alloc(ns));
            AVS_HIP(copy_in(src.p, idx, ns * sizeof(int32_t), where, c->stream));
            sp = src.p;
        }
        AVS_TRY(pad_lattice_i32(sp, s3[0], s3[1], s3[2], c->ridx[axis].p, r[0], r[1], r[2], AVS_UNASSIGNED, c->stream));
        AVS_HIP(hipStreamSynchronize(c->stream));
    }
    if (!flagged) {
        hipLaunchKernelGGL(k_ridx_tile_flags<false>, dim3((unsigned)tg.vol()), dim3(kBlock), 0, c->stream, (const int32_t *)c->ridx[axis].p,
                           (int32_t *)nullptr, fr3, tg, c->ridx_tiles[axis].p);
        AVS_HIP(hipGetLastError());
    }
    c->have_ridx[axis] = true;
    return AVS_OK;
}

// the pre-pass's regular-grid index lattice, by reference (avs_prepass_apply): only the tile flags are computed (one read)
avs_status avs::adopt_regular_index_lattice(avs_ctx *c, int32_t axis, std::shared_ptr<DevBuf<int32_t>> handle, const uint8_t *occ16, const int occ16_tiles[3])
{
    AVS_REQUIRE(c && handle && handle->p && axis >= 0 && axis < 3, AVS_EINVAL, "bad argument");
    int r[3] = {c->desc.nx, c->desc.ny, c->desc.nz};
    r[axis] += 1;
    const size_t n = (size_t)r[0] * r[1] * r[2];
    AVS_REQUIRE(handle->n == n, AVS_EINVAL, "lent regular-grid index lattice of axis %d does not match the context", axis);
    c->ridx[axis].adopt(std::move(handle));
    const RTileGrid tg = rtile_grid(r);
    const I3 fr3{{r[0], r[1], r[2]}};
    AVS_TRY(c->ridx_tiles[axis].alloc(tg.vol()));
    hipLaunchKernelGGL(k_ridx_tile_flags<false>, dim3((unsigned)tg.vol()), dim3(kBlock), 0, c->stream, (const int32_t *)c->ridx[axis].p,
                       (int32_t *)nullptr, fr3, tg, c->ridx_tiles[axis].p, occ16,
                       occ16 ? I3{{occ16_tiles[0], occ16_tiles[1], occ16_tiles[2]}} : I3{{0, 0, 0}});
    AVS_HIP(hipGetLastError());
    c->have_ridx[axis] = true;
    return AVS_OK;
}

extern "C" {

static avs_status transfer_impl(avs_ctx *c, float *out_x, float *out_y, float *out_z, avs_memspace where, bool in_place);
avs_status avs_transfer_to_regular_grid(avs_ctx *c, float *out_x, float *out_y, float *out_z, avs_memspace where)
{
    return transfer_impl(c, out_x, out_y, out_z, where, false);
}
avs_status avs_transfer_to_regular_grid_in_place(avs_ctx *c, float *vel_x, float *vel_y, float *vel_z)
{
    return transfer_impl(c, vel_x, vel_y, vel_z, AVS_MEM_DEVICE, true);
}
static avs_status transfer_impl(avs_ctx *c, float *out_x, float *out_y, float *out_z, avs_memspace where, bool in_place)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && out_x && out_y && out_z, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->solved, AVS_ESTATE, "no solution: call avs_solve first");
    AVS_REQUIRE(c->have_ridx[0] && c->have_ridx[1] && c->have_ridx[2], AVS_ESTATE, "regular-grid index fields missing (avs_set_regular_index_field)");
    AVS_REQUIRE(c->tables_ready, AVS_ESTATE, "dof tables missing");
    AVS_HIP(hipSetDevice(c->desc.device));
    Scope scope("Apply Octree Solution to Regular Grid"); // cpp:662
    hipStream_t st = c->stream;
    const int L = c->desc.levels;
    PyramidView P = c->view();
    PostView W{};
    W.levels = L;
    // allocate + zero the per-level fields (makeConstant(0) / INACTIVENODE, interp.h:65-81, cpp:688)
    const bool temporal = c->opt.prepass_temporal != 0;
    for (int l = 0; l < L; ++l) {
        const size_t nn = (size_t)((c->desc.nx >> l) + 1) * ((c->desc.ny >> l) + 1) * ((c->desc.nz >> l) + 1);
        if (c->post_nlab[l].n != nn) c->post_nodes_sparse[l] = nullptr; // (a new allocation may get the old address)
        AVS_TRY(c->post_nlab[l].alloc(nn));
        AVS_TRY(c->post_nf[l].alloc(nn));
        // (node flags and weights are only ever read for nodes that k_nodes_sample marked active, and it writes them: no fill)
        W.nlab[l] = c->post_nlab[l].p;
        W.nf[l] = c->post_nf[l].p;
        for (int a = 0; a < 3; ++a) {
            int fr[3] = {c->desc.nx >> l, c->desc.ny >> l, c->desc.nz >> l};
            fr[a] += 1;
            const size_t nf = (size_t)fr[0] * fr[1] * fr[2];
            if (c->post_vel[l][a].n != nf) c->post_vel_zero[l][a] = nullptr;
            if (c->post_nval[l][a].n != nn) c->post_nodes_sparse[l] = nullptr;
            AVS_TRY(c->post_vel[l][a].alloc(nf));
            AVS_TRY(c->post_nval[l][a].alloc(nn));
            AVS_TRY(c->post_nw[l][a].alloc(nn));
            if (!(temporal && c->post_vel_zero[l][a] == c->post_vel[l][a].p)) AVS_HIP(hipMemsetAsync(c->post_vel[l][a].p, 0, nf * sizeof(float), st));
            c->post_vel_zero[l][a] = nullptr; // (claimed again once this transfer has zeroed what it scatters)
            W.vel[l][a] = c->post_vel[l][a].p;
            W.nval[l][a] = c->post_nval[l][a].p;
            W.nw[l][a] = c->post_nw[l][a].p;
        }
        // node labels and values: the grids of the previous transfer are cleared where it labelled nodes; anything else is zero-filled
        const bool sparse = temporal && c->post_nodes_sparse[l] == c->post_nlab[l].p;
        c->post_nodes_sparse[l] = nullptr;
        if (sparse && c->post_lists_valid && c->post_list[l].p) // ... by the list of the nodes it labelled
            hipLaunchKernelGGL(k_nodes_clear, dim3(grid_for(c->post_list_n[l])), dim3(kBlock), 0, st, W, l, (size_t)c->post_list_n[l], (const uint32_t *)c->post_list[l].p);
        else if (sparse) hipLaunchKernelGGL(k_nodes_clear, dim3(grid_for(nn)), dim3(kBlock), 0, st, W, l, nn);
        else {
            AVS_HIP(hipMemsetAsync(c->post_nlab[l].p, 0, nn, st));
            for (int a = 0; a < 3; ++a) AVS_HIP(hipMemsetAsync(c->post_nval[l][a].p, 0, nn * sizeof(float), st));
        }
    }
    // Slab-local context (round 6): the DOFs of the rank's window are scattered and sampled, the faces of the rank's slab are written -- the
    // node values they read depend on faces <= 2 cells (of every level up to the top) away, well inside the window's 12-cell margin; the
    // solution is the whole vector (avs_dist_get_solution gathers it on every rank: n doubles, the one global-sized piece of the frame)
    const bool slab = c->slab.on;
    const int32_t *ids = slab ? (const int32_t *)c->wlist[0].p : nullptr;
    const int64_t n = slab ? c->n_window[0] : c->n_vel;
    if (slab) AVS_REQUIRE(ids && c->x.n >= (size_t)c->n_vel, AVS_ESTATE, "slab-local transfer: gather the solution first (avs_dist_get_solution)");
    int s_axis = -1, s_lo = 0, s_hi = 0;
    if (slab) {
        s_axis = c->slab.axis;
        s_lo = c->slab.cuts[c->slab.rank];
        s_hi = c->slab.rank == c->slab.world - 1 ? INT32_MAX : c->slab.cuts[c->slab.rank + 1]; // (the last rank takes the lattice's extra face)
    }
    if (n) hipLaunchKernelGGL(k_scatter_velocity, dim3(grid_for((size_t)n)), dim3(kBlock), 0, st, P, W, c->vdof.p, n, c->x.p, ids);
    auto nodes = [&](int l) { return (size_t)((c->desc.nx >> l) + 1) * ((c->desc.ny >> l) + 1) * ((c->desc.nz >> l) + 1); };
    // T2 lists the nodes it labels, per level, and the later passes walk the lists; sweeps over the node lattices remain for a context
    // without DOFs, node lattices beyond 2^32 entries, a list that overflows, or AVS_PREPASS_TEMPORAL=0
    bool lists = (temporal || slab) && n > 0 && nodes(0) < (1ull << 32);
    c->post_lists_valid = false;
    unsigned hcount[AVS_MAX_LEVELS] = {};
    if (lists) {
        NodeLists NL{};
        AVS_TRY(c->post_list_count.alloc(AVS_MAX_LEVELS));
        AVS_HIP(hipMemsetAsync(c->post_list_count.p, 0, AVS_MAX_LEVELS * sizeof(unsigned), st));
        NL.count = c->post_list_count.p;
        for (int l = 0; l < L; ++l) {
            const size_t cap = nodes(l) < (size_t)(2 * n + 1024) ? nodes(l) : (size_t)(2 * n + 1024); // (an active node is a corner of a DOF face)
            AVS_TRY(c->post_list[l].reserve(cap));
            NL.list[l] = c->post_list[l].p;
            NL.cap[l] = (unsigned)(c->post_list[l].n < (1ull << 32) - 1 ? c->post_list[l].n : (1ull << 32) - 1);
        }
        const int ds = c->opt.post_dof_sample;
        if (slab || ds > 0 || (ds < 0 && nodes(0) >= (size_t)32 * (size_t)n)) // sparse (and every slab-local context): from the DOFs
            hipLaunchKernelGGL(k_nodes_sample_dofs, dim3(grid_for((size_t)(4 * n), 1u << 30)), dim3(kBlock), 0, st, P, W, c->vdof.p, n, NL, ids);
        else
            for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_nodes_sample, dim3(grid_for(nodes(l))), dim3(kBlock), 0, st, P, W, l, NL);
        AVS_HIP(hipMemcpyAsync(hcount, c->post_list_count.p, sizeof(hcount), hipMemcpyDeviceToHost, st));
        AVS_HIP(hipStreamSynchronize(st));
        for (int l = 0; l < L; ++l) lists = lists && hcount[l] <= NL.cap[l];
        AVS_REQUIRE(lists || !slab, AVS_EINTERNAL, "slab-local transfer: a node list overflowed");
    } else {
        AVS_REQUIRE(!slab, AVS_ESTATE, "slab-local transfer: no DOF inside the window / node lattice too large for the list-driven passes");
        for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_nodes_sample, dim3(grid_for(nodes(l))), dim3(kBlock), 0, st, P, W, l, NodeLists{});
    }
    if (lists) {
        auto lp = [&](int l) { return (const uint32_t *)c->post_list[l].p; };
        for (int l = 0; l < L - 1; ++l) hipLaunchKernelGGL(k_nodes_bubble, dim3(grid_for(hcount[l + 1])), dim3(kBlock), 0, st, P, W, l, lp(l + 1), (size_t)hcount[l + 1]);
        for (int l = 0; l < L - 1; ++l) hipLaunchKernelGGL(k_nodes_finish, dim3(grid_for(hcount[l])), dim3(kBlock), 0, st, P, W, l, lp(l), (size_t)hcount[l]);
        for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_nodes_normalize, dim3(grid_for(hcount[l])), dim3(kBlock), 0, st, P, W, l, lp(l), (size_t)hcount[l]);
        for (int l = L - 2; l >= 0; --l) hipLaunchKernelGGL(k_nodes_distribute, dim3(grid_for(hcount[l])), dim3(kBlock), 0, st, P, W, l, lp(l), (size_t)hcount[l]);
    } else {
        for (int l = 0; l < L - 1; ++l) hipLaunchKernelGGL(k_nodes_bubble, dim3(grid_for(nodes(l + 1))), dim3(kBlock), 0, st, P, W, l, (const uint32_t *)nullptr, (size_t)0);
        for (int l = 0; l < L - 1; ++l) hipLaunchKernelGGL(k_nodes_finish, dim3(grid_for(nodes(l))), dim3(kBlock), 0, st, P, W, l, (const uint32_t *)nullptr, (size_t)0);
        for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_nodes_normalize, dim3(grid_for(nodes(l))), dim3(kBlock), 0, st, P, W, l, (const uint32_t *)nullptr, (size_t)0);
        for (int l = L - 2; l >= 0; --l) hipLaunchKernelGGL(k_nodes_distribute, dim3(grid_for(nodes(l))), dim3(kBlock), 0, st, P, W, l, (const uint32_t *)nullptr, (size_t)0);
    }
    AVS_HIP(hipGetLastError());
    float *outs[3] = {out_x, out_y, out_z};
    const bool padded = c->desc.field_nx != c->desc.nx || c->desc.field_ny != c->desc.ny || c->desc.field_nz != c->desc.nz;
    for (int a = 0; a < 3; ++a) {
        int fr[3] = {c->desc.nx, c->desc.ny, c->desc.nz};
        int sr[3] = {c->desc.field_nx, c->desc.field_ny, c->desc.field_nz};
        fr[a] += 1;
        sr[a] += 1;
        const size_t nf = (size_t)fr[0] * fr[1] * fr[2], ns = (size_t)sr[0] * sr[1] * sr[2];
        // a device destination on an unpadded grid is written in place; otherwise through a staging grid kept in the context
        float *work = outs[a];
        const bool inpl = in_place && !padded && !c->vel[a].is_const; // (a padded grid goes through the staging grid: every face is written)
        if (where == AVS_MEM_HOST || padded) {
            AVS_TRY(c->post_out[a].alloc(nf));
            work = c->post_out[a].p;
        }
        // the regular velocity field is updated in place in the reference: untouched faces keep the input velocity (copied tile by tile
        // in the same pass: k_apply_regular_tiled)
        const RTileGrid tg = rtile_grid(fr);
        int ntx_launch = 0;
        size_t blocks = tg.vol();
        if (slab && s_axis == 0 && inpl) { // (the in-place form writes nothing outside the slab: rows of 64 faces from the slab's first face)
            const int hi = s_hi < fr[0] ? s_hi : fr[0];
            ntx_launch = hi > s_lo ? (hi - s_lo + kRtX - 1) / kRtX : 1;
            blocks = (size_t)ntx_launch * tg.t[1] * tg.t[2];
        }
        hipLaunchKernelGGL(k_apply_regular_tiled, dim3((unsigned)blocks), dim3(kBlock), 0, st, P, W, a, (const int32_t *)c->ridx[a].p,
                           (const uint8_t *)c->ridx_tiles[a].p, tg, (const double *)c->x.p,
                           c->vel[a].is_const ? (const float *)nullptr : (const float *)c->vel[a].buf.p, (float)c->vel[a].cval, work, inpl ? 1 : 0,
                           s_axis, s_lo, s_hi, ntx_launch);
        AVS_HIP(hipGetLastError());
        if (padded) { // hand back the simulation grid's faces only
            if (where == AVS_MEM_DEVICE) AVS_TRY(crop_lattice_f32(work, fr[0], fr[1], fr[2], outs[a], sr[0], sr[1], sr[2], st));
            else {
                DevBuf<float> crop;
                AVS_TRY(crop.alloc(ns));
                AVS_TRY(crop_lattice_f32(work, fr[0], fr[1], fr[2], crop.p, sr[0], sr[1], sr[2], st));
                AVS_HIP(copy_out(outs[a], crop.p, ns * sizeof(float), where, st));
                AVS_HIP(hipStreamSynchronize(st));
            }
        } else if (where == AVS_MEM_HOST) AVS_HIP(copy_out(outs[a], work, nf * sizeof(float), where, st));
    }
    if (n) hipLaunchKernelGGL(k_unscatter_velocity, dim3(grid_for((size_t)n)), dim3(kBlock), 0, st, P, W, c->vdof.p, n, ids);
    AVS_HIP(hipGetLastError());
    AVS_HIP(hipStreamSynchronize(st));
    c->post_lists_valid = lists; // (the lists name every node whose label / values are non-zero now)
    for (int l = 0; l < L; ++l) { // what the staging grids hold now (see post_vel_zero / post_nodes_sparse)
        c->post_list_n[l] = hcount[l];
        c->post_nodes_sparse[l] = c->post_nlab[l].p;
        for (int a = 0; a < 3; ++a) c->post_vel_zero[l][a] = c->post_vel[l][a].p;
    }
    c->post_ready = true;
    return AVS_OK;
}

// node grids after all passes (parity tests): labels int8, values fp32, (n+1)^3 per level
avs_status avs_get_node_grid(avs_ctx *c, int32_t level, int8_t *labels, float *vx, float *vy, float *vz, avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->post_ready && level >= 0 && level < c->desc.levels, AVS_ESTATE, "call avs_transfer_to_regular_grid first");
    AVS_HIP(hipSetDevice(c->desc.device));
    const size_t nn = (size_t)((c->desc.nx >> level) + 1) * ((c->desc.ny >> level) + 1) * ((c->desc.nz >> level) + 1);
    if (labels) AVS_HIP(copy_out(labels, c->post_nlab[level].p, nn, where, c->stream));
    float *o[3] = {vx, vy, vz};
    for (int a = 0; a < 3; ++a)
        if (o[a]) AVS_HIP(copy_out(o[a], c->post_nval[level][a].p, nn * sizeof(float), where, c->stream));
    AVS_HIP(hipStreamSynchronize(c->stream));
    return AVS_OK;
}

} // extern "C"
