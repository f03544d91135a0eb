// avs_device_common.hpp -- lattice helpers shared by the assembly and pre-pass kernels (device code).
// Grid conventions: SURVEY.md App. A.1 (sample resolutions of SIM_RawField::init, x fastest).
#pragma once

#include "avs_internal.hpp"

namespace avs {

struct I3 {
    int v[3];
    __device__ __forceinline__ int &operator[](int a) { return v[a]; }
    __device__ __forceinline__ const int &operator[](int a) const { return v[a]; }
};

__device__ __forceinline__ I3 cell_res(const PyramidView &P, int l)
{
    return I3{{P.n[0] >> l, P.n[1] >> l, P.n[2] >> l}};
}
__device__ __forceinline__ I3 face_res(const PyramidView &P, int l, int a)
{
    I3 r = cell_res(P, l);
    r[a] += 1;
    return r;
}
__device__ __forceinline__ I3 edge_res(const PyramidView &P, int l, int a)
{
    I3 r = cell_res(P, l);
    r[0] += (a != 0);
    r[1] += (a != 1);
    r[2] += (a != 2);
    return r;
}
__device__ __forceinline__ size_t lin(const I3 &r, const I3 &p)
{
    return (size_t)p[0] + (size_t)r[0] * ((size_t)p[1] + (size_t)r[1] * (size_t)p[2]);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ float lerp32(float a, float b, float t)
{
    const float s = 1.0f - t;
    const float pa = a * s;
    const float pb = b * t;
    return pa + pb;
}

// SIM_RawField::getValue restated in exact index space: P2 = position in half fine cells,
// off2[a] = 1 where the lattice is cell-centred along a.  fp32, x then y then z.
__device__ float sample_f32(const FieldView &F, const I3 &r, const I3 &off2, const I3 &P2)
{
    if (F.is_const) return F.cval;
    int i0[3], i1[3];
    float t[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int q2 = P2[a] - off2[a];
        const int fl = q2 >> 1;
        t[a] = (q2 & 1) ? 0.5f : 0.0f;
        i0[a] = clampi(fl, 0, r[a] - 1);
        i1[a] = clampi(fl + 1, 0, r[a] - 1);
    }
    const float *d = F.data;
    const size_t sx = 1, sy = (size_t)r[0], sz = (size_t)r[0] * (size_t)r[1];
    const float c00 = lerp32(d[i0[0] * sx + i0[1] * sy + i0[2] * sz], d[i1[0] * sx + i0[1] * sy + i0[2] * sz], t[0]);
    const float c10 = lerp32(d[i0[0] * sx + i1[1] * sy + i0[2] * sz], d[i1[0] * sx + i1[1] * sy + i0[2] * sz], t[0]);
    const float c01 = lerp32(d[i0[0] * sx + i0[1] * sy + i1[2] * sz], d[i1[0] * sx + i0[1] * sy + i1[2] * sz], t[0]);
    const float c11 = lerp32(d[i0[0] * sx + i1[1] * sy + i1[2] * sz], d[i1[0] * sx + i1[1] * sy + i1[2] * sz], t[0]);
    const float c0 = lerp32(c00, c10, t[1]);
    const float c1 = lerp32(c01, c11, t[1]);
    return lerp32(c0, c1, t[2]);
}

__device__ __forceinline__ float field_at(const FieldView &F, const I3 &r, const I3 &p)
{
    return F.is_const ? F.cval : F.data[lin(r, p)];
}

__device__ __forceinline__ I3 pos2_center(int l, const I3 &p)
{
    return I3{{(2 * p[0] + 1) << l, (2 * p[1] + 1) << l, (2 * p[2] + 1) << l}};
}
__device__ __forceinline__ I3 pos2_face(int l, int axis, const I3 &p)
{
    I3 o;
#pragma unroll
    for (int a = 0; a < 3; ++a) o[a] = (a == axis) ? ((2 * p[a]) << l) : ((2 * p[a] + 1) << l);
    return o;
}
__device__ __forceinline__ I3 pos2_edge(int l, int axis, const I3 &p)
{
    I3 o;
#pragma unroll
    for (int a = 0; a < 3; ++a) o[a] = (a == axis) ? ((2 * p[a] + 1) << l) : ((2 * p[a]) << l);
    return o;
}
__device__ __forceinline__ I3 off_face(int a) { return I3{{a != 0, a != 1, a != 2}}; }

__device__ __forceinline__ int32_t vidx_at(const PyramidView &P, int l, int a, const I3 &f)
{
    return P.vidx[l][a][lin(face_res(P, l, a), f)];
}
__device__ __forceinline__ int32_t eidx_at(const PyramidView &P, int l, int a, const I3 &e)
{
    return P.eidx[l][a][lin(edge_res(P, l, a), e)];
}
__device__ __forceinline__ int label_at(const PyramidView &P, int l, const I3 &c)
{
    return P.labels[l][lin(cell_res(P, l), c)];
}

// oct.h:94-106 / 108-117 / 126-142
__device__ __forceinline__ I3 child_face(const I3 &f, int axis, int ci)
{
    I3 o{{2 * f[0], 2 * f[1], 2 * f[2]}};
    if (ci & 1) ++o[(axis + 1) % 3];
    if (ci & 2) ++o[(axis + 2) % 3];
    return o;
}
__device__ __forceinline__ I3 half3(const I3 &f) { return I3{{f[0] / 2, f[1] / 2, f[2] / 2}}; }


} // namespace avs
