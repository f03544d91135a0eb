// avs_prepass.hip -- the steps of solveGasSubclass BEFORE the hot path, on the device
// (SURVEY.md 8(f) "next #1 / #4"; reference: HDK_AdaptiveViscosity.cpp "cpp:", HDK_OctreeGrid.cpp "oct.cpp:").
//
//   P1 k_sdf_weights_far + k_sdf_weights   integration weights (cpp:712-766; HDK computeSDFWeightsSampled is not in the
//                        reference tree: fraction of n^3 sub-samples with interpolated SDF < 0, defined in
//                        oracle/avs_oracle.c and restated identically here, fp32, x then y then z); bricks whose SDF
//                        window has one sign are finished by the first kernel, the surface bricks by the second
//   P2 k_mask_labels     refinement mask + level-0 labels (cpp:815-867, oct.cpp:383-388); marks the tiles of the level-0 face lattices
//                        on the way (the SDF rule, cpp:907)
//   P3 k_oct_*           label pyramid, three passes per level (oct.cpp:93-189) + top level (oct.cpp:843-875)
//   P4 k_mark_tiles_all + k_tile_worklists + k_classify_batch  tile occupancy (all lattices of a level from one read of its labels; the
//                        level's "has an ACTIVE cell" flag too) + face / edge / centre / regular-face classification (cpp:886-1443) of the
//                        LISTED tiles -- occupied now, or visited by the allocation's last classification (reset) -- all lattices
//                        of a level in one launch
//   P5 scan + k_tile_select_ids + k_tile_ids   serial numbering in HDK 16^3 tile order (cpp:1566-1593, 1635-1660, 1688-1712): the per-tile
//                        DOF counts come out of the classification launch, exclusive scan over all tiles, ranks inside a tile by wave ballots
//   slab-local mode (avs_prepass_set_slab): every step on the rank's window only (Box3 sweeps, TileGrid launch boxes),
//                        the reference's GLOBAL ids through one all-reduce of the per-tile counts
//
// Every per-voxel rule is a gather from the finer / same level: one thread per output voxel (or per parent cell), deterministic (the
// atomics append to lists -- surface bricks, tiles -- whose order does not matter).  All integer outputs are bit-identical to the oracle
// and to the tensor-op restatement tests/prepass_torch.py (tests/test_gpu_prepass.py).
#include <cmath>
#include <climits>
#include <new>

#include "avs_device_common.hpp"

namespace avs {

static constexpr int kBlock = 256;
static constexpr int kTile = 16;
static_assert(kBlock == kTile * kTile, "tile kernels: one thread per (x, y) column of a 16^3 tile");
static_assert(AVS_UNASSIGNED == -1, "index lattices are pre-filled with memset(0xFF)");
static constexpr int kMaxSuper = 8;

struct SubConsts { // per axis: integer cell offset and fp32 fraction of every sub-sample
    int di[3][kMaxSuper];
    float fr[3][kMaxSuper];
    int n;
};

struct Grid3 {
    int r[3];
    __host__ __device__ size_t vol() const { return (size_t)r[0] * r[1] * r[2]; }
};

__device__ __forceinline__ size_t lin3(const Grid3 &g, int i, int j, int k)
{
    return (size_t)i + (size_t)g.r[0] * ((size_t)j + (size_t)g.r[1] * (size_t)k);
}

// Sub-box of a lattice (round 6, slab-local pre-pass): the dense kernels sweep `box` instead of the whole lattice -- a rank of a
// partitioned solve fills its window only.  The full lattice as a box gives the old sweep, index for index.
struct Box3 {
    int lo[3], n[3];
    __host__ __device__ size_t vol() const { return (n[0] > 0 && n[1] > 0 && n[2] > 0) ? (size_t)n[0] * n[1] * n[2] : 0; }
};
__device__ __forceinline__ void box_coords(const Box3 &b, size_t t, int &i, int &j, int &k)
{
    i = b.lo[0] + (int)(t % b.n[0]);
    const size_t q = t / b.n[0];
    j = b.lo[1] + (int)(q % b.n[1]);
    k = b.lo[2] + (int)(q / b.n[1]);
}

static inline unsigned grid_for(size_t n, unsigned cap = 1u << 20)
{
    size_t b = (n + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

// ---------------------------------------------------------------------------------------------
// P1: weights
// ---------------------------------------------------------------------------------------------
// One 512-thread workgroup = one 32x4x4 brick of sample indices, for ALL SEVEN weight fields at once (centre, 3 edge,
// 3 face lattices differ only in which axes are cell-centred).  The SDF cells the brick can touch (di = -2 .. +1 around
// every sample => a 35x7x7 window, clamped at the border) are staged in LDS once; the sign shortcuts and the 27 trilinear
// sub-samples of every field then read LDS only.  The brick is long in x: a wave stores two full 128-B runs per field
// (an 8^3 brick wrote 32-B pieces, and the kernel -- mostly bricks far from the surface, which only store -- ran at 1 TB/s).
static constexpr int kWBX = 32, kWBY = 4, kWBZ = 4;                    // brick extents
static constexpr int kWHX = kWBX + 3, kWHY = kWBY + 3, kWHZ = kWBZ + 3; // staged window: offsets -2 .. extent
static constexpr int kWThreads = kWBX * kWBY * kWBZ;
static constexpr int kWFields = 7;

struct WeightFields {
    float *out[kWFields];
    Grid3 tgt[kWFields];
    int centered[kWFields]; // bit a: samples are cell-centred along axis a
    int di[2][kMaxSuper];   // [centred?][sub-sample]: integer cell offset ...
    float fr[2][kMaxSuper]; // ... and fp32 fraction (the same along every axis)
    int n;
};

// Pass 1 (every brick): the brick-level form of the exact sign shortcut -- a window of one sign gives every sample of the
// brick n^3 or 0.  Away from the surface that is almost every brick; they are finished here (seven coalesced stores per
// thread, few registers, full occupancy).  Bricks whose window changes sign go on a list for k_sdf_weights, whose 150
// registers allow ONE workgroup per CU: run over all bricks it spent most of its time waiting for the window loads of bricks
// that needed no arithmetic (5.8 ms at 512^3, 45.6 ms at 1024^3).
// Pass 0 (round 5): the signs met inside every brick's OWN 32x4x4 cells (bit 0: a negative value, bit 1: a non-negative one), each SDF
// cell read exactly once, coalesced.  A brick's window (cells -2 .. extent around it, clamped at the border) lies inside the 27 bricks
// around it, so when those agree on one sign the window has that sign and pass 1 needs no SDF read at all; only where they disagree does
// it scan the window itself (the exact test, as before) -- it used to read the window of EVERY brick, 3.35x the lattice.
// One WAVE per brick (eight cells per lane, a ballot instead of a block barrier), each workgroup walking many bricks: a workgroup per
// brick is 2.2 M workgroups at 1024^3, and dispatching them -- ~3.4 ns each -- took as long as the old pass itself.
__global__ __launch_bounds__(kBlock) void k_sdf_sign_blocks(const float *__restrict__ sdf, Grid3 src, Grid3 bricks, Box3 box, uint8_t *__restrict__ signs)
{
    const size_t nb = box.vol();
    const int lane = threadIdx.x & 63;
    for (size_t t = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); t < nb; t += (size_t)gridDim.x * (kBlock / 64)) {
        int bc[3];
        box_coords(box, t, bc[0], bc[1], bc[2]);
        const size_t b = (size_t)bc[0] + (size_t)bricks.r[0] * ((size_t)bc[1] + (size_t)bricks.r[1] * (size_t)bc[2]);
        const int o0[3] = {bc[0] * kWBX, bc[1] * kWBY, bc[2] * kWBZ};
        int neg = 0, pos = 0;
#pragma unroll
        for (int u = 0; u < kWThreads / 64; ++u) { // lanes 0..31: a 128-B row of x, lanes 32..63 the next y
            const int t = u * 64 + lane;
            const int c[3] = {o0[0] + t % kWBX, o0[1] + (t / kWBX) % kWBY, o0[2] + t / (kWBX * kWBY)};
            if (c[0] < src.r[0] && c[1] < src.r[1] && c[2] < src.r[2]) {
                const float v = sdf[lin3(src, c[0], c[1], c[2])];
                if (v < 0.f) neg = 1;
                else pos = 1;
            }
        }
        const bool any_neg = __ballot(neg) != 0ull, any_pos = __ballot(pos) != 0ull;
        if (lane == 0) signs[b] = (uint8_t)((any_neg ? 1 : 0) | (any_pos ? 2 : 0));
    }
}

// Pass 1, one THREAD per brick: where the 27 bricks around it agree on a sign and the lattices already hold that constant for it
// (`state`, temporal reuse below) there is nothing to do -- in a running simulation that is nearly every brick; the others go on the work
// list of k_sdf_weights_far.
__global__ __launch_bounds__(kBlock) void k_brick_triage(Grid3 bricks, Box3 box, const uint8_t *__restrict__ signs, const uint8_t *__restrict__ state,
                                                         int32_t *__restrict__ work_list /* [0]: count, then brick ids */)
{
    const size_t nb = box.vol();
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < nb; t += (size_t)gridDim.x * kBlock) {
        int bc[3];
        box_coords(box, t, bc[0], bc[1], bc[2]);
        const size_t b = (size_t)bc[0] + (size_t)bricks.r[0] * ((size_t)bc[1] + (size_t)bricks.r[1] * (size_t)bc[2]);
        unsigned sg = 0u;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int q[3] = {bc[0] + dx, bc[1] + dy, bc[2] + dz};
                    if (q[0] >= 0 && q[1] >= 0 && q[2] >= 0 && q[0] < bricks.r[0] && q[1] < bricks.r[1] && q[2] < bricks.r[2])
                        sg |= signs[q[0] + (size_t)bricks.r[0] * (q[1] + (size_t)bricks.r[1] * q[2])];
                }
        const uint8_t now = sg == 1u ? 2 : (sg == 2u ? 1 : 0); // unanimous: the constant the brick gets (k_sdf_weights_far's encoding)
        if (now != 0 && state[b] == now) continue;
        work_list[1 + atomicAdd(work_list, 1)] = (int32_t)b; // (the order of the list does not matter)
    }
}

// Temporal reuse (round 5): `state[b]` remembers what the seven lattices hold for brick b since the last time THESE allocations were
// filled -- 1: 0.0 everywhere, 2: 1.0 everywhere, 3: computed values, 0: unknown.  A far brick whose constant has not changed (a
// simulation's next frame: every brick the surface has not reached or left) is not stored again: the pass then only READS the SDF
// windows, instead of writing seven full-size lattices (30 GB at 1024^3) whatever the liquid's volume.
__global__ __launch_bounds__(kWThreads) void k_sdf_weights_far(const float *__restrict__ sdf, Grid3 src, Grid3 bricks, WeightFields F,
                                                               int32_t *__restrict__ near_list /* [0]: count, then brick ids */,
                                                               uint8_t *__restrict__ state, const uint8_t *__restrict__ signs,
                                                               const int32_t *__restrict__ work_list)
{
    const int n = F.n;
    const int b = work_list[1 + blockIdx.x];
    const uint8_t held = state[b]; // (read by every thread before the barriers below, written by thread 0 behind them)
    const int bc[3] = {b % bricks.r[0], (b / bricks.r[0]) % bricks.r[1], b / (bricks.r[0] * bricks.r[1])};
    const int o0[3] = {bc[0] * kWBX, bc[1] * kWBY, bc[2] * kWBZ};
    int seen_neg = 0, seen_pos = 0;
    if (threadIdx.x < 27) { // the signs of the 27 bricks around this one: a superset of the window's
        const int q[3] = {bc[0] + (int)threadIdx.x % 3 - 1, bc[1] + ((int)threadIdx.x / 3) % 3 - 1, bc[2] + (int)threadIdx.x / 9 - 1};
        if (q[0] >= 0 && q[1] >= 0 && q[2] >= 0 && q[0] < bricks.r[0] && q[1] < bricks.r[1] && q[2] < bricks.r[2]) {
            const uint8_t sg = signs[q[0] + bricks.r[0] * (q[1] + bricks.r[1] * q[2])];
            seen_neg = sg & 1;
            seen_pos = (sg >> 1) & 1;
        }
    }
    int any_neg = __syncthreads_or(seen_neg);
    int any_pos = __syncthreads_or(seen_pos);
    if (any_neg && any_pos) { // the neighbourhood is mixed: the exact test on the window itself
        seen_neg = seen_pos = 0;
        for (int w = threadIdx.x; w < kWHX * kWHY * kWHZ; w += kWThreads) {
            const int wx = w % kWHX, wy = (w / kWHX) % kWHY, wz = w / (kWHX * kWHY);
            const float v = sdf[lin3(src, clampi(o0[0] - 2 + wx, 0, src.r[0] - 1), clampi(o0[1] - 2 + wy, 0, src.r[1] - 1),
                                     clampi(o0[2] - 2 + wz, 0, src.r[2] - 1))];
            if (v < 0.f) seen_neg = 1;
            else seen_pos = 1;
        }
        any_neg = __syncthreads_or(seen_neg);
        any_pos = __syncthreads_or(seen_pos);
    }
    if (any_pos && any_neg) {
        if (threadIdx.x == 0) {
            near_list[1 + atomicAdd(near_list, 1)] = b; // (the order of the list does not matter)
            state[b] = 3;
        }
        return;
    }
    const uint8_t now = any_neg ? 2 : 1;
    if (held == now) return; // the lattices already hold this brick's constant
    const float value = (float)(any_neg ? n * n * n : 0) / (float)(n * n * n);
    const int t = threadIdx.x;
    if (t == 0) state[b] = now;
    const int p[3] = {o0[0] + t % kWBX, o0[1] + (t / kWBX) % kWBY, o0[2] + t / (kWBX * kWBY)};
    for (int f = 0; f < kWFields; ++f) {
        const Grid3 tgt = F.tgt[f];
        if (p[0] < tgt.r[0] && p[1] < tgt.r[1] && p[2] < tgt.r[2]) F.out[f][lin3(tgt, p[0], p[1], p[2])] = value;
    }
}

// Pass 2: the bricks near the surface (near_list), everything from LDS.
__global__ __launch_bounds__(kWThreads) void k_sdf_weights(const float *__restrict__ sdf, Grid3 src, Grid3 bricks, WeightFields F,
                                                           const int32_t *__restrict__ near_list)
{
    __shared__ float win[kWHX * kWHY * kWHZ];
    const int n = F.n;
    const float n3 = (float)(n * n * n);
    const int b = near_list[1 + blockIdx.x];
    const int o0[3] = {(b % bricks.r[0]) * kWBX, ((b / bricks.r[0]) % bricks.r[1]) * kWBY, (b / (bricks.r[0] * bricks.r[1])) * kWBZ};
    // window cell (wx, wy, wz) holds sdf at clamp(o0 - 2 + w): clamping here reproduces the clamped reads below
    int seen_neg = 0, seen_pos = 0;
    for (int w = threadIdx.x; w < kWHX * kWHY * kWHZ; w += kWThreads) {
        const int wx = w % kWHX, wy = (w / kWHX) % kWHY, wz = w / (kWHX * kWHY);
        const float v = sdf[lin3(src, clampi(o0[0] - 2 + wx, 0, src.r[0] - 1), clampi(o0[1] - 2 + wy, 0, src.r[1] - 1),
                                 clampi(o0[2] - 2 + wz, 0, src.r[2] - 1))];
        win[w] = v;
        if (v < 0.f) seen_neg = 1;
        else seen_pos = 1;
    }
    // brick-level form of the exact sign shortcut below: a window of one sign gives every sample of the brick n^3 / 0
    // (away from the surface that is almost every brick, and it saves the per-sample 4^3 scan of the window)
    const int any_neg = __syncthreads_or(seen_neg);
    const int any_pos = __syncthreads_or(seen_pos);
    const int t = threadIdx.x;
    const int p[3] = {o0[0] + t % kWBX, o0[1] + (t / kWBX) % kWBY, o0[2] + t / (kWBX * kWBY)};
    // NB: unclamped source coordinates c in [o0-2, o0+extent] map to the window cell that was filled with the clamped
    // coordinate, so `at` returns exactly what the reference's clamped read returns.
    auto at = [&](int cx, int cy, int cz) {
        const int ix = clampi(cx - (o0[0] - 2), 0, kWHX - 1), iy = clampi(cy - (o0[1] - 2), 0, kWHY - 1), iz = clampi(cz - (o0[2] - 2), 0, kWHZ - 1);
        return win[ix + kWHX * (iy + kWHY * iz)];
    };
    for (int f = 0; f < kWFields; ++f) {
        const Grid3 tgt = F.tgt[f];
        if (p[0] >= tgt.r[0] || p[1] >= tgt.r[1] || p[2] >= tgt.r[2]) continue;
        float *out = F.out[f] + lin3(tgt, p[0], p[1], p[2]);
        if (!any_pos || !any_neg) {
            *out = (float)(any_neg ? n * n * n : 0) / n3;
            continue;
        }
        const int *di[3];
        const float *fr[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int c = (F.centered[f] >> a) & 1;
            di[a] = F.di[c];
            fr[a] = F.fr[c];
        }
        int lo[3], hi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = p[a] + di[a][0];
            hi[a] = p[a] + di[a][n - 1] + 1;
        }
        // exact shortcut: interpolation preserves the sign, so an all-negative (all non-negative) neighbourhood
        // gives n^3 (0).
        bool allneg = true, allpos = true;
        for (int kk = lo[2]; kk <= hi[2]; ++kk)
            for (int jj = lo[1]; jj <= hi[1]; ++jj)
                for (int ii = lo[0]; ii <= hi[0]; ++ii) {
                    const float v = at(ii, jj, kk);
                    if (v < 0.f) allpos = false;
                    else allneg = false;
                }
        int count;
        if (allneg) count = n * n * n;
        else if (allpos) count = 0;
        else {
            count = 0;
            for (int sz = 0; sz < n; ++sz)
                for (int sy = 0; sy < n; ++sy)
                    for (int sx = 0; sx < n; ++sx) {
                        const int bx0 = p[0] + di[0][sx], by0 = p[1] + di[1][sy], bz0 = p[2] + di[2][sz];
                        const float tx = fr[0][sx], ty = fr[1][sy], tz = fr[2][sz];
                        const float c00 = lerp32(at(bx0, by0, bz0), at(bx0 + 1, by0, bz0), tx);
                        const float c10 = lerp32(at(bx0, by0 + 1, bz0), at(bx0 + 1, by0 + 1, bz0), tx);
                        const float c01 = lerp32(at(bx0, by0, bz0 + 1), at(bx0 + 1, by0, bz0 + 1), tx);
                        const float c11 = lerp32(at(bx0, by0 + 1, bz0 + 1), at(bx0 + 1, by0 + 1, bz0 + 1), tx);
                        const float c0 = lerp32(c00, c10, ty);
                        const float c1 = lerp32(c01, c11, ty);
                        if (lerp32(c0, c1, tz) < 0.f) ++count;
                    }
        }
        *out = (float)count / n3;
    }
}

// ---------------------------------------------------------------------------------------------
// P4: tile occupancy + classification
// ---------------------------------------------------------------------------------------------
struct TileGrid {
    int tr[3];
    // the tiles a launch covers (round 6, slab-local pre-pass: the tiles of the rank's window; full grid: bn == tr, lo == 0)
    int lo[3], bn[3];
    __host__ __device__ size_t vol() const { return (size_t)tr[0] * tr[1] * tr[2]; }
    __host__ __device__ size_t launch() const { return (bn[0] > 0 && bn[1] > 0 && bn[2] > 0) ? (size_t)bn[0] * bn[1] * bn[2] : 0; }
};
static TileGrid tile_grid(const int gr[3])
{
    TileGrid t;
    for (int a = 0; a < 3; ++a) {
        t.tr[a] = (gr[a] + kTile - 1) / kTile;
        t.lo[a] = 0;
        t.bn[a] = t.tr[a];
    }
    return t;
}
// workgroup b of a tile kernel -> linear tile id on the full tile grid
__device__ __forceinline__ unsigned launch_tile(const TileGrid &t, unsigned b)
{
    const int x = t.lo[0] + (int)(b % (unsigned)t.bn[0]), y = t.lo[1] + (int)((b / (unsigned)t.bn[0]) % (unsigned)t.bn[1]),
              z = t.lo[2] + (int)(b / ((unsigned)t.bn[0] * (unsigned)t.bn[1]));
    return (unsigned)x + (unsigned)t.tr[0] * ((unsigned)y + (unsigned)t.tr[1] * (unsigned)z);
}
__device__ __forceinline__ size_t tile_of(const TileGrid &t, int i, int j, int k)
{
    return (size_t)(i / kTile) + (size_t)t.tr[0] * ((size_t)(j / kTile) + (size_t)t.tr[1] * (size_t)(k / kTile));
}

// (slab-local pre-pass: a tile outside the launch box of its grid belongs to another rank's window and is never marked)
__device__ __forceinline__ bool in_launch(const TileGrid &t, int i, int j, int k)
{
    const int x = i / kTile - t.lo[0], y = j / kTile - t.lo[1], z = k / kTile - t.lo[2];
    return x >= 0 && x < t.bn[0] && y >= 0 && y < t.bn[1] && z >= 0 && z < t.bn[2];
}
// Tile occupancy.  Kind 0: faces (both directions along `axis`) of hit cells, cpp:887-1000; kind 1: the 4 `axis` edges of
// ACTIVE cells, cpp:1003-1057.  All six occupancy grids of one level (3 face lattices, 3 edge lattices) come from ONE read of
// the cell lattice: one launch per lattice re-read the level-0 SDF / labels nine times (2.5 ms of the 7.7 ms classification
// at 512^3)
struct TileSets {
    TileGrid tg[2][3];  // [kind][axis]
    uint8_t *occ[2][3];
};
// ---------------------------------------------------------------------------------------------
// P2: mask + base labels
// ---------------------------------------------------------------------------------------------
// Marks of one group of V consecutive x cells (first cell (i, j, k), bit u of `hits`: cell i + u marks) on a lattice's tile occupancy:
// the tiles of the entries cell + (0 | dx, 0 | dy, 0 | dz).  Round 6: a wave of 64 cells used to issue up to 18 one-byte stores per
// marking cell, nearly all to the same two or three bytes (k_mark_tiles_all: 3.3 ms at 1024^3 for 5 GB of reads); now the tile of the
// group itself is stored once per run of lanes that share it, and an offset entry only where it leaves that tile.
__device__ __forceinline__ void mark_group(const TileGrid &tg, uint8_t *__restrict__ occ, int i, int j, int k, int V, unsigned hits, int dx, int dy, int dz)
{
    const bool any = hits != 0u;
    const int key = (any && in_launch(tg, i, j, k)) ? (int)tile_of(tg, i, j, k) : -1;
    const int prev = __shfl_up(key, 1, 64);
    if (key >= 0 && ((threadIdx.x & 63) == 0 || prev != key)) occ[key] = 1;
    if (!any) return;
    // offsets that leave the group's tile (the group is aligned: only its last cell can leave along x)
    const bool ox = dx && (((i + V) & (kTile - 1)) == 0) && ((hits >> (V - 1)) & 1u);
    const bool oy = dy && (((j + 1) & (kTile - 1)) == 0);
    const bool oz = dz && (((k + 1) & (kTile - 1)) == 0);
    for (int m = 1; m < 8; ++m) {
        const bool ux = m & 1, uy = m & 2, uz = m & 4;
        if ((ux && !ox) || (uy && !oy) || (uz && !oz)) continue;
        const int ei = ux ? i + V : i, ej = j + (uy ? 1 : 0), ek = k + (uz ? 1 : 0); // (any entry of the neighbouring tile names it)
        if (in_launch(tg, ei, ej, ek)) occ[tile_of(tg, ei, ej, ek)] = 1;
    }
}

// Level-0 labels and mask; V cells per thread along x (V = 4: one 16-B load, two 4-B stores -- one-byte stores per thread kept the kernel at
// 2.7 TB/s).  `faces`: also marks the tiles of the level-0 face lattices by the SDF rule (cpp:907) -- the classification's occupancy pass
// used to read the 4.3-GB SDF of a 1024^3 grid a second time for exactly this comparison.
template <int V>
__global__ __launch_bounds__(kBlock) void k_mask_labels(const float *__restrict__ liquid, const float *__restrict__ solid,
                                                        Box3 box, double dx, double extrapolation, int8_t *__restrict__ mask,
                                                        int8_t *__restrict__ labels, Grid3 g, Grid3 sim, TileSets T, double occ_sdf, int faces)
{
    const double inner = dx * 2., outer = 3. * dx; // cpp:259-262 (fine bandwidth getter mismatch => 2)
    Box3 bx = box;
    bx.n[0] /= V; // groups of V cells
    const size_t n = bx.vol();
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;; t += (size_t)gridDim.x * kBlock) {
        if (t - (threadIdx.x & 63) >= n) break; // (whole waves stay: mark_group shuffles between neighbouring lanes)
        const bool live = t < n;
        int gi = 0, cj = 0, ck = 0;
        if (live) {
            gi = (int)(t % bx.n[0]);
            const size_t q = t / bx.n[0];
            cj = bx.lo[1] + (int)(q % bx.n[1]);
            ck = bx.lo[2] + (int)(q / bx.n[1]);
        }
        const int ci = box.lo[0] + gi * V;
        unsigned hits = 0u;
        if (live) {
            const size_t o = lin3(g, ci, cj, ck);
            float sv[V], so[V];
            if (V == 4) {
                const float4 a = *reinterpret_cast<const float4 *>(liquid + o);
                sv[0] = a.x; sv[1 % V] = a.y; sv[2 % V] = a.z; sv[3 % V] = a.w;
                if (solid) {
                    const float4 b = *reinterpret_cast<const float4 *>(solid + o);
                    so[0] = b.x; so[1 % V] = b.y; so[2 % V] = b.z; so[3 % V] = b.w;
                }
            } else {
                sv[0] = liquid[o];
                if (solid) so[0] = solid[o];
            }
            int8_t mk[V], lb[V];
#pragma unroll
            for (int u = 0; u < V; ++u) {
                const double sdf = (double)sv[u];
                int m;
                if (ci + u >= sim.r[0] || cj >= sim.r[1] || ck >= sim.r[2]) m = 1; // outside the simulation grid: stays INACTIVE (oct.cpp:375-379)
                else if (sdf > 0 && sdf < outer) m = 0;
                else if (sdf <= 0.) {
                    if (sdf > -inner) m = 0;
                    else {
                        const double sd = solid ? (double)so[u] : -1.0;
                        m = (sd > (-inner - extrapolation)) ? 0 : -1;
                    }
                } else m = 1;
                mk[u] = (int8_t)m;
                lb[u] = (int8_t)(m == 0 ? AVS_ACTIVE : (m < 0 ? AVS_UP : AVS_INACTIVE)); // oct.cpp:383-388
                if (sdf < occ_sdf) hits |= 1u << u;
            }
            if (V == 4) {
                *reinterpret_cast<uint32_t *>(mask + o) = (uint32_t)(uint8_t)mk[0] | ((uint32_t)(uint8_t)mk[1 % V] << 8) | ((uint32_t)(uint8_t)mk[2 % V] << 16) | ((uint32_t)(uint8_t)mk[3 % V] << 24);
                *reinterpret_cast<uint32_t *>(labels + o) = (uint32_t)(uint8_t)lb[0] | ((uint32_t)(uint8_t)lb[1 % V] << 8) | ((uint32_t)(uint8_t)lb[2 % V] << 16) | ((uint32_t)(uint8_t)lb[3 % V] << 24);
            } else {
                mask[o] = mk[0];
                labels[o] = lb[0];
            }
        }
        if (faces) { // faces of hit cells, both directions along each axis (cpp:887-1000)
            if (!live) hits = 0u;
            mark_group(T.tg[0][0], T.occ[0][0], ci, cj, ck, V, hits, 1, 0, 0);
            mark_group(T.tg[0][1], T.occ[0][1], ci, cj, ck, V, hits, 0, 1, 0);
            mark_group(T.tg[0][2], T.occ[0][2], ci, cj, ck, V, hits, 0, 0, 1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// P3: octree passes, one thread per PARENT cell (reads / writes only its own 8 children and itself)
// ---------------------------------------------------------------------------------------------
// pass 1 (oct.cpp:395-565): UP with an ACTIVE sibling -> ACTIVE; a parent with an ACTIVE child -> DOWN
__global__ __launch_bounds__(kBlock) void k_oct_pass1(int8_t *__restrict__ lab, Grid3 g, int8_t *__restrict__ par, Grid3 pg, Box3 box)
{
    const size_t total = box.vol();
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        int i, j, k;
        box_coords(box, t, i, j, k);
        const size_t o = lin3(pg, i, j, k);
        bool any = false;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
            any |= lab[lin3(g, 2 * i + (ci & 1), 2 * j + ((ci >> 1) & 1), 2 * k + ((ci >> 2) & 1))] == AVS_ACTIVE;
        if (!any) continue;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const size_t c = lin3(g, 2 * i + (ci & 1), 2 * j + ((ci >> 1) & 1), 2 * k + ((ci >> 2) & 1));
            if (lab[c] == AVS_UP) lab[c] = AVS_ACTIVE;
        }
        par[o] = AVS_DOWN;
    }
}

// pass 2 (oct.cpp:657-754): DOWN children -> parent DOWN (list applied first, oct.cpp:145), then face
// grading: an UP child with an ACTIVE face neighbour -> parent ACTIVE (oct.cpp:162)
__global__ __launch_bounds__(kBlock) void k_oct_pass2(const int8_t *__restrict__ lab, Grid3 g, int8_t *__restrict__ par, Grid3 pg, Box3 box)
{
    const size_t total = box.vol();
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        int i, j, k;
        box_coords(box, t, i, j, k);
        const size_t o = lin3(pg, i, j, k);
        bool down = false, grade = false;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const int c[3] = {2 * i + (ci & 1), 2 * j + ((ci >> 1) & 1), 2 * k + ((ci >> 2) & 1)};
            const int8_t l = lab[lin3(g, c[0], c[1], c[2])];
            if (l == AVS_DOWN) down = true;
            if (l == AVS_UP) {
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int d = -1; d <= 1; d += 2) {
                        int q[3] = {c[0], c[1], c[2]};
                        q[a] += d;
                        if (q[a] < 0 || q[a] >= g.r[a]) continue;
                        if (lab[lin3(g, q[0], q[1], q[2])] == AVS_ACTIVE) grade = true;
                    }
            }
        }
        if (grade) par[o] = AVS_ACTIVE;
        else if (down) par[o] = AVS_DOWN;
    }
}

// pass 3 (oct.cpp:757-840): an UP child under a still INACTIVE parent -> parent UP
__global__ __launch_bounds__(kBlock) void k_oct_pass3(const int8_t *__restrict__ lab, Grid3 g, int8_t *__restrict__ par, Grid3 pg, Box3 box)
{
    const size_t total = box.vol();
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
        int i, j, k;
        box_coords(box, t, i, j, k);
        const size_t o = lin3(pg, i, j, k);
        if (par[o] != AVS_INACTIVE) continue;
        bool up = false;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
            up |= lab[lin3(g, 2 * i + (ci & 1), 2 * j + ((ci >> 1) & 1), 2 * k + ((ci >> 2) & 1))] == AVS_UP;
        if (up) par[o] = AVS_UP;
    }
}

__global__ __launch_bounds__(kBlock) void k_oct_top(int8_t *__restrict__ lab, Grid3 g, Box3 box)
{
    const size_t n = box.vol();
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += (size_t)gridDim.x * kBlock) {
        int i, j, k;
        box_coords(box, t, i, j, k);
        const size_t o = lin3(g, i, j, k);
        if (lab[o] == AVS_UP) lab[o] = AVS_ACTIVE;
    }
}


// Tile occupancy of a level from its labels: the 4 `axis` edges of ACTIVE cells on the three edge lattices (cpp:1003-1057) and -- above
// level 0, where the faces follow the labels too -- both faces along every axis (at level 0 the SDF rule marks them: k_mask_labels).
// Also the level's "has an ACTIVE cell" flag (oct.cpp:198-211): k_any_active read every label lattice a second time for it.
template <int V>
__global__ __launch_bounds__(kBlock) void k_mark_tiles_all(const int8_t *__restrict__ lab, Grid3 cg, Box3 box, TileSets T, int faces, int *__restrict__ flag)
{
    Box3 bx = box;
    bx.n[0] /= V;
    const size_t n = bx.vol();
    bool seen = false;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;; t += (size_t)gridDim.x * kBlock) {
        if (t - (threadIdx.x & 63) >= n) break; // (whole waves stay: mark_group shuffles between neighbouring lanes)
        const bool live = t < n;
        int gi = 0, cj = 0, ck = 0;
        if (live) {
            gi = (int)(t % bx.n[0]);
            const size_t q = t / bx.n[0];
            cj = bx.lo[1] + (int)(q % bx.n[1]);
            ck = bx.lo[2] + (int)(q / bx.n[1]);
        }
        const int ci = box.lo[0] + gi * V;
        unsigned hits = 0u;
        if (live) {
            const size_t o = lin3(cg, ci, cj, ck);
            if (V == 4) {
                const uint32_t w = *reinterpret_cast<const uint32_t *>(lab + o);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if ((int8_t)((w >> (8 * u)) & 0xffu) == AVS_ACTIVE) hits |= 1u << u;
            } else if (lab[o] == AVS_ACTIVE) hits = 1u;
        }
        seen |= hits != 0u;
        if (!__ballot(hits != 0u)) continue; // nothing ACTIVE in the wave's 64 groups
#pragma unroll
        for (int axis = 0; axis < 3; ++axis) {
            if (faces) mark_group(T.tg[0][axis], T.occ[0][axis], ci, cj, ck, V, hits, axis == 0, axis == 1, axis == 2);
            mark_group(T.tg[1][axis], T.occ[1][axis], ci, cj, ck, V, hits, axis != 0, axis != 1, axis != 2); // HDKcellToEdge: + 0 | 1 along the two other axes
        }
    }
    if (__ballot(seen) && (threadIdx.x & 63) == 0) *flag = 1; // benign race: every writer stores 1
}

// Temporal reuse (round 5): an index lattice is AVS_UNASSIGNED outside the tiles its last classification visited, so the next frame
// resets those tiles only -- `occ` is the occupancy the allocation was last classified with -- instead of a memset of the whole lattice
// (39 GB of fills at 1024^3, 11 ms, for a sheet that occupies one tile in ten).
__device__ __forceinline__ void reset_tile(int32_t *__restrict__ out, const Grid3 &fg, const TileGrid &tg, unsigned tb)
{
    const int tile_x = tb % tg.tr[0], tile_y = (tb / tg.tr[0]) % tg.tr[1], tile_z = tb / (tg.tr[0] * tg.tr[1]);
    const int i = tile_x * kTile + (threadIdx.x & (kTile - 1)), j = tile_y * kTile + (threadIdx.x >> 4);
    if (i >= fg.r[0] || j >= fg.r[1]) return;
    for (int z = 0; z < kTile && tile_z * kTile + z < fg.r[2]; ++z) out[lin3(fg, i, j, tile_z * kTile + z)] = AVS_UNASSIGNED;
}

// Round 6: the tiles a classification has to touch -- occupied now (classified) or visited by the allocation's last classification (reset) --
// are LISTED first, all lattices of a level in one launch (blockIdx.y), and the classification workgroups walk their lattice's list.  A
// workgroup per tile was 274 k workgroups per lattice at 1024^3 -- twice (reset, classify), ten lattices at level 0 -- for a sheet that
// occupies one tile in ten: the classification's 9 ms were mostly the dispatch of workgroups that left at once.
constexpr int kWorkLattices = 7;
struct WorkLists {
    TileGrid tg[kWorkLattices];
    const uint8_t *now[kWorkLattices], *prev[kWorkLattices]; // prev: null = the lattice was filled afresh
    int32_t *cnt[kWorkLattices], *items[kWorkLattices];      // count (the counters lie behind each other: one memset zeroes them) and tile ids; null = lattice unused
};
__global__ __launch_bounds__(kBlock) void k_tile_worklists(WorkLists W)
{
    const int k = blockIdx.y;
    if (!W.items[k]) return;
    const TileGrid tg = W.tg[k];
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= tg.launch()) return;
    const unsigned tb = launch_tile(tg, (unsigned)t);
    if (W.now[k][tb] || (W.prev[k] && W.prev[k][tb])) W.items[k][atomicAdd(W.cnt[k], 1)] = (int32_t)tb; // (the order does not matter)
}
constexpr unsigned kListGrid = 8192; // workgroups walking a list

struct ClassifyArgs {
    int n[3]; // level-0 resolution
    int level, axis;
    double extrapolation;
    const int8_t *lab;
    const float *centerw, *edgew[3], *solid;
};

// classifyOctreeVelocityFacesPartial, cpp:1167-1323
__device__ __forceinline__ int classify_velocity_tile(const ClassifyArgs &A, const Grid3 &fg, const TileGrid &tg, const uint8_t *__restrict__ occ, int32_t *__restrict__ out, unsigned tb)
{
    const int l = A.level, axis = A.axis;
    const Grid3 cg{{A.n[0] >> l, A.n[1] >> l, A.n[2] >> l}};
    const Grid3 c0{{A.n[0], A.n[1], A.n[2]}};
    // one workgroup per 16^3 tile of the lattice; the lattice was pre-filled with AVS_UNASSIGNED and a tile nobody marked
    // keeps that value ("constant tiles are never visited", cpp:1197): it is neither read nor written here
    if (!occ[tb]) { // listed because the allocation's LAST classification visited it: back to AVS_UNASSIGNED
        reset_tile(out, fg, tg, tb);
        return 0;
    }
    const int tile_x = tb % tg.tr[0], tile_y = (tb / tg.tr[0]) % tg.tr[1], tile_z = tb / (tg.tr[0] * tg.tr[1]);
    const int vi_ = tile_x * kTile + (threadIdx.x & (kTile - 1)), vj_ = tile_y * kTile + (threadIdx.x >> 4);
    if (vi_ >= fg.r[0] || vj_ >= fg.r[1]) return 0;
    int cnt = 0; // the tile's DOFs (entries classified 0): what the numbering counts
    for (int vz_ = 0; vz_ < kTile && tile_z * kTile + vz_ < fg.r[2]; ++vz_) {
        int f[3] = {vi_, vj_, tile_z * kTile + vz_};
        const size_t o = lin3(fg, f[0], f[1], f[2]);
        int32_t v = AVS_UNASSIGNED;
        if (occ[tile_of(tg, f[0], f[1], f[2])]) { // constant tiles are never visited (cpp:1197)
            int b[3] = {f[0], f[1], f[2]};
            --b[axis];
            if (b[axis] < 0 || f[axis] >= cg.r[axis]) { // cpp:1210-1215
                if (l == 0) v = AVS_OUTSIDE;
            } else {
                const int bl = A.lab[lin3(cg, b[0], b[1], b[2])], fl = A.lab[lin3(cg, f[0], f[1], f[2])];
                if (l == 0) {
                    if (bl == AVS_ACTIVE && fl == AVS_ACTIVE) { // cpp:1232-1272
                        bool active = A.centerw[lin3(c0, b[0], b[1], b[2])] > 0.f || A.centerw[lin3(c0, f[0], f[1], f[2])] > 0.f;
                        for (int ea = 0; ea < 3 && !active; ++ea) {
                            if (ea == axis) continue;
                            Grid3 eg = c0;
                            eg.r[0] += (ea != 0);
                            eg.r[1] += (ea != 1);
                            eg.r[2] += (ea != 2);
                            const int oa = 3 - axis - ea;
                            for (int d = 0; d < 2; ++d) {
                                int e[3] = {f[0], f[1], f[2]};
                                e[oa] += d;
                                if (A.edgew[ea][lin3(eg, e[0], e[1], e[2])] > 0.f) { active = true; break; }
                            }
                        }
                        if (active) {
                            // solidSurface.getValue(face position) = average of the two axial cells (exact index space)
                            const float s = A.solid ? lerp32(A.solid[lin3(c0, b[0], b[1], b[2])], A.solid[lin3(c0, f[0], f[1], f[2])], 0.5f) : -1.f;
                            v = ((double)s > -A.extrapolation) ? AVS_SOLIDBOUNDARY : 0;
                        } else v = AVS_OUTSIDE;
                    } else if (bl == AVS_INACTIVE || fl == AVS_INACTIVE) v = AVS_OUTSIDE;
                    else if ((bl == AVS_UP && fl == AVS_ACTIVE) || (bl == AVS_ACTIVE && fl == AVS_UP)) v = 0;
                } else if ((bl == AVS_ACTIVE && fl == AVS_ACTIVE) || (bl == AVS_UP && fl == AVS_ACTIVE) || (bl == AVS_ACTIVE && fl == AVS_UP))
                    v = 0; // cpp:1301-1319
            }
        }
        out[o] = v;
        cnt += v == 0;
    }
    return cnt;
}

// classifyRegularVelocityFacesPartial, cpp:1087-1165 (no octree labels involved)
__device__ __forceinline__ int classify_regular_tile(const ClassifyArgs &A, const Grid3 &fg, const TileGrid &tg, const uint8_t *__restrict__ occ, int32_t *__restrict__ out, unsigned tb)
{
    const int axis = A.axis;
    const Grid3 c0{{A.n[0], A.n[1], A.n[2]}};
    // one workgroup per 16^3 tile of the lattice; the lattice was pre-filled with AVS_UNASSIGNED and a tile nobody marked
    // keeps that value ("constant tiles are never visited", cpp:1197): it is neither read nor written here
    if (!occ[tb]) { // listed because the allocation's LAST classification visited it: back to AVS_UNASSIGNED
        reset_tile(out, fg, tg, tb);
        return 0;
    }
    const int tile_x = tb % tg.tr[0], tile_y = (tb / tg.tr[0]) % tg.tr[1], tile_z = tb / (tg.tr[0] * tg.tr[1]);
    const int vi_ = tile_x * kTile + (threadIdx.x & (kTile - 1)), vj_ = tile_y * kTile + (threadIdx.x >> 4);
    if (vi_ >= fg.r[0] || vj_ >= fg.r[1]) return 0;
    int cnt = 0; // the tile's DOFs (entries classified 0): what the numbering counts
    for (int vz_ = 0; vz_ < kTile && tile_z * kTile + vz_ < fg.r[2]; ++vz_) {
        int f[3] = {vi_, vj_, tile_z * kTile + vz_};
        const size_t o = lin3(fg, f[0], f[1], f[2]);
        int32_t v = AVS_UNASSIGNED;
        int b[3] = {f[0], f[1], f[2]};
        --b[axis];
        if (occ[tile_of(tg, f[0], f[1], f[2])] && b[axis] >= 0 && f[axis] < c0.r[axis]) {
            bool active = A.centerw[lin3(c0, b[0], b[1], b[2])] > 0.f || A.centerw[lin3(c0, f[0], f[1], f[2])] > 0.f;
            for (int ea = 0; ea < 3 && !active; ++ea) {
                if (ea == axis) continue;
                Grid3 eg = c0;
                eg.r[0] += (ea != 0);
                eg.r[1] += (ea != 1);
                eg.r[2] += (ea != 2);
                const int oa = 3 - axis - ea;
                for (int d = 0; d < 2; ++d) {
                    int e[3] = {f[0], f[1], f[2]};
                    e[oa] += d;
                    if (A.edgew[ea][lin3(eg, e[0], e[1], e[2])] > 0.f) { active = true; break; }
                }
            }
            if (active) {
                const float s = A.solid ? lerp32(A.solid[lin3(c0, b[0], b[1], b[2])], A.solid[lin3(c0, f[0], f[1], f[2])], 0.5f) : -1.f;
                v = ((double)s > -A.extrapolation) ? AVS_SOLIDBOUNDARY : 0;
            }
        }
        out[o] = v;
        cnt += v == 0;
    }
    return cnt;
}

// classifyEdgeStressesPartial, cpp:1325-1405
__device__ __forceinline__ int classify_edges_tile(const ClassifyArgs &A, const Grid3 &eg, const TileGrid &tg, const uint8_t *__restrict__ occ, int32_t *__restrict__ out, unsigned tb)
{
    const int l = A.level, axis = A.axis;
    const Grid3 cg{{A.n[0] >> l, A.n[1] >> l, A.n[2] >> l}};
    // one workgroup per 16^3 tile of the lattice; the lattice was pre-filled with AVS_UNASSIGNED and a tile nobody marked
    // keeps that value ("constant tiles are never visited", cpp:1197): it is neither read nor written here
    if (!occ[tb]) { // listed because the allocation's LAST classification visited it: back to AVS_UNASSIGNED
        reset_tile(out, eg, tg, tb);
        return 0;
    }
    const int tile_x = tb % tg.tr[0], tile_y = (tb / tg.tr[0]) % tg.tr[1], tile_z = tb / (tg.tr[0] * tg.tr[1]);
    const int vi_ = tile_x * kTile + (threadIdx.x & (kTile - 1)), vj_ = tile_y * kTile + (threadIdx.x >> 4);
    if (vi_ >= eg.r[0] || vj_ >= eg.r[1]) return 0;
    int cnt = 0; // the tile's DOFs (entries classified 0): what the numbering counts
    for (int vz_ = 0; vz_ < kTile && tile_z * kTile + vz_ < eg.r[2]; ++vz_) {
        int e[3] = {vi_, vj_, tile_z * kTile + vz_};
        const size_t o = lin3(eg, e[0], e[1], e[2]);
        int32_t v = AVS_UNASSIGNED;
        if (occ[tile_of(tg, e[0], e[1], e[2])]) {
            bool active = false;
            for (int ci = 0; ci < 4; ++ci) { // HDKedgeToCell, util.h:169-185
                int c[3] = {e[0], e[1], e[2]};
                if (!(ci & 1)) --c[(axis + 1) % 3];
                if (!(ci & 2)) --c[(axis + 2) % 3];
                if (c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= cg.r[0] || c[1] >= cg.r[1] || c[2] >= cg.r[2]) {
                    v = AVS_OUTSIDE;
                    break; // isStressActive keeps whatever it was (cpp:1368-1369)
                }
                const int lb = A.lab[lin3(cg, c[0], c[1], c[2])];
                if (lb == AVS_DOWN) { active = false; break; }
                if (lb == AVS_ACTIVE) active = true;
            }
            if (active) v = (l == 0) ? ((A.edgew[axis][o] > 0.f) ? 0 : AVS_OUTSIDE) : 0;
        }
        out[o] = v;
        cnt += v == 0;
    }
    return cnt;
}

// classifyCenterStressesPartial, cpp:1407-1443.  Round 5: the centre lattice by tiles, like the others.  A centre DOF needs an ACTIVE cell, and k_mark_tiles_all flags, for every ACTIVE
// cell c, the tile of edge c of the axis-0 edge lattice -- whose tile COORDINATES are the cell's (c / 16 per axis; only the tile grid's
// extents differ): the cell tiles with that flag are a superset of the tiles with a centre DOF, every other tile is AVS_UNASSIGNED
// throughout (reset sparsely like the index lattices, not read by the numbering).
__global__ __launch_bounds__(kBlock) void k_center_tiles(const uint8_t *__restrict__ occ_edge0, TileGrid te, TileGrid tc, uint8_t *__restrict__ occ_c)
{
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= tc.vol()) return;
    const int tx = (int)(t % tc.tr[0]), ty = (int)((t / tc.tr[0]) % tc.tr[1]), tz = (int)(t / ((size_t)tc.tr[0] * tc.tr[1]));
    occ_c[t] = occ_edge0[(size_t)tx + (size_t)te.tr[0] * ((size_t)ty + (size_t)te.tr[1] * (size_t)tz)];
}
__device__ __forceinline__ int classify_centers_tiled_tile(const int8_t *__restrict__ lab, const float *__restrict__ centerw, int level, const Grid3 &cg, const TileGrid &tg, const uint8_t *__restrict__ occ, int32_t *__restrict__ out, unsigned tb)
{
    if (!occ[tb]) { // listed because the allocation's LAST classification visited it: back to AVS_UNASSIGNED
        reset_tile(out, cg, tg, tb);
        return 0;
    }
    const int tile_x = tb % tg.tr[0], tile_y = (tb / tg.tr[0]) % tg.tr[1], tile_z = tb / (tg.tr[0] * tg.tr[1]);
    const int i = tile_x * kTile + (threadIdx.x & (kTile - 1)), j = tile_y * kTile + (threadIdx.x >> 4);
    if (i >= cg.r[0] || j >= cg.r[1]) return 0;
    int cnt = 0; // the tile's DOFs (entries classified 0): what the numbering counts
    for (int z = 0; z < kTile && tile_z * kTile + z < cg.r[2]; ++z) {
        const size_t o = lin3(cg, i, j, tile_z * kTile + z);
        const int32_t v = (lab[o] == AVS_ACTIVE && (level != 0 || centerw[o] > 0.f)) ? 0 : AVS_UNASSIGNED;
        out[o] = v;
        cnt += v == 0;
    }
    return cnt;
}

// Slab-local pre-pass (round 6): several ranks classify a tile that straddles their windows (identically: the classification is
// deterministic); ONE of them -- the rank whose slab holds the tile's first plane along the cut axis -- reports its count, the others
// report 0, and the sum over the ranks gives every rank the count of every tile.
struct SlabOwn {
    int on, axis, world, rank;
    int cuts[kMaxRanks + 1]; // fine cells along the cut axis: rank r owns [cuts[r], cuts[r + 1])
};
__device__ __forceinline__ bool slab_owner_is_me(const SlabOwn &own, int tile_coord_along_axis, int level)
{
    long long pos = ((long long)tile_coord_along_axis * kTile) << level;
    if (pos > own.cuts[own.world] - 1) pos = own.cuts[own.world] - 1; // (the extra last tile of a face lattice)
    int r = 0;
    while (r + 1 < own.world && pos >= own.cuts[r + 1]) ++r;
    return r == own.rank;
}

// All lattices of a level in ONE launch (blockIdx.y names the lattice; round 6): a launch per lattice -- seven per level and three for the
// regular grid, each with its own list-counter memset and occupancy copy -- made the classification of the reference's own scenes (304 x 80 x 80:
// a dozen occupied tiles per lattice) a hundred launches of fixed cost.  The same launch keeps the occupancy the lattice was classified
// with (`remember`: the record the NEXT frame's reset list is built from; nobody reads the old record any more once the lists exist).
struct ClassifyBatch {
    ClassifyArgs A;                  // level, labels, weights, solid; the axis comes per lattice
    int kind[kWorkLattices];         // 0 velocity faces, 1 edges, 2 centres, 3 regular-grid faces, -1 unused
    int axis[kWorkLattices];
    Grid3 g[kWorkLattices];
    TileGrid tg[kWorkLattices];
    const uint8_t *occ[kWorkLattices];
    uint8_t *remember[kWorkLattices];
    int32_t *out[kWorkLattices];
    const int32_t *cnt[kWorkLattices], *items[kWorkLattices];
    int32_t *dofs[kWorkLattices];    // per tile of the lattice: its DOF count, where the numbering's scan reads it (zero-filled by the caller)
    SlabOwn own;                     // slab-local mode: a tile is counted by the rank that owns its first plane
    unsigned occ_cap;
};
__global__ __launch_bounds__(kBlock) void k_classify_batch(ClassifyBatch B)
{
    const int k = blockIdx.y;
    const int kind = B.kind[k];
    if (kind < 0) return;
    ClassifyArgs A = B.A;
    A.axis = B.axis[k];
    const Grid3 g = B.g[k];
    const TileGrid tg = B.tg[k];
    const uint8_t *occ = B.occ[k];
    int32_t *out = B.out[k];
    const int n_list = *B.cnt[k];
    for (int li = (int)blockIdx.x; li < n_list; li += (int)gridDim.x) {
        const unsigned tb = (unsigned)B.items[k][li];
        int c;
        if (kind == 0) c = classify_velocity_tile(A, g, tg, occ, out, tb);
        else if (kind == 1) c = classify_edges_tile(A, g, tg, occ, out, tb);
        else if (kind == 2) c = classify_centers_tiled_tile(A.lab, A.centerw, A.level, g, tg, occ, out, tb);
        else c = classify_regular_tile(A, g, tg, occ, out, tb);
        // the tile's DOF count for the numbering (k_tile_counts read every occupied tile of every lattice a second time for it)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if ((threadIdx.x & 63) == 0 && c) {
            bool mine = true;
            if (B.own.on) {
                const int tcoord = B.own.axis == 0 ? (int)(tb % tg.tr[0]) : (B.own.axis == 1 ? (int)((tb / tg.tr[0]) % tg.tr[1]) : (int)(tb / (tg.tr[0] * tg.tr[1])));
                mine = slab_owner_is_me(B.own, tcoord, kind == 3 ? 0 : A.level);
            }
            if (mine) atomicAdd(&B.dofs[k][tb], c);
        }
    }
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < B.occ_cap; i += gridDim.x * kBlock) B.remember[k][i] = occ[i];
}

// ---------------------------------------------------------------------------------------------
// P5: numbering in HDK tile order = exclusive scan over the tile-major flag sequence
// ---------------------------------------------------------------------------------------------
// position of voxel (i, j, k) in the "for each 16^3 tile (x fastest), for each voxel (x fastest)" sweep
__device__ __forceinline__ size_t tilemajor_pos(const Grid3 &g, int i, int j, int k)
{
    const int tx = i / kTile, ty = j / kTile, tz = k / kTile;
    const int ex = min(kTile, g.r[0] - kTile * tx), ey = min(kTile, g.r[1] - kTile * ty), ez = min(kTile, g.r[2] - kTile * tz);
    const int lx = i - kTile * tx, ly = j - kTile * ty, lz = k - kTile * tz;
    return (size_t)g.r[0] * g.r[1] * kTile * tz + (size_t)g.r[0] * kTile * ty * ez + (size_t)kTile * tx * ey * ez +
           ((size_t)lz * ey + ly) * ex + lx;
}

// The sweep visits tiles x-fastest and voxels x-fastest inside a tile, so the id of a flagged voxel is
//   base + (flagged voxels in earlier tiles) + (flagged voxels earlier in its own tile).
// One 256-thread workgroup per tile; in step z = 0..15 thread t reads voxel (lx, ly, lz) = (t & 15, t >> 4, z) of the tile:
// 16 coalesced 64-B rows per step, and (step, thread) order == sweep order inside the tile.  k_tile_counts -> exclusive
// scan over the tiles -> k_tile_ids: the grid is read twice and written where flagged (12 B per voxel instead of the
// 28 B of flag array + full-lattice scan + apply).
__device__ __forceinline__ unsigned tile_flags(const int32_t *__restrict__ grid, const Grid3 &g, int ntx, int nty, size_t *first,
                                               size_t *zstride, int tile)
{
    const int t = threadIdx.x;
    const int tx = tile % ntx, ty = (tile / ntx) % nty, tz = tile / (ntx * nty);
    const int i = tx * kTile + (t & (kTile - 1)), j = ty * kTile + (t >> 4);
    *zstride = (size_t)g.r[0] * g.r[1];
    *first = lin3(g, i, j, tz * kTile);
    unsigned bits = 0u; // bit z: voxel (i, j, tz*16 + z) is a DOF
    if (i < g.r[0] && j < g.r[1]) {
        const int ez = min(kTile, g.r[2] - kTile * tz);
        int32_t v[kTile];
#pragma unroll
        for (int z = 0; z < kTile; ++z) v[z] = z < ez ? grid[*first + (size_t)z * *zstride] : -1;
#pragma unroll
        for (int z = 0; z < kTile; ++z) bits |= (v[z] == 0 ? 1u : 0u) << z;
    }
    return bits;
}

// Round 5: ALL lattices of one kind in one launch (they used to be numbered one after the other: counts, a three-kernel scan, ids and a
// base update per lattice -- 370 launches per frame, more time in launch gaps than in the kernels for every grid below 512^3).  The
// tiles of the lattices are concatenated in numbering order (level-major, then axis), so ONE exclusive scan over the concatenated counts
// gives every tile the id of its first DOF: the same ids as the sequential numbering, bit for bit.
constexpr int kNumLattices = 3 * AVS_MAX_LEVELS;
struct NumLattice {
    int32_t *grid;
    const uint8_t *occ; // tile occupancy the lattice was classified with (null: every tile is read)
    Grid3 g;
    int ntx, nty, tile0, tag; // tiles per row / column, first tile in the concatenated space, dof-table tag (level | axis << 8)
};
struct NumBatch {
    NumLattice lat[kNumLattices];
    int count, total_tiles;
};
// (the batch lives in device memory: a by-value kernel argument indexed with a run-time index is copied to scratch by every thread;
// the lattices' first tiles come by value, compared in an unrolled loop on scalar registers: most workgroups belong to tiles without a
// DOF and must find that out with ONE load -- a search through device memory made the 3 M mostly empty tiles of a 1024^3 batch slower
// than the 12 separate launches had been)
struct NumStarts {
    int tile0[kNumLattices]; // unused entries: INT_MAX
};
__device__ __forceinline__ NumLattice batch_lattice(const NumStarts &S, const NumBatch *__restrict__ B, int tile)
{
    int li = 0;
#pragma unroll
    for (int k = 1; k < kNumLattices; ++k)
        if (tile >= S.tile0[k]) li = k;
    return B->lat[li];
}

__device__ __forceinline__ bool slab_owns_tile(const SlabOwn &own, const NumLattice &Lt, int tile)
{
    const int tc = own.axis == 0 ? tile % Lt.ntx : (own.axis == 1 ? (tile / Lt.ntx) % Lt.nty : tile / (Lt.ntx * Lt.nty));
    return slab_owner_is_me(own, tc, Lt.tag & 0xff);
}

// Round 6: the per-tile DOF counts come out of the classification launch (k_classify_batch); behind the scan the tiles with a DOF that
// this rank classified are LISTED and the numbering workgroups walk the list -- a workgroup per tile of the concatenated space was 376 k
// workgroups at 512^3 (2.9 M at 1024^3), nine in ten of which found nothing to do and left: dispatching them cost more than the work.
__global__ __launch_bounds__(kBlock) void k_tile_select_ids(NumStarts S, const NumBatch *__restrict__ B, int tiles, const int32_t *__restrict__ tile_off,
                                                            int32_t *__restrict__ list)
{
    const int t = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (t >= tiles) return;
    if (tile_off[t + 1] == tile_off[t]) return;
    const NumLattice Lt = batch_lattice(S, B, t);
    if (Lt.occ && !Lt.occ[t - Lt.tile0]) return; // (slab-local pre-pass: a tile of another rank's window -- counted there, not classified here)
    list[1 + atomicAdd(list, 1)] = t;
}

// `table` (optional): the dof table of this kind -- record (level | axis << 8, i, j, k) of every id handed out, what the solver context
// otherwise rebuilds with a sweep over every index lattice (k_dof_table: 5 ms per frame at 1024^3); ids beyond `cap` are not recorded
// (the table was sized from the previous frame's count: the caller then discards it).  total: where the kind's DOF count goes.
__global__ __launch_bounds__(kBlock) void k_tile_ids(NumStarts S, const NumBatch *__restrict__ B, const int32_t *__restrict__ tile_off,
                                                     int32_t *__restrict__ table, long long cap, long long *__restrict__ total, int total_tiles,
                                                     const int32_t *__restrict__ list)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *total = (long long)tile_off[total_tiles];
    __shared__ int cnt[kTile * (kBlock / 64)]; // flagged voxels of (step z, wave w), then their exclusive prefix
    const int n_list = list[0];
    for (int li = (int)blockIdx.x; li < n_list; li += (int)gridDim.x) {
        const int gt = list[1 + li];
        const NumLattice Lt = batch_lattice(S, B, gt);
        const int tile = gt - Lt.tile0;
        size_t first, zs;
        const unsigned bits = tile_flags(Lt.grid, Lt.g, Lt.ntx, Lt.nty, &first, &zs, tile);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        unsigned long long below[kTile]; // ballot of step z restricted to the lanes below this one
#pragma unroll
        for (int z = 0; z < kTile; ++z) {
            const unsigned long long m = __ballot((bits >> z) & 1u);
            if (lane == 0) cnt[z * (kBlock / 64) + wave] = __popcll(m);
            below[z] = m & ((1ull << lane) - 1ull);
        }
        __syncthreads();
        if (wave == 0) { // exclusive scan of the 64 (step, wave) counters, in sweep order
            const int c = cnt[lane];
            int incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            cnt[lane] = incl - c;
        }
        __syncthreads();
        if (bits) {
            const int32_t id0 = tile_off[gt];
            const int ti = (tile % Lt.ntx) * kTile + (threadIdx.x & (kTile - 1)), tj = ((tile / Lt.ntx) % Lt.nty) * kTile + (threadIdx.x >> 4);
            const int tk = (tile / (Lt.ntx * Lt.nty)) * kTile;
#pragma unroll
            for (int z = 0; z < kTile; ++z)
                if ((bits >> z) & 1u) {
                    const int32_t id = id0 + cnt[z * (kBlock / 64) + wave] + __popcll(below[z]);
                    Lt.grid[first + (size_t)z * zs] = id;
                    if (table && id < cap) reinterpret_cast<int4 *>(table)[id] = make_int4(Lt.tag, ti, tj, tk + z);
                }
        }
        __syncthreads(); // cnt is reused by the next tile
    }
}

// Window lists (slab-local pre-pass): the ids of the DOFs in the tiles this rank classified, ascending -- what the rank's index-only
// assembly sweeps instead of 0 .. n.  A tile's DOFs are the consecutive ids tile_off[t] .. tile_off[t + 1].
__global__ __launch_bounds__(kBlock) void k_win_counts(NumStarts S, const NumBatch *__restrict__ B, const int32_t *__restrict__ tile_off, int tiles,
                                                       int32_t *__restrict__ wcnt)
{
    const int t = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (t >= tiles) return;
    const NumLattice Lt = batch_lattice(S, B, t);
    wcnt[t] = (!Lt.occ || Lt.occ[t - Lt.tile0]) ? tile_off[t + 1] - tile_off[t] : 0;
}
__global__ __launch_bounds__(kBlock) void k_win_fill(const int32_t *__restrict__ tile_off, const int32_t *__restrict__ wcnt,
                                                     const int32_t *__restrict__ woff, int32_t *__restrict__ list)
{
    const int t = (int)blockIdx.x;
    const int c = wcnt[t];
    for (int i = threadIdx.x; i < c; i += kBlock) list[woff[t] + i] = tile_off[t] + i;
}
// 32-bit fill of a sub-box of a lattice (the window of an index lattice -> AVS_UNASSIGNED)
__global__ __launch_bounds__(kBlock) void k_fill_box_i32(int32_t *__restrict__ out, Grid3 g, Box3 box, int32_t value)
{
    const size_t n = box.vol();
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += (size_t)gridDim.x * kBlock) {
        int i, j, k;
        box_coords(box, t, i, j, k);
        out[lin3(g, i, j, k)] = value;
    }
}

} // namespace avs

// ---------------------------------------------------------------------------------------------
// host object
// ---------------------------------------------------------------------------------------------
using namespace avs;

struct avs_prepass {
    avs_prepass_desc desc{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int levels = 0; // after capping
    int max_levels = 0;
    DevBuf<float> liquid, solid;
    // the lattices avs_prepass_apply LENDS to solver contexts (SharedBuf, avs_internal.hpp: two allocations each, the one nobody else
    // references is filled)
    SharedBuf<float> centerw, edgew[3], facew[3];
    bool have_solid = false;
    DevBuf<int8_t> mask;
    SharedBuf<int8_t> labels[AVS_MAX_LEVELS];
    SharedBuf<int32_t> vidx[AVS_MAX_LEVELS][3], eidx[AVS_MAX_LEVELS][3], cidx[AVS_MAX_LEVELS], ridx[3];
    DevBuf<int32_t> near_list; // k_sdf_weights_far: [0] = count, then the bricks whose SDF window changes sign
    DevBuf<uint8_t> brick_signs; // k_sdf_sign_blocks: the signs inside every 32x4x4 brick of cells
    DevBuf<uint8_t> occ_store;                             // tile occupancy of every lattice of the run (kept: an allocation + release per frame cost more than the flags' work)
    DevBuf<int> lvl_flags;                                 // "level l has an ACTIVE cell"
    DevBuf<int32_t> win_cnt, win_off;                      // window lists: DOFs per classified tile, their scan
    DevBuf<int32_t> work_lists;                            // classification: per lattice of the level [count | tiles to classify / reset]
    DevBuf<int32_t> num_lists;                             // numbering: the tiles that can hold a DOF / that hold one, per batch
    DevBuf<int32_t> num_counts, num_offsets, num_scan_tmp; // numbering: DOFs per tile of the concatenated lattices of one kind, their exclusive scan
    DevBuf<long long> num_totals;                          // ... and the four DOF counts
    DevBuf<struct avs::NumBatch> num_batches;              // ... the lattice lists of the four batches
    DevBuf<int32_t> work_list;   // k_brick_triage: [0] = count, then the bricks k_sdf_weights_far has to look at
    // temporal reuse: what an ALLOCATION (SharedBuf::id) holds since it was last filled
    struct TileState { uint64_t id = 0; DevBuf<uint8_t> occ; };          // index lattice: the tiles its classification visited
    TileState vstate[AVS_MAX_LEVELS][3][2], estate[AVS_MAX_LEVELS][3][2], rstate[3][2], cstate[AVS_MAX_LEVELS][2];
    struct BrickState { uint64_t ids[7] = {}; DevBuf<uint8_t> st; };     // weight lattices: k_sdf_weights_far's per-brick record
    BrickState wstate[2];
    SharedBuf<int32_t> dof[3];                 // dof tables written by the numbering pass (velocity, edge, centre), lent with the lattices
    long long dof_cap[3] = {0, 0, 0};          // ... sized from the previous run's counts (0: none this run)
    bool dof_valid[3] = {false, false, false}; // ... and complete (the count did not outgrow the capacity)
    int64_t prev_counts[3] = {0, 0, 0};
    bool temporal = true; // AVS_PREPASS_TEMPORAL=0: every run fills everything (measurement / tests)
    // slab-local mode (avs_prepass_set_slab)
    avs::SlabWindow slab;
    avs_allreduce_i32_fn slab_fn = nullptr;
    void *slab_user = nullptr;
    uint64_t slab_sig = 0, state_sig = 0; // temporal-reuse records describe fillings of ONE window: a new window voids them
    SharedBuf<int32_t> wlist[3];          // ids of the DOFs inside the window (velocity, edge, centre), ascending
    int64_t n_window[3] = {0, 0, 0};
    const float *liq = nullptr, *sol = nullptr; // the SDFs of the running call (the caller's device arrays in slab mode, else the copies above)
    int64_t counts[4] = {0, 0, 0, 0}; // velocity, edge, centre, regular
    double ms[4] = {0, 0, 0, 0};
    bool ready = false;
};

static void pp_res(const avs_prepass_desc &d, int kind, int level, int axis, int r[3])
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
static Grid3 g3(const int r[3]) { return Grid3{{r[0], r[1], r[2]}}; }

static void sub_consts(int n, bool centered, int s, int *di, float *fr)
{
    // identical expression to oracle/avs_oracle.c subsample_consts (evaluated in double, rounded once)
    const double d = ((centered ? 0.5 : 0.0) - 0.5) + (((double)s + 0.5) / (double)n - 0.5);
    const double fl = std::floor(d);
    *di = (int)fl;
    *fr = (float)(d - fl);
}

// all seven weight fields in one launch: bricks cover the largest lattice (n + 1 samples per axis)
static Box3 full_box(const int r[3]) { return Box3{{0, 0, 0}, {r[0], r[1], r[2]}}; }
// [lo, hi) along `axis`, everything across
static Box3 slab_box(const int r[3], int axis, int lo, int hi)
{
    Box3 b = full_box(r);
    if (lo < 0) lo = 0;
    if (hi > r[axis]) hi = r[axis];
    b.lo[axis] = lo;
    b.n[axis] = hi > lo ? hi - lo : 0;
    return b;
}

static avs_status run_weights(avs_prepass *p, WeightFields &F)
{
    F.n = p->desc.n_super;
    for (int c = 0; c < 2; ++c)
        for (int s = 0; s < F.n; ++s) sub_consts(F.n, c == 1, s, &F.di[c][s], &F.fr[c][s]);
    int sr[3];
    pp_res(p->desc, 2, 0, 0, sr);
    const Grid3 bricks{{(sr[0] + 1 + kWBX - 1) / kWBX, (sr[1] + 1 + kWBY - 1) / kWBY, (sr[2] + 1 + kWBZ - 1) / kWBZ}};
    const size_t nb = bricks.vol();
    AVS_TRY(p->near_list.reserve(nb + 1));
    AVS_HIP(hipMemsetAsync(p->near_list.p, 0, sizeof(int32_t), p->stream));
    // the record of the seven allocations being filled (keyed on their ids), or a cleared one
    const uint64_t ids[7] = {p->centerw.id, p->edgew[0].id, p->facew[0].id, p->edgew[1].id, p->facew[1].id, p->edgew[2].id, p->facew[2].id};
    avs_prepass::BrickState *ws = nullptr;
    for (int k = 0; k < 2 && !ws; ++k)
        if (p->temporal && p->wstate[k].st.n == nb && memcmp(p->wstate[k].ids, ids, sizeof(ids)) == 0) ws = &p->wstate[k];
    if (!ws) { // an unused record, else the one that does NOT describe the other allocation of the pair (that one is lent out, its record stays valid)
        const uint64_t other = p->centerw.ids[p->centerw.cur ^ 1];
        const int k = p->wstate[0].ids[0] == 0 ? 0 : (p->wstate[1].ids[0] == 0 ? 1 : (p->wstate[0].ids[0] == other ? 1 : 0));
        ws = &p->wstate[k];
        AVS_TRY(ws->st.alloc(nb));
        AVS_HIP(hipMemsetAsync(ws->st.p, 0, nb, p->stream));
        memcpy(ws->ids, ids, sizeof(ids));
    }
    AVS_TRY(p->brick_signs.reserve(nb));
    AVS_TRY(p->work_list.reserve(nb + 1));
    AVS_HIP(hipMemsetAsync(p->work_list.p, 0, sizeof(int32_t), p->stream));
    // slab-local mode: the bricks that hold the samples [win_lo - 1, win_hi + 1] of the level-0 window (the classification reads one
    // entry beyond a tile), their neighbours' signs for the triage
    Box3 tri = full_box(bricks.r), sgn = tri;
    if (p->slab.on) {
        const int a = p->slab.axis, bs = a == 0 ? kWBX : (a == 1 ? kWBY : kWBZ);
        const int lo = (p->slab.win_lo[0] - 1 < 0 ? 0 : p->slab.win_lo[0] - 1) / bs;
        const int hi_s = p->slab.win_hi[0] >= sr[a] ? sr[a] + 1 : p->slab.win_hi[0] + 2; // one past the last sample needed
        const int hi = (hi_s + bs - 1) / bs;
        tri = slab_box(bricks.r, a, lo, hi);
        sgn = slab_box(bricks.r, a, lo - 1, hi + 1);
    }
    hipLaunchKernelGGL(k_sdf_sign_blocks, dim3(grid_for((sgn.vol() + 3) / 4 * kBlock, 1u << 15)), dim3(kBlock), 0, p->stream, p->liq, g3(sr), bricks, sgn, p->brick_signs.p);
    hipLaunchKernelGGL(k_brick_triage, dim3(grid_for(tri.vol(), 1u << 15)), dim3(kBlock), 0, p->stream, bricks, tri, (const uint8_t *)p->brick_signs.p,
                       (const uint8_t *)ws->st.p, p->work_list.p);
    int32_t n_work = 0;
    AVS_HIP(hipMemcpyAsync(&n_work, p->work_list.p, sizeof(int32_t), hipMemcpyDeviceToHost, p->stream));
    AVS_HIP(hipStreamSynchronize(p->stream));
    if (n_work > 0)
        hipLaunchKernelGGL(k_sdf_weights_far, dim3((unsigned)n_work), dim3(kWThreads), 0, p->stream, p->liq, g3(sr), bricks, F, p->near_list.p, ws->st.p,
                           (const uint8_t *)p->brick_signs.p, (const int32_t *)p->work_list.p);
    int32_t n_near = 0;
    AVS_HIP(hipMemcpyAsync(&n_near, p->near_list.p, sizeof(int32_t), hipMemcpyDeviceToHost, p->stream));
    AVS_HIP(hipStreamSynchronize(p->stream));
    if (n_near > 0)
        hipLaunchKernelGGL(k_sdf_weights, dim3((unsigned)n_near), dim3(kWThreads), 0, p->stream, p->liq, g3(sr), bricks, F,
                           (const int32_t *)p->near_list.p);
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

// Index lattice -> AVS_UNASSIGNED everywhere.  `states`: the two records of this lattice (one per allocation of its SharedBuf).  Returns
// the record the classification launch fills in (k_classify_batch keeps the occupancy; the id is set behind it); until then the record is void (id 0), so a run that
// fails in between leaves no stale claim about the allocation.
static avs_status unassign_lattice(avs_prepass *p, SharedBuf<int32_t> &buf, Grid3 g, TileGrid tg, size_t occ_cap, avs_prepass::TileState states[2],
                                   avs_prepass::TileState **out, const Box3 *window, const uint8_t **prev_occ)
{
    // *prev_occ: the tiles the allocation's last classification visited (they go on the classification's work list and are reset by it),
    // or null when the lattice was filled afresh here
    avs_prepass::TileState *ts = nullptr;
    for (int k = 0; k < 2 && !ts; ++k)
        if (p->temporal && states[k].id != 0 && states[k].id == buf.id && states[k].occ.n == occ_cap) ts = &states[k];
    (void)tg;
    *prev_occ = nullptr;
    if (ts) {
        *prev_occ = ts->occ.p;
    } else {
        const uint64_t other = buf.ids[buf.cur ^ 1];
        ts = &states[states[0].id == 0 ? 0 : (states[1].id == 0 ? 1 : (states[0].id == other ? 1 : 0))];
        AVS_TRY(ts->occ.alloc(occ_cap));
        if (window) { // slab-local mode: the rank's window of the lattice only (the rest is never read)
            if (window->vol()) hipLaunchKernelGGL(k_fill_box_i32, dim3(grid_for(window->vol(), 1u << 16)), dim3(kBlock), 0, p->stream, buf.p, g, *window, (int32_t)AVS_UNASSIGNED);
        } else AVS_HIP(hipMemsetAsync(buf.p, 0xFF, g.vol() * sizeof(int32_t), p->stream));
    }
    ts->id = 0;
    AVS_HIP(hipGetLastError());
    *out = ts;
    return AVS_OK;
}

struct EvTimer {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    explicit EvTimer(hipStream_t st) : s(st) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
    ~EvTimer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    void start() { (void)hipEventRecord(a, s); }
    double stop() { (void)hipEventRecord(b, s); (void)hipEventSynchronize(b); float ms = 0.f; (void)hipEventElapsedTime(&ms, a, b); return ms; }
};

extern "C" {

avs_status avs_prepass_create(const avs_prepass_desc *d, avs_prepass **out)
{
    AVS_REQUIRE(d && out, AVS_EINVAL, "null argument");
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    AVS_REQUIRE(pow2(d->nx) && pow2(d->ny) && pow2(d->nz), AVS_EINVAL, "base resolution must be a power of two per axis");
    AVS_REQUIRE(d->desired_levels >= 1 && d->desired_levels <= AVS_MAX_LEVELS, AVS_EINVAL, "desired_levels out of range");
    AVS_REQUIRE(d->n_super >= 1 && d->n_super <= kMaxSuper, AVS_EINVAL, "n_super must be in [1, %d]", kMaxSuper);
    AVS_REQUIRE(d->dx > 0., AVS_EINVAL, "dx must be positive");
    AVS_REQUIRE(d->field_nx >= 0 && d->field_nx <= d->nx && d->field_ny >= 0 && d->field_ny <= d->ny && d->field_nz >= 0 && d->field_nz <= d->nz,
                AVS_EINVAL, "simulation grid %d %d %d must lie in [0, octree grid]", d->field_nx, d->field_ny, d->field_nz);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    AVS_REQUIRE(e == hipSuccess && ndev > 0, AVS_EHIP, "no HIP device available (%s)", hipGetErrorString(e));
    AVS_REQUIRE(d->device >= 0 && d->device < ndev, AVS_EINVAL, "device out of range");
    AVS_HIP(hipSetDevice(d->device));
    avs_prepass *p = new (std::nothrow) avs_prepass();
    AVS_REQUIRE(p, AVS_ENOMEM, "out of host memory");
    p->desc = *d;
    if (p->desc.field_nx == 0) p->desc.field_nx = d->nx;
    if (p->desc.field_ny == 0) p->desc.field_ny = d->ny;
    if (p->desc.field_nz == 0) p->desc.field_nz = d->nz;
    if (d->stream) p->stream = reinterpret_cast<hipStream_t>(d->stream);
    else {
        if (hipStreamCreate(&p->stream) != hipSuccess) { delete p; set_error("hipStreamCreate failed"); return AVS_EHIP; }
        p->own_stream = true;
    }
    // level cap, oct.cpp:32-40
    int L = d->desired_levels;
    for (int n : {d->nx, d->ny, d->nz}) {
        int lg = 0;
        while ((1 << (lg + 1)) <= n) ++lg;
        if (lg < L) L = lg;
    }
    p->max_levels = L < 1 ? 1 : L;
    p->temporal = avs::options_from_env().prepass_temporal != 0;
    *out = p;
    return AVS_OK;
}

avs_status avs_prepass_set_slab(avs_prepass *p, int32_t cut_axis, const int32_t *cuts, int32_t world, int32_t rank, avs_allreduce_i32_fn fn, void *user)
{
    AVS_REQUIRE(p, AVS_EINVAL, "null argument");
    if (!cuts || world <= 1) { // off
        p->slab = avs::SlabWindow{};
        p->slab_fn = nullptr;
        p->slab_user = nullptr;
        p->slab_sig = 0;
        return AVS_OK;
    }
    AVS_REQUIRE(cut_axis >= 0 && cut_axis < 3 && world <= avs::kMaxRanks && rank >= 0 && rank < world && fn, AVS_EINVAL,
                "slab: axis in 0..2, 2 <= world <= %d, 0 <= rank < world and an all-reduce callback are required", avs::kMaxRanks);
    const int extent = cut_axis == 0 ? p->desc.nx : (cut_axis == 1 ? p->desc.ny : p->desc.nz);
    AVS_REQUIRE(cuts[0] == 0 && cuts[world] == extent, AVS_EINVAL, "slab: cuts must run from 0 to the axis' extent (%d)", extent);
    for (int r = 0; r < world; ++r) AVS_REQUIRE(cuts[r] <= cuts[r + 1], AVS_EINVAL, "slab: cuts must ascend");
    avs::SlabWindow w;
    w.on = true;
    w.axis = cut_axis;
    w.world = world;
    w.rank = rank;
    uint64_t sig = 1469598103934665603ull; // FNV-1a over what defines the window
    auto mix = [&](uint64_t v) { sig = (sig ^ v) * 1099511628211ull; };
    mix((uint64_t)cut_axis); mix((uint64_t)world); mix((uint64_t)rank);
    for (int r = 0; r <= world; ++r) { w.cuts[r] = cuts[r]; mix((uint64_t)(uint32_t)cuts[r]); }
    p->slab = w;
    p->slab_fn = fn;
    p->slab_user = user;
    p->slab_sig = sig | 1ull;
    return AVS_OK;
}

avs_status avs_prepass_get_window(avs_prepass *p, int32_t *lo, int32_t *hi, int64_t *n_window)
{
    AVS_REQUIRE(p, AVS_EINVAL, "null argument");
    AVS_REQUIRE(p->ready, AVS_ESTATE, "call avs_prepass_run first");
    for (int l = 0; l < AVS_MAX_LEVELS; ++l) {
        const int nl = l < p->max_levels ? ((p->slab.axis == 0 ? p->desc.nx : (p->slab.axis == 1 ? p->desc.ny : p->desc.nz)) >> l) : 0;
        if (lo) lo[l] = p->slab.on ? p->slab.win_lo[l] : 0;
        if (hi) hi[l] = p->slab.on ? p->slab.win_hi[l] : nl;
    }
    if (n_window)
        for (int k = 0; k < 3; ++k) n_window[k] = p->slab.on ? p->n_window[k] : p->counts[k];
    return AVS_OK;
}

void avs_prepass_destroy(avs_prepass *p)
{
    if (!p) return;
    (void)hipSetDevice(p->desc.device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

avs_status avs_prepass_run(avs_prepass *p, const float *liquid, const float *solid, avs_memspace where)
{
    (void)hipGetLastError(); // (a stale error of the host application's own HIP calls on this thread is not ours: see OptScope)
    AVS_REQUIRE(p && liquid, AVS_EINVAL, "null argument");
    AVS_HIP(hipSetDevice(p->desc.device));
    hipStream_t st = p->stream;
    const avs_prepass_desc &d = p->desc;
    int r0[3];
    pp_res(d, 2, 0, 0, r0);
    const size_t n0 = g3(r0).vol();
    const double extrapolation = d.dx * d.extrapolation_scale; // cpp:243
    p->ready = false;
    for (int k = 0; k < 3; ++k) p->dof_valid[k] = false;
    EvTimer t(st);

    // the SDFs arrive on the simulation grid; outside it they are read with clamped coordinates (border replication)
    const int s0[3] = {d.field_nx, d.field_ny, d.field_nz};
    const bool padded = s0[0] != r0[0] || s0[1] != r0[1] || s0[2] != r0[2];
    const size_t ns = g3(s0).vol();
    auto take = [&](DevBuf<float> &dst, const float *src) -> avs_status {
        AVS_TRY(dst.alloc(n0));
        if (!padded) {
            AVS_HIP(copy_in(dst.p, src, n0 * sizeof(float), where, st));
            return AVS_OK;
        }
        DevBuf<float> tmp;
        const float *sp = src;
        if (where == AVS_MEM_HOST) {
            AVS_TRY(tmp.alloc(ns));
            AVS_HIP(copy_in(tmp.p, src, ns * sizeof(float), where, st));
            sp = tmp.p;
        }
        AVS_TRY(pad_lattice_f32(sp, s0[0], s0[1], s0[2], dst.p, r0[0], r0[1], r0[2], true, 0.f, st));
        AVS_HIP(hipStreamSynchronize(st)); // tmp dies here
        return AVS_OK;
    };
    // device arrays on the octree lattice are read in place (round 6; they are only read while this call runs): the copy was a 4.3-GB
    // pass per SDF and frame at 1024^3 -- and in slab-local mode a full-lattice pass for a window
    const bool alias = where == AVS_MEM_DEVICE && !padded;
    if (alias) p->liq = liquid;
    else {
        AVS_TRY(take(p->liquid, liquid));
        p->liq = p->liquid.p;
    }
    p->have_solid = solid != nullptr;
    p->sol = nullptr;
    if (solid) {
        if (alias) p->sol = solid;
        else {
            AVS_TRY(take(p->solid, solid));
            p->sol = p->solid.p;
        }
    }
    if (where == AVS_MEM_HOST) AVS_HIP(hipStreamSynchronize(st));

    // ---- slab-local mode: the window of every level along the cut axis, and what the octree passes must cover for it -------------
    const int L = p->max_levels;
    const bool slab = p->slab.on;
    const int sa = p->slab.axis;
    int win_lo[AVS_MAX_LEVELS] = {}, win_hi[AVS_MAX_LEVELS] = {}; // index window (level-l cells; multiples of 16 / the level's end)
    int nl_lo[AVS_MAX_LEVELS] = {}, nl_hi[AVS_MAX_LEVELS] = {};   // labels the classification of the window reads
    int v_lo[AVS_MAX_LEVELS] = {}, v_hi[AVS_MAX_LEVELS] = {};     // labels valid before the level serves as children
    int p1_lo[AVS_MAX_LEVELS] = {}, p1_hi[AVS_MAX_LEVELS] = {};   // parents of pass 1 at level l (pass 2 / 3: v_lo .. v_hi)
    {
        const int na = sa == 0 ? d.nx : (sa == 1 ? d.ny : d.nz);
        for (int l = 0; l < L; ++l) {
            const int nl = na >> l;
            if (!slab) { win_lo[l] = nl_lo[l] = v_lo[l] = p1_lo[l] = 0; win_hi[l] = nl_hi[l] = v_hi[l] = p1_hi[l] = nl; continue; }
            const int s_lo = p->slab.cuts[p->slab.rank] >> l, s_hi = (p->slab.cuts[p->slab.rank + 1] + (1 << l) - 1) >> l;
            int lo = s_lo - kSlabIndexMargin, hi = s_hi + kSlabIndexMargin;
            lo = lo < 0 ? 0 : lo / kTile * kTile;
            hi = (hi + kTile - 1) / kTile * kTile;
            if (hi > nl) hi = nl;
            win_lo[l] = lo; win_hi[l] = hi;
            nl_lo[l] = lo - 2 < 0 ? 0 : lo - 2;
            nl_hi[l] = hi + 2 > nl ? nl : hi + 2;
        }
        if (slab) {
            v_lo[L - 1] = nl_lo[L - 1]; v_hi[L - 1] = nl_hi[L - 1];
            for (int l = L - 2; l >= 0; --l) { // pass 2 of a parent reads the face neighbours of its children AFTER their own parents' pass 1
                const int np = na >> (l + 1);
                int a = v_lo[l + 1] - 1, b = v_hi[l + 1] + 1;
                if (nl_lo[l] / 2 < a) a = nl_lo[l] / 2;
                if ((nl_hi[l] + 1) / 2 > b) b = (nl_hi[l] + 1) / 2;
                p1_lo[l + 1] = a < 0 ? 0 : a;
                p1_hi[l + 1] = b > np ? np : b;
                v_lo[l] = 2 * p1_lo[l + 1];
                v_hi[l] = 2 * p1_hi[l + 1];
            }
            // (the mask kernel is pointwise: its box may be any superset -- whole groups of four cells keep it on its vector path along x)
            v_lo[0] = v_lo[0] / 4 * 4;
            v_hi[0] = (v_hi[0] + 3) / 4 * 4 > na ? na : (v_hi[0] + 3) / 4 * 4;
            for (int l = 0; l < AVS_MAX_LEVELS; ++l) { p->slab.win_lo[l] = l < L ? win_lo[l] : 0; p->slab.win_hi[l] = l < L ? win_hi[l] : 0; }
            if (p->state_sig != p->slab_sig) { // the records of what the allocations hold describe another window: void
                for (int l = 0; l < AVS_MAX_LEVELS; ++l)
                    for (int k = 0; k < 2; ++k) {
                        p->cstate[l][k].id = 0;
                        for (int a = 0; a < 3; ++a) p->vstate[l][a][k].id = p->estate[l][a][k].id = 0;
                    }
                for (int a = 0; a < 3; ++a)
                    for (int k = 0; k < 2; ++k) p->rstate[a][k].id = 0;
                for (int k = 0; k < 2; ++k) memset(p->wstate[k].ids, 0, sizeof(p->wstate[k].ids));
                p->state_sig = p->slab_sig;
            }
        }
    }
    // entries [lo, hi) of a lattice with extents gr along the cut axis that the window covers / the tiles of a launch over them
    auto win_box = [&](const int gr[3], int l) {
        if (!slab) return full_box(gr);
        return slab_box(gr, sa, win_lo[l], win_hi[l] >= ((sa == 0 ? d.nx : (sa == 1 ? d.ny : d.nz)) >> l) ? gr[sa] : win_hi[l]);
    };
    auto win_tiles = [&](const int gr[3], int l) {
        TileGrid t = tile_grid(gr);
        if (slab) {
            const Box3 b = win_box(gr, l);
            t.lo[sa] = b.lo[sa] / kTile;
            t.bn[sa] = (b.lo[sa] + b.n[sa] + kTile - 1) / kTile - t.lo[sa];
            if (b.n[sa] <= 0) t.bn[sa] = 0;
        }
        return t;
    };
    auto cell_box = [&](int l, int lo, int hi) {
        int r[3];
        pp_res(d, 2, l, 0, r);
        return slab ? slab_box(r, sa, lo, hi) : full_box(r);
    };

    // ---- P1 weights ------------------------------------------------------------------------
    PhaseScope phase;
    phase.next("Compute Surface Weights"); // cpp:759 (+ "Compute Collision Weights", cpp:776: one fused launch here)
    t.start();
    {
        WeightFields F{};
        int nf = 0;
        auto add = [&](float *out, const int res[3], bool cx, bool cy, bool cz) {
            F.out[nf] = out;
            F.tgt[nf] = g3(res);
            F.centered[nf] = (cx ? 1 : 0) | (cy ? 2 : 0) | (cz ? 4 : 0);
            ++nf;
        };
        AVS_TRY(p->centerw.alloc(n0));
        add(p->centerw.p, r0, true, true, true);
        for (int a = 0; a < 3; ++a) {
            int er[3], fr[3];
            pp_res(d, 1, 0, a, er);
            pp_res(d, 0, 0, a, fr);
            AVS_TRY(p->edgew[a].alloc(g3(er).vol()));
            add(p->edgew[a].p, er, a == 0, a == 1, a == 2); // centred along the edge only
            AVS_TRY(p->facew[a].alloc(g3(fr).vol()));
            add(p->facew[a].p, fr, a != 0, a != 1, a != 2); // centred across the face
        }
        AVS_TRY(run_weights(p, F));
    }
    p->ms[0] = t.stop();

    // ---- P2 + P3 octree --------------------------------------------------------------------
    phase.next("Build Octree"); // cpp:874 (+ "Build Mask for Octree", cpp:813)
    t.start();
    AVS_TRY(p->mask.alloc(n0));
    for (int l = 0; l < L; ++l) {
        int r[3];
        pp_res(d, 2, l, 0, r);
        AVS_TRY(p->labels[l].alloc(g3(r).vol()));
        if (l > 0) AVS_HIP(hipMemsetAsync(p->labels[l].p, 0, g3(r).vol(), st)); // INACTIVE, oct.cpp:59,69 (level 0 is written by the mask kernel wherever it is read)
    }
    // tile-occupancy flags of the lattices of every level ([level][kind][axis][occ_cap] + the cell tiles in slot 6), kept until the numbering,
    // which skips the tiles nobody visited; the level-0 face lattices' flags are set by the mask kernel (the SDF rule, cpp:907) and serve the
    // regular-grid classification too
    const double occ_sdf = 2. * d.dx; // cpp:907
    const size_t occ_cap = (size_t)(d.nx / kTile + 2) * (size_t)(d.ny / kTile + 2) * (size_t)(d.nz / kTile + 2);
    DevBuf<uint8_t> &occ_all = p->occ_store;
    AVS_TRY(occ_all.reserve((size_t)L * 7 * occ_cap));
    AVS_HIP(hipMemsetAsync(occ_all.p, 0, (size_t)L * 7 * occ_cap, st));
    auto tile_sets = [&](int l) {
        TileSets T;
        uint8_t *ob = occ_all.p + (size_t)l * 7 * occ_cap;
        for (int kind = 0; kind < 2; ++kind)
            for (int a = 0; a < 3; ++a) {
                int gr[3];
                pp_res(d, kind, l, a, gr);
                T.tg[kind][a] = win_tiles(gr, l);
                T.occ[kind][a] = ob + (size_t)(kind * 3 + a) * occ_cap;
            }
        return T;
    };
    {
        const Box3 b0 = cell_box(0, v_lo[0], v_hi[0]);
        const TileSets T0 = tile_sets(0);
        if ((b0.lo[0] & 3) == 0 && (b0.n[0] & 3) == 0 && (r0[0] & 3) == 0)
            hipLaunchKernelGGL(k_mask_labels<4>, dim3(grid_for(b0.vol() / 4)), dim3(kBlock), 0, st, p->liq, p->sol, b0, d.dx, extrapolation, p->mask.p,
                               p->labels[0].p, g3(r0), g3(s0), T0, occ_sdf, 1);
        else
            hipLaunchKernelGGL(k_mask_labels<1>, dim3(grid_for(b0.vol())), dim3(kBlock), 0, st, p->liq, p->sol, b0, d.dx, extrapolation, p->mask.p,
                               p->labels[0].p, g3(r0), g3(s0), T0, occ_sdf, 1);
    }
    for (int l = 0; l < L - 1; ++l) {
        int r[3], rp[3];
        pp_res(d, 2, l, 0, r);
        pp_res(d, 2, l + 1, 0, rp);
        const Box3 b1 = cell_box(l + 1, p1_lo[l + 1], p1_hi[l + 1]), b2 = cell_box(l + 1, v_lo[l + 1], v_hi[l + 1]);
        hipLaunchKernelGGL(k_oct_pass1, dim3(grid_for(b1.vol())), dim3(kBlock), 0, st, p->labels[l].p, g3(r), p->labels[l + 1].p, g3(rp), b1);
        hipLaunchKernelGGL(k_oct_pass2, dim3(grid_for(b2.vol())), dim3(kBlock), 0, st, p->labels[l].p, g3(r), p->labels[l + 1].p, g3(rp), b2);
        hipLaunchKernelGGL(k_oct_pass3, dim3(grid_for(b2.vol())), dim3(kBlock), 0, st, p->labels[l].p, g3(r), p->labels[l + 1].p, g3(rp), b2);
    }
    {
        int r[3];
        pp_res(d, 2, L - 1, 0, r);
        const Box3 bt = cell_box(L - 1, v_lo[L - 1], v_hi[L - 1]);
        hipLaunchKernelGGL(k_oct_top, dim3(grid_for(bt.vol())), dim3(kBlock), 0, st, p->labels[L - 1].p, g3(r), bt);
    }
    // cap at the first level without ACTIVE cells, oct.cpp:198-211 (slab mode: a property of the WHOLE octree -- the flags of the
    // rank's part travel with the tile counts, the cap is known after the exchange; until then every level is classified)
    DevBuf<int> &flags = p->lvl_flags;
    AVS_TRY(flags.reserve(AVS_MAX_LEVELS));
    AVS_HIP(hipMemsetAsync(flags.p, 0, AVS_MAX_LEVELS * sizeof(int), st));
    AVS_HIP(hipGetLastError());
    int hflags[AVS_MAX_LEVELS] = {};
    int capped = L; // every level is classified: the flags come out of the classification's occupancy pass (k_mark_tiles_all)
    p->ms[1] = t.stop();
    for (int k = 0; k < 4; ++k) p->counts[k] = 0;

    // ---- P4 classification -----------------------------------------------------------------
    phase.next("Build Octree Velocity and Stress Labels"); // cpp:360 (+ "Build Regular Grid Velocity Labels", cpp:306)
    t.start();
    size_t max_vol = 0;
    // The numbering's tile space, laid out now: the classification launch drops every tile's DOF count where the scan will read it.  One
    // batch per counter -- velocity faces, edges, centres (each over ALL levels, level-major, then axis: numbering order), regular-grid
    // faces (cpp:1486-1509: one counter over the three axes); the four count arrays lie behind each other in ONE buffer (+ the levels'
    // flags in slab mode: what the ranks sum).  Levels beyond the cap are the tail of their batch: the scans stop before them.
    DevBuf<int32_t> &fl = p->num_counts;
    int64_t seg[5] = {0, 0, 0, 0, 0};          // first count of every batch in the buffer (each batch: tiles + 1 entries)
    int level_tile0[4][AVS_MAX_LEVELS + 1];     // first tile of every level inside a batch (counters 0 .. 2)
    int tile0_of[4][AVS_MAX_LEVELS][3] = {};    // first tile of every lattice inside its batch
    for (int counter = 0; counter < 4; ++counter) {
        int64_t tiles = 0;
        auto add = [&](int kind, int l, int a) {
            int gr[3];
            pp_res(d, kind, l, a, gr);
            tile0_of[counter][l][a] = (int)tiles;
            tiles += (int64_t)((gr[0] + kTile - 1) / kTile) * ((gr[1] + kTile - 1) / kTile) * ((gr[2] + kTile - 1) / kTile);
        };
        if (counter < 3) {
            for (int l = 0; l < L; ++l) {
                level_tile0[counter][l] = (int)tiles;
                for (int a = 0; a < (counter == 2 ? 1 : 3); ++a) add(counter, l, a);
            }
            level_tile0[counter][L] = (int)tiles;
        } else
            for (int a = 0; a < 3; ++a) add(0, 0, a);
        AVS_REQUIRE(tiles < (1ll << 31) - 1, AVS_EINVAL, "too many tiles");
        seg[counter + 1] = seg[counter] + tiles + 1;
    }
    const int64_t n_exchange = seg[4] + AVS_MAX_LEVELS;
    AVS_REQUIRE(n_exchange < (1ll << 31) - 1, AVS_EINVAL, "too many tiles");
    AVS_TRY(fl.reserve((size_t)n_exchange));
    AVS_HIP(hipMemsetAsync(fl.p, 0, (size_t)n_exchange * sizeof(int32_t), st));
    SlabOwn own{};
    if (slab) {
        own.on = 1; own.axis = sa; own.world = p->slab.world; own.rank = p->slab.rank;
        for (int r = 0; r <= p->slab.world; ++r) own.cuts[r] = p->slab.cuts[r];
    }
    TileGrid tg0[3];
    for (int l = 0; l < capped; ++l) {
        int cr[3];
        pp_res(d, 2, l, 0, cr);
        uint8_t *ob = occ_all.p + (size_t)l * 7 * occ_cap; // six lattices + the cell tiles (slot 6)
        const TileSets T = tile_sets(l);
        if (l == 0)
            for (int a = 0; a < 3; ++a) tg0[a] = T.tg[0][a];
        {   // (slab mode: the cells whose faces / edges can lie in a tile of the window)
            const Box3 mb = cell_box(l, win_lo[l] - 1, win_hi[l]);
            if ((mb.lo[0] & 3) == 0 && (mb.n[0] & 3) == 0 && (cr[0] & 3) == 0)
                hipLaunchKernelGGL(k_mark_tiles_all<4>, dim3(grid_for(mb.vol() / 4)), dim3(kBlock), 0, st, p->labels[l].p, g3(cr), mb, T, l > 0 ? 1 : 0, flags.p + l);
            else
                hipLaunchKernelGGL(k_mark_tiles_all<1>, dim3(grid_for(mb.vol())), dim3(kBlock), 0, st, p->labels[l].p, g3(cr), mb, T, l > 0 ? 1 : 0, flags.p + l);
        }
        // the seven lattices of the level: fresh fill or the record of the last classification, ONE launch that lists the tiles to touch,
        // then one classification launch per lattice over its list
        WorkLists W{};
        ClassifyBatch B{};
        avs_prepass::TileState *tss[kWorkLattices] = {};
        const TileGrid tc = win_tiles(cr, l);
        uint8_t *occ_c = ob + 6 * occ_cap;
        hipLaunchKernelGGL(k_center_tiles, dim3(grid_for(tc.vol())), dim3(kBlock), 0, st, (const uint8_t *)T.occ[1][0], T.tg[1][0], tc, occ_c);
        AVS_TRY(p->work_lists.reserve(8 + (size_t)kWorkLattices * occ_cap)); // [8 counters | 7 x tile ids]
        AVS_HIP(hipMemsetAsync(p->work_lists.p, 0, 8 * sizeof(int32_t), st));
        B.A.n[0] = d.nx; B.A.n[1] = d.ny; B.A.n[2] = d.nz;
        B.A.level = l;
        B.A.extrapolation = extrapolation;
        B.A.lab = p->labels[l].p;
        B.A.centerw = p->centerw.p;
        for (int b2 = 0; b2 < 3; ++b2) B.A.edgew[b2] = p->edgew[b2].p;
        B.A.solid = p->sol;
        B.occ_cap = (unsigned)occ_cap;
        B.own = own;
        size_t max_launch = 0;
        for (int k = 0; k < kWorkLattices; ++k) {
            const int kind = k < 3 ? 0 : (k < 6 ? 1 : 2), a = k < 6 ? k % 3 : 0;
            int gr[3];
            pp_res(d, kind, l, a, gr);
            SharedBuf<int32_t> &buf = kind == 0 ? p->vidx[l][a] : (kind == 1 ? p->eidx[l][a] : p->cidx[l]);
            AVS_TRY(buf.alloc(g3(gr).vol()));
            if (g3(gr).vol() > max_vol) max_vol = g3(gr).vol();
            W.tg[k] = kind == 2 ? tc : T.tg[kind][a];
            W.now[k] = kind == 2 ? occ_c : T.occ[kind][a];
            W.cnt[k] = p->work_lists.p + k;
            W.items[k] = p->work_lists.p + 8 + (size_t)k * occ_cap;
            const Box3 wb = win_box(gr, l);
            AVS_TRY(unassign_lattice(p, buf, g3(gr), W.tg[k], occ_cap, kind == 0 ? p->vstate[l][a] : (kind == 1 ? p->estate[l][a] : p->cstate[l]), &tss[k],
                                     slab ? &wb : nullptr, &W.prev[k]));
            if (W.tg[k].launch() > max_launch) max_launch = W.tg[k].launch();
            B.kind[k] = kind; B.axis[k] = a; B.g[k] = g3(gr); B.tg[k] = W.tg[k]; B.occ[k] = W.now[k]; B.remember[k] = tss[k]->occ.p; B.out[k] = buf.p;
            B.cnt[k] = W.cnt[k]; B.items[k] = W.items[k];
            B.dofs[k] = fl.p + seg[kind] + tile0_of[kind][l][a];
        }
        if (max_launch) hipLaunchKernelGGL(k_tile_worklists, dim3(grid_for(max_launch), kWorkLattices), dim3(kBlock), 0, st, W);
        {   // (with no tile to launch over the occupancy is still remembered: the grid has at least one workgroup per lattice)
            const unsigned grid = (unsigned)(max_launch < kListGrid ? (max_launch ? max_launch : 1) : kListGrid);
            hipLaunchKernelGGL(k_classify_batch, dim3(grid, kWorkLattices), dim3(kBlock), 0, st, B);
            AVS_HIP(hipGetLastError());
        }
        for (int k = 0; k < kWorkLattices; ++k) { // the records describe these allocations from here on
            const int kind = k < 3 ? 0 : (k < 6 ? 1 : 2), a = k < 6 ? k % 3 : 0;
            SharedBuf<int32_t> &buf = kind == 0 ? p->vidx[l][a] : (kind == 1 ? p->eidx[l][a] : p->cidx[l]);
            tss[k]->id = buf.id;
        }
    }
    { // regular-grid faces, cpp:1457-1481: the occupancy of the level-0 face lattices, marked above by the same rule (kind 0, SDF)
        WorkLists W{};
        ClassifyBatch B{};
        avs_prepass::TileState *tss[3] = {};
        size_t max_launch = 0;
        AVS_TRY(p->work_lists.reserve(8 + (size_t)kWorkLattices * occ_cap));
        AVS_HIP(hipMemsetAsync(p->work_lists.p, 0, 8 * sizeof(int32_t), st));
        B.A.n[0] = d.nx; B.A.n[1] = d.ny; B.A.n[2] = d.nz;
        B.A.level = 0;
        B.A.extrapolation = extrapolation;
        B.A.lab = p->labels[0].p;
        B.A.centerw = p->centerw.p;
        for (int b2 = 0; b2 < 3; ++b2) B.A.edgew[b2] = p->edgew[b2].p;
        B.A.solid = p->sol;
        B.occ_cap = (unsigned)occ_cap;
        B.own = own;
        for (int k = 0; k < kWorkLattices; ++k) B.kind[k] = -1;
        for (int a = 0; a < 3; ++a) {
            int gr[3];
            pp_res(d, 0, 0, a, gr);
            AVS_TRY(p->ridx[a].alloc(g3(gr).vol()));
            W.tg[a] = tg0[a];
            W.now[a] = occ_all.p + (size_t)a * occ_cap;
            W.cnt[a] = p->work_lists.p + a;
            W.items[a] = p->work_lists.p + 8 + (size_t)a * occ_cap;
            const Box3 wb = win_box(gr, 0);
            AVS_TRY(unassign_lattice(p, p->ridx[a], g3(gr), W.tg[a], occ_cap, p->rstate[a], &tss[a], slab ? &wb : nullptr, &W.prev[a]));
            if (W.tg[a].launch() > max_launch) max_launch = W.tg[a].launch();
            B.kind[a] = 3; B.axis[a] = a; B.g[a] = g3(gr); B.tg[a] = W.tg[a]; B.occ[a] = W.now[a]; B.remember[a] = tss[a]->occ.p; B.out[a] = p->ridx[a].p;
            B.cnt[a] = W.cnt[a]; B.items[a] = W.items[a];
            B.dofs[a] = fl.p + seg[3] + tile0_of[3][0][a];
        }
        if (max_launch) hipLaunchKernelGGL(k_tile_worklists, dim3(grid_for(max_launch), kWorkLattices), dim3(kBlock), 0, st, W);
        {
            const unsigned grid = (unsigned)(max_launch < kListGrid ? (max_launch ? max_launch : 1) : kListGrid);
            hipLaunchKernelGGL(k_classify_batch, dim3(grid, kWorkLattices), dim3(kBlock), 0, st, B);
            AVS_HIP(hipGetLastError());
        }
        for (int a = 0; a < 3; ++a) tss[a]->id = p->ridx[a].id;
    }
    AVS_HIP(hipGetLastError());
    if (!slab) { // cap at the first level without ACTIVE cells, oct.cpp:198-211
        AVS_HIP(hipMemcpyAsync(hflags, flags.p, sizeof(hflags), hipMemcpyDeviceToHost, st));
        AVS_HIP(hipStreamSynchronize(st));
        capped = 0;
        while (capped < L && hflags[capped]) ++capped;
    }
    p->levels = capped;
    p->ms[2] = t.stop();
    if (capped == 0) { // no liquid in the refinement band: nothing to solve (the reference asserts here, oct.cpp:206)
        p->ms[3] = 0.;
        p->ready = true;
        return AVS_OK;
    }

    // ---- P5 numbering ----------------------------------------------------------------------
    t.start();
    DevBuf<int32_t> &ids = p->num_offsets, &scan_tmp = p->num_scan_tmp; // (scratch kept across frames)
    DevBuf<long long> &base = p->num_totals;
    (void)max_vol;
    AVS_TRY(base.alloc(4));
    AVS_HIP(hipMemsetAsync(base.p, 0, 4 * sizeof(long long), st));
    PhaseTrace tr(st, "numbering", avs::options_from_env().trace_phases != 0);
    // the batches' lattice lists (the layout is the one the classification counted into: tile0_of, seg)
    NumBatch host_batches[4] = {}; // (alive until the synchronisation behind the last batch: they are the sources of asynchronous copies)
    for (int counter = 0; counter < 4; ++counter) {
        NumBatch &B = host_batches[counter];
        auto add = [&](int32_t *grid, const int gr[3], const uint8_t *occ, int tag, int tile0) {
            NumLattice &Lt = B.lat[B.count++];
            Lt.grid = grid;
            Lt.occ = occ;
            Lt.g = g3(gr);
            Lt.ntx = (gr[0] + kTile - 1) / kTile;
            Lt.nty = (gr[1] + kTile - 1) / kTile;
            Lt.tile0 = tile0;
            Lt.tag = tag;
        };
        if (counter < 3) {
            const int kind = counter;
            for (int l = 0; l < L; ++l)
                for (int a = 0; a < (kind == 2 ? 1 : 3); ++a) {
                    int gr[3];
                    pp_res(d, kind, l, a, gr);
                    const uint8_t *oc = occ_all.p + ((size_t)l * 7 + (kind == 2 ? 6 : (size_t)kind * 3 + a)) * occ_cap; // (centres: the cell tiles with an ACTIVE cell, k_center_tiles)
                    add(kind == 0 ? p->vidx[l][a].p : (kind == 1 ? p->eidx[l][a].p : p->cidx[l].p), gr, oc, l | (a << 8), tile0_of[counter][l][a]);
                }
        } else {
            for (int a = 0; a < 3; ++a) {
                int gr[3];
                pp_res(d, 0, 0, a, gr);
                add(p->ridx[a].p, gr, occ_all.p + (size_t)a * occ_cap, 0, tile0_of[3][0][a]); // classified with the level-0 face occupancy
            }
        }
        B.total_tiles = (int)(seg[counter + 1] - seg[counter] - 1);
    }
    AVS_TRY(ids.reserve((size_t)seg[4]));
    AVS_TRY(p->num_lists.reserve((size_t)seg[4] + 4)); // per batch: [count | tile ids]  (a batch's segment has tiles + 1 entries)
    constexpr unsigned kNumGrid = 4096;                 // workgroups walking a list
    AVS_TRY(p->num_batches.alloc(4));
    NumStarts S[4];
    for (int counter = 0; counter < 4; ++counter) {
        const NumBatch &B = host_batches[counter];
        AVS_TRY(scan_tmp.reserve(scan_tmp_elems((int64_t)B.total_tiles + 1)));
        AVS_HIP(hipMemcpyAsync(p->num_batches.p + counter, &B, sizeof(NumBatch), hipMemcpyHostToDevice, st)); // (pageable source: staged before the call returns)
        for (int k = 0; k < kNumLattices; ++k) S[counter].tile0[k] = k < B.count ? B.lat[k].tile0 : INT_MAX;
    }
    AVS_HIP(hipGetLastError());
    tr.mark("tile counts");
    int eff_tiles[4]; // tiles of the levels below the cap (slab mode: the cap is only known now)
    for (int c = 0; c < 4; ++c) eff_tiles[c] = host_batches[c].total_tiles;
    if (slab) { // ONE exchange: every tile's count from the rank that owns it, every level's "has an ACTIVE cell" from whoever saw one
        AVS_HIP(hipMemcpyAsync(fl.p + seg[4], flags.p, AVS_MAX_LEVELS * sizeof(int), hipMemcpyDeviceToDevice, st));
        const avs_status ex = p->slab_fn(fl.p, n_exchange, (void *)st, p->slab_user);
        AVS_REQUIRE(ex == AVS_OK, ex, "the all-reduce callback of avs_prepass_set_slab failed (%d)", (int)ex);
        AVS_HIP(hipMemcpyAsync(hflags, fl.p + seg[4], sizeof(hflags), hipMemcpyDeviceToHost, st));
        AVS_HIP(hipStreamSynchronize(st));
        capped = 0;
        while (capped < L && hflags[capped]) ++capped;
        p->levels = capped;
        if (capped == 0) {
            p->ms[3] = t.stop();
            for (int k = 0; k < 3; ++k) p->n_window[k] = 0;
            p->ready = true;
            return AVS_OK;
        }
    }
    for (int c = 0; c < 3; ++c) eff_tiles[c] = level_tile0[c][capped]; // (levels are the outer order of a batch: the capped ones are its tail)
    tr.mark("exchange");
    for (int counter = 0; counter < 4; ++counter)
        AVS_TRY(exclusive_scan_i32(fl.p + seg[counter], ids.p + seg[counter], eff_tiles[counter], scan_tmp.p, scan_tmp.n, st));
    // dof tables: slab mode sizes them exactly (the totals are read first); else room for the previous frame's count + 25 % (a first
    // run has no estimate: the context builds them)
    long long hb[4] = {0, 0, 0, 0};
    if (slab) {
        int32_t tot[4];
        for (int c = 0; c < 4; ++c) AVS_HIP(hipMemcpyAsync(&tot[c], ids.p + seg[c] + eff_tiles[c], sizeof(int32_t), hipMemcpyDeviceToHost, st));
        AVS_HIP(hipStreamSynchronize(st));
        for (int c = 0; c < 4; ++c) {
            AVS_REQUIRE(tot[c] >= 0, AVS_EINVAL, "DOF count exceeds int32");
            hb[c] = tot[c];
        }
    }
    for (int k = 0; k < 3; ++k) {
        p->dof_valid[k] = false;
        if (slab) p->dof_cap[k] = hb[k] > 0 ? hb[k] : 1;
        else p->dof_cap[k] = (p->temporal && p->prev_counts[k] > 0) ? p->prev_counts[k] + p->prev_counts[k] / 4 + 4096 : 0;
        if (p->dof_cap[k] > 0) AVS_TRY(p->dof[k].alloc((size_t)p->dof_cap[k] * 4));
        if (slab) AVS_HIP(hipMemsetAsync(p->dof[k].p, 0xFF, (size_t)p->dof_cap[k] * 4 * sizeof(int32_t), st)); // entries outside the window stay recognisably void
    }
    for (int counter = 0; counter < 4; ++counter) {
        const bool tab = counter < 3 && p->dof_cap[counter] > 0;
        if (eff_tiles[counter]) {
            int32_t *list = p->num_lists.p + seg[counter];
            AVS_HIP(hipMemsetAsync(list, 0, sizeof(int32_t), st));
            hipLaunchKernelGGL(k_tile_select_ids, dim3(grid_for((size_t)eff_tiles[counter])), dim3(kBlock), 0, st, S[counter], p->num_batches.p + counter,
                               eff_tiles[counter], (const int32_t *)(ids.p + seg[counter]), list);
            hipLaunchKernelGGL(k_tile_ids, dim3(kNumGrid), dim3(kBlock), 0, st, S[counter], p->num_batches.p + counter,
                               (const int32_t *)(ids.p + seg[counter]), tab ? p->dof[counter].p : (int32_t *)nullptr,
                               tab ? p->dof_cap[counter] : 0ll, base.p + counter, eff_tiles[counter], (const int32_t *)list);
        }
    }
    AVS_HIP(hipGetLastError());
    tr.mark("scans, tables, ids");
    if (slab) { // the window's DOFs of every kind, ascending: the tiles this rank classified, each a run of consecutive ids
        DevBuf<int32_t> &wcnt = p->win_cnt, &woff = p->win_off;
        for (int k = 0; k < 3; ++k) {
            const int tiles = eff_tiles[k];
            AVS_TRY(wcnt.reserve((size_t)tiles + 1));
            AVS_TRY(woff.reserve((size_t)tiles + 1));
            hipLaunchKernelGGL(k_win_counts, dim3(grid_for((size_t)tiles)), dim3(kBlock), 0, st, S[k], p->num_batches.p + k, (const int32_t *)(ids.p + seg[k]), tiles, wcnt.p);
            AVS_TRY(exclusive_scan_i32(wcnt.p, woff.p, tiles, scan_tmp.p, scan_tmp.n, st));
            int32_t nw = 0;
            AVS_HIP(hipMemcpyAsync(&nw, woff.p + tiles, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            AVS_HIP(hipStreamSynchronize(st));
            p->n_window[k] = nw;
            AVS_TRY(p->wlist[k].alloc((size_t)(nw > 0 ? nw : 1)));
            if (tiles && nw) hipLaunchKernelGGL(k_win_fill, dim3((unsigned)tiles), dim3(kBlock), 0, st, (const int32_t *)(ids.p + seg[k]), (const int32_t *)wcnt.p, (const int32_t *)woff.p, p->wlist[k].p);
            AVS_HIP(hipGetLastError());
            AVS_HIP(hipStreamSynchronize(st)); // wcnt / woff are reused
        }
    } else {
        AVS_HIP(hipMemcpyAsync(hb, base.p, sizeof(hb), hipMemcpyDeviceToHost, st));
        AVS_HIP(hipStreamSynchronize(st));
    }
    tr.mark("window lists");
    for (int k = 0; k < 4; ++k) p->counts[k] = hb[k];
    for (int k = 0; k < 3; ++k) {
        p->dof_valid[k] = p->dof_cap[k] > 0 && hb[k] <= p->dof_cap[k];
        p->prev_counts[k] = hb[k];
    }
    p->ms[3] = t.stop();
    p->ready = true;
    return AVS_OK;
}

avs_status avs_prepass_get_info(avs_prepass *p, avs_prepass_info *info)
{
    AVS_REQUIRE(p && info, AVS_EINVAL, "null argument");
    AVS_REQUIRE(p->ready, AVS_ESTATE, "call avs_prepass_run first");
    info->levels = p->levels;
    info->n_velocity = p->counts[0];
    info->n_edge = p->counts[1];
    info->n_center = p->counts[2];
    info->n_regular = p->counts[3];
    info->weights_ms = p->ms[0];
    info->octree_ms = p->ms[1];
    info->classify_ms = p->ms[2];
    info->number_ms = p->ms[3];
    return AVS_OK;
}

avs_status avs_prepass_get_labels(avs_prepass *p, int32_t level, int8_t *out, avs_memspace where)
{
    AVS_REQUIRE(p && out, AVS_EINVAL, "null argument");
    AVS_REQUIRE(p->ready && level >= 0 && level < p->levels, AVS_EINVAL, "level out of range");
    AVS_HIP(hipSetDevice(p->desc.device));
    int r[3];
    pp_res(p->desc, 2, level, 0, r);
    AVS_HIP(copy_out(out, p->labels[level].p, g3(r).vol(), where, p->stream));
    AVS_HIP(hipStreamSynchronize(p->stream));
    return AVS_OK;
}

avs_status avs_prepass_get_mask(avs_prepass *p, int8_t *out, avs_memspace where)
{
    AVS_REQUIRE(p && out, AVS_EINVAL, "null argument");
    AVS_REQUIRE(p->ready, AVS_ESTATE, "call avs_prepass_run first");
    AVS_HIP(hipSetDevice(p->desc.device));
    int r[3];
    pp_res(p->desc, 2, 0, 0, r);
    AVS_HIP(copy_out(out, p->mask.p, g3(r).vol(), where, p->stream));
    AVS_HIP(hipStreamSynchronize(p->stream));
    return AVS_OK;
}

avs_status avs_prepass_get_index(avs_prepass *p, avs_index_kind kind, int32_t level, int32_t axis, int32_t *out, avs_memspace where)
{
    AVS_REQUIRE(p && out, AVS_EINVAL, "null argument");
    AVS_REQUIRE(p->ready && level >= 0 && level < p->levels && axis >= 0 && axis < 3, AVS_EINVAL, "level / axis out of range");
    AVS_HIP(hipSetDevice(p->desc.device));
    int r[3];
    pp_res(p->desc, (int)kind, level, axis, r);
    const int32_t *src = kind == AVS_INDEX_VELOCITY ? p->vidx[level][axis].p : (kind == AVS_INDEX_EDGE ? p->eidx[level][axis].p : p->cidx[level].p);
    AVS_HIP(copy_out(out, src, g3(r).vol() * sizeof(int32_t), where, p->stream));
    AVS_HIP(hipStreamSynchronize(p->stream));
    return AVS_OK;
}

avs_status avs_prepass_get_regular_index(avs_prepass *p, int32_t axis, int32_t *out, avs_memspace where)
{
    AVS_REQUIRE(p && out, AVS_EINVAL, "null argument");
    AVS_REQUIRE(p->ready && p->levels >= 1 && axis >= 0 && axis < 3, AVS_EINVAL, "not ready / axis out of range");
    AVS_HIP(hipSetDevice(p->desc.device));
    int r[3];
    pp_res(p->desc, 0, 0, axis, r);
    AVS_HIP(copy_out(out, p->ridx[axis].p, g3(r).vol() * sizeof(int32_t), where, p->stream));
    AVS_HIP(hipStreamSynchronize(p->stream));
    return AVS_OK;
}

avs_status avs_prepass_get_weights(avs_prepass *p, avs_field_kind kind, int32_t axis, float *out, avs_memspace where)
{
    AVS_REQUIRE(p && out, AVS_EINVAL, "null argument");
    AVS_REQUIRE(p->ready && axis >= 0 && axis < 3, AVS_EINVAL, "not ready / axis out of range");
    AVS_HIP(hipSetDevice(p->desc.device));
    int r[3];
    const float *src;
    switch (kind) {
    case AVS_FIELD_CENTER_WEIGHTS: pp_res(p->desc, 2, 0, 0, r); src = p->centerw.p; break;
    case AVS_FIELD_EDGE_WEIGHTS: pp_res(p->desc, 1, 0, axis, r); src = p->edgew[axis].p; break;
    case AVS_FIELD_FACE_WEIGHTS: pp_res(p->desc, 0, 0, axis, r); src = p->facew[axis].p; break;
    default: set_error("not a weight field"); return AVS_EINVAL;
    }
    AVS_HIP(copy_out(out, src, g3(r).vol() * sizeof(float), where, p->stream));
    AVS_HIP(hipStreamSynchronize(p->stream));
    return AVS_OK;
}

avs_status avs_prepass_apply(avs_prepass *p, avs_ctx *ctx)
{
    (void)hipGetLastError(); // (a stale error of the host application's own HIP calls on this thread is not ours: see OptScope)
    AVS_REQUIRE(p && ctx, AVS_EINVAL, "null argument");
    AVS_REQUIRE(p->ready && p->levels >= 1, AVS_ESTATE, "pre-pass has no levels (no liquid in the refinement band)");
    AVS_REQUIRE(ctx->desc.levels == p->levels && ctx->desc.nx == p->desc.nx && ctx->desc.ny == p->desc.ny && ctx->desc.nz == p->desc.nz &&
                    ctx->desc.device == p->desc.device,
                AVS_EINVAL, "context (levels %d) does not match the pre-pass (levels %d)", ctx->desc.levels, p->levels);
    AVS_REQUIRE(ctx->desc.field_nx == p->desc.field_nx && ctx->desc.field_ny == p->desc.field_ny && ctx->desc.field_nz == p->desc.field_nz,
                AVS_EINVAL, "context and pre-pass disagree on the simulation grid (%d %d %d vs %d %d %d)", ctx->desc.field_nx,
                ctx->desc.field_ny, ctx->desc.field_nz, p->desc.field_nx, p->desc.field_ny, p->desc.field_nz);
    AVS_HIP(hipStreamSynchronize(p->stream));
    // No copy (round 5): the context takes references on the lattices this pre-pass just filled; the next avs_prepass_run fills the
    // OTHER allocation of every lattice, so what the context reads stays as it is until the next apply.  (The weights and the
    // regular-grid indices live on the padded octree lattices here, which is what the context stores.)
    avs::PrepassLoan loan;
    loan.levels = p->levels;
    for (int l = 0; l < p->levels; ++l) {
        loan.labels[l] = p->labels[l].handle();
        loan.cidx[l] = p->cidx[l].handle();
        for (int a = 0; a < 3; ++a) {
            loan.vidx[l][a] = p->vidx[l][a].handle();
            loan.eidx[l][a] = p->eidx[l][a].handle();
        }
    }
    loan.centerw = p->centerw.handle();
    for (int a = 0; a < 3; ++a) {
        loan.edgew[a] = p->edgew[a].handle();
        loan.facew[a] = p->facew[a].handle();
        loan.ridx[a] = p->ridx[a].handle();
    }
    for (int k = 0; k < 3; ++k) loan.counts[k] = p->counts[k];
    if (p->slab.on) {
        loan.slab = p->slab;
        for (int k = 0; k < 3; ++k) {
            loan.wlist[k] = p->wlist[k].handle();
            loan.n_window[k] = p->n_window[k];
        }
    }
    if (p->dof_valid[0] && p->dof_valid[1] && p->dof_valid[2])
        for (int k = 0; k < 3; ++k) loan.dof[k] = p->dof[k].handle();
    for (int a = 0; a < 3; ++a) // the occupancy each regular-grid lattice was classified with: the context flags its transfer tiles without reading the rest
        for (int k = 0; k < 2; ++k)
            if ((p->temporal || p->slab.on) && p->rstate[a][k].id != 0 && p->rstate[a][k].id == p->ridx[a].id) { // (slab-local: the lattice is only defined inside those tiles)
                int gr[3];
                pp_res(p->desc, 0, 0, a, gr);
                loan.ridx_occ[a] = p->rstate[a][k].occ.p;
                for (int b = 0; b < 3; ++b) loan.ridx_occ_tiles[a][b] = (gr[b] + kTile - 1) / kTile;
            }
    AVS_TRY(avs::adopt_prepass_lattices(ctx, loan));
    AVS_HIP(hipStreamSynchronize(ctx->stream));
    return AVS_OK;
}

} // extern "C"
