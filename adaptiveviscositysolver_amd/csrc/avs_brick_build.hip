// avs_brick_build.hip -- builds the brick-structured SpMV form (avs_brick.hip) of the brick-major system on the device.
//
// Input (BrickSource): a CSR in the solver's numbering -- the permuted global system (p_row_ptr / p_col) or, round 5, the LOCAL system of
// a partitioned solve, whose columns [n_rows, n_cols) are halo entries -- its value codes and packed words (ValueIndex: one dictionary of
// < 2047 values, code and column in one 32-bit word), and for every column the reference DOF behind it + the dof table (which face a
// column is: level, axis, cell -- cpp:1566-1593).  Output: the arrays behind BrickView.  A fill run names its first column absolutely
// (8 B per run), which makes the form independent of how the rows of a brick are laid out: the [interior | halo-reading] split order
// of a rank's rows simply gives a brick two runs of rows, i.e. two sets of tiles, and halo columns are just columns >= n_rows.  Lossless by construction: a row is stored as a pattern only after its words have been compared, one by one,
// with the words of the pattern it hashed to (K5); anything that does not fit a limit (lattice, extra slots, LDS budgets, runs) is kept
// as packed words ("streamed" rows) -- tests/test_gpu_matrix_formats.py and avs_bench_spmv compare y with the plain CSR kernel bit for bit.
//
//   K1  geometry of every row (brick, level, axis, cell) + brick starts               [one host round trip: the tile table, O(bricks)]
//   K3  per G tile: extra slots (off-lattice columns in the 27 neighbour bricks), per entry its slot and pattern word, per row a 64-bit
//       hash of the word sequence
//   K4  global pattern set: hash-table insert, pattern ids + storage, copy of the representative rows' words    [counts read back]
//   K5  every row against its pattern, word for word (hash collisions become streamed rows)
//   K6  per tile: patterns used (LDS budgets), execution order (sorted by length, pattern), row descriptors, halo fill runs, own slots,
//       descriptor block
//   K7  streamed rows: their packed words compacted tile after tile, headers patched
#include <algorithm>
#include <chrono>
#include <vector>

#include "avs_internal.hpp"

namespace avs {

namespace {
constexpr int kBlk = 256;
constexpr int kTileBlk = 512;
constexpr int kHashBitsPat = 21;                 // pattern hash table: 2 M slots
constexpr int kMaxPatterns = 1 << 19;            // more distinct patterns: the scene is not regular, the form is not built
constexpr int kBlockStride = 1728;               // words reserved per descriptor block (48 + 2 x 320 + 640 + 384 = 1712 at the limits)
constexpr int kXsRows = kBrickXSlots;

struct TileInfo {                                // built on the device from the brick starts
    int32_t row0, nrows, is_g, obx, oby, obz, pad0, pad1;
};
__device__ __forceinline__ uint64_t geo_pack(uint32_t brick, int level, int axis, int i, int j, int k)
{
    return ((uint64_t)brick << 38) | ((uint64_t)(level & 7) << 35) | ((uint64_t)(axis & 3) << 33) | ((uint64_t)(i & 0x7ff) << 22) |
           ((uint64_t)(j & 0x7ff) << 11) | (uint64_t)(k & 0x7ff);
}
struct Geo { int brick, level, axis, i, j, k; };
__device__ __forceinline__ Geo geo_unpack(uint64_t g)
{
    Geo r;
    r.brick = (int)(g >> 38); r.level = (int)((g >> 35) & 7); r.axis = (int)((g >> 33) & 3);
    r.i = (int)((g >> 22) & 0x7ff); r.j = (int)((g >> 11) & 0x7ff); r.k = (int)(g & 0x7ff);
    return r;
}

// K1: brick, level, axis, level-cell of every row of the brick-major system (the key of k_brick_keys, avs_reorder.hip)
__global__ __launch_bounds__(kBlk) void k_bk_geo(const int32_t *__restrict__ vdof, const int32_t *__restrict__ ref_id, int64_t n, int nx, int ny,
                                                int nz, uint64_t *__restrict__ geo)
{
    const int64_t r = (int64_t)blockIdx.x * kBlk + threadIdx.x;
    if (r >= n) return;
    const int4 rec = reinterpret_cast<const int4 *>(vdof)[ref_id[r]];
    const int level = rec.x & 0xff, axis = (rec.x >> 8) & 0xff;
    int px = rec.y << level, py = rec.z << level, pz = rec.w << level;
    px = px < nx ? px : nx - 1;
    py = py < ny ? py : ny - 1;
    pz = pz < nz ? pz : nz - 1;
    const uint32_t nbx = (uint32_t)((nx + 7) >> 3), nby = (uint32_t)((ny + 7) >> 3);
    const uint32_t brick = ((uint32_t)(pz >> 3) * nby + (uint32_t)(py >> 3)) * nbx + (uint32_t)(px >> 3);
    geo[r] = geo_pack(brick, level, axis, rec.y, rec.z, rec.w);
}
__global__ __launch_bounds__(kBlk) void k_bk_first(const uint64_t *__restrict__ geo, int64_t n, int32_t *__restrict__ first)
{
    const int64_t r = (int64_t)blockIdx.x * kBlk + threadIdx.x;
    if (r >= n) return;
    first[r] = (r == 0 || (geo[r] >> 38) != (geo[r - 1] >> 38)) ? 1 : 0;
}
__global__ __launch_bounds__(kBlk) void k_bk_brick_starts(const uint64_t *__restrict__ geo, const int32_t *__restrict__ first,
                                                         const int32_t *__restrict__ bidx, int64_t n, int32_t *__restrict__ bstart,
                                                         int32_t *__restrict__ bbrick)
{
    const int64_t r = (int64_t)blockIdx.x * kBlk + threadIdx.x;
    if (r >= n || !first[r]) return;
    bstart[bidx[r]] = (int32_t)r;
    bbrick[bidx[r]] = (int32_t)(geo[r] >> 38);
}

// ---- tiles, from the brick starts (all O(bricks)): a brick with >= kBrickMinRows rows is a G tile (cut every kBrickMaxRows rows), the rows
//      of a run of smaller bricks are cut into E tiles of kBrickETileRows rows
__global__ __launch_bounds__(kBlk) void k_bk_run_first(const int32_t *__restrict__ bstart, int nbricks, int32_t *__restrict__ run_first)
{
    const int b = blockIdx.x * kBlk + threadIdx.x;
    if (b >= nbricks) return;
    const bool big = bstart[b + 1] - bstart[b] >= kBrickMinRows;
    const bool prev_big = b == 0 || bstart[b] - bstart[b - 1] >= kBrickMinRows;
    run_first[b] = (!big && prev_big) ? 1 : 0;
}
__global__ __launch_bounds__(kBlk) void k_bk_run_starts(const int32_t *__restrict__ bstart, const int32_t *__restrict__ run_first,
                                                       const int32_t *__restrict__ run_id, int nbricks, int32_t *__restrict__ run_start)
{
    const int b = blockIdx.x * kBlk + threadIdx.x;
    if (b < nbricks && run_first[b]) run_start[run_id[b]] = bstart[b];
}
// tiles a brick starts: a big brick ceil(rows / kBrickMaxRows); a small one the multiples of kBrickETileRows (counted from its run's
// first row) that fall inside it
__device__ __forceinline__ int tiles_of_brick(const int32_t *bstart, const int32_t *run_first, const int32_t *run_id, const int32_t *run_start, int b,
                                              int *first_off)
{
    const int r0 = bstart[b], rows = bstart[b + 1] - r0;
    if (rows >= kBrickMinRows) { *first_off = 0; return (rows + kBrickMaxRows - 1) / kBrickMaxRows; }
    const int rid = run_id[b] + (run_first[b] ? 0 : -1); // exclusive scan: the run this brick belongs to
    const int off = r0 - run_start[rid];
    const int k0 = (off + kBrickETileRows - 1) / kBrickETileRows, k1 = (off + rows + kBrickETileRows - 1) / kBrickETileRows;
    *first_off = k0 * kBrickETileRows - off;
    return k1 - k0;
}
__global__ __launch_bounds__(kBlk) void k_bk_tile_counts(const int32_t *__restrict__ bstart, const int32_t *__restrict__ run_first,
                                                        const int32_t *__restrict__ run_id, const int32_t *__restrict__ run_start, int nbricks,
                                                        int32_t *__restrict__ tcount, int *__restrict__ too_big)
{
    const int b = blockIdx.x * kBlk + threadIdx.x;
    if (b >= nbricks) return;
    if (bstart[b + 1] - bstart[b] >= 2048) atomicExch(too_big, 1); // (a run offset names at most 2047 rows of a brick)
    int fo;
    tcount[b] = tiles_of_brick(bstart, run_first, run_id, run_start, b, &fo);
}
__global__ __launch_bounds__(kBlk) void k_bk_tile_rows(const int32_t *__restrict__ bstart, const int32_t *__restrict__ bbrick,
                                                      const int32_t *__restrict__ run_first, const int32_t *__restrict__ run_id,
                                                      const int32_t *__restrict__ run_start, const int32_t *__restrict__ tile0, int nbricks, int nbx,
                                                      int nby, TileInfo *__restrict__ tiles)
{
    const int b = blockIdx.x * kBlk + threadIdx.x;
    if (b >= nbricks) return;
    int fo;
    const int cnt = tiles_of_brick(bstart, run_first, run_id, run_start, b, &fo);
    const int r0 = bstart[b], rows = bstart[b + 1] - r0;
    const bool big = rows >= kBrickMinRows;
    const int id = bbrick[b];
    for (int i = 0; i < cnt; ++i) {
        TileInfo &T = tiles[tile0[b] + i];
        T.row0 = r0 + (big ? i * kBrickMaxRows : fo + i * kBrickETileRows);
        T.is_g = big ? 1 : 0;
        T.obx = id % nbx; T.oby = (id / nbx) % nby; T.obz = id / (nbx * nby);
    }
}
// rows of a tile = up to the next tile's first row
__global__ __launch_bounds__(kBlk) void k_bk_tile_finish(int ntiles, int64_t n, TileInfo *__restrict__ tiles)
{
    const int t = blockIdx.x * kBlk + threadIdx.x;
    if (t >= ntiles) return;
    tiles[t].nrows = (t + 1 < ntiles ? tiles[t + 1].row0 : (int32_t)n) - tiles[t].row0;
}

// slot of a face on the lattices of a tile whose brick is (obx, oby, obz): -1 when it is not on them
__device__ __forceinline__ int lattice_slot(const Geo &g, int obx, int oby, int obz)
{
    if (g.level > 3) return -1;
    const int w = 8 >> g.level, S = w + 2;
    const int rx = g.i - w * obx + 1, ry = g.j - w * oby + 1, rz = g.k - w * obz + 1;
    if (rx < 0 || rx >= S || ry < 0 || ry >= S || rz < 0 || rz >= S) return -1;
    return kBrickLoff[g.level] + ((rz * S + ry) * S + rx) * 3 + g.axis;
}
// base of a row (level lr, local level-lr cell c = cell - w_lr * origin) on the lattice of level lc (as k_spmv_brick computes it)
__device__ __forceinline__ int base_slot(int lr, int cx, int cy, int cz, int lc)
{
    const int up = lc > lr ? lc - lr : 0, dn = lr > lc ? lr - lc : 0;
    const int S = (8 >> lc) + 2;
    const int bx = ((cx >> up) << dn) + 1, by = ((cy >> up) << dn) + 1, bz = ((cz >> up) << dn) + 1;
    return kBrickLoff[lc] + ((bz * S + by) * S + bx) * 3;
}

__device__ __forceinline__ uint64_t mix64(uint64_t h)
{
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return h;
}

// global pattern set (K4a): insert a row's hash; the winner of a slot leaves the row as the pattern's representative
__device__ __forceinline__ void pattern_insert(uint64_t h, int row, unsigned long long *__restrict__ keys, int32_t *__restrict__ rep, int *overflow)
{
    unsigned s = (unsigned)(h >> (64 - kHashBitsPat));
    for (int probe = 0; probe < 64; ++probe) {
        const unsigned long long cur = __hip_atomic_load(&keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == h) return;
        if (cur == 0ull) {
            const unsigned long long old = atomicCAS(&keys[s], 0ull, (unsigned long long)h);
            if (old == 0ull) { rep[s] = (int32_t)row; return; }
            if (old == h) return;
        }
        s = (s + 1) & ((1u << kHashBitsPat) - 1u);
    }
    atomicExch(overflow, 1); // (the row will not find its hash in K5 and becomes a streamed row)
}

// K3: one workgroup per tile, 16 lanes per row.  Pass A gives every entry its lattice slot and collects the off-lattice columns that lie in
// the 27 neighbour bricks (extra slots, ascending column order so that the fill finds runs); pass B forms the pattern words and every
// row's hash (a sum of position-keyed word hashes: the lanes of a row add their shares), and the tile's DISTINCT hashes go into the
// global pattern set (a few dozen inserts per tile instead of one per row).
__global__ __launch_bounds__(kTileBlk) void k_bk_rows(const TileInfo *__restrict__ tiles, const int32_t *__restrict__ row_ptr,
                                                     const int32_t *__restrict__ col, const uint16_t *__restrict__ codes,
                                                     const uint64_t *__restrict__ geo, int nbx, int nby, int zero_code,
                                                     uint32_t *__restrict__ ewords, uint16_t *__restrict__ eslot, uint64_t *__restrict__ row_hash,
                                                     uint32_t *__restrict__ rgeo, uint16_t *__restrict__ ownslot,
                                                     unsigned long long *__restrict__ keys, int32_t *__restrict__ rep, int *__restrict__ overflow)
{
    __shared__ int xset[512];       // hash set of extra columns
    __shared__ int xlist[512];      // ... compacted, then sorted
    __shared__ int xcount;
    __shared__ unsigned long long hset[2048]; // the tile's distinct row hashes
    const TileInfo &T = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    const int row0 = T.row0, nrows = T.nrows;
    if (!T.is_g) {
        for (int r = tid; r < nrows; r += kTileBlk) {
            row_hash[row0 + r] = 0ull;
            rgeo[row0 + r] = 0u;
            ownslot[row0 + r] = 0xffffu;
        }
        return;
    }
    const int obx = T.obx, oby = T.oby, obz = T.obz;
    const int sub = tid & 15, grp = tid >> 4;
    constexpr int GRPS = kTileBlk / 16;
    xset[tid] = -1;
    for (int i = tid; i < 2048; i += kTileBlk) hset[i] = 0ull;
    if (tid == 0) xcount = 0;
    __syncthreads();
    for (int r = grp; r < nrows; r += GRPS) {
        const int rs = row_ptr[row0 + r], re = row_ptr[row0 + r + 1];
        for (int k = rs + sub; k < re; k += 16) {
            const int c = col[k];
            const Geo g = geo_unpack(geo[c]);
            int slot = lattice_slot(g, obx, oby, obz);
            if (slot < 0) {
                slot = 0xffff;
                const int cbx = g.brick % nbx, cby = (g.brick / nbx) % nby, cbz = g.brick / (nbx * nby);
                if (abs(cbx - obx) <= 1 && abs(cby - oby) <= 1 && abs(cbz - obz) <= 1) {
                    slot = 0xfffe; // candidate for an extra slot
                    unsigned h = ((unsigned)c * 2654435761u) >> 23; // 9 bits
                    for (int probe = 0; probe < 512; ++probe) {
                        const int old = atomicCAS(&xset[h], -1, c);
                        if (old == -1 || old == c) break;
                        h = (h + 1) & 511u;
                    }
                }
            }
            eslot[k] = (uint16_t)slot;
        }
    }
    __syncthreads();
    if (xset[tid] >= 0) xlist[atomicAdd(&xcount, 1)] = xset[tid];
    __syncthreads();
    const int nxs = xcount; // (a full set means more than 512 candidates: the surplus never got in and stays off-lattice)
    int mine = -1, rank = 0;
    if (tid < nxs) {
        mine = xlist[tid];
        for (int i = 0; i < nxs; ++i) rank += xlist[i] < mine ? 1 : 0;
    }
    __syncthreads();
    if (tid < nxs) xlist[rank] = mine; // ascending
    __syncthreads();
    const int nxk = nxs < kXsRows ? nxs : kXsRows; // the smallest kXsRows columns get a slot
    for (int r = grp; r < nrows; r += GRPS) {
        const int row = row0 + r;
        const Geo gr = geo_unpack(geo[row]);
        const int rs = row_ptr[row], re = row_ptr[row + 1];
        const int own = lattice_slot(gr, obx, oby, obz);
        const int lr = gr.level > 3 ? 3 : gr.level;
        const int w = 8 >> lr;
        const int cx = gr.i - w * obx, cy = gr.j - w * oby, cz = gr.k - w * obz; // local cell, -1 .. w for a lattice row
        bool ok = own >= 0 && re > rs && re - rs <= kBrickPatLen && cx >= -1 && cx < 15 && cy >= -1 && cy < 15 && cz >= -1 && cz < 15;
        uint64_t h = 0ull;
        bool simple = true;
        for (int k = rs + sub; k < re; k += 16) {
            int slot = eslot[k];
            int tag = slot < kBrickLoff[1] ? 0 : slot < kBrickLoff[2] ? 1 : slot < kBrickLoff[3] ? 2 : 3;
            if (slot == 0xfffe) { // extra slot?
                const int c = col[k];
                int lo = 0, hi = nxk;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (xlist[mid] < c) lo = mid + 1; else hi = mid;
                }
                slot = (lo < nxk && xlist[lo] == c) ? kBrickSlotsPad + lo : 0xffff;
                tag = 0;
                eslot[k] = (uint16_t)slot;
            }
            const unsigned code = codes ? codes[k] : 0u;   // (value-code variant: the patterns carry the geometry only)
            uint32_t word = 0u;
            if (slot != 0xffff && gr.level <= 3) {
                const int delta = slot - base_slot(lr, cx, cy, cz, tag);
                if (delta < -4096 || delta > 4095 || (codes && code >= (unsigned)zero_code)) ok = false;
                word = ((uint32_t)(delta & 0x1fff) << 19) | ((uint32_t)tag << 14) | (code << 3);
                if (tag) simple = false;
            } else {
                ok = false;
            }
            ewords[k] = word;
            h += mix64(((uint64_t)word << 8) ^ (uint64_t)(k - rs) ^ 0x9e3779b97f4a7c15ull);
        }
        // the 16 lanes of the row: sum of the hashes, AND of the flags
        int okf = ok ? 1 : 0, sf = simple ? 1 : 0;
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) {
            h += __shfl_xor(h, m, 16);
            okf &= __shfl_xor(okf, m, 16);
            sf &= __shfl_xor(sf, m, 16);
        }
        h = mix64(h ^ (uint64_t)(re - rs));
        if (h == 0ull) h = 1ull;
        if (sub == 0) {
            row_hash[row] = okf ? h : 0ull;
            rgeo[row] = okf ? (((uint32_t)gr.level << 18) | ((uint32_t)gr.axis << 16) | ((uint32_t)(cz + 1) << 8) | ((uint32_t)(cy + 1) << 4) |
                               (uint32_t)(cx + 1) | (sf ? 0x80000000u : 0u))
                            : 0u;
            ownslot[row] = own >= 0 ? (uint16_t)own : (uint16_t)0xffffu;
            if (okf) { // first sighting of the hash in this tile: into the global set
                unsigned s2 = (unsigned)(h >> 53); // 11 bits
                bool fresh = false;
                for (int probe = 0; probe < 2048; ++probe) {
                    const unsigned long long old = atomicCAS(&hset[s2], 0ull, (unsigned long long)h);
                    if (old == 0ull) { fresh = true; break; }
                    if (old == h) break;
                    s2 = (s2 + 1) & 2047u;
                }
                if (fresh) pattern_insert(h, row, keys, rep, overflow);
            }
        }
    }
}

// K4b: every occupied slot becomes a pattern: id, length, storage
__global__ __launch_bounds__(kBlk) void k_bk_assign(const unsigned long long *__restrict__ keys, const int32_t *__restrict__ rep,
                                                   const int32_t *__restrict__ row_ptr, int32_t *__restrict__ slot_id,
                                                   int32_t *__restrict__ pat_rep, int32_t *__restrict__ pat_off, int *__restrict__ counters)
{
    const int s = blockIdx.x * kBlk + threadIdx.x;
    if (s >= (1 << kHashBitsPat)) return;
    slot_id[s] = -1;
    if (!keys[s]) return;
    const int id = atomicAdd(&counters[0], 1);
    if (id >= kMaxPatterns) return;
    const int r = rep[s];
    const int len4 = (row_ptr[r + 1] - row_ptr[r] + 3) & ~3;
    const int off = atomicAdd(&counters[1], len4);
    slot_id[s] = id;
    pat_rep[id] = r;
    pat_off[id] = off;
}
// K4c: the representative rows' words become the pattern table (padded to quads: first entry's slot, code of 0.0)
__global__ __launch_bounds__(kBlk) void k_bk_copy_patterns(int npat, const int32_t *__restrict__ pat_rep, const int32_t *__restrict__ pat_off,
                                                          const int32_t *__restrict__ row_ptr, const uint32_t *__restrict__ ewords,
                                                          int zero_code, uint32_t *__restrict__ pwords)
{
    const int sub = threadIdx.x & 15;
    const int p = (blockIdx.x * kBlk + threadIdx.x) >> 4;
    if (p >= npat) return;
    const int r = pat_rep[p], rs = row_ptr[r], len = row_ptr[r + 1] - rs, off = pat_off[p];
    const int len4 = (len + 3) & ~3;
    const uint32_t pad = (ewords[rs] & ~(0x7ffu << 3)) | ((uint32_t)zero_code << 3);
    for (int j = sub; j < len4; j += 16) pwords[off + j] = j < len ? ewords[rs + j] : pad;
}
// K5: a row keeps its pattern only if its words ARE the pattern's words
__global__ __launch_bounds__(kBlk) void k_bk_verify(const uint64_t *__restrict__ row_hash, int64_t n, const unsigned long long *__restrict__ keys,
                                                   const int32_t *__restrict__ slot_id, const int32_t *__restrict__ pat_off,
                                                   const int32_t *__restrict__ pat_rep, const int32_t *__restrict__ row_ptr,
                                                   const uint32_t *__restrict__ ewords, const uint32_t *__restrict__ pwords,
                                                   int32_t *__restrict__ row_pid)
{
    const int64_t r = (int64_t)blockIdx.x * kBlk + threadIdx.x;
    if (r >= n) return;
    const uint64_t h = row_hash[r];
    int pid = -1;
    if (h) {
        unsigned s = (unsigned)(h >> (64 - kHashBitsPat));
        for (int probe = 0; probe < 64; ++probe) {
            const unsigned long long cur = keys[s];
            if (cur == h) { pid = slot_id[s]; break; }
            if (cur == 0ull) break;
            s = (s + 1) & ((1u << kHashBitsPat) - 1u);
        }
        if (pid >= 0) {
            const int rs = row_ptr[r], len = row_ptr[r + 1] - rs;
            const int q = pat_rep[pid];
            if (row_ptr[q + 1] - row_ptr[q] != len) pid = -1;
            else {
                const int off = pat_off[pid];
                for (int j = 0; j < len; ++j)
                    if (ewords[rs + j] != pwords[off + j]) { pid = -1; break; }
            }
        }
    }
    row_pid[r] = pid;
}

// K6: one workgroup per tile writes everything the SpMV reads for it
__global__ __launch_bounds__(kTileBlk) void k_bk_tile(const TileInfo *__restrict__ tiles, const int32_t *__restrict__ row_ptr,
                                                     const int32_t *__restrict__ col, const uint16_t *__restrict__ eslot,
                                                     const int32_t *__restrict__ row_pid, const uint32_t *__restrict__ rgeo,
                                                     const int32_t *__restrict__ pat_off, const int32_t *__restrict__ pat_rep,
                                                     int n_rows, int has_halo, int pat_words_cap, int vc, const uint8_t *__restrict__ force_e,
                                                     uint32_t *__restrict__ blocks, uint2 *__restrict__ tile_blk, uint2 *__restrict__ rdesc,
                                                     uint2 *__restrict__ sdesc, int32_t *__restrict__ slen, int32_t *__restrict__ tile_info_out,
                                                     uint8_t *__restrict__ tile_bnd_out, int *__restrict__ fallbacks)
{
    __shared__ int smap[4096];                 // slot -> column of the halo fill
    __shared__ int pkey[1024], pidx[1024];     // distinct global pattern ids of the tile (hash set) -> list index
    __shared__ int plist[1024], plen4[1024], pstart[1024]; // ... in list order: id, padded length, first word in the LDS image
    __shared__ unsigned skey[1024];            // execution order sort
    __shared__ int scan[kTileBlk];
    __shared__ int counters[8];                // 0 patterns listed, 1 patterns kept, 2 pattern rows, 3 streamed rows, 4 streamed words, 5 runs
    const int t = blockIdx.x;
    const TileInfo &T = tiles[t];
    const int tid = threadIdx.x;
    const int row0 = T.row0, nrows = T.nrows;
    uint32_t *blk = blocks + (int64_t)t * kBlockStride;
    const bool gtile = T.is_g && !(force_e && force_e[t]);
    for (int i = tid; i < 4096; i += kTileBlk) smap[i] = -1;
    for (int i = tid; i < 1024; i += kTileBlk) { pkey[i] = -1; skey[i] = 0xffffffffu; }
    if (tid < 8) counters[tid] = 0;
    int bnd = 0; // partitioned systems: does any row of the tile read a halo column?
    if (has_halo)
        for (int r = tid; r < nrows; r += kTileBlk)
            for (int e = row_ptr[row0 + r]; e < row_ptr[row0 + r + 1]; ++e) bnd |= col[e] >= n_rows ? 1 : 0;
    bnd = __syncthreads_or(bnd);
    // ---- distinct patterns
    int my_pid[2] = {-1, -1};
    if (gtile)
        for (int k = 0; k < 2; ++k) {
            const int r = tid + k * kTileBlk;
            if (r >= nrows) break;
            const int pid = row_pid[row0 + r];
            my_pid[k] = pid;
            if (pid < 0) continue;
            unsigned h = ((unsigned)pid * 2654435761u) >> 22; // 10 bits
            for (;;) {
                const int old = atomicCAS(&pkey[h], -1, pid);
                if (old == -1) { const int li = atomicAdd(&counters[0], 1); pidx[h] = li; plist[li] = pid; break; }
                if (old == pid) break;
                h = (h + 1) & 1023u;
            }
        }
    __syncthreads();
    const int nlist = counters[0];
    // The list order above is the order of arrival (atomics): make it CANONICAL -- patterns ordered by the first row of the tile that uses
    // them -- so that which patterns are kept, the local pattern numbers, the execution order of the rows and with it the lane that sums
    // a row's share of p.Ap are the same in every build of the same matrix (y never depended on it; the solver's iteration counts did).
    for (int i = tid; i < nlist; i += kTileBlk) pstart[i] = 0x7fffffff;
    __syncthreads();
    if (gtile)
        for (int k = 0; k < 2; ++k) {
            const int pid = my_pid[k];
            if (pid < 0) continue;
            unsigned h = ((unsigned)pid * 2654435761u) >> 22;
            while (pkey[h] != pid) h = (h + 1) & 1023u;
            atomicMin(&pstart[pidx[h]], tid + k * kTileBlk);
        }
    __syncthreads();
    for (int i = tid; i < nlist; i += kTileBlk) {
        const int m = pstart[i];
        int rk = 0;
        for (int j = 0; j < nlist; ++j) rk += pstart[j] < m ? 1 : 0;
        plen4[i] = rk;                                   // canonical index of list entry i
        skey[rk] = (unsigned)plist[i];
    }
    __syncthreads();
    for (int h = tid; h < 1024; h += kTileBlk)
        if (pkey[h] != -1) pidx[h] = plen4[pidx[h]];
    __syncthreads();
    for (int i = tid; i < nlist; i += kTileBlk) plist[i] = (int)skey[i];
    __syncthreads();
    for (int i = tid; i < nlist; i += kTileBlk) skey[i] = 0xffffffffu;
    __syncthreads();
    for (int i = tid; i < nlist; i += kTileBlk) {
        const int q = pat_rep[plist[i]];
        plen4[i] = (row_ptr[q + 1] - row_ptr[q] + 3) & ~3;
    }
    __syncthreads();
    if (tid == 0) { // list order prefix (<= 1024 entries): which patterns fit the LDS image
        int acc = 0, kept = 0;
        for (int i = 0; i < nlist; ++i) {
            if (kept < kBrickPatMax && acc + plen4[i] <= pat_words_cap && i == kept) { pstart[i] = acc; acc += plen4[i]; ++kept; }
            else pstart[i] = -1;
        }
        counters[1] = kept;
        counters[6] = acc; // words
    }
    __syncthreads();
    const int npat = counters[1], npq = counters[6] >> 2;
    // ---- rows: pattern row (local index) or streamed
    int my_li[2] = {-1, -1}, my_len[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {
        const int r = tid + k * kTileBlk;
        if (r >= nrows) break;
        my_len[k] = row_ptr[row0 + r + 1] - row_ptr[row0 + r];
        const int pid = my_pid[k];
        if (pid >= 0) {
            unsigned h = ((unsigned)pid * 2654435761u) >> 22;
            while (pkey[h] != pid) h = (h + 1) & 1023u;
            const int li = pidx[h];
            if (pstart[li] >= 0) my_li[k] = li;
        }
        if (my_li[k] >= 0) {
            atomicAdd(&counters[2], 1);
            skey[r] = ((unsigned)my_len[k] << 20) | ((unsigned)my_li[k] << 10) | (unsigned)r; // len <= 64, li < 512, r < 1024
        }
    }
    __syncthreads();
    const int nprow = counters[2];
    // ---- execution order: bitonic sort of 1024 keys (0xffffffff = not a pattern row, sorts last)
    for (int size = 2; size <= 1024; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < 512; i += kTileBlk) {
                const int lo = ((i / stride) * stride * 2) + (i % stride), hi = lo + stride;
                const bool up = ((lo / size) & 1) == 0;
                const unsigned a = skey[lo], b = skey[hi];
                if ((a > b) == up) { skey[lo] = b; skey[hi] = a; }
            }
            __syncthreads();
        }
    for (int i = tid; i < nprow; i += kTileBlk) {
        const unsigned key = skey[i];
        const int r = (int)(key & 1023u), li = (int)((key >> 10) & 1023u);
        rdesc[row0 + i] = uint2{((uint32_t)li << 20) | (rgeo[row0 + r] & 0x000fffffu), (uint32_t)r};
    }
    // value-code variant: the code stream of the tile is laid out wave by wave of the execution order -- the wave of rows
    // [512 k + 64 w, + 64) owns a block of (quads of its longest row, the last one: the order is ascending in length) x 64 lanes x 8 B
    __shared__ int cblock[16];
    if (vc && tid < 16) {
        const int lo = (tid >> 3) * kTileBlk + (tid & 7) * 64, hi = lo + 64 < nprow ? lo + 64 : nprow;
        cblock[tid] = hi > lo ? (int)(((skey[hi - 1] >> 20) + 3u) >> 2) * 64 : 0;
    }
    __syncthreads();
    // ---- streamed rows: descriptors in row order, words counted (their offsets follow from a scan over all rows: K7)
    int sflag[2], slenv[2];
    for (int k = 0; k < 2; ++k) {
        const int r = tid + k * kTileBlk;
        sflag[k] = (r < nrows && my_li[k] < 0) ? 1 : 0;
        slenv[k] = sflag[k] ? my_len[k] : 0;
        if (r < nrows) slen[row0 + r] = slenv[k];
    }
    // block scan of (count, words) over the rows in row order: rows tid (k = 0) come before rows tid + 512 (k = 1)
    int base_cnt = 0, base_w = 0;
    for (int k = 0; k < 2; ++k) {
        scan[tid] = sflag[k];
        __syncthreads();
        for (int o = 1; o < kTileBlk; o <<= 1) {
            const int v = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const int my_cnt = scan[tid] - sflag[k] + base_cnt, tot_cnt = scan[kTileBlk - 1];
        __syncthreads();
        scan[tid] = slenv[k];
        __syncthreads();
        for (int o = 1; o < kTileBlk; o <<= 1) {
            const int v = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const int my_w = scan[tid] - slenv[k] + base_w, tot_w = scan[kTileBlk - 1];
        __syncthreads();
        if (sflag[k]) sdesc[row0 + my_cnt] = uint2{(uint32_t)(tid + k * kTileBlk) | ((uint32_t)my_len[k] << 16), (uint32_t)my_w};
        base_cnt += tot_cnt;
        base_w += tot_w;
    }
    const int nsrows = base_cnt, nsw = base_w;
    // ---- halo fill: slot -> column of every entry of a pattern row that is not one of the tile's own rows
    if (gtile)
        for (int k = 0; k < 2; ++k) {
            if (my_li[k] < 0) continue;
            const int row = row0 + tid + k * kTileBlk;
            for (int e = row_ptr[row]; e < row_ptr[row + 1]; ++e) {
                const int c = col[e];
                if (c >= row0 && c < row0 + nrows) continue;
                smap[eslot[e]] = c; // (a slot has one column: every writer stores the same value)
            }
        }
    __syncthreads();
    // run starts: slot s filled and not the continuation of the run through s - 1 (consecutive slots with consecutive columns, at most 16 long)
    constexpr int SPT = 4096 / kTileBlk; // 8 consecutive slots per thread
    unsigned startmask = 0;
    {
        const int s0 = tid * SPT;
        for (int u = 0; u < SPT; ++u) {
            const int s = s0 + u, c = smap[s];
            if (c < 0) continue;
            bool st = true;
            if (s > 0) {
                const int cp = smap[s - 1];
                if (cp >= 0 && cp + 1 == c) st = false;
            }
            if (st) startmask |= 1u << u;
        }
    }
    // natural runs are cut every 16 slots: position inside the natural run needs the last natural start at or before s
    __shared__ int laststart[kTileBlk];
    {
        int last = -1;
        for (int u = 0; u < SPT; ++u)
            if (startmask & (1u << u)) last = tid * SPT + u;
        laststart[tid] = last;
        __syncthreads();
        for (int o = 1; o < kTileBlk; o <<= 1) { // inclusive max scan
            const int v = tid >= o ? laststart[tid - o] : -1;
            __syncthreads();
            if (v > laststart[tid]) laststart[tid] = v;
            __syncthreads();
        }
    }
    unsigned cutmask = 0;
    int ncut = 0;
    {
        int last = tid > 0 ? laststart[tid - 1] : -1;
        for (int u = 0; u < SPT; ++u) {
            const int s = tid * SPT + u;
            if (smap[s] < 0) continue;
            if (startmask & (1u << u)) last = s;
            if (((s - last) & (kBrickRunLen - 1)) == 0) { cutmask |= 1u << u; ++ncut; }
        }
    }
    scan[tid] = ncut;
    __syncthreads();
    for (int o = 1; o < kTileBlk; o <<= 1) {
        const int v = tid >= o ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const int nruns = scan[kTileBlk - 1];
    int my_run = scan[tid] - ncut;
    const bool fits = nruns <= kBrickMaxRuns && kBlkHdrWords + 2 * nruns + npq + npat <= kBlockStride;
    __syncthreads();
    if (!fits) { // (block-uniform) a tile over a limit is redone as an E tile
        if (tid == 0) atomicExch(fallbacks, 1);
        if (tid == 0) tile_info_out[t] = -1;
        return;
    }
    if (gtile) {
        for (int u = 0; u < SPT; ++u) {
            if (!(cutmask & (1u << u))) continue;
            const int s = tid * SPT + u;
            int len = 1;
            const int c = smap[s];
            while (len < kBrickRunLen && s + len < 4096 && smap[s + len] == c + len) ++len;
            // (the loop above may run past the next cut of the SAME natural run only if that cut is 16 away: len < 16 stops it)
            // a run is 8 B: the absolute first column | slot << 4 | length - 1 (round 5; it was 4 B naming a neighbour brick and an offset,
            // which cost the kernel a second, dependent LDS read per batch; halo columns of a partitioned system are just columns >= n_rows)
            blk[kBlkHdrWords + 2 * my_run] = (uint32_t)c;
            blk[kBlkHdrWords + 2 * my_run + 1] = ((uint32_t)s << 4) | (uint32_t)(len - 1);
            ++my_run;
        }
        // pattern quads + pinfo in list order
        for (int i = tid; i < npat; i += kTileBlk) {
            const int id = plist[i];
            const uint32_t simple = rgeo[pat_rep[id]] >> 31;
            blk[kBlkHdrWords + 2 * nruns + npq + i] = (uint32_t)pstart[i] | ((uint32_t)(plen4[i] >> 2) << 16) | (simple << 31);
            for (int q = 0; q < (plen4[i] >> 2); ++q) blk[kBlkHdrWords + 2 * nruns + (pstart[i] >> 2) + q] = (uint32_t)(pat_off[id] + 4 * q);
        }
    }
    // header
    if (tid < 16) {
        int v = 0;
        switch (tid) {
        case 0: v = row0; break;
        case 1: v = nrows; break;
        case 2: v = gtile ? npat : 0; break;
        case 3: v = gtile ? nruns : 0; break;
        case 4: v = gtile ? npq : 0; break;
        case 5: v = gtile ? nprow : 0; break;
        case 6: v = row0; break;   // srow0: the tile's streamed-row descriptors start at its first row
        case 7: v = nsrows; break;
        case 8: v = 0; break;      // sword0: K7
        case 9: v = nsw; break;
        case 10: v = row0; break;  // rd0: the tile's pattern-row descriptors start at its first row
        case 11: v = bnd; break;   // the tile's rows read halo columns (partitioned solve, direct transport: wait for the peers' flags)
        case 12:                   // value-code variant: size of the tile's code stream in 8-B quads (K7 turns it into its start)
            if (vc && gtile) for (int i = 0; i < 16; ++i) v += cblock[i];
            break;
        case 13: v = t * kBrickTileStride; break; // ... the tile's value table in ttab
        default: break;            // (14: the table's size, written by k_bk_tile_codes)
        }
        blk[tid] = (uint32_t)v;
    }
    if (tid < 32) { // words 16 .. 31: first quad of every wave's block of the code stream, relative to the tile's start (value-code variant)
        int off = 0;
        if (vc && gtile && tid < 16) for (int i = 0; i < tid; ++i) off += cblock[i];
        blk[16 + tid] = (uint32_t)off;
    }
    if (tid == 0) {
        tile_bnd_out[t] = (uint8_t)bnd;
        const int words = kBlkHdrWords + (gtile ? 2 * nruns + npq + npat : 0);
        tile_blk[t] = uint2{(uint32_t)((int64_t)t * (kBlockStride / 4)), (uint32_t)((words + 3) >> 2)};
        tile_info_out[t] = nprow;
    }
}

// K7: streamed words tile after tile (sstart = exclusive scan of slen over the rows) + the headers' sword0.  A word is the packed
// stream's (code << col_bits | column) when code and column share 32 bits, else 64 bits: column | code << 32 (WIDE: matrices of more
// than 2^(32 - bits(table)) columns -- the 1024^3 thin sheet -- and the unpacked 6-B form).
template <bool WIDE>
__global__ __launch_bounds__(kBlk) void k_bk_copy_streamed(int64_t n, const int32_t *__restrict__ slen, const int32_t *__restrict__ sstart,
                                                          const int32_t *__restrict__ row_ptr, const uint32_t *__restrict__ packed,
                                                          const uint16_t *__restrict__ codes, const int32_t *__restrict__ col,
                                                          uint32_t *__restrict__ swords)
{
    const int sub = threadIdx.x & 15;
    const int64_t groups = ((int64_t)gridDim.x * kBlk) >> 4;
    for (int64_t r = ((int64_t)blockIdx.x * kBlk + threadIdx.x) >> 4; r < n; r += groups) {
        const int len = slen[r];
        if (!len) continue;
        const int src = row_ptr[r], dst = sstart[r];
        for (int j = sub; j < len; j += 16) {
            if (WIDE) reinterpret_cast<uint2 *>(swords)[dst + j] = uint2{(uint32_t)col[src + j], (uint32_t)codes[src + j]};
            else swords[dst + j] = packed[src + j];
        }
    }
}
__global__ __launch_bounds__(kBlk) void k_bk_patch(int ntiles, const TileInfo *__restrict__ tiles, const int32_t *__restrict__ sstart,
                                                  uint32_t *__restrict__ blocks)
{
    const int t = blockIdx.x * kBlk + threadIdx.x;
    if (t >= ntiles) return;
    blocks[(int64_t)t * kBlockStride + 8] = (uint32_t)sstart[tiles[t].row0];
}
// ---- value-code variant (variable viscosity) -------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned vhash(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
    return (unsigned)k & 1023u;
}
constexpr unsigned long long kNoValue = 0xFFFFFFFFFFFFFFFFull; // (a NaN pattern: never a matrix value)
// inserts `key` into the 1024-slot set; false when the set is full
__device__ __forceinline__ bool vset_insert(unsigned long long *set, unsigned long long key)
{
    unsigned h = vhash(key);
    for (int probe = 0; probe < 1024; ++probe) {
        const unsigned long long old = atomicCAS(&set[h], kNoValue, key);
        if (old == kNoValue || old == key) return true;
        h = (h + 1) & 1023u;
    }
    return false;
}
// a G tile whose rows hold more distinct values than a tile's LDS table takes becomes an E tile (counted over ALL its rows: the pattern
// rows are not known yet)
__global__ __launch_bounds__(kTileBlk) void k_bk_tile_valcount(const TileInfo *__restrict__ tiles, const int32_t *__restrict__ row_ptr,
                                                              const double *__restrict__ val, uint8_t *__restrict__ force_e)
{
    __shared__ unsigned long long set[1024];
    __shared__ int count, full;
    const TileInfo &T = tiles[blockIdx.x];
    if (!T.is_g) return;
    const int tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
    for (int i = tid; i < 1024; i += kTileBlk) set[i] = kNoValue;
    if (tid == 0) { count = 0; full = 0; }
    __syncthreads();
    for (int r = grp; r < T.nrows; r += kTileBlk / 16)
        for (int k = row_ptr[T.row0 + r] + sub; k < row_ptr[T.row0 + r + 1]; k += 16)
            if (!vset_insert(set, (unsigned long long)__double_as_longlong(val[k]))) full = 1;
    __syncthreads();
    int c = 0;
    for (int i = tid; i < 1024; i += kTileBlk) c += set[i] != kNoValue ? 1 : 0;
    atomicAdd(&count, c);
    __syncthreads();
    if (tid == 0 && (full || count > kBrickTileVals)) force_e[blockIdx.x] = 1;
}
// per G tile, after K6: the value table of its pattern rows (ttab, entry ntv = nothing: the kernel puts 0.0 there) and the rows' codes
// (byte offsets into the table) at their places in the code stream
__global__ __launch_bounds__(kTileBlk) void k_bk_tile_codes(int ntiles, const uint2 *__restrict__ tile_blk, uint32_t *__restrict__ blocks,
                                                           const int32_t *__restrict__ row_ptr, const double *__restrict__ val,
                                                           const uint2 *__restrict__ rdesc, double *__restrict__ ttab, uint16_t *__restrict__ vcodes,
                                                           int *__restrict__ overflow)
{
    __shared__ unsigned long long set[1024];
    __shared__ int code_of[1024];
    __shared__ int scan[kTileBlk];
    const int t = blockIdx.x, tid = threadIdx.x;
    uint32_t *bw = blocks + (int64_t)tile_blk[t].x * 4;
    const int row0 = (int)bw[0], npat = (int)bw[2], nprow = (int)bw[5], rd0 = (int)bw[10];
    const int64_t cw0 = (int64_t)bw[12];
    if (npat == 0 || nprow == 0) { if (tid == 0) bw[14] = 0u; return; }
    for (int i = tid; i < 1024; i += kTileBlk) set[i] = kNoValue;
    __syncthreads();
    for (int i = tid; i < nprow; i += kTileBlk) {
        const int r = row0 + (int)rdesc[rd0 + i].y;
        for (int k = row_ptr[r]; k < row_ptr[r + 1]; ++k)
            if (!vset_insert(set, (unsigned long long)__double_as_longlong(val[k]))) atomicExch(overflow, 1);
    }
    __syncthreads();
    // slot -> code: exclusive count of the occupied slots before it (two slots per thread)
    const int o0 = set[2 * tid] != kNoValue ? 1 : 0, o1 = set[2 * tid + 1] != kNoValue ? 1 : 0;
    scan[tid] = o0 + o1;
    __syncthreads();
    for (int o = 1; o < kTileBlk; o <<= 1) {
        const int v = tid >= o ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const int before = scan[tid] - o0 - o1, ntv = scan[kTileBlk - 1];
    code_of[2 * tid] = before;
    code_of[2 * tid + 1] = before + o0;
    double *tab = ttab + (int64_t)t * kBrickTileStride;
    if (ntv <= kBrickTileVals) {
        if (o0) tab[before] = __longlong_as_double((long long)set[2 * tid]);
        if (o1) tab[before + o0] = __longlong_as_double((long long)set[2 * tid + 1]);
    } else if (tid == 0) {
        atomicExch(overflow, 1); // (cannot happen: k_bk_tile_valcount counted a superset)
    }
    if (tid == 0) {
        bw[14] = (uint32_t)ntv;
        atomicAdd(overflow + 1, ntv);                        // (total table entries: what the form stores)
    }
    __syncthreads();
    const unsigned pad = (unsigned)(ntv * 8);                // the table's 0.0 entry
    for (int i = tid; i < nprow; i += kTileBlk) {
        const int r = row0 + (int)rdesc[rd0 + i].y;
        const int rs = row_ptr[r], len = row_ptr[r + 1] - rs;
        const int kw = (i / kTileBlk) * (kTileBlk / 64) + ((i % kTileBlk) >> 6), lane = i & 63;
        uint16_t *dst = vcodes + 4 * (cw0 + (int64_t)bw[16 + kw] + lane);   // quad q of this row at + 4 * 64 q
        const int nq = (len + 3) >> 2;
        for (int j = 0; j < 4 * nq; ++j) {
            unsigned code = pad;
            if (j < len) {
                const unsigned long long key = (unsigned long long)__double_as_longlong(val[rs + j]);
                unsigned h = vhash(key);
                while (set[h] != key) h = (h + 1) & 1023u;
                code = (unsigned)code_of[h] * 8u;
            }
            dst[(int64_t)(j >> 2) * 256 + (j & 3)] = (uint16_t)code;
        }
    }
}
// streamed rows of the value-code variant: 12 B per entry, column | value
__global__ __launch_bounds__(kBlk) void k_bk_copy_streamed_vc(int64_t n, const int32_t *__restrict__ slen, const int32_t *__restrict__ sstart,
                                                             const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                             const double *__restrict__ val, uint32_t *__restrict__ swords)
{
    const int sub = threadIdx.x & 15;
    const int64_t groups = ((int64_t)gridDim.x * kBlk) >> 4;
    for (int64_t r = ((int64_t)blockIdx.x * kBlk + threadIdx.x) >> 4; r < n; r += groups) {
        const int len = slen[r];
        if (!len) continue;
        const int src = row_ptr[r], dst = sstart[r];
        for (int j = sub; j < len; j += 16) {
            const long long v = __double_as_longlong(val[src + j]);
            swords[3 * (int64_t)(dst + j)] = (uint32_t)col[src + j];
            swords[3 * (int64_t)(dst + j) + 1] = (uint32_t)(v & 0xffffffffll);
            swords[3 * (int64_t)(dst + j) + 2] = (uint32_t)((unsigned long long)v >> 32);
        }
    }
}
__global__ __launch_bounds__(kBlk) void k_bk_code_sizes(int ntiles, const uint32_t *__restrict__ blocks, int32_t *__restrict__ csize)
{
    const int t = blockIdx.x * kBlk + threadIdx.x;
    if (t < ntiles) csize[t] = (int32_t)blocks[(int64_t)t * kBlockStride + 12];
}
__global__ __launch_bounds__(kBlk) void k_bk_patch_codes(int ntiles, const int32_t *__restrict__ cstart, uint32_t *__restrict__ blocks)
{
    const int t = blockIdx.x * kBlk + threadIdx.x;
    if (t < ntiles) blocks[(int64_t)t * kBlockStride + 12] = (uint32_t)cstart[t];
}

// what the cost model of the planned walk reads of a tile: the counts in its block header
__global__ __launch_bounds__(kBlk) void k_bk_tile_features(int ntiles, const uint2 *__restrict__ tile_blk, const uint32_t *__restrict__ blocks,
                                                          int4 *__restrict__ feat)
{
    const int t = blockIdx.x * kBlk + threadIdx.x;
    if (t >= ntiles) return;
    const uint32_t *bw = blocks + (int64_t)tile_blk[t].x * 4;
    feat[2 * t] = int4{(int)bw[1], (int)bw[3], (int)bw[5], (int)bw[9]};      // rows, fill runs, pattern rows, streamed words
    feat[2 * t + 1] = int4{(int)bw[2], (int)bw[4], (int)bw[7], 0};            // patterns, pattern quads, streamed rows
}
__global__ __launch_bounds__(kBlk) void k_bk_count_regular(int ntiles, const int32_t *__restrict__ tile_nprow, unsigned long long *__restrict__ total)
{
    const int t = blockIdx.x * kBlk + threadIdx.x;
    if (t < ntiles && tile_nprow[t] > 0) atomicAdd(total, (unsigned long long)tile_nprow[t]);
}
} // namespace

// pattern word for 8-B elements (delta << 19 | level << 14 | code << 3: the delta and the code as byte offsets) -> for 4-B elements
// (delta << 18 | level << 14 | code << 2)
__global__ void k_bk_pwords32(int n, const uint32_t *__restrict__ w8, uint32_t *__restrict__ w4)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = w8[i];
    w4[i] = ((uint32_t)((int32_t)w >> 1) & 0xffff0000u) | (w & 0xc000u) | ((w & 0x3ff8u) >> 1);
}

void BrickForm::clear()
{
    ready = false;
    ntiles = 0;
    regular_rows = 0;
    patterns = 0;
    wgrid = 0;
    vc = false;
}

void BrickScratch::release()
{
    geo.release(); row_hash.release();
    first.release(); bidx.release(); scan_tmp.release(); bstart.release(); bbrick.release(); run_first.release(); run_id.release();
    run_start.release(); tcount.release(); tile0.release(); rep.release(); slot_id.release(); pat_rep.release(); pat_off.release();
    row_pid.release(); slen.release(); sstart.release(); tile_nprow.release();
    ewords.release(); rgeo.release(); eslot.release(); keys.release(); total.release(); counters.release(); tiles.release(); force_e.release();
}

void BrickForm::release()
{
    clear();
    tile_blk.release(); rdesc.release(); sdesc.release(); blocks.release(); pwords.release(); pwords32.release(); swords.release(); ownslot.release(); tile_flags.release(); wlist.release(); wptr.release(); vcodes.release(); ttab.release();
    scratch.release();
}

// The persistent grid's tile walk, planned: the static strided walks end with their slowest workgroup 15-20 % after the average one
// (profiles/r04_notes.md: E tiles cost two G tiles, G tiles differ by their rows; a dynamic queue was slower).  Here every workgroup's
// sequence is fixed in advance by list scheduling on ESTIMATED tile costs: the tiles of an XCD, in their order (neighbouring bricks stay
// neighbours in time: a halo value is fetched by one L2), each go to the workgroup of that XCD that has the least work so far.
// Deterministic (a function of the form and the grid), so the fold order of p.Ap and with it the iteration count are reproducible.
avs_status BrickForm::plan_walk(int grid, int xcd_mode, const BrickCost &cm, hipStream_t st)
{
    wgrid = 0;
    plan_makespan = plan_mean = 0.;
    if (!ready || ntiles <= 0 || grid <= 0 || (grid & 7) != 0 || ntiles < grid) return AVS_OK;
    DevBuf<int4> dfeat;
    AVS_TRY(dfeat.alloc((size_t)ntiles * 2));
    hipLaunchKernelGGL(k_bk_tile_features, dim3((unsigned)((ntiles + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, ntiles, (const uint2 *)tile_blk.p,
                       (const uint32_t *)blocks.p, dfeat.p);
    std::vector<int4> feat((size_t)ntiles * 2);
    std::vector<uint2> tb((size_t)ntiles);
    AVS_HIP(hipMemcpyAsync(feat.data(), dfeat.p, feat.size() * sizeof(int4), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipMemcpyAsync(tb.data(), tile_blk.p, tb.size() * sizeof(uint2), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    std::vector<double> cost((size_t)ntiles);
    double total = 0.;
    for (int t = 0; t < ntiles; ++t) {
        const int4 a = feat[2 * (size_t)t], b = feat[2 * (size_t)t + 1];
        const bool etile = b.x == 0; // no patterns: packed words only
        const double c = cm.tile + cm.row * a.z + cm.run * a.y + cm.word * a.w + cm.quad * b.y + (etile ? cm.etile : 0.);
        cost[(size_t)t] = c;
        total += c;
    }
    // tiles of every XCD, in walk order
    const int gx = grid >> 3;
    std::vector<std::vector<int>> xt(8);
    if (xcd_mode == 0) { // contiguous ranges of equal COST
        double acc = 0.;
        int c = 0;
        for (int t = 0; t < ntiles; ++t) {
            while (c < 7 && acc >= total * (c + 1) / 8.) ++c;
            xt[(size_t)c].push_back(t);
            acc += cost[(size_t)t];
        }
    } else {             // chunks of gx tiles dealt to the XCDs in turn
        for (int t = 0; t < ntiles; ++t) xt[(size_t)((t / gx) & 7)].push_back(t);
    }
    std::vector<std::vector<int>> seq((size_t)grid);
    std::vector<double> load((size_t)grid, 0.);
    for (int c = 0; c < 8; ++c) {
        // a binary heap over the XCD's workgroups keyed by (work so far, workgroup): ties go to the lower workgroup -- deterministic
        std::vector<std::pair<double, int>> heap;
        for (int j = 0; j < gx; ++j) heap.emplace_back(0., c + 8 * j);
        auto cmp = [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a > b; };
        std::make_heap(heap.begin(), heap.end(), cmp);
        for (int t : xt[(size_t)c]) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            auto &w = heap.back();
            seq[(size_t)w.second].push_back(t);
            w.first += cost[(size_t)t];
            load[(size_t)w.second] = w.first;
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
    }
    std::vector<int32_t> hptr((size_t)grid + 1, 0);
    std::vector<uint2> hlist;
    hlist.reserve((size_t)ntiles);
    for (int b = 0; b < grid; ++b) {
        if (seq[(size_t)b].empty()) return AVS_OK; // (an XCD with fewer tiles than workgroups: the strided walk serves such a matrix)
        for (int t : seq[(size_t)b]) hlist.push_back(tb[(size_t)t]);
        hptr[(size_t)b + 1] = (int32_t)hlist.size();
        plan_makespan = std::max(plan_makespan, load[(size_t)b]);
    }
    plan_mean = total / grid;
    AVS_TRY(wlist.alloc(hlist.size() + 1));
    AVS_TRY(wptr.alloc(hptr.size()));
    AVS_HIP(hipMemcpyAsync(wlist.p, hlist.data(), hlist.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
    AVS_HIP(hipMemcpyAsync(wptr.p, hptr.data(), hptr.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    AVS_HIP(hipStreamSynchronize(st)); // (the host vectors die here)
    wgrid = grid;
    return AVS_OK;
}

int64_t BrickForm::stored_bytes(int64_t n) const
{
    // descriptor blocks (used words), row descriptors 8 B, own slots 2 B per row, streamed words + descriptors, pattern table, tile list
    return 4 * block_words + 8 * regular_rows + 2 * n + (vc ? 12 : (wide ? 8 : 4)) * streamed_words + 8 * streamed_rows + 4 * pattern_words + 8 * (int64_t)ntiles +
           (vc ? 8 * code_quads + 8 * table_values : 0);
}

void BrickForm::view(BrickView &B, const ValueIndex &vi) const
{
    B = BrickView();
    if (!ready) return;
    B.ntiles = ntiles;
    B.tile_blk = tile_blk.p;
    B.blocks = blocks.p;
    B.rdesc = rdesc.p;
    B.ownslot = ownslot.p;
    B.pwords = pwords.p;
    B.pwords32 = pwords32.p;
    B.sdesc = sdesc.p;
    B.swords = swords.p;
    B.table = vi.table.p;
    B.table_size = vi.table_size;
    B.col_bits = wide ? 0 : vi.col_bits;   // 0: 64-bit streamed words
    if (vc) { B.vc = 1; B.vcodes = vcodes.p; B.ttab = ttab.p; B.table = nullptr; B.table_size = kBrickTileVals; }
    B.n_rows = (int)n_rows;
    if (wgrid > 0) { B.wlist = wlist.p; B.wptr = wptr.p; B.wgrid = wgrid; }
}

// BrickSource -> bf.  Leaves bf.ready = false (and AVS_OK) when the matrix does not qualify or is not regular enough; an error status
// only for real failures.
avs_status build_brick_form(BrickForm &bf, const BrickSource &src, const Options &opt, hipStream_t st)
{
    bf.clear();
    const int64_t n = src.n_rows, nnz = src.nnz, n_cols = src.n_cols;
    if (n <= 0 || !src.vi || !src.row_ptr || !src.col || !src.ref_id || !src.vdof) return AVS_OK;
    const ValueIndex &vi = *src.vi;
    // One dictionary of few values (uniform viscosity): the patterns carry value codes into it.  Anything else -- tile-local dictionaries,
    // thousands of values, no dictionary at all: spatially varying viscosity, cpp:2148-2150 -- takes the VALUE-CODE variant (round 5): the
    // patterns carry the geometry only, every tile gets its own table of <= kBrickTileVals values in LDS and every pattern row a stream of
    // 2-B codes into it.
    bool vc = vi.tile_tables || !vi.codes.p || vi.table_size <= 0 || vi.table_size + 1 >= kBrickTableMax;
    if (!vc) {
        BrickView probe;
        probe.table_size = vi.table_size;
        if (!brick_lds_fits(probe)) vc = true; // the one table next to the lattice would exceed a workgroup's LDS
    }
    if (vc && (!src.val || !opt.brick_value_codes)) return AVS_OK;
    const bool wide = vc || vi.col_bits <= 0 || vi.col_windows; // no (code << col_bits | column) stream to copy the streamed rows from
    if (src.brick_shift != 3 || src.levels < 1) return AVS_OK;
    if (src.nx > 1024 || src.ny > 1024 || src.nz > 1024 || nnz >= (1ll << 31) || n_cols >= (1ll << 31) - 64) return AVS_OK;
    const bool timing = opt.brick_timing != 0;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "brick build: %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    const int nx = src.nx, ny = src.ny, nz = src.nz;
    const int nbx = (nx + 7) >> 3, nby = (ny + 7) >> 3;
    BrickScratch &S = bf.scratch;
    AVS_TRY(S.geo.reserve((size_t)n_cols));
    AVS_TRY(S.counters.reserve(16));
    AVS_HIP(hipMemsetAsync(S.counters.p, 0, 16 * sizeof(int), st));
    AVS_TRY(S.first.reserve((size_t)n + 1));
    AVS_TRY(S.bidx.reserve((size_t)n + 1));
    AVS_TRY(S.scan_tmp.reserve(scan_tmp_elems(n)));
    const unsigned gn = (unsigned)((n + kBlk - 1) / kBlk);
    hipLaunchKernelGGL(k_bk_geo, dim3((unsigned)((n_cols + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, src.vdof, src.ref_id, n_cols, nx, ny, nz, S.geo.p);
    hipLaunchKernelGGL(k_bk_first, dim3(gn), dim3(kBlk), 0, st, S.geo.p, n, S.first.p);
    AVS_TRY(exclusive_scan_i32(S.first.p, S.bidx.p, n, S.scan_tmp.p, S.scan_tmp.n, st));
    int32_t nbricks = 0;
    AVS_HIP(hipMemcpyAsync(&nbricks, S.bidx.p + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    AVS_TRY(S.bstart.reserve((size_t)nbricks + 2));
    AVS_TRY(S.bbrick.reserve((size_t)nbricks + 2));
    AVS_TRY(S.run_first.reserve((size_t)nbricks + 2));
    AVS_TRY(S.run_id.reserve((size_t)nbricks + 2));
    AVS_TRY(S.run_start.reserve((size_t)nbricks + 2));
    AVS_TRY(S.tcount.reserve((size_t)nbricks + 2));
    AVS_TRY(S.tile0.reserve((size_t)nbricks + 2));
    hipLaunchKernelGGL(k_bk_brick_starts, dim3(gn), dim3(kBlk), 0, st, S.geo.p, S.first.p, S.bidx.p, n, S.bstart.p, S.bbrick.p);
    {
        const int32_t n32 = (int32_t)n;
        AVS_HIP(hipMemcpyAsync(S.bstart.p + nbricks, &n32, sizeof(int32_t), hipMemcpyHostToDevice, st)); // (pageable source: copied at the call)
    }
    const unsigned gb = (unsigned)((nbricks + kBlk - 1) / kBlk);
    hipLaunchKernelGGL(k_bk_run_first, dim3(gb), dim3(kBlk), 0, st, S.bstart.p, nbricks, S.run_first.p);
    AVS_TRY(exclusive_scan_i32(S.run_first.p, S.run_id.p, nbricks, S.scan_tmp.p, S.scan_tmp.n, st));
    hipLaunchKernelGGL(k_bk_run_starts, dim3(gb), dim3(kBlk), 0, st, S.bstart.p, S.run_first.p, S.run_id.p, nbricks, S.run_start.p);
    hipLaunchKernelGGL(k_bk_tile_counts, dim3(gb), dim3(kBlk), 0, st, S.bstart.p, S.run_first.p, S.run_id.p, S.run_start.p, nbricks, S.tcount.p, S.counters.p + 3);
    AVS_TRY(exclusive_scan_i32(S.tcount.p, S.tile0.p, nbricks, S.scan_tmp.p, S.scan_tmp.n, st));
    int32_t ntiles = 0;
    AVS_HIP(hipMemcpyAsync(&ntiles, S.tile0.p + nbricks, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    lap("geometry + brick starts");
    if (ntiles <= 0) return AVS_OK;
    AVS_TRY(S.tiles.reserve((size_t)ntiles * sizeof(TileInfo)));
    TileInfo *wtiles = reinterpret_cast<TileInfo *>(S.tiles.p);
    hipLaunchKernelGGL(k_bk_tile_rows, dim3(gb), dim3(kBlk), 0, st, S.bstart.p, S.bbrick.p, S.run_first.p, S.run_id.p, S.run_start.p, S.tile0.p, nbricks,
                       nbx, nby, wtiles);
    hipLaunchKernelGGL(k_bk_tile_finish, dim3((unsigned)((ntiles + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, ntiles, n, wtiles);
    const TileInfo *dtiles = reinterpret_cast<const TileInfo *>(S.tiles.p);
    lap("tiles");
    // ---- K3
    AVS_TRY(S.ewords.reserve((size_t)nnz));
    AVS_TRY(S.eslot.reserve((size_t)nnz));
    AVS_TRY(S.row_hash.reserve((size_t)n));
    AVS_TRY(S.rgeo.reserve((size_t)n));
    AVS_TRY(bf.ownslot.alloc((size_t)n + 8));
    const int zero_code = vc ? 0 : vi.table_size;
    // ---- K4
    const size_t hslots = (size_t)1 << kHashBitsPat;
    AVS_TRY(S.keys.reserve(hslots));
    AVS_TRY(S.rep.reserve(hslots));
    AVS_TRY(S.slot_id.reserve(hslots));
    AVS_TRY(S.pat_rep.reserve((size_t)kMaxPatterns));
    AVS_TRY(S.pat_off.reserve((size_t)kMaxPatterns));
    AVS_HIP(hipMemsetAsync(S.keys.p, 0, hslots * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_bk_rows, dim3(ntiles), dim3(kTileBlk), 0, st, dtiles, src.row_ptr, src.col, vc ? (const uint16_t *)nullptr : (const uint16_t *)vi.codes.p, S.geo.p, nbx, nby, zero_code,
                       S.ewords.p, S.eslot.p, S.row_hash.p, S.rgeo.p, bf.ownslot.p, S.keys.p, S.rep.p, S.counters.p + 2);
    lap("K3 rows + pattern insert");
    hipLaunchKernelGGL(k_bk_assign, dim3((unsigned)(hslots / kBlk)), dim3(kBlk), 0, st, S.keys.p, S.rep.p, src.row_ptr, S.slot_id.p, S.pat_rep.p,
                       S.pat_off.p, S.counters.p);
    int hc[4] = {};
    AVS_HIP(hipMemcpyAsync(hc, S.counters.p, sizeof(hc), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    const int npat = hc[0], nwords = hc[1];
    lap("K4 assign");
    if (npat <= 0 || npat > kMaxPatterns || hc[3]) return AVS_OK; // nothing regular / not a regular scene
    AVS_TRY(bf.pwords.alloc((size_t)nwords + 16));
    hipLaunchKernelGGL(k_bk_copy_patterns, dim3((unsigned)(((size_t)npat * 16 + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, npat, S.pat_rep.p, S.pat_off.p,
                       src.row_ptr, S.ewords.p, zero_code, bf.pwords.p);
    // ---- K5
    AVS_TRY(S.row_pid.reserve((size_t)n));
    hipLaunchKernelGGL(k_bk_verify, dim3(gn), dim3(kBlk), 0, st, S.row_hash.p, n, S.keys.p, S.slot_id.p, S.pat_off.p, S.pat_rep.p, src.row_ptr,
                       S.ewords.p, bf.pwords.p, S.row_pid.p);
    lap("K4c copy + K5 verify");
    // ---- K6 (+ a second round for the tiles that exceeded a limit: they become E tiles)
    AVS_TRY(bf.blocks.alloc((size_t)ntiles * kBlockStride + 64));
    AVS_TRY(bf.tile_blk.alloc((size_t)ntiles + 1));
    AVS_TRY(bf.rdesc.alloc((size_t)n + 8));
    AVS_TRY(bf.sdesc.alloc((size_t)n + 8));
    AVS_TRY(S.slen.reserve((size_t)n + 1));
    AVS_TRY(S.sstart.reserve((size_t)n + 1));
    AVS_TRY(S.tile_nprow.reserve((size_t)ntiles));
    AVS_TRY(S.force_e.reserve((size_t)ntiles));
    AVS_TRY(bf.tile_flags.alloc((size_t)ntiles + 1));
    AVS_HIP(hipMemsetAsync(S.force_e.p, 0, (size_t)ntiles, st));
    if (vc) hipLaunchKernelGGL(k_bk_tile_valcount, dim3(ntiles), dim3(kTileBlk), 0, st, dtiles, src.row_ptr, src.val, S.force_e.p);
    for (int round = 0; round < 2; ++round) {
        AVS_HIP(hipMemsetAsync(S.counters.p + 4, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_bk_tile, dim3(ntiles), dim3(kTileBlk), 0, st, dtiles, src.row_ptr, src.col, S.eslot.p, S.row_pid.p, S.rgeo.p,
                           S.pat_off.p, S.pat_rep.p, (int)n, n_cols > n ? 1 : 0, vc ? kBrickPatWordsVc : kBrickPatWords, vc ? 1 : 0, S.force_e.p, bf.blocks.p, bf.tile_blk.p, bf.rdesc.p, bf.sdesc.p, S.slen.p,
                           S.tile_nprow.p, bf.tile_flags.p, S.counters.p + 4);
        int fb = 0;
        AVS_HIP(hipMemcpyAsync(&fb, S.counters.p + 4, sizeof(int), hipMemcpyDeviceToHost, st));
        AVS_HIP(hipStreamSynchronize(st));
        if (!fb) break;
        AVS_REQUIRE(round == 0, AVS_EINTERNAL, "brick form: a tile exceeds its limits as an E tile");
        std::vector<int32_t> hn((size_t)ntiles);
        std::vector<uint8_t> hf((size_t)ntiles, 0);
        AVS_HIP(hipMemcpy(hn.data(), S.tile_nprow.p, (size_t)ntiles * sizeof(int32_t), hipMemcpyDeviceToHost));
        std::vector<uint8_t> he((size_t)ntiles, 0);
        AVS_HIP(hipMemcpy(he.data(), S.force_e.p, (size_t)ntiles, hipMemcpyDeviceToHost)); // (tiles already forced: too many values for a tile's table)
        for (int t = 0; t < ntiles; ++t) hf[(size_t)t] = (hn[(size_t)t] < 0 || he[(size_t)t]) ? 1 : 0;
        AVS_HIP(hipMemcpy(S.force_e.p, hf.data(), (size_t)ntiles, hipMemcpyHostToDevice));
    }
    lap("K6 tiles");
    // ---- K7
    AVS_TRY(exclusive_scan_i32(S.slen.p, S.sstart.p, n, S.scan_tmp.p, S.scan_tmp.n, st));
    AVS_TRY(S.total.reserve(1));
    AVS_HIP(hipMemsetAsync(S.total.p, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_bk_count_regular, dim3((unsigned)((ntiles + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, ntiles, S.tile_nprow.p, S.total.p);
    int32_t total_sw = 0;
    unsigned long long regular = 0;
    AVS_HIP(hipMemcpyAsync(&total_sw, S.sstart.p + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipMemcpyAsync(&regular, S.total.p, sizeof(regular), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    bf.wide = wide;
    bf.vc = vc;
    bf.code_quads = 0;
    bf.table_values = 0;
    AVS_TRY(bf.swords.alloc(((size_t)total_sw + 16) * (vc ? 3 : (wide ? 2 : 1))));
    if (vc) {
        hipLaunchKernelGGL(k_bk_copy_streamed_vc, dim3(4096), dim3(kBlk), 0, st, n, S.slen.p, S.sstart.p, src.row_ptr, src.col, src.val, bf.swords.p);
        // the code stream: per-tile sizes (header word 12) -> starts, then tables and codes tile by tile
        AVS_TRY(S.tcount.reserve((size_t)ntiles + 2));
        AVS_TRY(S.tile0.reserve((size_t)ntiles + 2));
        const unsigned gt = (unsigned)((ntiles + kBlk - 1) / kBlk);
        hipLaunchKernelGGL(k_bk_code_sizes, dim3(gt), dim3(kBlk), 0, st, ntiles, (const uint32_t *)bf.blocks.p, S.tcount.p);
        AVS_TRY(exclusive_scan_i32(S.tcount.p, S.tile0.p, ntiles, S.scan_tmp.p, S.scan_tmp.n, st));
        int32_t total_q = 0;
        AVS_HIP(hipMemcpyAsync(&total_q, S.tile0.p + ntiles, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        AVS_HIP(hipStreamSynchronize(st));
        AVS_REQUIRE(total_q >= 0, AVS_EINTERNAL, "brick form: code stream offsets overflow");
        bf.code_quads = total_q;
        AVS_TRY(bf.vcodes.alloc((size_t)total_q + 4096));
        AVS_TRY(bf.ttab.alloc((size_t)ntiles * kBrickTileStride + 8));
        AVS_HIP(hipMemsetAsync(bf.vcodes.p, 0, ((size_t)total_q + 4096) * sizeof(uint2), st));
        hipLaunchKernelGGL(k_bk_patch_codes, dim3(gt), dim3(kBlk), 0, st, ntiles, (const int32_t *)S.tile0.p, bf.blocks.p);
        AVS_HIP(hipMemsetAsync(S.counters.p + 5, 0, 2 * sizeof(int), st));
        hipLaunchKernelGGL(k_bk_tile_codes, dim3(ntiles), dim3(kTileBlk), 0, st, ntiles, (const uint2 *)bf.tile_blk.p, bf.blocks.p, src.row_ptr, src.val,
                           (const uint2 *)bf.rdesc.p, bf.ttab.p, reinterpret_cast<uint16_t *>(bf.vcodes.p), S.counters.p + 5);
        int ovf[2] = {0, 0};
        AVS_HIP(hipMemcpyAsync(ovf, S.counters.p + 5, sizeof(ovf), hipMemcpyDeviceToHost, st));
        AVS_HIP(hipStreamSynchronize(st));
        AVS_REQUIRE(!ovf[0], AVS_EINTERNAL, "brick form: a tile's value table overflowed");
        bf.table_values = ovf[1];
    } else if (wide)
        hipLaunchKernelGGL(k_bk_copy_streamed<true>, dim3(4096), dim3(kBlk), 0, st, n, S.slen.p, S.sstart.p, src.row_ptr, (const uint32_t *)nullptr,
                           vi.codes.p, src.col, bf.swords.p);
    else
        hipLaunchKernelGGL(k_bk_copy_streamed<false>, dim3(4096), dim3(kBlk), 0, st, n, S.slen.p, S.sstart.p, src.row_ptr, vi.packed.p,
                           (const uint16_t *)nullptr, (const int32_t *)nullptr, bf.swords.p);
    hipLaunchKernelGGL(k_bk_patch, dim3((unsigned)((ntiles + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, ntiles, dtiles, S.sstart.p, bf.blocks.p);
    AVS_HIP(hipGetLastError());
    lap("K7 streamed");
    bf.ntiles = ntiles;
    bf.regular_rows = (int64_t)regular;
    bf.patterns = npat;
    bf.streamed_words = total_sw;
    bf.pattern_words = nwords;
    bf.streamed_rows = n - (int64_t)regular;
    bf.n_rows = n;
    bf.halo_tiles = 0;
    {
        std::vector<uint2> hb((size_t)ntiles);
        AVS_HIP(hipMemcpy(hb.data(), bf.tile_blk.p, (size_t)ntiles * sizeof(uint2), hipMemcpyDeviceToHost));
        int64_t w = 0;
        for (const uint2 &b : hb) w += 4 * (int64_t)b.y;
        bf.block_words = w;
        if (n_cols > n) {
            // partitioned system: the tiles whose rows read halo columns go to the END of the walk order (a stable partition of the
            // tile list -- a tile is named by its descriptor block alone), so that a persistent workgroup multiplies all its other
            // tiles while the peers' entries travel and meets a flag wait, if at all, last
            std::vector<uint8_t> hf((size_t)ntiles), hf2((size_t)ntiles);
            AVS_HIP(hipMemcpy(hf.data(), bf.tile_flags.p, (size_t)ntiles, hipMemcpyDeviceToHost));
            std::vector<uint2> hb2;
            hb2.reserve((size_t)ntiles);
            for (int pass = 0; pass < 2; ++pass)
                for (int t = 0; t < ntiles; ++t)
                    if ((hf[(size_t)t] != 0) == (pass == 1)) { hf2[hb2.size()] = hf[(size_t)t]; hb2.push_back(hb[(size_t)t]); }
            for (uint8_t f : hf) bf.halo_tiles += f ? 1 : 0;
            AVS_HIP(hipMemcpy(bf.tile_blk.p, hb2.data(), (size_t)ntiles * sizeof(uint2), hipMemcpyHostToDevice));
            AVS_HIP(hipMemcpy(bf.tile_flags.p, hf2.data(), (size_t)ntiles, hipMemcpyHostToDevice));
        }
    }
    // the pattern table once more with byte offsets for 4-B elements (the float-vector kernel of AVS_PRECISION_F32: a few thousand words)
    AVS_TRY(bf.pwords32.alloc((size_t)nwords + 16));
    hipLaunchKernelGGL(k_bk_pwords32, dim3((unsigned)((nwords + 16 + kBlk - 1) / kBlk)), dim3(kBlk), 0, st, nwords + 16, bf.pwords.p, bf.pwords32.p);
    AVS_HIP(hipGetLastError());
    // worth it only where most rows are patterns (a curved surface with ~10^4 distinct values gives every row its own)
    // (value-code variant, round 6: a streamed row costs it 12 B per entry and a tile's table is small -- on a curved surface, 72 % pattern rows
    //  and 10^5 patterns, it LOST to the word stream, 3,138 against 4,869 it/s on the 512^3 sphere, which the round-5 rule let through;
    //  smoothly varying viscosity has 98 %: AUTO asks the variant for kBrickMinRegularVc)
    const double min_frac = (bf.vc && opt.brick != 1) ? std::max(opt.brick_min_regular, kBrickMinRegularVc) : opt.brick_min_regular;
    bf.ready = (double)regular >= min_frac * (double)n;
    return AVS_OK;
}

// the single-GPU system of a context: c->p_row_ptr / p_col / vi (codes, packed) / perm / vdof -> c->brick
avs_status build_brick_form(avs_ctx *c)
{
    c->brick.clear();
    if (!c->reordered) return AVS_OK;
    BrickSource src;
    src.n_rows = src.n_cols = c->n_vel;
    src.nnz = c->nnz;
    src.row_ptr = c->p_row_ptr.p;
    src.col = c->p_col.p;
    src.vi = &c->vi;
    src.val = c->p_val.p;
    src.vdof = c->vdof.p;
    src.ref_id = c->perm.p;
    src.nx = c->desc.nx; src.ny = c->desc.ny; src.nz = c->desc.nz;
    src.levels = c->desc.levels;
    src.brick_shift = c->brick_shift;
    return build_brick_form(c->brick, src, c->opt, c->stream);
}

} // namespace avs
