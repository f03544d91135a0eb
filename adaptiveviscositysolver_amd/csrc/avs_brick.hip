// avs_brick.hip -- brick-structured SpMV for the brick-major system (round 4).
//
// The matrix of this problem is translation invariant along the liquid surface: the level-0 shell (80 % of the rows of the 512^3 beam) and
// the transitions below it repeat a few hundred row PATTERNS -- but only when a column is named by WHERE its face lies relative to the
// row's face, not by its index.  This form does that:
//
//   * tile = the rows of one 8^3 brick of fine cells (consecutive ids in the brick-major numbering, avs_reorder.hip; at most 1024 rows, a
//     fuller brick is cut in two); runs of small bricks (coarse interior: 3 .. 24 rows each) are cut into "E" tiles of 128 rows that keep
//     today's packed words;
//   * every face a tile may read has a tile-LOCAL slot on four padded lattices -- level l, brick-relative level-l cell (-1 .. 8>>l)^3,
//     axis a -> kBrickLoff[l] + ((z S + y) S + x) 3 + a with S = (8>>l) + 2: 3000 + 648 + 192 + 81 = 3921 slots of x in LDS (31 KB),
//     filled from <= 16-entry runs of consecutive (slot, column) pairs (quarter-wave loads; a run names its column as an offset into one
//     of the 27 neighbour bricks, whose first rows are in the tile header: 4 B per run);
//   * a row of a G tile whose columns all have a slot is stored as ONE 32-bit descriptor (tile-local pattern id, level, axis, cell) + its
//     16-bit position in the tile; its pattern -- the sequence of (slot delta relative to the row's base on the column's lattice,
//     lattice level, value code) in the STORED column order -- comes from a global table (a few thousand words, L2 resident), the tile's
//     patterns staged in LDS quad by quad.  The rows are EXECUTED sorted by (pattern length, pattern): a wave's 64 rows then have
//     the same trip count (one lane per row; unsorted, every wave pays for its longest row: 40 entries against 15 on average);
//   * any other row is "streamed": today's packed words (code << col_bits | column) in CSR order; the workgroup gathers x for them
//     from global memory (requested together with the fill, so the latencies overlap), parks the products in LDS and one lane per
//     row adds its segment -- the round-3 kernel's scheme, for the few rows that need it.  E tiles and G tiles without any pattern row use
//     the lattice's LDS for the products (3936 per pass instead of 1024).
//
// PERSISTENT workgroups (three per CU) walk the tiles: a workgroup launch per tile cost more than the tile (17 k launches of 8 waves with
// 53 KB of LDS: 68 us of a 170 us kernel with the arithmetic switched off).  Inside the walk the next tile's header (scalar) and ALL its
// descriptors are requested before the current tile is multiplied, so a tile waits for one global round trip (x and pattern words).
// One lane sums one row left to right with a multiply and an add per entry (no FMA, -ffp-contract=off): y is bit-identical to the
// plain CSR kernel.
// Round 5 -- partitioned systems: a fill run names its column as (one of <= 32 windows of 2^11 columns in the tile header, offset), so
// the local system of a rank ([owned | halo] columns) is just another matrix; with the halo in the tail of the vector (RCCL transport)
// the kernel is unchanged.  Direct transport: the halo lives in the rank's comm block, written by the peers; a small launch in front of
// this kernel (k_halo_gather, avs_pcg.hip) waits for the peers' flags and copies it behind the owned entries of the vector, and the
// persistent workgroups' x.y partials go to the stage slots the finalizer launch behind it folds (k_halo_finalize).  (A HALO
// instantiation of this kernel -- flag wait per flagged tile, comm-block loads in the fill -- was built first: 4-23 spilled registers
// and 139-167 us against 118 for the plain kernel; the two small launches cost ~5 us.)  Reference: Eigen's `mat * p` inside ConjugateGradient (cpp:618-630 of HDK_AdaptiveViscosity.cpp); the row
// structure is that of cpp:2537-2745.
#include <mutex>

#include "avs_internal.hpp"

namespace avs {

constexpr int kBrickBlk = 512;
#ifndef AVS_BRICK_TAIL
#define AVS_BRICK_TAIL 4      // chunks of the execution order that count as a tile's tail (in-loop A/B with four-lane fill runs: 2 -> 94.0 us, 4 -> 91.9 us)
#endif
#ifndef AVS_BRICK_TAIL_PRIO
#define AVS_BRICK_TAIL_PRIO 1
#endif
#ifndef AVS_BRICK_LOAD_PRIO
#define AVS_BRICK_LOAD_PRIO 2
#endif
#ifndef AVS_BRICK_PRIO
#define AVS_BRICK_PRIO 3 // s_setprio placements (see the load phase below): 0 none, 1 load phase, 2 + waves 6 and 7 in the walk, 3 + the waves of the tile's last two chunks
#endif

#ifdef AVS_PROBES
// measurement only (AVS_BRICK_DEBUG & 16): wall_clock64 stamps of workgroup phases, 8 per tile, first kStampTiles tiles of every workgroup
constexpr int kStampTiles = 40, kStampWgs = 1024;
__device__ long long g_brick_stamps[kStampWgs * kStampTiles * 8];
#define BRICK_DBG(bit) (B.debug & (bit))
#define BRICK_STAMP(slot)                                                                                            \
    do {                                                                                                             \
        if (BRICK_DBG(16) && threadIdx.x == 0 && iter < kStampTiles && blockIdx.x < kStampWgs) {                             \
            g_brick_stamps[((int)blockIdx.x * kStampTiles + iter) * 8 + (slot)] = wall_clock64();                    \
        }                                                                                                            \
    } while (0)

#else
#define BRICK_DBG(bit) 0
#define BRICK_STAMP(slot) do { } while (0)
#endif

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double brick_dpp_add(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double brick_wave_sum_dpp(double v) // lane 63 holds the sum (fixed order)
{
    v = brick_dpp_add<0x111, 0xf>(v);
    v = brick_dpp_add<0x112, 0xf>(v);
    v = brick_dpp_add<0x114, 0xf>(v);
    v = brick_dpp_add<0x118, 0xf>(v);
    v = brick_dpp_add<0x142, 0xa>(v);
    v = brick_dpp_add<0x143, 0xc>(v);
    return v;
}

// i-th streamed row of a tile -> thread: consecutive rows go to different waves (a handful of rows per tile: every wave gets a few lanes)
__device__ __forceinline__ int brick_srow_of_thread(int tid, int k) { return k * kBrickBlk + ((tid & 63) << 3) + (tid >> 6); }

// The kernel has no static LDS: the dynamic segment starts at LDS address 0, and a read from a BYTE ADDRESS spares the "+ base" the
// compiler otherwise leaves in the instruction stream per access (v_add_u32 v, 0, v: one of six VALU instructions per matrix entry).
template <typename T>
__device__ __forceinline__ T lds_abs(unsigned byte_addr) { return *(const T __attribute__((address_space(3))) *)(uintptr_t)byte_addr; }
// LDS layout in BYTES for vectors of T (double: the PCG loops; float: the float-vector loop of AVS_PRECISION_F32, avs_pcg_f32.inl):
//   [0, kBrickSlotsPad T) the lattice | kBrickXSlots T extra x slots | kBrickBlockBytes: the tile's descriptor block, then the products of
//   its streamed rows | vals (value table, T) | pattern words | pinfo | row-base table.  With T = float the lattice is 15.4 KB instead of
//   31 KB: the LDS would hold FOUR workgroups per CU instead of three -- but only at 64 registers per lane, where the fused-dot
//   instantiation spills 13 of them (113 MB of scratch traffic per launch at 512^3): measured equal on the 512^3 beam (99 against 100.5 us),
//   8 % slower on the 1024^3 sheet (317 against 288 us).  The kernel keeps the 80-register budget of the double instantiation: three
//   workgroups per CU, no spills.
constexpr unsigned kBrickBlockBytes = (unsigned)((kBrickPark - kBrickXSlots) * sizeof(double));
template <typename T> constexpr unsigned brick_vals_byte() { return (unsigned)((kBrickSlotsPad + kBrickXSlots) * sizeof(T)) + kBrickBlockBytes; } // `vals` behind the lattice and `park`
template <typename T> constexpr unsigned brick_vals_elems(int table_size) { return (unsigned)((table_size + 4) & ~3); } // keeps the pattern image 16-B aligned for both T

// element `idx` of an array whose base is workgroup-uniform: a 32-bit byte offset in a VGPR + the base in SGPRs (global_load ... saddr)
// instead of 64-bit address arithmetic per lane
template <class T>
__device__ __forceinline__ T ld_u32(const T *base, unsigned idx)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)(unsigned)(idx * (unsigned)sizeof(T)));
}


// A tile's descriptors are ONE contiguous block of 32-bit words (<= 8 KB), fetched with one 16-B load per thread and parked in LDS:
//   [0] row0 [1] nrows [2] npat [3] nruns [4] npq (pattern quads) [5] nprow (pattern rows) [6] srow0 [7] nsrows [8] sword0 [9] nsw
//   [16 .. 48) first rows of the 27 neighbour bricks
//   [10] rd0 (first pattern-row descriptor of the tile in rdesc)
//   runs[nruns] (8 B each: absolute first column | slot << 4 | length - 1) | pquads[npq] | pinfo[npat]
//   [11] the tile's rows read halo columns (partitioned systems)
// The per-row arrays (rdesc: descriptor + position of every pattern row in execution order; ownslot) stay in global memory.
constexpr int kBlkHdr = kBlkHdrWords;

template <bool DOT, bool VC = false, typename T = double>
__global__ __launch_bounds__(kBrickBlk) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_spmv_brick(BrickView B, const T *__restrict__ x, T *__restrict__ y,
                                                         double *__restrict__ partial, const int *__restrict__ done_flag)
{
    if (DOT && done_flag && *done_flag) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr unsigned ES = (unsigned)sizeof(T);                         // bytes per vector element: every LDS byte offset below scales with it
    constexpr unsigned kValsByte = brick_vals_byte<T>();
    T *xs = reinterpret_cast<T *>(smem_raw);                             // kBrickSlotsPad
    T *park = xs + kBrickSlotsPad;                                       // right behind the lattice: [0, kBrickXSlots) extra x slots (off-lattice
                                                                        // columns), behind them kBrickBlockBytes: first the tile's descriptor
                                                                        // block, then the products of its streamed rows
    T *vals = reinterpret_cast<T *>(smem_raw + kValsByte);               // table_size + 1 (the last entry is 0.0: padding words), padded to 16 B
    uint32_t *pw = reinterpret_cast<uint32_t *>(vals + brick_vals_elems<T>(B.table_size)); // kBrickPatWords (VC: kBrickPatWordsVc) + 8
    uint32_t *pinfo = pw + (VC ? kBrickPatWordsVc : kBrickPatWords) + 8; // kBrickPatMax: local start | quads << 16
    uint2 *rbt = reinterpret_cast<uint2 *>(pinfo + kBrickPatMax);       // kBrickRowBase entries: a row's bases on the four lattices, per axis
    const uint32_t *bw = reinterpret_cast<const uint32_t *>(park + kBrickXSlots);
    const int tid = threadIdx.x;
    constexpr int RL = kBrickRunLen;                                    // lanes per fill run
    const int lane = tid & 63, l16 = tid & (RL - 1), qw = tid / RL;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);          // (scalar)
    const bool wide = VC || B.col_bits == 0;                             // 64-bit streamed words: column | code << 32 (VC: 96 bits, column | value)
    const unsigned cmask = wide ? 0xffffffffu : (1u << B.col_bits) - 1u;
    const int cbits = B.col_bits;
    constexpr int QW = kBrickBlk / RL;
    static_assert((kBrickPark - kBrickXSlots) * 2 <= kBrickBlk * 4, "one 16-B load per thread fetches a whole descriptor block");

    if (!VC)
        for (int i = tid; i <= B.table_size; i += kBrickBlk) vals[i] = (i < B.table_size) ? (T)B.table[i] : (T)0; // once per workgroup (VC: a table per tile; float: the values ARE floats, AVS_PRECISION_F32)
    // A row of level lr at local cell (cx, cy, cz) has the byte offset 8 (kBrickLoff[lc] + 3 ((bz S + by) S + bx)) on lattice lc, with
    // b = ((c >> up) << down) + 1 per axis: the sum of one term per axis.  Table entry (axis d, lr, c + 1) = the four lattices' terms as
    // 16-bit fields {lattice 0 | lattice 1 << 16, lattice 2 | lattice 3 << 16} (the sums stay below 2^15: no carries between the fields);
    // the x entries carry the lattice offsets.  Three 8-B LDS reads and four adds per row instead of ~60 VALU instructions.
    if (tid < kBrickRowBase) {
        const int d = tid / 40, lr = (tid / 10) & 3, cc = tid % 10 - 1;
        unsigned f[4];
#pragma unroll
        for (int lc = 0; lc < 4; ++lc) {
            const int up = lc > lr ? lc - lr : 0, dn2 = lr > lc ? lr - lc : 0;
            const int S = (8 >> lc) + 2;
            const int b = ((cc >> up) << dn2) + 1;
            f[lc] = (unsigned)(3 * (int)ES * b * (d == 0 ? 1 : d == 1 ? S : S * S) + (d == 0 ? (int)ES * kBrickLoff[lc] : 0));
        }
        rbt[tid] = uint2{f[0] | (f[1] << 16), f[2] | (f[3] << 16)};
    }

    // Tile -> workgroup.  Workgroup b runs on XCD b mod 8 (round-robin dispatch); consecutive tiles are neighbouring bricks and read
    // each other's rows as halo, so what an XCD works on at any time is one contiguous run of gridDim / 8 tiles, its workgroups side by
    // side (a halo value is fetched by one L2; tile = b + k gridDim: by up to eight).  Two walks, chosen per matrix (BrickView::walk,
    // measured at the first assembly of a matrix size, avs_reorder.hip):
    //   0  one contiguous EIGHTH of the tiles per XCD -- best where tiles cost the same everywhere (512^3 beam: 2 % over walk 1 in the loop);
    //   1  chunks of gridDim / 8 tiles dealt to the XCDs in turn -- every XCD sees the same mix of tiles.  A thin sheet's eighths are its z
    //      layers (full surface bricks here, coarse interior there) and the kernel waits for the slowest XCD: 133 (walk 0) against 94 us.
    //   2  (round 5) a PLANNED walk: every workgroup's tile sequence is laid out by the host from a cost model of the tiles (BrickForm::plan_walk:
    //      cost-equal contiguous ranges per XCD, then the tiles of a range dealt, in order, to whichever of the XCD's workgroups has
    //      the least work so far) -- the static walks end 15-20 % after their average workgroup (E tiles cost two G tiles, G tiles
    //      differ by their rows); deterministic: the plan is a function of the matrix and the grid.
    const uint2 *seq;                                                    // this workgroup's tiles: seq[0], seq[step], ... (cnt of them)
    int step, cnt;
    if (B.wlist && (int)gridDim.x == B.wgrid && !BRICK_DBG(32)) {
        const int w0 = B.wptr[blockIdx.x];
        seq = B.wlist + w0;
        step = 1;
        cnt = B.wptr[blockIdx.x + 1] - w0;
    } else {
        int tile = blockIdx.x, tstep = (int)gridDim.x, tend = B.ntiles;
        if ((gridDim.x & 7u) == 0u && B.ntiles >= (int)gridDim.x && !BRICK_DBG(32)) {
            const int c = (int)(blockIdx.x & 7u), gx = (int)(gridDim.x >> 3);
            tile = c * gx + (int)(blockIdx.x >> 3);
            if (B.walk == 0) { // one contiguous eighth of the tiles per XCD
                tstep = gx;
                tile = (int)(((int64_t)B.ntiles * c) >> 3) + (int)(blockIdx.x >> 3);
                tend = (int)(((int64_t)B.ntiles * (c + 1)) >> 3);
            }
        }
        seq = B.tile_blk + tile;
        step = tstep;
        cnt = tile < tend ? (tend - tile + tstep - 1) / tstep : 0;
    }
    if (cnt <= 0) {                                                      // (a workgroup without tiles: only when the grid exceeds what the plan could fill)
        if (DOT && tid == 0) partial[blockIdx.x] = 0.;
        return;
    }
    const uint4 *blocks16 = reinterpret_cast<const uint4 *>(B.blocks);
    uint2 tb = seq[0];                                                   // first 16-B unit, units
    uint4 blk = blocks16[(int64_t)tb.x + (tid < (int)tb.y ? tid : 0)];
    int nsw_prev = 0;

    int iter = 0;
    T dot = 0;                                                           // x.y of this lane's rows, all tiles of the workgroup
    for (;;) {
        BRICK_STAMP(0);
        if (nsw_prev > 0) __syncthreads();                               // the previous tile's streamed sums have read `park`
        if (tid < (int)tb.y) reinterpret_cast<uint4 *>(park + kBrickXSlots)[tid] = blk;
        __syncthreads();                                                 // every wave is done with the previous tile; the block is visible
        const int row0 = __builtin_amdgcn_readfirstlane((int)bw[0]), nrows = __builtin_amdgcn_readfirstlane((int)bw[1]);
        const int npat = __builtin_amdgcn_readfirstlane((int)bw[2]), nruns = __builtin_amdgcn_readfirstlane((int)bw[3]);
        const int npq = __builtin_amdgcn_readfirstlane((int)bw[4]), nprow = __builtin_amdgcn_readfirstlane((int)bw[5]);
        const int srow0 = __builtin_amdgcn_readfirstlane((int)bw[6]), nsrows = __builtin_amdgcn_readfirstlane((int)bw[7]);
        const int sword0 = __builtin_amdgcn_readfirstlane((int)bw[8]), nsw = __builtin_amdgcn_readfirstlane((int)bw[9]);
        const int rd0 = __builtin_amdgcn_readfirstlane((int)bw[10]);
        // VC (variable viscosity: the patterns carry the geometry only): the tile's value codes start at vcodes[cw0] (one 8-B quad of four
        // 16-bit byte offsets into the tile's value table per row quad, wave-interleaved in execution order), its ntv values at ttab[tt0]
        const int cw0 = VC ? __builtin_amdgcn_readfirstlane((int)bw[12]) : 0, tt0 = VC ? __builtin_amdgcn_readfirstlane((int)bw[13]) : 0;
        const int ntv = VC ? __builtin_amdgcn_readfirstlane((int)bw[14]) : 0;
        const int o_runs = kBlkHdr, o_pq = o_runs + 2 * nruns, o_pi = o_pq + npq; // (runs are 8 B)
        const bool emode = npat == 0;                 // no pattern rows: the products of the streamed rows may use the x lattice's LDS
        T *prod = emode ? xs : park + kBrickXSlots;
        const int cap = emode ? kBrickSlotsPad : kBrickPark - kBrickXSlots;
        BRICK_STAMP(1);
        // Round 6: a CU runs three workgroups on the same SIMDs and the kernel is bound by instruction ISSUE (38 M VALU wave-instructions per
        // launch at 512^3, 80 % of them outside the row walk).  A wave in this phase -- descriptors, the tile's ONE round trip of loads, LDS
        // writes, with the whole workgroup waiting at the barrier behind it -- goes ahead of the other workgroups' row walks (s_setprio 2
        // until the barrier), and so do, in the walk, the two waves with the tile's LONGEST rows (the execution order is sorted by pattern
        // length: the slowest wave walks 8.9 quads per tile against a mean of 4.3 on the 512^3 beam, 5.5 against 2.7 on the 1024^3 sheet,
        // and the next tile cannot start before it is done).  Measured (profiles/r06_notes.md): 512^3 beam 111.5 -> 104.2 us stand-alone,
        // 1024^3 sheet 324 -> 305 us.
        if (AVS_BRICK_PRIO >= 1) __builtin_amdgcn_s_setprio(AVS_BRICK_LOAD_PRIO);
#ifdef AVS_PROBES
        if (BRICK_DBG(16) && threadIdx.x == 0 && iter < kStampTiles && blockIdx.x < kStampWgs) // tile kind: 1 E tile, 2 G tile with streamed rows, 0 G tile
        {
            long long *sp = &g_brick_stamps[((int)blockIdx.x * kStampTiles + iter) * 8];
            sp[5] = emode ? 1 : (nsw > 0 ? 2 : 0);
            sp[6] = (long long)nprow | ((long long)nruns << 16) | ((long long)npq << 32) | ((long long)nrows << 48);   // what a cost model can see
            sp[7] = (long long)nsw | ((long long)nsrows << 32);
        }
#endif
        // ---- this tile's loads, ONE round trip: x of the halo runs, x of the tile's own rows, the pattern quads, the row descriptors.
        // Round 5: (i) a fill run is 8 B -- absolute first column | slot << 4 | length - 1 -- so a batch costs ONE LDS read (it was two
        // dependent ones: run word, then the base of its neighbour brick); (ii) every LDS read the loads depend on (run words, pattern
        // quad offsets) is issued before the first global load -- one wait instead of one or two per batch; (iii) nothing is SELECTED
        // right behind a load (`valid ? loaded : default` made the compiler wait for that load on the spot: the round-4 binary stalled
        // for the own-slot load, the row descriptors and the streamed descriptors one after the other, three serialised round trips);
        // validity is a predicate recomputed where the value is used.
        constexpr int RPT = kBrickMaxRows / kBrickBlk;
        constexpr int PQ = (kBrickPatWords / 4 + kBrickBlk - 1) / kBrickBlk;
#ifndef AVS_BRICK_RUFAST
#define AVS_BRICK_RUFAST (RL == 16 ? 6 : (RL == 8 ? 4 : (RL == 4 ? 3 : 2)))
#endif
        constexpr int kRuFast = AVS_BRICK_RUFAST; // fill batches held in registers (192 / 256 / 384 / 512 runs: nearly every tile); the rest, rare, go run by run
        const uint2 *runs2 = reinterpret_cast<const uint2 *>(bw + o_runs);
        T fv[kRuFast];
        uint32_t rdsc[kRuFast];
        uint32_t pqo[PQ];
#pragma unroll
        for (int u = 0; u < PQ; ++u) {
            pqo[u] = 0u;
            if (u * kBrickBlk < npq) {
                const int q = tid + u * kBrickBlk;
                pqo[u] = bw[o_pq + (q < npq ? q : 0)];
            }
        }
        const uint32_t pinf = bw[o_pi + (tid < npat ? tid : 0)];
        {
            uint2 rw[kRuFast];
#pragma unroll
            for (int u = 0; u < kRuFast; ++u) {
                rw[u] = uint2{0u, 0u};
                if (u * QW < nruns) {                  // block-uniform: a batch nobody needs is not requested
                    const int q = u * QW + qw;
                    rw[u] = runs2[q < nruns ? q : 0];
                }
            }
#pragma unroll
            for (int u = 0; u < kRuFast; ++u) {
                fv[u] = 0;
                rdsc[u] = 0xffffffffu;
                if (u * QW < nruns) {
                    const int q = u * QW + qw;
                    const bool on = q < nruns && l16 <= (int)(rw[u].y & 15u) && !BRICK_DBG(1);
                    rdsc[u] = on ? rw[u].y : 0xffffffffu;
                    fv[u] = ld_u32(x, on ? rw[u].x + (unsigned)l16 : (unsigned)row0);
                }
            }
        }
        T xo[RPT];
        uint32_t os[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            xo[k] = 0;
            os[k] = 0xffffu;
            if (!emode && k * kBrickBlk < nrows) {
                const int r = tid + k * kBrickBlk;
                xo[k] = ld_u32(x + row0, (unsigned)(r < nrows ? r : 0));
                os[k] = ld_u32(B.ownslot + row0, (unsigned)(r < nrows ? r : 0));   // (a clamped lane re-reads row 0's slot: it is not written, see below)
            }
        }
        uint4 pqv[PQ];
#pragma unroll
        for (int u = 0; u < PQ; ++u) {
            pqv[u] = uint4{0u, 0u, 0u, 0u};
            if (u * kBrickBlk < npq) pqv[u] = *reinterpret_cast<const uint4 *>(B.pwords + pqo[u]);
        }
        T tval = 0;
        if (VC && ntv > 0) tval = (T)ld_u32(B.ttab + tt0, (unsigned)(tid < ntv ? tid : 0));   // the tile's value table (<= kBrickTileVals entries)
        int cbo[RPT];                                  // VC: first quad of this wave's blocks of the code stream (read NOW: the streamed rows'
                                                       // products overwrite the block image while slower waves still walk their rows)
#pragma unroll
        for (int k = 0; k < RPT; ++k) cbo[k] = VC ? __builtin_amdgcn_readfirstlane((int)bw[16 + k * (kBrickBlk / 64) + (tid >> 6)]) : 0;
        uint2 rdv[RPT]; // descriptor, position in the tile (valid for tid + k * kBrickBlk < nprow)
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            rdv[k] = uint2{0u, 0u};
            if (k * kBrickBlk < nprow) {
                const int i = tid + k * kBrickBlk;
#ifdef AVS_EXP_MNT
                {
                    typedef unsigned u2v __attribute__((ext_vector_type(2)));
                    const u2v v = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(reinterpret_cast<const char *>(B.rdesc + rd0) + (size_t)((unsigned)(i < nprow ? i : 0) * 8u)));
                    rdv[k] = uint2{v.x, v.y};
                }
#else
                rdv[k] = ld_u32(B.rdesc + rd0, (unsigned)(i < nprow ? i : 0));
#endif
            }
        }
        // streamed rows (few tiles): descriptors (valid for brick_srow_of_thread(tid, k) < nsrows), and for a G tile the first pass of words with their x
        uint2 sd[kBrickMaxRows / kBrickBlk];
#pragma unroll
        for (int k = 0; k < kBrickMaxRows / kBrickBlk; ++k) sd[k] = uint2{0u, 0u};
        uint32_t w0[2] = {0u, 0u};
        T xv0[2] = {0, 0};
        if (nsrows > 0) {
#pragma unroll
            for (int k = 0; k < kBrickMaxRows / kBrickBlk; ++k) {
                if (k * kBrickBlk >= nsrows) break;
                const int i = brick_srow_of_thread(tid, k);
                sd[k] = B.sdesc[srow0 + (i < nsrows ? i : 0)];
            }
            if (!emode && !wide) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = tid + u * kBrickBlk;
                    w0[u] = B.swords[(int64_t)sword0 + (e < nsw ? e : 0)];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) xv0[u] = x[w0[u] & cmask];
            }
        }
        // the next tile's block address: a scalar load, in flight while this tile's data arrives
        const bool more = iter + 1 < cnt;
        const uint2 tbn = seq[(int64_t)(more ? iter + 1 : iter) * step];
        // ---- LDS writes
        if (VC && tid <= ntv) vals[tid] = tid < ntv ? tval : (T)0;          // (entry ntv = 0.0: the code of the padding words)
        if (tid < npat) pinfo[tid] = pinf;
#pragma unroll
        for (int u = 0; u < PQ; ++u) {
            const int q = tid + u * kBrickBlk;
            if (q < npq) reinterpret_cast<uint4 *>(pw)[q] = pqv[u];
        }
#pragma unroll
        for (int u = 0; u < kRuFast; ++u)
            if (rdsc[u] != 0xffffffffu) xs[((rdsc[u] >> 4) & 0xfffu) + l16] = fv[u]; // (extra slots: xs runs on into `park`)
        if (nruns > kRuFast * QW && !BRICK_DBG(1)) {   // (block-uniform, rare) the runs beyond the register batches: straight into LDS
#pragma unroll 1
            for (int q = kRuFast * QW + qw; q < nruns; q += QW) {
                const uint2 r = runs2[q];
                if (l16 <= (int)(r.y & 15u)) xs[((r.y >> 4) & 0xfffu) + l16] = ld_u32(x, r.x + (unsigned)l16);
            }
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (!emode && tid + k * kBrickBlk < nrows && os[k] != 0xffffu) xs[os[k]] = xo[k];
        __syncthreads();
        BRICK_STAMP(2);

        if (AVS_BRICK_PRIO >= 1) __builtin_amdgcn_s_setprio(0);
        // ---- the next tile's block travels while this tile is multiplied (16 B per thread)
        blk = blocks16[(int64_t)tbn.x + (tid < (int)tbn.y ? tid : 0)];

        if (!emode) {
            // products of the first streamed pass (the block in `park` is dead now)
            if (nsw > 0 && !wide) {
                if (tid < nsw && tid < cap) prod[tid] = vals[w0[0] >> cbits] * xv0[0];
                if (tid + kBrickBlk < nsw && tid + kBrickBlk < cap) prod[tid + kBrickBlk] = vals[w0[1] >> cbits] * xv0[1];
            }
            // pattern rows: one lane per row, everything from LDS
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
            if (k * kBrickBlk >= nprow) break;
            if (tid + k * kBrickBlk < nprow && !BRICK_DBG(2)) {
                const uint32_t rd = rdv[k].x, ro = rdv[k].y;
                const unsigned pid = rd >> 20;
                const int lr = (int)((rd >> 18) & 3u), ax = (int)((rd >> 16) & 3u);
                const uint2 *rb = rbt + lr * 10;
                const uint2 tx = rb[rd & 15u], ty = rb[40 + ((rd >> 4) & 15u)], tz = rb[80 + ((rd >> 8) & 15u)];
                const unsigned P01 = tx.x + ty.x + tz.x, P23 = tx.y + ty.y + tz.y; // byte offsets of the row's base on the four lattices
                const unsigned own8 = __builtin_amdgcn_perm(P23, P01, (unsigned)lr * 0x0202u + 0x0c0c0100u) + ES * (unsigned)ax;
                const unsigned pi = pinfo[pid];
                const uint4 *wq = reinterpret_cast<const uint4 *>(pw + (pi & 0xffffu));
                const int nq = (int)((pi >> 16) & 0x7fffu);
                // (the wave's chunk of the execution order is one of the tile's last two: a scalar test -- as a per-lane test on the row index it
                //  cost two more spilled registers and made the kernel 4 % SLOWER than no priority at all)
#ifdef AVS_BRICK_GRADED
                {
                    const int back = ((nprow + 63) >> 6) - (k * (kBrickBlk / 64) + wave_id); // 1 = the last chunk
                    if (back <= 2) __builtin_amdgcn_s_setprio(2);
                    else if (back <= 4) __builtin_amdgcn_s_setprio(1);
                }
#else
                if (AVS_BRICK_PRIO == 2 ? wave_id >= 6 : (AVS_BRICK_PRIO >= 3 && k * (kBrickBlk / 64) + wave_id + AVS_BRICK_TAIL >= ((nprow + 63) >> 6))) __builtin_amdgcn_s_setprio(AVS_BRICK_TAIL_PRIO);
#endif
                // word: delta << 19 (signed 13) | 000 | lattice level << 14 | code << 3 -- byte offsets for 8-B elements; T = float reads
                // the 4-B image of the table (BrickView::pwords32: delta << 18 | level << 14 | code << 2).  A pattern is padded to whole
                // quads with words that repeat its first entry's slot with the code of 0.0: +-0.0 added to a sum that is never -0.0.
                constexpr unsigned CM = ES == 8 ? 0x3ff8u : 0x1ffcu;       // the code field as a byte offset into `vals`
                constexpr int VS = ES == 8 ? 0 : 1;                       // VC: the code stream holds byte offsets for 8-B values
                auto addr = [&](uint32_t w) -> unsigned {
                    const unsigned sel = ((w >> 14) & 3u) * 0x0202u + 0x0c0c0100u;
                    return (unsigned)((int)w >> 16) + __builtin_amdgcn_perm(P23, P01, sel); // + 16-bit field number `level` of P23:P01
                };
                T sum = 0;
                // software pipeline: the words of quad q + 1 and the value / x reads of quad q are in flight while quad q - 1 is added
                // (reading the words one quad past the pattern is harmless: the LDS image ends with spare quads; they are never decoded)
                auto walk = [&](auto adr) {
                    T v0, v1, v2, v3, x0, x1, x2, x3;
                    const uint4 w = wq[0];
                    uint4 wn = wq[1];
                    v0 = lds_abs<T>(kValsByte + (w.x & CM)); x0 = lds_abs<T>(adr(w.x));
                    v1 = lds_abs<T>(kValsByte + (w.y & CM)); x1 = lds_abs<T>(adr(w.y));
                    v2 = lds_abs<T>(kValsByte + (w.z & CM)); x2 = lds_abs<T>(adr(w.z));
                    v3 = lds_abs<T>(kValsByte + (w.w & CM)); x3 = lds_abs<T>(adr(w.w));
                    for (int q = 1; q < nq; ++q) {
                        const uint4 wnn = wq[q + 1];
                        const T a0 = lds_abs<T>(kValsByte + (wn.x & CM)), c0 = lds_abs<T>(adr(wn.x));
                        const T a1 = lds_abs<T>(kValsByte + (wn.y & CM)), c1 = lds_abs<T>(adr(wn.y));
                        const T a2 = lds_abs<T>(kValsByte + (wn.z & CM)), c2 = lds_abs<T>(adr(wn.z));
                        const T a3 = lds_abs<T>(kValsByte + (wn.w & CM)), c3 = lds_abs<T>(adr(wn.w));
                        sum += v0 * x0;
                        sum += v1 * x1;
                        sum += v2 * x2;
                        sum += v3 * x3;
                        v0 = a0; x0 = c0; v1 = a1; x1 = c1; v2 = a2; x2 = c2; v3 = a3; x3 = c3;
                        wn = wnn;
                    }
                    sum += v0 * x0;
                    sum += v1 * x1;
                    sum += v2 * x2;
                    sum += v3 * x3;
                };
                // VC: the value of an entry comes from the row's own code stream -- quad q of this lane at cp[64 q] (the wave's rows are
                // interleaved, so a wave load is one 512-B run) -- instead of from the pattern word; same pipeline, same order of additions
                auto walk_vc = [&](auto adr) {
                    const uint2 *cp = B.vcodes + (int64_t)cw0 + cbo[k] + lane;
                    T v0, v1, v2, v3, x0, x1, x2, x3;
                    const uint4 w = wq[0];
                    uint4 wn = wq[1];
                    uint2 c = cp[0];
                    uint2 cn = cp[nq > 1 ? 64 : 0];
                    v0 = lds_abs<T>(kValsByte + ((c.x & 0xffffu) >> VS)); x0 = lds_abs<T>(adr(w.x));
                    v1 = lds_abs<T>(kValsByte + ((c.x >> 16) >> VS)); x1 = lds_abs<T>(adr(w.y));
                    v2 = lds_abs<T>(kValsByte + ((c.y & 0xffffu) >> VS)); x2 = lds_abs<T>(adr(w.z));
                    v3 = lds_abs<T>(kValsByte + ((c.y >> 16) >> VS)); x3 = lds_abs<T>(adr(w.w));
                    for (int q = 1; q < nq; ++q) {
                        const uint4 wnn = wq[q + 1];
                        const uint2 cnn = cp[(q + 1 < nq ? q + 1 : q) * 64];
                        const T a0 = lds_abs<T>(kValsByte + ((cn.x & 0xffffu) >> VS)), c0 = lds_abs<T>(adr(wn.x));
                        const T a1 = lds_abs<T>(kValsByte + ((cn.x >> 16) >> VS)), c1 = lds_abs<T>(adr(wn.y));
                        const T a2 = lds_abs<T>(kValsByte + ((cn.y & 0xffffu) >> VS)), c2 = lds_abs<T>(adr(wn.z));
                        const T a3 = lds_abs<T>(kValsByte + ((cn.y >> 16) >> VS)), c3 = lds_abs<T>(adr(wn.w));
                        sum += v0 * x0;
                        sum += v1 * x1;
                        sum += v2 * x2;
                        sum += v3 * x3;
                        v0 = a0; x0 = c0; v1 = a1; x1 = c1; v2 = a2; x2 = c2; v3 = a3; x3 = c3;
                        wn = wnn;
                        cn = cnn;
                    }
                    sum += v0 * x0;
                    sum += v1 * x1;
                    sum += v2 * x2;
                    sum += v3 * x3;
                };
                // (a queue of four code quads per lane, the first four loaded with the tile's other loads, was measured: 20 spilled registers, 288 us)
                if (pi >> 31) { // every column of the pattern on the level-0 lattice (rows are executed sorted by pattern: waves rarely mix)
                    const unsigned b0 = P01 & 0xffffu;
                    if (VC) walk_vc([&](uint32_t w) -> unsigned { return (unsigned)((int)w >> 16) + b0; });
                    else walk([&](uint32_t w) -> unsigned { return (unsigned)((int)w >> 16) + b0; });
                } else {
                    if (VC) walk_vc(addr);
                    else walk(addr);
                }
#ifdef AVS_EXP_YNT
                __builtin_nontemporal_store(sum, reinterpret_cast<T *>(reinterpret_cast<char *>(y + row0) + (size_t)(unsigned)(ro * ES)));
#else
                *reinterpret_cast<T *>(reinterpret_cast<char *>(y + row0) + (size_t)(unsigned)(ro * ES)) = sum;
#endif
                if (AVS_BRICK_PRIO >= 2) __builtin_amdgcn_s_setprio(0);
                if (DOT) dot += sum * lds_abs<T>(own8);
            }
            }
        }
        BRICK_STAMP(3);
        // streamed rows: passes of `cap` products parked in LDS, then every row adds its segment left to right
        if (nsw > 0 && !BRICK_DBG(4)) {
#ifdef AVS_EXP_SPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
            T ssum[kBrickMaxRows / kBrickBlk];
#pragma unroll
            for (int k = 0; k < kBrickMaxRows / kBrickBlk; ++k) ssum[k] = 0;
            for (int ts = 0; ts < nsw; ts += cap) {
                const int te = (ts + cap < nsw) ? ts + cap : nsw;
                if (emode || wide || ts > 0) {
                    if (ts > 0) __syncthreads(); // the previous pass has been summed
                    constexpr int SU = 4; // (8 entries per thread -- a whole pass of an E tile in one round of loads -- was measured SLOWER: 7.1 -> 7.7 us per E tile)
                    for (int e0 = ts + tid; e0 < te; e0 += SU * kBrickBlk) {
                        uint32_t w4[SU], c4[SU];
                        T x4[SU], v4[SU];
#pragma unroll
                        for (int u = 0; u < SU; ++u) {
                            const int e = e0 + u * kBrickBlk;
                            const int64_t at = (int64_t)sword0 + (e < te ? e : ts);
                            v4[u] = 0;
                            if (VC) { // 12 B per entry: column, value
                                const uint32_t *w = B.swords + 3 * at;
                                w4[u] = w[0];
                                c4[u] = 0u;
                                v4[u] = (T)__hiloint2double((int)w[2], (int)w[1]);
                            } else if (wide) {
                                const uint2 w = reinterpret_cast<const uint2 *>(B.swords)[at];
                                w4[u] = w.x;
                                c4[u] = w.y;
                            } else {
                                const uint32_t w = B.swords[at];
                                w4[u] = w & cmask;
                                c4[u] = w >> cbits;
                            }
                        }
#pragma unroll
                        for (int u = 0; u < SU; ++u) x4[u] = x[w4[u]];
#pragma unroll
                        for (int u = 0; u < SU; ++u) {
                            const int e = e0 + u * kBrickBlk;
                            if (e < te) prod[e - ts] = (VC ? v4[u] : vals[c4[u]]) * x4[u];
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < kBrickMaxRows / kBrickBlk; ++k) {
                    const int len = brick_srow_of_thread(tid, k) < nsrows ? (int)(sd[k].x >> 16) : 0, st = (int)sd[k].y;
                    const int a = st > ts ? st : ts, b = (st + len < te) ? st + len : te;
                    for (int j = a; j < b; ++j) ssum[k] += prod[j - ts];
                }
            }
#pragma unroll
            for (int k = 0; k < kBrickMaxRows / kBrickBlk; ++k)
                if (brick_srow_of_thread(tid, k) < nsrows && (sd[k].x >> 16) > 0) {
                    const int64_t row = (int64_t)row0 + (int)(sd[k].x & 0xffffu);
                    y[row] = ssum[k];
                    if (DOT) dot += ssum[k] * x[row];
                }
        }
#ifdef AVS_EXP_SPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        BRICK_STAMP(4);
        if (!more) break;
        ++iter;
        tb = tbn;
        nsw_prev = (nsw > 0 && !BRICK_DBG(4)) ? 1 : 0;
    }
    if (DOT) { // ONE partial per workgroup (fixed order: the tile walk is static, the waves' sums are added in wave order): few enough
               // for the vector kernel that follows to fold them itself instead of a reduction launch in between
        const double dsum = brick_wave_sum_dpp((double)dot);            // (float: a lane's few rows in float, everything across lanes in double)
        double *wsum = reinterpret_cast<double *>(park + kBrickXSlots);
        __syncthreads();                                                 // (the last tile's streamed sums may still be reading `park`)
        if (lane == 63) wsum[tid >> 6] = dsum;
        __syncthreads();
        if (tid == 0) {
            double s = wsum[0];
#pragma unroll
            for (int k = 1; k < kBrickBlk / 64; ++k) s += wsum[k];
            partial[blockIdx.x] = s;
        }
    }
}

size_t brick_lds_bytes(const BrickView &B, int elem_bytes)
{
    const size_t es = (size_t)elem_bytes;
    const size_t vals = elem_bytes == 4 ? brick_vals_elems<float>(B.table_size) : brick_vals_elems<double>(B.table_size);
    return (size_t)(kBrickSlotsPad + kBrickXSlots) * es + kBrickBlockBytes + vals * es +
           (size_t)((B.vc ? kBrickPatWordsVc : kBrickPatWords) + 8 + kBrickPatMax + 2 * kBrickRowBase) * sizeof(uint32_t);
}
size_t brick_lds_bytes(const BrickView &B) { return brick_lds_bytes(B, B.f32 ? 4 : 8); }

// LDS a workgroup may ask for on this device (the kernel opts in to more than the default 48 KiB, hipFuncSetAttribute below)
constexpr size_t kBrickLdsLimit = 64 * 1024;
bool brick_lds_fits(const BrickView &B) { return brick_lds_bytes(B, 8) <= kBrickLdsLimit; }

template <typename T>
static const void *brick_kernel(bool dot, bool vc)
{
    if (vc) return dot ? (const void *)k_spmv_brick<true, true, T> : (const void *)k_spmv_brick<false, true, T>;
    return dot ? (const void *)k_spmv_brick<true, false, T> : (const void *)k_spmv_brick<false, false, T>;
}

// persistent grid: as many workgroups as the device keeps resident for THIS LDS size (a larger value table costs a workgroup per CU;
// the float kernel's lattice is half as large: four per CU); queried once per (device, LDS size, variant), guarded: contexts of several
// host threads share the cache
static int brick_grid(const BrickView &B, size_t lds, int elem_bytes)
{
    struct Entry { int dev; size_t lds; int vc, es; int grid; };
    static std::mutex mu;
    static std::vector<Entry> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    int g = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (const Entry &e : cache)
            if (e.dev == dev && e.lds == lds && e.vc == B.vc && e.es == elem_bytes) g = e.grid;
        if (!g) {
            int per_cu = 0, cus = 0;
            const void *fn = elem_bytes == 4 ? brick_kernel<float>(true, B.vc != 0) : brick_kernel<double>(true, B.vc != 0);
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBrickBlk, lds);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (per_cu < 1) per_cu = 1;
            if (cus < 1) cus = 256;
            g = per_cu * cus;
            cache.push_back(Entry{dev, lds, B.vc, elem_bytes, g});
        }
    }
#ifdef AVS_PROBES
    if (const char *e = getenv("AVS_BRICK_GRID")) g = atoi(e) > 0 ? atoi(e) : g;
#endif
    return g < B.ntiles ? g : B.ntiles;
}

// the kernel's dynamic LDS limit is a per-device attribute of the loaded code object: raised once per device
static avs_status brick_raise_lds_limit()
{
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (done[dev]) return AVS_OK;
    for (int dot = 0; dot < 2; ++dot)
        for (int vc = 0; vc < 2; ++vc) {
            AVS_HIP(hipFuncSetAttribute(brick_kernel<double>(dot, vc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBrickLdsLimit));
            AVS_HIP(hipFuncSetAttribute(brick_kernel<float>(dot, vc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBrickLdsLimit));
        }
    done[dev] = true;
    return AVS_OK;
}

#ifdef AVS_PROBES
// phase times of the last launch with AVS_BRICK_DEBUG & 16, in 10-ns ticks: mean over workgroups and tiles
static void brick_print_stamps()
{
    std::vector<long long> hs((size_t)kStampWgs * kStampTiles * 8);
    if (hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(g_brick_stamps), hs.size() * sizeof(long long)) != hipSuccess) return;
    if (const char *path = getenv("AVS_BRICK_STAMP_FILE")) { // raw stamps [workgroup][tile of its walk][8] for tools/probes/brick_cost_fit.py
        if (FILE *f = fopen(path, "wb")) {
            const int hdr[4] = {kStampWgs, kStampTiles, 8, 0};
            fwrite(hdr, sizeof(hdr), 1, f);
            fwrite(hs.data(), sizeof(long long), hs.size(), f);
            fclose(f);
        }
    }
    static const char *kind[3] = {"G tiles", "E tiles", "G tiles with streamed rows"};
    for (int ty = 0; ty < 3; ++ty) {
        double acc[6] = {};
        long cnt = 0;
        for (int w = 0; w < kStampWgs; ++w)
            for (int i = 1; i + 1 < kStampTiles; ++i) {
                const long long *p = &hs[((size_t)w * kStampTiles + i) * 8], *pn = p + 8;
                if (!p[0] || !p[4] || !pn[0] || p[5] != ty) continue;
                acc[0] += (double)(p[1] - p[0]); acc[1] += (double)(p[2] - p[1]); acc[2] += (double)(p[3] - p[2]);
                acc[3] += (double)(p[4] - p[3]); acc[4] += (double)(pn[0] - p[4]); acc[5] += (double)(pn[0] - p[0]);
                ++cnt;
            }
        if (cnt)
            fprintf(stderr, "brick phases (us, mean of %ld %s): block to LDS + barrier %.2f | issue loads, LDS writes, barrier %.2f | pattern rows %.2f | streamed + dot %.2f | loop %.2f | tile %.2f\n",
                    cnt, kind[ty], acc[0] / cnt / 100, acc[1] / cnt / 100, acc[2] / cnt / 100, acc[3] / cnt / 100, acc[4] / cnt / 100, acc[5] / cnt / 100);
    }
}
#endif

// partial sums of x.y the fused-dot launch writes: one per workgroup of the persistent grid (of the kernel the view is laid out for:
// BrickView::f32 -- the planned walk and the partial arrays follow that grid)
int brick_partial_count(const BrickView &B)
{
    const int es = B.f32 ? 4 : 8;
    return B.ntiles > 0 ? brick_grid(B, brick_lds_bytes(B, es), es) : 0;
}
int brick_partial_count(const BrickView &B, int elem_bytes)
{
    return B.ntiles > 0 ? brick_grid(B, brick_lds_bytes(B, elem_bytes), elem_bytes) : 0;
}

template <typename T>
static avs_status spmv_brick_launch_t(const BrickView &B0, const T *x, T *y, double *partial, const int *done_flag, hipStream_t stream)
{
    if (B0.ntiles <= 0) return AVS_OK;
    constexpr int es = (int)sizeof(T);
    const size_t lds = brick_lds_bytes(B0, es);
    AVS_REQUIRE(lds <= kBrickLdsLimit, AVS_EINTERNAL, "brick form: %zu bytes of LDS per workgroup exceed the limit (value table of %d entries)", lds,
                B0.table_size);
    AVS_TRY(brick_raise_lds_limit());
    const int grid = brick_grid(B0, lds, es);
    BrickView B = B0;
    if (es == 4) {
        AVS_REQUIRE(B.pwords32, AVS_EINTERNAL, "brick form: no 4-byte pattern image for the float kernel");
        B.pwords = B.pwords32;
    }
#ifdef AVS_PROBES
    const int dbg = getenv("AVS_BRICK_DEBUG") ? atoi(getenv("AVS_BRICK_DEBUG")) : 0; // phase switches / stamps (measurement builds only; read per launch: a probe script flips it)
    B.debug |= dbg;
#endif
    if (B.vc) {
        if (partial) hipLaunchKernelGGL((k_spmv_brick<true, true, T>), dim3(grid), dim3(kBrickBlk), lds, stream, B, x, y, partial, done_flag);
        else hipLaunchKernelGGL((k_spmv_brick<false, true, T>), dim3(grid), dim3(kBrickBlk), lds, stream, B, x, y, partial, done_flag);
    } else if (partial) hipLaunchKernelGGL((k_spmv_brick<true, false, T>), dim3(grid), dim3(kBrickBlk), lds, stream, B, x, y, partial, done_flag);
    else hipLaunchKernelGGL((k_spmv_brick<false, false, T>), dim3(grid), dim3(kBrickBlk), lds, stream, B, x, y, partial, done_flag);
    AVS_HIP(hipGetLastError());
#ifdef AVS_PROBES
    if (B.debug & 64) { // print the phase stamps of THIS launch (synchronises: not for timing loops)
        AVS_HIP(hipStreamSynchronize(stream));
        brick_print_stamps();
    }
#endif
    return AVS_OK;
}
avs_status spmv_brick_launch(const BrickView &B, const double *x, double *y, double *partial, const int *done_flag, hipStream_t stream)
{
    return spmv_brick_launch_t<double>(B, x, y, partial, done_flag, stream);
}
avs_status spmv_brick_launch_f32(const BrickView &B, const float *x, float *y, double *partial, const int *done_flag, hipStream_t stream)
{
    return spmv_brick_launch_t<float>(B, x, y, partial, done_flag, stream);
}

} // namespace avs

#ifdef AVS_PROBES
// Measurement / test entry: SpMV on a brick form whose arrays the caller built (tools/brick_build.py: the torch reference builder the
// device builder is tested against).  Device pointers.
extern "C" avs_status avs_brick_spmv_probe(const avs_brick_arrays *a, const double *x, double *y, double *partial, int32_t repeats,
                                           void *stream, double *ms_per_launch)
{
    AVS_REQUIRE(a && x && y && repeats > 0, AVS_EINVAL, "bad argument");
    AVS_REQUIRE(a->table_size > 0 && a->table_size < avs::kBrickTableMax, AVS_EINVAL, "value table size %d out of range", a->table_size);
    avs::BrickView B;
    B.ntiles = a->ntiles;
    B.tile_blk = reinterpret_cast<const uint2 *>(a->tile_blk);
    B.blocks = a->blocks;
    B.rdesc = reinterpret_cast<const uint2 *>(a->rdesc);
    B.ownslot = a->ownslot;
    B.pwords = a->pwords;
    B.sdesc = reinterpret_cast<const uint2 *>(a->sdesc);
    B.swords = a->swords;
    B.table = a->table;
    B.table_size = a->table_size;
    B.col_bits = a->col_bits;
    if (const char *e = getenv("AVS_BRICK_DEBUG")) B.debug = atoi(e);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    AVS_TRY(avs::spmv_brick_launch(B, x, y, partial, nullptr, st));
    avs::Timer t(st);
    t.start();
    for (int i = 0; i < repeats; ++i) AVS_TRY(avs::spmv_brick_launch(B, x, y, partial, nullptr, st));
    const double ms = t.stop() / repeats;
    if (ms_per_launch) *ms_per_launch = ms;
    if (B.debug & 16) avs::brick_print_stamps();
    return AVS_OK;
}
#endif // AVS_PROBES
