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
// plain CSR kernel.  Reference: Eigen's `mat * p` inside ConjugateGradient (cpp:618-630 of HDK_AdaptiveViscosity.cpp); the row
// structure is that of cpp:2537-2745.
#include "avs_internal.hpp"

namespace avs {

constexpr int kBrickBlk = 512;

// measurement only (AVS_BRICK_DEBUG & 16): wall_clock64 stamps of workgroup phases, 8 per tile, first kStampTiles tiles of every workgroup
constexpr int kStampTiles = 24, kStampWgs = 1024;
__device__ long long g_brick_stamps[kStampWgs * kStampTiles * 8];
#define BRICK_STAMP(slot)                                                                                            \
    do {                                                                                                             \
        if ((B.debug & 16) && tid == 0 && iter < kStampTiles && blockIdx.x < kStampWgs) {                             \
            if ((slot) == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
            g_brick_stamps[((int)blockIdx.x * kStampTiles + iter) * 8 + (slot)] = wall_clock64();                    \
        }                                                                                                            \
    } while (0)

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

__device__ __forceinline__ double lds_f64(const double *base, unsigned byte_off)
{
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off);
}

constexpr int kBrickRpt = kBrickMaxRows / kBrickBlk;                    // rows per thread
constexpr int kBrickRu = 6;                                             // halo fill runs per quarter wave held in registers
constexpr int kBrickPq = (kBrickPatWords / 4 + kBrickBlk - 1) / kBrickBlk; // pattern quads per thread
constexpr int kBrickSu = kBrickPark / kBrickBlk;                        // words per thread of a G tile's first streamed pass

struct BrickHdr { int row0, nrows, npat, pat0, run0, nruns, pq0, npq, srow0, nsrows, sword0, nsw, rd0, nprow; };
// what a tile's x / pattern loads need (requested one tile ahead, while the previous tile is multiplied)
struct BrickAhead {
    uint32_t rdsc[kBrickRu];
    uint32_t pqo[kBrickPq];
    uint32_t w0[kBrickSu];
    uint32_t pinf;
    int nbreg;
};

__device__ __forceinline__ BrickHdr brick_load_hdr(const BrickTile &T)
{
    BrickHdr h;
    h.row0 = T.row0; h.nrows = T.nrows; h.npat = T.npat; h.pat0 = T.pat0; h.run0 = T.run0; h.nruns = T.nruns; h.pq0 = T.pq0; h.npq = T.npq;
    h.srow0 = T.srow0; h.nsrows = T.nsrows; h.sword0 = T.sword0; h.nsw = T.nsw; h.rd0 = T.rd0; h.nprow = T.nprow;
    return h;
}

__device__ __forceinline__ void brick_load_ahead(const BrickView &B, const BrickTile &T, const BrickHdr &h, int tid, BrickAhead &d)
{
    const bool emode = h.npat == 0;
#pragma unroll
    for (int u = 0; u < kBrickSu; ++u) {
        const int e = tid + u * kBrickBlk;
        d.w0[u] = 0u;
        if (!emode && u * kBrickBlk < h.nsw) // block-uniform: a batch nobody needs is not requested (a load instruction costs the same
                                             // address-unit time with one active lane as with 64)
        d.w0[u] = B.swords[(int64_t)h.sword0 + ((!emode && e < h.nsw) ? e : 0)]; // (unconditional loads from a valid address: a conditional
                                                                                   //  load is a branch + s_waitcnt vmcnt(0) per load)
    }
    d.nbreg = T.nb[tid & 31];                                           // first rows of the 27 neighbour bricks, one per lane
    const int qw = tid >> 4;
#pragma unroll
    for (int u = 0; u < kBrickRu; ++u) {
        const int q = u * (kBrickBlk / 16) + qw;
        d.rdsc[u] = 0xffffffffu;
        if (u * (kBrickBlk / 16) < h.nruns) {
            const uint32_t r = B.runs[h.run0 + (q < h.nruns ? q : 0)];
            d.rdsc[u] = (q < h.nruns) ? r : 0xffffffffu;
        }
    }
#pragma unroll
    for (int u = 0; u < kBrickPq; ++u) {
        const int q = tid + u * kBrickBlk;
        d.pqo[u] = 0u;
        if (u * kBrickBlk < h.npq) d.pqo[u] = B.pquads[h.pq0 + (q < h.npq ? q : 0)];
    }
    d.pinf = 0u;
    if (h.npat > 0) d.pinf = B.pinfo[h.pat0 + (tid < h.npat ? tid : 0)];
}

template <bool DOT, int PERSIST>
__global__ __launch_bounds__(kBrickBlk) void k_spmv_brick(BrickView B, const double *__restrict__ x, double *__restrict__ y,
                                                         double *__restrict__ partial, const int *__restrict__ done_flag)
{
    if (DOT && done_flag && *done_flag) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *xs = smem;                                                  // kBrickSlotsPad
    double *vals = smem + kBrickSlotsPad;                               // table_size + 1 (the last entry is 0.0: padding words), even
    double *park = vals + ((B.table_size + 2) & ~1);                    // kBrickPark products of streamed rows
    uint32_t *pw = reinterpret_cast<uint32_t *>(park + kBrickPark);     // kBrickPatWords + 8
    uint32_t *pinfo = pw + kBrickPatWords + 8;                          // kBrickPatMax: local start | quads << 16 | simple << 31
    const int tid = threadIdx.x;
    const int lane = tid & 63, l16 = tid & 15, qw = tid >> 4;
    const unsigned cmask = (1u << B.col_bits) - 1u;
    const int cbits = B.col_bits;
    constexpr int QW = kBrickBlk / 16;

    for (int i = tid; i <= B.table_size; i += kBrickBlk) vals[i] = (i < B.table_size) ? B.table[i] : 0.; // once per workgroup

    int tile = blockIdx.x;
    if (tile >= B.ntiles) return;
    BrickHdr h = brick_load_hdr(B.tiles[tile]);
    BrickAhead d;
    brick_load_ahead(B, B.tiles[tile], h, tid, d);

    int iter = 0;
    for (;;) {
        BRICK_STAMP(0);
        const bool emode = h.npat == 0;                 // no pattern rows: the products of the streamed rows may use the x lattice's LDS
        double *prod = emode ? xs : park;
        const int cap = emode ? kBrickSlotsPad : kBrickPark;
        // ---- this tile's loads, one round trip: x of the halo runs (a run's neighbour-brick base comes from the lane that holds it), x of
        //      the tile's own rows with their slots, the pattern quads, x of the first streamed pass, the row descriptors
        double fv[kBrickRu];
#pragma unroll
        for (int u = 0; u < kBrickRu; ++u) {
            const uint32_t r = d.rdsc[u];
            const int nbase = __builtin_amdgcn_ds_bpermute((int)(r >> 27) << 2, d.nbreg);
            const int len = (int)(r & 15u) + 1;
            const bool on = (r != 0xffffffffu) && l16 < len && !(B.debug & 1);
            fv[u] = 0.;
            if (u * QW < h.nruns) fv[u] = x[on ? (int64_t)nbase + (int)((r >> 16) & 0x7ffu) + l16 : (int64_t)h.row0];
        }
        double xo[kBrickRpt];
        uint32_t os[kBrickRpt];
#pragma unroll
        for (int k = 0; k < kBrickRpt; ++k) {
            const int r = tid + k * kBrickBlk;
            const bool on = !emode && r < h.nrows;
            xo[k] = 0.;
            os[k] = 0xffffu;
            if (!emode && k * kBrickBlk < h.nrows) {
                xo[k] = x[(int64_t)h.row0 + (on ? r : 0)];
                const uint32_t o = B.ownslot[(int64_t)h.row0 + (on ? r : 0)];
                os[k] = on ? o : 0xffffu;
            }
        }
        uint4 pqv[kBrickPq];
#pragma unroll
        for (int u = 0; u < kBrickPq; ++u) {
            const int q = tid + u * kBrickBlk;
            pqv[u] = uint4{0u, 0u, 0u, 0u};
            if (u * kBrickBlk < h.npq) pqv[u] = *reinterpret_cast<const uint4 *>(B.pwords + (q < h.npq ? d.pqo[u] : 0u));
        }
        double xv0[kBrickSu];
#pragma unroll
        for (int u = 0; u < kBrickSu; ++u) {
            const int e = tid + u * kBrickBlk;
            xv0[u] = 0.;
            if (!emode && u * kBrickBlk < h.nsw) xv0[u] = x[(e < h.nsw) ? (d.w0[u] & cmask) : (uint32_t)h.row0];
        }
        uint32_t rd[kBrickRpt], ro[kBrickRpt];
#pragma unroll
        for (int k = 0; k < kBrickRpt; ++k) {
            const int i = tid + k * kBrickBlk;
            rd[k] = 0u;
            ro[k] = 0u;
            if (k * kBrickBlk < h.nprow) {
                rd[k] = B.rdesc[h.rd0 + (i < h.nprow ? i : 0)];
                ro[k] = (uint32_t)B.rorder[h.rd0 + (i < h.nprow ? i : 0)];
            }
        }
        uint2 sd[kBrickRpt];
#pragma unroll
        for (int k = 0; k < kBrickRpt; ++k) {
            const int i = brick_srow_of_thread(tid, k);
            sd[k] = uint2{0u, 0u};
            if (k * kBrickBlk < h.nsrows) {
                const uint2 t = B.sdesc[h.srow0 + (i < h.nsrows ? i : 0)];
                sd[k] = (i < h.nsrows) ? t : uint2{0u, 0u};
            }
        }
        // the next tile's header: scalar loads, in flight while this tile's data arrives
        const int tnext = tile + (int)gridDim.x;
        const bool more = PERSIST != 0 && tnext < B.ntiles;
        const BrickTile &Tn = B.tiles[more ? tnext : tile];
        const BrickHdr hn = brick_load_hdr(Tn);
        BRICK_STAMP(1);
        // ---- LDS writes
        if (tid < h.npat) pinfo[tid] = d.pinf;
#pragma unroll
        for (int u = 0; u < kBrickPq; ++u) {
            const int q = tid + u * kBrickBlk;
            if (q < h.npq) reinterpret_cast<uint4 *>(pw)[q] = pqv[u];
        }
#pragma unroll
        for (int u = 0; u < kBrickRu; ++u) {
            const uint32_t r = d.rdsc[u];
            const int len = (int)(r & 15u) + 1;
            if (r != 0xffffffffu && l16 < len) xs[((r >> 4) & 0xfffu) + l16] = fv[u];
        }
#pragma unroll
        for (int k = 0; k < kBrickRpt; ++k)
            if (os[k] != 0xffffu) xs[os[k]] = xo[k];
        // tiles with more halo runs than the registers hold (few): the rest in a plain loop
        for (int q0 = kBrickRu * QW; q0 < h.nruns; q0 += QW) {
            const int q = q0 + qw;
            const uint32_t r = B.runs[h.run0 + (q < h.nruns ? q : 0)];
            const int nbase = __builtin_amdgcn_ds_bpermute((int)(r >> 27) << 2, d.nbreg);
            const int len = (int)(r & 15u) + 1;
            const bool on = q < h.nruns && l16 < len;
            const double xv1 = x[on ? (int64_t)nbase + (int)((r >> 16) & 0x7ffu) + l16 : (int64_t)h.row0];
            if (on) xs[((r >> 4) & 0xfffu) + l16] = xv1;
        }
        const uint32_t w0a = d.w0[0], w0b = kBrickSu > 1 ? d.w0[kBrickSu - 1] : 0u;
        __syncthreads();
        BRICK_STAMP(2);

        // ---- what the next tile's loads need travels while this tile is multiplied
        if (PERSIST == 1) brick_load_ahead(B, Tn, hn, tid, d);

        double dot = 0.;
        if (!emode) {
            // products of the first streamed pass
            static_assert(kBrickSu == 2, "two words per thread in the first streamed pass");
            if (tid < h.nsw) prod[tid] = vals[w0a >> cbits] * xv0[0];
            if (tid + kBrickBlk < h.nsw) prod[tid + kBrickBlk] = vals[w0b >> cbits] * xv0[1];
            // pattern rows: one lane per row, everything from LDS
#pragma unroll
            for (int k = 0; k < kBrickRpt; ++k) {
                if (k * kBrickBlk >= h.nprow) break; // block-uniform
                const int i = tid + k * kBrickBlk;
                if (i < h.nprow && !(B.debug & 2)) {
                    const uint32_t rdv = rd[k];
                    const unsigned pid = rdv >> 20;
                    const int lr = (int)((rdv >> 18) & 3u), ax = (int)((rdv >> 16) & 3u);
                    const int cx = (int)(rdv & 15u) - 1, cy = (int)((rdv >> 4) & 15u) - 1, cz = (int)((rdv >> 8) & 15u) - 1; // local level-lr cell
                    unsigned b8[4]; // byte offsets of the row's base on the four lattices
#pragma unroll
                    for (int lc = 0; lc < 4; ++lc) {
                        const int up = lc > lr ? lc - lr : 0, dn2 = lr > lc ? lr - lc : 0;
                        const int S = (8 >> lc) + 2;
                        const int bx = ((cx >> up) << dn2) + 1, by = ((cy >> up) << dn2) + 1, bz = ((cz >> up) << dn2) + 1;
                        b8[lc] = (unsigned)((kBrickLoff[lc] + ((bz * S + by) * S + bx) * 3) * 8);
                    }
                    const unsigned own8 = (lr == 0 ? b8[0] : lr == 1 ? b8[1] : lr == 2 ? b8[2] : b8[3]) + 8u * (unsigned)ax;
                    const unsigned P01 = (b8[0] & 0xffffu) | (b8[1] << 16), P23 = (b8[2] & 0xffffu) | (b8[3] << 16);
                    const unsigned pi = pinfo[pid];
                    const uint4 *wq = reinterpret_cast<const uint4 *>(pw + (pi & 0xffffu));
                    const int nq = (int)((pi >> 16) & 0x7fffu);
                    // word: delta << 19 (signed 13) | 000 | lattice level << 14 | code << 3.  A pattern is padded to whole quads with
                    // words that repeat its first entry's slot with the code of 0.0: +-0.0 added to a sum that is never -0.0.
                    auto addr = [&](uint32_t w) -> unsigned {
                        const unsigned sel = ((w >> 14) & 3u) * 0x0202u + 0x0c0c0100u;
                        return (unsigned)((int)w >> 16) + __builtin_amdgcn_perm(P23, P01, sel); // + 16-bit field number `level` of P23:P01
                    };
                    double sum = 0.;
                    // software pipeline: the words of quad q + 1 and the value / x reads of quad q are in flight while quad q - 1 is added
                    // (reading the words one quad past the pattern is harmless: the LDS image ends with spare quads; they are never decoded)
                    double v0, v1, v2, v3, x0, x1, x2, x3;
                    uint4 w = wq[0];
                    uint4 wn = wq[1];
                    v0 = lds_f64(vals, w.x & 0x3ff8u); x0 = lds_f64(xs, addr(w.x));
                    v1 = lds_f64(vals, w.y & 0x3ff8u); x1 = lds_f64(xs, addr(w.y));
                    v2 = lds_f64(vals, w.z & 0x3ff8u); x2 = lds_f64(xs, addr(w.z));
                    v3 = lds_f64(vals, w.w & 0x3ff8u); x3 = lds_f64(xs, addr(w.w));
                    for (int q = 1; q < nq; ++q) {
                        const uint4 wnn = wq[q + 1];
                        const double a0 = lds_f64(vals, wn.x & 0x3ff8u), c0 = lds_f64(xs, addr(wn.x));
                        const double a1 = lds_f64(vals, wn.y & 0x3ff8u), c1 = lds_f64(xs, addr(wn.y));
                        const double a2 = lds_f64(vals, wn.z & 0x3ff8u), c2 = lds_f64(xs, addr(wn.z));
                        const double a3 = lds_f64(vals, wn.w & 0x3ff8u), c3 = lds_f64(xs, addr(wn.w));
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
                    y[(int64_t)h.row0 + (int)ro[k]] = sum;
                    if (DOT) dot += sum * lds_f64(xs, own8);
                }
            }
        }
        BRICK_STAMP(3);
        // streamed rows: passes of `cap` products parked in LDS, then every row adds its segment left to right
        if (h.nsw > 0 && !(B.debug & 4)) {
            double ssum[kBrickRpt];
#pragma unroll
            for (int k = 0; k < kBrickRpt; ++k) ssum[k] = 0.;
            for (int ts = 0; ts < h.nsw; ts += cap) {
                const int te = (ts + cap < h.nsw) ? ts + cap : h.nsw;
                if (emode || ts > 0) {
                    if (ts > 0) __syncthreads(); // the previous pass has been summed
                    for (int e0 = ts + tid; e0 < te; e0 += 4 * kBrickBlk) {
                        uint32_t w4[4];
                        double x4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int e = e0 + u * kBrickBlk;
                            w4[u] = B.swords[(int64_t)h.sword0 + (e < te ? e : ts)];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int e = e0 + u * kBrickBlk;
                            x4[u] = x[w4[u] & cmask];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int e = e0 + u * kBrickBlk;
                            if (e < te) prod[e - ts] = vals[w4[u] >> cbits] * x4[u];
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < kBrickRpt; ++k) {
                    const int len = (int)(sd[k].x >> 16), st = (int)sd[k].y;
                    const int a = st > ts ? st : ts, b = (st + len < te) ? st + len : te;
                    for (int j = a; j < b; ++j) ssum[k] += prod[j - ts];
                }
            }
#pragma unroll
            for (int k = 0; k < kBrickRpt; ++k) {
                const int len = (int)(sd[k].x >> 16);
                if (len > 0) {
                    const int64_t row = (int64_t)h.row0 + (int)(sd[k].x & 0xffffu);
                    y[row] = ssum[k];
                    if (DOT) dot += ssum[k] * x[row];
                }
            }
        }
        if (DOT) {
            const double dsum = brick_wave_sum_dpp(dot);
            if (lane == 63) partial[(int64_t)tile * (kBrickBlk / 64) + (tid >> 6)] = dsum;
        }
        BRICK_STAMP(4);
        if (!more) break;
        __syncthreads(); // every wave is done with this tile's LDS
        BRICK_STAMP(5);
        ++iter;
        tile = tnext;
        h = hn;
        if (PERSIST == 2) brick_load_ahead(B, B.tiles[tile], h, tid, d);
    }
}

size_t brick_lds_bytes(const BrickView &B)
{
    return (size_t)(kBrickSlotsPad + ((B.table_size + 2) & ~1) + kBrickPark) * sizeof(double) + (size_t)(kBrickPatWords + 8 + kBrickPatMax) * sizeof(uint32_t);
}

// persistent grid: as many workgroups as the device keeps resident (queried once per process and device)
static int brick_grid(const BrickView &B, size_t lds)
{
    static int cached[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!cached[dev]) {
        int per_cu = 0, cus = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_spmv_brick<true, 1>, kBrickBlk, lds);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (per_cu < 1) per_cu = 1;
        if (cus < 1) cus = 256;
        cached[dev] = per_cu * cus;
    }
    int g = cached[dev];
    if (const char *e = getenv("AVS_BRICK_GRID")) g = atoi(e) > 0 ? atoi(e) : g;
    return g < B.ntiles ? g : B.ntiles;
}

avs_status spmv_brick_launch(const BrickView &B, const double *x, double *y, double *partial, const int *done_flag, hipStream_t stream)
{
    if (B.ntiles <= 0) return AVS_OK;
    const size_t lds = brick_lds_bytes(B);
    static bool attr_set = false;
    if (lds > 48 * 1024 && !attr_set) {
        AVS_HIP(hipFuncSetAttribute((const void *)k_spmv_brick<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        AVS_HIP(hipFuncSetAttribute((const void *)k_spmv_brick<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        AVS_HIP(hipFuncSetAttribute((const void *)k_spmv_brick<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        AVS_HIP(hipFuncSetAttribute((const void *)k_spmv_brick<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        AVS_HIP(hipFuncSetAttribute((const void *)k_spmv_brick<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        AVS_HIP(hipFuncSetAttribute((const void *)k_spmv_brick<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr_set = true;
    }
    int persist = 1;
    if (const char *e = getenv("AVS_BRICK_PERSIST")) persist = atoi(e);
    if (persist == 2) {
        int per_cu = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_spmv_brick<true, 2>, kBrickBlk, lds);
        int grid = 256 * (per_cu > 0 ? per_cu : 1);
        if (const char *e = getenv("AVS_BRICK_GRID")) grid = atoi(e);
        if (grid > B.ntiles) grid = B.ntiles;
        if (partial) hipLaunchKernelGGL((k_spmv_brick<true, 2>), dim3(grid), dim3(kBrickBlk), lds, stream, B, x, y, partial, done_flag);
        else hipLaunchKernelGGL((k_spmv_brick<false, 2>), dim3(grid), dim3(kBrickBlk), lds, stream, B, x, y, partial, done_flag);
    } else if (persist) {
        const int grid = brick_grid(B, lds);
        if (partial) hipLaunchKernelGGL((k_spmv_brick<true, 1>), dim3(grid), dim3(kBrickBlk), lds, stream, B, x, y, partial, done_flag);
        else hipLaunchKernelGGL((k_spmv_brick<false, 1>), dim3(grid), dim3(kBrickBlk), lds, stream, B, x, y, partial, done_flag);
    } else {
        if (partial) hipLaunchKernelGGL((k_spmv_brick<true, 0>), dim3(B.ntiles), dim3(kBrickBlk), lds, stream, B, x, y, partial, done_flag);
        else hipLaunchKernelGGL((k_spmv_brick<false, 0>), dim3(B.ntiles), dim3(kBrickBlk), lds, stream, B, x, y, partial, done_flag);
    }
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

} // namespace avs

// Measurement / test entry: SpMV on a brick form whose arrays the caller built (tools/brick_build.py: the torch reference builder the
// device builder is tested against).  Device pointers.
extern "C" avs_status avs_brick_spmv_probe(const avs_brick_arrays *a, const double *x, double *y, double *partial, int32_t repeats,
                                           void *stream, double *ms_per_launch)
{
    AVS_REQUIRE(a && x && y && repeats > 0, AVS_EINVAL, "bad argument");
    AVS_REQUIRE(a->table_size > 0 && a->table_size < avs::kBrickTableMax, AVS_EINVAL, "value table size %d out of range", a->table_size);
    static_assert(sizeof(avs::BrickTile) == 192, "tile header layout");
    avs::BrickView B;
    B.ntiles = a->ntiles;
    B.tiles = reinterpret_cast<const avs::BrickTile *>(a->tiles);
    B.rdesc = a->rdesc;
    B.rorder = a->rorder;
    B.ownslot = a->ownslot;
    B.runs = a->runs;
    B.pquads = a->pquads;
    B.pinfo = a->pinfo;
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
    if (B.debug & 16) { // phase times of the last launch, in 10-ns ticks: mean over workgroups and tiles
        std::vector<long long> hs((size_t)avs::kStampWgs * avs::kStampTiles * 8);
        AVS_HIP(hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(avs::g_brick_stamps), hs.size() * sizeof(long long)));
        double acc[6] = {};
        long cnt = 0;
        for (int w = 0; w < avs::kStampWgs; ++w)
            for (int i = 1; i + 1 < avs::kStampTiles; ++i) {
                const long long *p = &hs[((size_t)w * avs::kStampTiles + i) * 8], *pn = p + 8;
                if (!p[0] || !p[5] || !pn[0]) continue;
                acc[0] += (double)(p[1] - p[0]); acc[1] += (double)(p[2] - p[1]); acc[2] += (double)(p[3] - p[2]);
                acc[3] += (double)(p[4] - p[3]); acc[4] += (double)(p[5] - p[4]); acc[5] += (double)(pn[0] - p[0]);
                ++cnt;
            }
        if (cnt)
            fprintf(stderr, "brick phases (us, mean of %ld tiles): wait data %.2f | lds writes + barrier %.2f | pattern rows %.2f | streamed %.2f | barrier B %.2f | tile %.2f\n",
                    cnt, acc[0] / cnt / 100, acc[1] / cnt / 100, acc[2] / cnt / 100, acc[3] / cnt / 100, acc[4] / cnt / 100, acc[5] / cnt / 100);
    }
    return AVS_OK;
}
