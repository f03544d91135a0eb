// avs_pcg.hip -- device-resident Jacobi-PCG + CSR SpMV for gfx950 (MI355X).
//
// Replaces cpp:611-643 of the reference (HDK_AdaptiveViscosity.cpp): Eigen's
// ConjugateGradient<SparseMatrix<double>, Lower|Upper> with DiagonalPreconditioner and
// solveWithGuess.  The iteration follows Eigen 3.3/3.4 ConjugateGradient.h operation by operation
// (SURVEY.md 8(c)); only the order of additions inside dot products differs (block-wise tree).
//
// Roofline: every kernel here is HBM-bound.  Algorithmic bytes per SpMV launch:
//   12*nnz + 4*(n+1) + 16*n      (fp64 values, int32 columns / row pointers)
//
// SpMV kernels
//   variant 1  "stream": a 256-thread workgroup owns 256 consecutive rows.  Phase 1 streams the
//              rows' (val, col) pairs with fully coalesced loads, gathers x[col] and parks the
//              products in LDS; phase 2: one thread per row adds its LDS segment left to right
//              (the same order as the CPU oracle => bit-identical y).  Row stride ~15 doubles is
//              bank-conflict-free for ds_read_b64 (30*t mod 64 distinct for t < 32).
//   variant 2-4 "vector": 4 / 8 / 16 lanes per row, shuffle reduction.
//
// CG scalars never leave the device.  The host enqueues iterations in chunks and polls a
// `done` flag once per chunk; kernels of iterations enqueued past convergence exit immediately, so
// the iteration count and the solution are exactly those of the sequential algorithm.
#include "avs_internal.hpp"
#include "avs_halo.hpp"

namespace avs {

static constexpr int kBlock = 256;
static constexpr int kVecGrid = 2048;        // grid-stride vector kernels: 8 blocks per CU
static constexpr int kStreamCap = 4096;      // LDS products per tile (32 KiB)
static constexpr int kChunk = 32;            // iterations enqueued between host polls
static constexpr int kTimedChunkEvery = 8; // hipGraph replay: 1 chunk in 8 is enqueued launch by launch with timing events
static constexpr int kFuseAlphaMax = 4096;   // single-GPU loop: SpMV partial sums that k_update_r's workgroups fold themselves (<= 262 k rows;
                                             // 5.9 k partials at 380 k rows: 27.1 -> 25.4 k it/s, so not beyond)
static constexpr int kSampleEvery = 4;       // SpMV launches bracketed by HIP events: every 4th (events are not free)

struct PcgWork {
    int64_t n = 0, n_ext = 0;
    DevBuf<double> r, p, t, invd, partial;
    DevBuf<uint16_t> dcode;  // value-indexed matrix: code of every row's diagonal entry ...
    DevBuf<double> invtab;   // ... into the table of inverted values (2 B instead of 8 B per row and vector pass)
    DevBuf<float> f_x, f_r, f_p, f_t, f_b, f_invd, f_invtab; // float-vector loop of AVS_PRECISION_F32 contexts (avs_pcg_f32.inl)
    DevBuf<double> x_save;   // the initial guess while the CU-resident loop runs (restored if it faults)
    DevBuf<double> cancel_word; // partitioned solves: [0] this rank's avs_cancel request as the kernels / the all-reduce see it (0. / 1.)
    DevBuf<int> cancel_dev;     // ... and as an int for the finalizer of the direct transport
    int resident_faults = 0; // CU-resident launches of this workspace that ended in a timed-out wait
    DevBuf<unsigned long long> fused_bar; // k_update_fused: the grid barrier's ticket counter (monotonic)
    bool fused_off = false;  // ... a launch of it timed out at its barrier on this workspace: the two-launch form from then on
    int fused_faults = 0, fused_used = 0;
    DevBuf<double> s, u; // single-reduction variant (multi-GPU): s = A p recurrence, u = M^-1 r with halo tail
    DevBuf<PcgScalars> sc;
    DevBuf<double> stage2;         // direct transport: per-workgroup SpMV sums of the halo-touching launch
    DevBuf<double> stage;          // multi-block reduction: kRedBlocks x 4 block sums ...
    DevBuf<unsigned> ticket;       // ... and the arrival counter (reset by the last block)
    PcgScalars *host_sc = nullptr; // pinned
    // one chunk of kChunk iterations captured as a hipGraph (single-GPU loop): relaunched while the key matches
    hipGraphExec_t graph = nullptr;
    const void *graph_key[10] = {};
    double graph_tol = 0.;
    bool graph_broken = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evA[kChunk] = {}, evB[kChunk] = {}; // per-launch SpMV timing inside the solve
    size_t npartial = 0;
    struct ResidentPlan *resident = nullptr; // CU-resident loop (avs_pcg_resident.inl): lane plan of the current matrix
    int resident_used = 0;                   // the last solve ran it
};

// ---------------------------------------------------------------------------------------------
// block reduction helpers (wave64 shuffles, then LDS across the 4 waves of a 256-thread block)
// ---------------------------------------------------------------------------------------------
// wave64 sum without the LDS crossbar: inclusive scan inside every row of 16 lanes (row_shr 1, 2, 4, 8), then
// row_bcast15 / row_bcast31 carry the row totals forward; lane 63 ends up with the sum of all 64 lanes (fixed order)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo); // lanes without a source (or masked rows) add +0.0
}

__device__ __forceinline__ double wave_sum_dpp(double v) // result in lane 63
{
    v = dpp_add<0x111, 0xf>(v); // row_shr:1
    v = dpp_add<0x112, 0xf>(v); // row_shr:2
    v = dpp_add<0x114, 0xf>(v); // row_shr:4
    v = dpp_add<0x118, 0xf>(v); // row_shr:8
    v = dpp_add<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ double block_sum(double v, double *lds4)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds4[wave] = v;
    __syncthreads();
    double s = 0.;
    if (threadIdx.x == 0) s = ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
    return s; // valid in thread 0
}

// ---------------------------------------------------------------------------------------------
// SpMV: stream variant
// ---------------------------------------------------------------------------------------------
template <bool DOT>
__global__ __launch_bounds__(kBlock) void k_spmv_stream(CsrView A, const double *__restrict__ x,
                                                        double *__restrict__ y,
                                                        double *__restrict__ partial,
                                                        const PcgScalars *sc)
{
    if (DOT && sc && sc->done) return;
    __shared__ double prod[kStreamCap];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * kBlock;
    const int64_t row = row0 + tid;
    const int64_t rlast = (row0 + kBlock < A.n) ? row0 + kBlock : A.n;
    const int s_blk = A.row_ptr[row0];
    const int e_blk = A.row_ptr[rlast];
    int rs = 0, re = 0;
    if (row < A.n) {
        rs = A.row_ptr[row];
        re = A.row_ptr[row + 1];
    }
    double sum = 0.;
    for (int ts = s_blk; ts < e_blk; ts += kStreamCap) {
        const int te = (ts + kStreamCap < e_blk) ? ts + kStreamCap : e_blk;
        // phase 1: coalesced stream of (val, col), gather x, products to LDS.  4 independent
        // loads per thread in flight per trip.
        int k = ts + tid;
        for (; k + 3 * kBlock < te; k += 4 * kBlock) {
            const double v0 = A.val[k], v1 = A.val[k + kBlock], v2 = A.val[k + 2 * kBlock], v3 = A.val[k + 3 * kBlock];
            const int c0 = A.col[k], c1 = A.col[k + kBlock], c2 = A.col[k + 2 * kBlock], c3 = A.col[k + 3 * kBlock];
            const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
            prod[k - ts] = v0 * x0;
            prod[k - ts + kBlock] = v1 * x1;
            prod[k - ts + 2 * kBlock] = v2 * x2;
            prod[k - ts + 3 * kBlock] = v3 * x3;
        }
        for (; k < te; k += kBlock) prod[k - ts] = A.val[k] * x[A.col[k]];
        __syncthreads();
        // phase 2: each row adds its part of the tile, left to right
        const int a = rs > ts ? rs : ts;
        const int b = re < te ? re : te;
        for (int j = a; j < b; ++j) sum += prod[j - ts];
        __syncthreads();
    }
    if (row < A.n) y[row] = sum;
    if (DOT) {
        double d = (row < A.n) ? sum * x[row] : 0.;
        d = block_sum(d, red);
        if (tid == 0) partial[blockIdx.x] = d;
    }
}


// ---------------------------------------------------------------------------------------------
// SpMV: generic tile kernel for tuning.  One block = BLK rows; knobs: LDS capacity CAP, 16-B/8-B
// vector streaming (VEC), XCD-contiguous tile ownership (XCD: block b -> XCD b % 8 by the observed
// dispatch rule; each XCD then owns a contiguous eighth of the rows so x gathers of neighbouring
// tiles share one L2 -- used for speed only, never for correctness).
// ---------------------------------------------------------------------------------------------
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int i2_t __attribute__((ext_vector_type(2)));

// matrix words of the value-indexed kernels: non-temporal when the matrix is larger than the caches can hold between two products (the
// read-once stream must not evict x from L2), plain when it may stay in the 256 MB Infinity Cache from one iteration to the next
template <bool KEEP, typename T>
__device__ __forceinline__ T stream_load_k(const T *p)
{
    if (KEEP) return *p;
    return __builtin_nontemporal_load(p);
}
template <bool KEEP, typename T>
__device__ __forceinline__ void stream_store_k(T v, T *p)
{
    if (KEEP) *p = v;
    else __builtin_nontemporal_store(v, p);
}
// (the choice is a TEMPLATE parameter: with a run-time flag the optimiser merges the plain and the hinted load of one address into one
// plain load and the hint is gone)
template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T *p)
{
    if (NT) return __builtin_nontemporal_load(p); // read once: do not let the stream evict x from L2
    return *p;
}

template <int BLK, int CAP, bool DOT, bool VEC, bool XCD, bool NT, bool HALO = false>
__global__ __launch_bounds__(BLK) void k_spmv_tile(CsrView A, const double *__restrict__ xin,
                                                   double *__restrict__ y, double *__restrict__ partial,
                                                   const PcgScalars *sc, int chunk, const int32_t *__restrict__ tiles = nullptr,
                                                   HaloView hv = HaloView())
{
    if (DOT && sc && sc->done) return;
    __shared__ double prod[CAP + 2];
    __shared__ double red[BLK / 64];
    const int tid = threadIdx.x;
    // halo-touching launch of the direct transport: columns >= n_own live in the comm block's halo area
    struct Gather {
        const double *x, *hx;
        int n_own;
        __device__ __forceinline__ double operator[](int c) const { return (HALO && c >= n_own) ? ld_sys_f64(hx + (c - n_own)) : x[c]; }
    };
    const Gather x{xin, HALO ? hv.dd->my_halo : nullptr, HALO ? (int)hv.dd->n_own : 0};
    if (HALO) {
        if ((int)blockIdx.x >= hv.ntiles) { halo_finalizer<BLK>(hv, (int)blockIdx.x - hv.ntiles); return; }
        if (hv.tile_bnd[blockIdx.x]) halo_wait(hv); // block-uniform
    }
    int64_t tile = tiles ? (int64_t)tiles[blockIdx.x] : (int64_t)blockIdx.x; // tile lists: interior / halo-touching subsets
    if (XCD) {
        const int64_t ntiles = gridDim.x;
        if (chunk <= 0) { // each XCD owns one contiguous eighth
            const int64_t per = ntiles >> 3, rem = ntiles & 7;
            const int q = blockIdx.x & 7;
            const int64_t slot = blockIdx.x >> 3;
            tile = (int64_t)q * per + (q < rem ? q : rem) + slot;
        } else { // XCDs take turns on chunks of `chunk` consecutive tiles (one moving window)
            const int64_t super = (int64_t)chunk * 8;
            const int64_t full = (ntiles / super) * super; // tail keeps the identity mapping
            if (tile < full) {
                const int64_t sb = tile / super, r = tile % super;
                const int q = (int)(r & 7);
                const int64_t slot = r >> 3;
                tile = sb * super + (int64_t)q * chunk + slot;
            }
        }
    }
    const int64_t row0 = tile * BLK;
    const int64_t row = row0 + tid;
    const int64_t rlast = (row0 + BLK < A.n) ? row0 + BLK : A.n;
    const int s_blk = A.row_ptr[row0];
    const int e_blk = A.row_ptr[rlast];
    int rs = 0, re = 0;
    if (row < A.n) {
        rs = A.row_ptr[row];
        re = A.row_ptr[row + 1];
    }
    double sum = 0.;
    for (int ts = s_blk; ts < e_blk; ts += CAP) {
        const int te = (ts + CAP < e_blk) ? ts + CAP : e_blk;
        int base;
        if (VEC) {
            base = ts & ~1;          // even => 16-B aligned doubles, 8-B aligned ints
            const int te2 = te & ~1; // pairs fully below te
            int k = base + 2 * tid;
            for (; k + 2 * BLK < te2; k += 4 * BLK) {
                const d2_t v0 = stream_load<NT>(reinterpret_cast<const d2_t *>(A.val + k));
                const d2_t v1 = stream_load<NT>(reinterpret_cast<const d2_t *>(A.val + k + 2 * BLK));
                const i2_t c0 = stream_load<NT>(reinterpret_cast<const i2_t *>(A.col + k));
                const i2_t c1 = stream_load<NT>(reinterpret_cast<const i2_t *>(A.col + k + 2 * BLK));
                const double x00 = x[c0.x], x01 = x[c0.y], x10 = x[c1.x], x11 = x[c1.y];
                prod[k - base] = v0.x * x00;
                prod[k - base + 1] = v0.y * x01;
                prod[k - base + 2 * BLK] = v1.x * x10;
                prod[k - base + 2 * BLK + 1] = v1.y * x11;
            }
            for (; k < te2; k += 2 * BLK) {
                const d2_t v0 = stream_load<NT>(reinterpret_cast<const d2_t *>(A.val + k));
                const i2_t c0 = stream_load<NT>(reinterpret_cast<const i2_t *>(A.col + k));
                prod[k - base] = v0.x * x[c0.x];
                prod[k - base + 1] = v0.y * x[c0.y];
            }
            if (tid == 0 && (te & 1)) prod[te - 1 - base] = A.val[te - 1] * x[A.col[te - 1]];
        } else {
            base = ts;
            int k = ts + tid;
            for (; k + 3 * BLK < te; k += 4 * BLK) {
                const double v0 = stream_load<NT>(A.val + k), v1 = stream_load<NT>(A.val + k + BLK),
                             v2 = stream_load<NT>(A.val + k + 2 * BLK), v3 = stream_load<NT>(A.val + k + 3 * BLK);
                const int c0 = stream_load<NT>(A.col + k), c1 = stream_load<NT>(A.col + k + BLK),
                          c2 = stream_load<NT>(A.col + k + 2 * BLK), c3 = stream_load<NT>(A.col + k + 3 * BLK);
                const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
                prod[k - ts] = v0 * x0;
                prod[k - ts + BLK] = v1 * x1;
                prod[k - ts + 2 * BLK] = v2 * x2;
                prod[k - ts + 3 * BLK] = v3 * x3;
            }
            for (; k < te; k += BLK) prod[k - ts] = stream_load<NT>(A.val + k) * x[stream_load<NT>(A.col + k)];
        }
        __syncthreads();
        const int a = rs > ts ? rs : ts;
        const int b = re < te ? re : te;
        for (int j = a; j < b; ++j) sum += prod[j - base];
        __syncthreads();
    }
    if (row < A.n) {
        if (NT) __builtin_nontemporal_store(sum, y + row);
        else y[row] = sum;
    }
    if (DOT) {
        double d = (row < A.n) ? sum * xin[row] : 0.;
        d = wave_sum(d);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = d;
        __syncthreads();
        if (tid == 0) {
            double t = 0.;
            for (int w = 0; w < BLK / 64; ++w) t += red[w];
            if (!HALO) partial[blockIdx.x] = t; // slot = position in the launch (== tile without a tile list)
            else __hip_atomic_store(hv.stage + tile, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // fire and forget, see halo_finalizer
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SpMV: vector variants (LPR lanes per row)
// ---------------------------------------------------------------------------------------------
template <int LPR, bool DOT>
__global__ __launch_bounds__(kBlock) void k_spmv_vec(CsrView A, const double *__restrict__ x,
                                                     double *__restrict__ y,
                                                     double *__restrict__ partial,
                                                     const PcgScalars *sc)
{
    if (DOT && sc && sc->done) return;
    __shared__ double red[4];
    const int sub = threadIdx.x % LPR;
    const int64_t group = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int64_t ngroups = (int64_t)gridDim.x * kBlock / LPR;
    double dot = 0.;
    for (int64_t row = group; row < A.n; row += ngroups) {
        const int s = A.row_ptr[row], e = A.row_ptr[row + 1];
        double sum = 0.;
        for (int k = s + sub; k < e; k += LPR) sum += A.val[k] * x[A.col[k]];
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, LPR);
        if (sub == 0) {
            y[row] = sum;
            if (DOT) dot += sum * x[row];
        }
    }
    if (DOT) {
        dot = block_sum(dot, red);
        if (threadIdx.x == 0) partial[blockIdx.x] = dot;
    }
}

// ---------------------------------------------------------------------------------------------
// SpMV on the value-indexed matrix (CsrView::codes / table): same tile structure as k_spmv_tile, but the
// stream is (uint16 code, int32 col) = 6 B per non-zero instead of 12 B, and the value comes from the
// dictionary -- staged in LDS when it has at most TBL entries (the uniform-coefficient case: ~100
// entries), read through L1/L2 otherwise.  val == table[code] bit for bit, so products, their order and
// the row sums are those of the plain kernel.
// ---------------------------------------------------------------------------------------------
// Every lane issues ALL of its quad loads for the pass (8-B code quad + 16-B column quad, or one 16-B quad of
// packed words), then ALL 4*U gathers of x, before the first product is formed; products are parked with 16-B LDS
// stores.  Row sums stay left-to-right (oracle order).  U = CAP / (4 BLK) = 2 with 32 waves per CU is the measured
// optimum (profiles/r01_spmv_variants.md): U = 4 needs 70 VGPRs and twice the LDS, which halves the resident waves.
typedef unsigned short us4_t __attribute__((ext_vector_type(4)));
typedef int i4_t __attribute__((ext_vector_type(4)));

typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

// TLT > 0: tile-local dictionaries (CsrView::tab_ptr): the tile's own table is staged in LDS (its first TLT entries; a tile with
// more distinct values takes a separate path that reads the rest through L1), codes are tile-local.
// CWIN: windowed columns (CsrView::cbase): one 32-bit word per non-zero = code << 20 | window slot << 14 | offset.
template <int BLK, int CAP, bool DOT, bool LTAB, bool PACK, int WIN = 0, int TLT = 0, bool HALO = false, bool CWIN = false, bool KEEPW = false>
__global__ __launch_bounds__(BLK) void k_spmv_vi2(CsrView A, const double *__restrict__ x, double *__restrict__ y,
                                                  double *__restrict__ partial, const PcgScalars *sc,
                                                  const int32_t *__restrict__ tiles, HaloView hv = HaloView())
{
    if (DOT && sc && sc->done) return;
    if (HALO) {
        if ((int)blockIdx.x >= hv.ntiles) { halo_finalizer<BLK>(hv, (int)blockIdx.x - hv.ntiles); return; }
        if (hv.tile_bnd[blockIdx.x]) halo_wait(hv); // block-uniform
    }
    const double *__restrict__ hx = HALO ? hv.dd->my_halo : nullptr;
    const int n_own_cols = HALO ? (int)hv.dd->n_own : 0;
    constexpr int U = CAP / (4 * BLK); // quads per lane per pass
    static_assert(U >= 1 && U * 4 * BLK == CAP, "CAP must be a multiple of 4*BLK");
    static_assert(TLT == 0 || (!LTAB && !PACK && BLK == 512), "tile tables: 512-row tiles, no plain packing");
    static_assert(!CWIN || (!PACK && BLK == 512), "windowed columns: 512-row tiles");
    constexpr bool WORDS = PACK || CWIN; // the stream is one 32-bit word per non-zero
    constexpr bool PREF = WORDS;         // prefetch the next pass's words during this one (4 dwords per quad; the 6-B form would need 6)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    static_assert(WIN == 0 || (WIN >= BLK && WIN % BLK == 0), "window = a whole number of tiles");
    double *prod = smem;                 // CAP + 4
    double *xs = smem + CAP + 4;                           // WIN entries of x around the tile's rows
    int *cb = reinterpret_cast<int *>(smem + CAP + 4 + WIN); // CWIN: the tile's 64 window bases
    double *tbl = smem + CAP + 4 + WIN + (CWIN ? kCwinSlots / 2 : 0); // table_size (LTAB) / TLT entries (tile tables)
    const int tid = threadIdx.x;
    const int64_t tile = tiles ? (int64_t)tiles[blockIdx.x] : (int64_t)blockIdx.x;
    if (CWIN && tid < kCwinSlots) cb[tid] = A.cbase[tile * kCwinSlots + tid];
    const double *__restrict__ gtab = A.table;
    int tlen = A.table_size;
    if (TLT > 0) {
        const int t0 = A.tab_ptr[tile];
        gtab = A.table + t0;
        tlen = A.tab_ptr[tile + 1] - t0;
        if (tlen > TLT) tlen = TLT;
    }
    // Tile tables: the first TLT entries are in LDS.  The rare tile with more distinct values reads the rest through L1 -- on a
    // path of its own (`big`, wave-uniform), so that the common tile carries no global load (and no s_waitcnt on it) per product.
    bool big = false;
    if (TLT > 0) big = (A.tab_ptr[tile + 1] - A.tab_ptr[tile]) > TLT;
    auto value = [&](unsigned code) -> double { return (LTAB || TLT > 0) ? tbl[code] : gtab[code]; };
    auto value_big = [&](unsigned code) -> double { return code < (unsigned)TLT ? tbl[code] : gtab[code]; };
    const int64_t row0 = tile * BLK;
    const int64_t row = row0 + tid;
    const int64_t rlast = (row0 + BLK < A.n) ? row0 + BLK : A.n;
    // x window: 43 % (WIN = 256) / 58 % (512) of a tile's columns lie within WIN entries of its rows in the brick-major
    // numbering; those gathers are served from LDS instead of one L1 access each
    int w0 = 0;
    unsigned wlen = 0;
    if (WIN > 0) {
        const int64_t lo = row0 - (WIN - BLK) / 2;
        w0 = (int)(lo > 0 ? lo : 0);
        const int64_t left = A.n - w0;
        wlen = (unsigned)(left < WIN ? left : WIN);
#pragma unroll
        for (int i = 0; i < WIN / BLK; ++i) {
            const unsigned o = (unsigned)(tid + i * BLK);
            if (o < wlen) xs[o] = x[(int64_t)w0 + o];
        }
    }
    auto gather = [&](int col) -> double {
        if (WIN > 0 && !HALO) {
            // Both reads are issued for every lane, from a clamped address, and the value is selected afterwards.  As two exec-masked
            // branches writing ONE register the compiler put an `s_waitcnt lgkmcnt(0)` between the LDS read and the global load of every
            // gather (write-after-write on the destination): the eight global gathers of a pass went out one LDS round trip apart.
            const unsigned off = (unsigned)(col - w0);
            const bool in = off < wlen;
            const double a = xs[in ? off : 0u];
            const double b = x[in ? w0 : col];
            return in ? a : b;
        }
        if (WIN > 0) {
            const unsigned off = (unsigned)(col - w0);
            if (off < wlen) return xs[off];
        }
        if (HALO && col >= n_own_cols) return ld_sys_f64(hx + (col - n_own_cols)); // the comm block's halo area, written by the peers
        return x[col];
    };
    // the matrix stream of a pass: every lane's quads are requested in one go ...
    u4_t qn[U]; // value codes (WORDS: whole words, split at use)
    i4_t cn[U];
    auto request = [&](int ts, int e_blk) {
        const int te = (ts + CAP < e_blk) ? ts + CAP : e_blk;
        const int base = ts & ~3, te4 = te & ~3;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = base + 4 * (tid + u * BLK);
            qn[u] = u4_t{0, 0, 0, 0};
            cn[u] = i4_t{0, 0, 0, 0};
            if (kk < te4) {
                if (WORDS) {
                    qn[u] = stream_load_k<KEEPW>(reinterpret_cast<const u4_t *>(A.packed + kk));
                } else {
                    const us4_t h = stream_load_k<KEEPW>(reinterpret_cast<const us4_t *>(A.codes + kk));
                    qn[u] = u4_t{h.x, h.y, h.z, h.w};
                    cn[u] = stream_load_k<KEEPW>(reinterpret_cast<const i4_t *>(A.col + kk));
                }
            }
        }
    };
    const int s_blk = A.row_ptr[row0];
    const int e_blk = A.row_ptr[rlast];
    // ... and the FIRST pass is requested before the tile's tables / x window are staged: their latency (two dependent global
    // round trips for a tile-local dictionary) hides behind the stream instead of delaying it
    request(s_blk, e_blk);
    if (LTAB || TLT > 0)
        for (int i = tid; i < tlen; i += BLK) tbl[i] = gtab[i];
    int rs = 0, re = 0;
    double xr = 0.;
    if (row < A.n) {
        rs = A.row_ptr[row];
        re = A.row_ptr[row + 1];
        if (DOT && WIN == 0) xr = x[row]; // early: its latency hides behind the passes
    }
    if (LTAB || WIN > 0 || TLT > 0 || CWIN) __syncthreads();
    if (DOT && WIN > 0 && row < A.n) xr = xs[(int)(row - w0)]; // the tile's own rows are always inside the window
    double sum = 0.;
    for (int ts = s_blk; ts < e_blk; ts += CAP) {
        const int te = (ts + CAP < e_blk) ? ts + CAP : e_blk;
        const int base = ts & ~3; // 8-B aligned code quads, 16-B aligned column quads
        const int te4 = te & ~3;  // quads fully below te
        u4_t q[U];
        i4_t c[U];
        double xv[U][4];
        const unsigned cmask = PACK ? ((1u << A.col_bits) - 1u) : 0u;
        const int cbits = PACK ? A.col_bits : 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            q[u] = qn[u];
            c[u] = cn[u];
        }
        if (PREF && te < e_blk) request(te, e_blk); // software pipeline: the next pass travels while this one is multiplied
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = base + 4 * (tid + u * BLK);
            if (kk < te4) {
                if (PACK) {
                    c[u] = i4_t{(int)(q[u].x & cmask), (int)(q[u].y & cmask), (int)(q[u].z & cmask), (int)(q[u].w & cmask)};
                    q[u] = u4_t{q[u].x >> cbits, q[u].y >> cbits, q[u].z >> cbits, q[u].w >> cbits};
                }
                if (CWIN) {
                    constexpr unsigned om = (1u << kCwinOffBits) - 1u, sm = kCwinSlots - 1u;
                    c[u] = i4_t{cb[(q[u].x >> kCwinOffBits) & sm] + (int)(q[u].x & om), cb[(q[u].y >> kCwinOffBits) & sm] + (int)(q[u].y & om),
                                cb[(q[u].z >> kCwinOffBits) & sm] + (int)(q[u].z & om), cb[(q[u].w >> kCwinOffBits) & sm] + (int)(q[u].w & om)};
                    constexpr int sh = kCwinOffBits + kCwinSlotBits;
                    q[u] = u4_t{q[u].x >> sh, q[u].y >> sh, q[u].z >> sh, q[u].w >> sh};
                }
                if ((CWIN || TLT > 0) && kk < ts) {
                    // Head of the first pass: the aligned quad starts up to 3 entries BEFORE the tile's first non-zero.  Those
                    // entries belong to the previous tile -- they are never summed, but their words are relative to THAT tile's
                    // column windows / dictionary: decoded against this tile's they point anywhere (a gather up to 16 K entries
                    // past the end of x: found as a GPU memory fault on a 3-rank distributed solve by tools/stress_parity.py).
                    const int safe = (int)row0;
                    if (kk + 0 < ts) { c[u].x = safe; q[u].x = 0u; }
                    if (kk + 1 < ts) { c[u].y = safe; q[u].y = 0u; }
                    if (kk + 2 < ts) { c[u].z = safe; q[u].z = 0u; }
                }
                xv[u][0] = gather(c[u].x);
                xv[u][1] = gather(c[u].y);
                xv[u][2] = gather(c[u].z);
                xv[u][3] = gather(c[u].w);
            }
        }
        auto products = [&](auto val) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = base + 4 * (tid + u * BLK);
                if (kk < te4) {
                    d2_t lo, hi;
                    lo.x = val(q[u].x) * xv[u][0];
                    lo.y = val(q[u].y) * xv[u][1];
                    hi.x = val(q[u].z) * xv[u][2];
                    hi.y = val(q[u].w) * xv[u][3];
                    *reinterpret_cast<d2_t *>(prod + (kk - base)) = lo;
                    *reinterpret_cast<d2_t *>(prod + (kk - base) + 2) = hi;
                }
            }
            if (tid < te - te4) { // ragged end of the pass (at most 3 entries; entries below ts are never summed)
                const int kk = te4 + tid;
                if (kk >= ts) prod[kk - base] = val(A.codes[kk]) * gather(A.col[kk]); // the unpacked arrays stay resident (an entry
                                                                                       // below ts is the previous tile's: see above)
            }
        };
        if (TLT > 0 && big) products(value_big);
        else products(value);
        if (!PREF && te < e_blk) request(te, e_blk);
        __syncthreads();
        const int a = rs > ts ? rs : ts;
        const int b = re < te ? re : te;
        for (int j = a; j < b; ++j) sum += prod[j - base];
        __syncthreads();
    }
    if (DOT) {
        // x.y: one partial per WAVE, reduced with DPP row shifts / broadcasts -- no LDS traffic, no barrier, no atomic.
        // (Measured: parking the terms in LDS and letting the last-arriving wave fold them cost 9-10 us per launch, two
        // dependent LDS round trips at the end of every wave's life while the LDS pipe is busy; a block barrier 15 us.)
        const double d = wave_sum_dpp((row < A.n) ? sum * xr : 0.);
        if ((tid & 63) == 63) {
            if (!HALO) partial[(int64_t)blockIdx.x * (BLK / 64) + (tid >> 6)] = d; // slot = position in the launch
            else // direct transport: write-through, fire and forget -- a finalizer block watches the slot (halo_finalizer)
                __hip_atomic_store(hv.stage + tile * (BLK / 64) + (tid >> 6), d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (row < A.n) __builtin_nontemporal_store(sum, y + row);
}

static constexpr int kViLdsTable = 2048; // dictionary entries staged in LDS (16 KiB)
static constexpr int kTileRows = 512;    // rows per workgroup of the value-indexed kernel and of the dist tile lists
static constexpr int kTileCap = 4096;    // products parked per pass (U = 2 quads per lane)
static constexpr int kTileWin = 512;     // x entries staged in LDS (the tile's own rows): 38 KiB per workgroup => 32 waves per CU

template <int BLK, int CAP, bool DOT, bool LTAB, bool PACK, int WIN = 0, int TLT = 0, bool HALO = false, bool CWIN = false>
static avs_status spmv_vi2_launch_t(const CsrView &A, const double *x, double *y, double *partial, const PcgScalars *sc,
                                    const int32_t *tiles, int ntiles, size_t lds, hipStream_t stream, const HaloView &hv = HaloView())
{
    // > 48 KiB of dynamic LDS (value table of ~1.5 k+ entries) needs the opt-in; it is a per-device function attribute, so it
    // is set on every such launch (cheap, rare path) rather than cached in a process-wide flag
    if (A.keep_cached && !HALO) { // the words may stay in the Infinity Cache between two products: plain loads
        if (lds > 48 * 1024)
            AVS_HIP(hipFuncSetAttribute((const void *)k_spmv_vi2<BLK, CAP, DOT, LTAB, PACK, WIN, TLT, HALO, CWIN, !HALO>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
        hipLaunchKernelGGL((k_spmv_vi2<BLK, CAP, DOT, LTAB, PACK, WIN, TLT, HALO, CWIN, !HALO>), dim3(ntiles), dim3(BLK), lds, stream, A, x, y,
                           partial, sc, tiles, hv);
        AVS_HIP(hipGetLastError());
        return AVS_OK;
    }
    if (lds > 48 * 1024)
        AVS_HIP(hipFuncSetAttribute((const void *)k_spmv_vi2<BLK, CAP, DOT, LTAB, PACK, WIN, TLT, HALO, CWIN>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
    hipLaunchKernelGGL((k_spmv_vi2<BLK, CAP, DOT, LTAB, PACK, WIN, TLT, HALO, CWIN>), dim3(ntiles), dim3(BLK), lds, stream, A, x, y, partial,
                       sc, tiles, hv);
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

static constexpr int kTltLds = 1024;   // tile-table entries staged in LDS (8 KiB; tiles rarely hold more distinct values)

// tile-local dictionaries: 512-row tiles, CAP products per pass, the tile's table next to the x window
template <int CAP, bool DOT, int TCAP>
static avs_status spmv_tlt_launch(const CsrView &A, const double *x, double *y, double *partial, const PcgScalars *sc,
                                  const int32_t *tiles, int ntiles, hipStream_t stream)
{
    if (ntiles <= 0) return AVS_OK;
    const size_t lds = (size_t)(CAP + 4 + kTileWin + TCAP + (A.cbase ? kCwinSlots / 2 : 0)) * sizeof(double);
    if (A.cbase) return spmv_vi2_launch_t<kTileRows, CAP, DOT, false, false, kTileWin, TCAP, false, true>(A, x, y, partial, sc, tiles, ntiles, lds, stream);
    return spmv_vi2_launch_t<kTileRows, CAP, DOT, false, false, kTileWin, TCAP>(A, x, y, partial, sc, tiles, ntiles, lds, stream);
}

template <int BLK, int CAP, bool DOT, int WIN = 0>
static avs_status spmv_vi2_launch(const CsrView &A, const double *x, double *y, double *partial, const PcgScalars *sc,
                                  const int32_t *tiles, int ntiles, hipStream_t stream)
{
    if (ntiles <= 0) return AVS_OK;
    if (A.tab_ptr) { // tile-local codes: only the tile-table kernel can decode them
        if (BLK != kTileRows) { set_error("tile-local dictionaries need %d-row tiles", kTileRows); return AVS_EINVAL; }
        return spmv_tlt_launch<kTileCap, DOT, kTltLds>(A, x, y, partial, sc, tiles, ntiles, stream);
    }
    const bool ltab = A.table_size <= kViLdsTable;
    if (A.cbase) { // windowed columns with ONE dictionary (it has at most 2048 entries, see build_matrix_index)
        if (BLK != kTileRows || WIN != kTileWin || !ltab) { set_error("windowed columns need the default tile geometry"); return AVS_EINVAL; }
        const size_t ldsw = (size_t)(kTileCap + 4 + kTileWin + kCwinSlots / 2 + ((A.table_size + 1) & ~1)) * sizeof(double);
        return spmv_vi2_launch_t<kTileRows, kTileCap, DOT, true, false, kTileWin, 0, false, true>(A, x, y, partial, sc, tiles, ntiles, ldsw, stream);
    }
    const bool pack = A.packed != nullptr;
    const size_t lds = (size_t)(CAP + 4 + WIN + (ltab ? ((A.table_size + 1) & ~1) : 0)) * sizeof(double);
    if (ltab && pack) return spmv_vi2_launch_t<BLK, CAP, DOT, true, true, WIN>(A, x, y, partial, sc, tiles, ntiles, lds, stream);
    if (ltab) return spmv_vi2_launch_t<BLK, CAP, DOT, true, false, WIN>(A, x, y, partial, sc, tiles, ntiles, lds, stream);
    if (pack) return spmv_vi2_launch_t<BLK, CAP, DOT, false, true, WIN>(A, x, y, partial, sc, tiles, ntiles, lds, stream);
    return spmv_vi2_launch_t<BLK, CAP, DOT, false, false, WIN>(A, x, y, partial, sc, tiles, ntiles, lds, stream);
}

#ifdef AVS_PROBES
// ---------------------------------------------------------------------------------------------
// stream ceilings of this chip for the access widths the SpMV uses (measurement helpers)
// ---------------------------------------------------------------------------------------------
template <bool NT, int MODE>
__global__ __launch_bounds__(kBlock) void k_stream_probe(const d2_t *__restrict__ a, d2_t *__restrict__ b, int64_t n2,
                                                         double *__restrict__ sink)
{
    double acc = 0.;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        const d2_t v0 = stream_load<NT>(a + i), v1 = stream_load<NT>(a + i + stride), v2 = stream_load<NT>(a + i + 2 * stride),
                   v3 = stream_load<NT>(a + i + 3 * stride);
        if (MODE == 1) { b[i] = v0; b[i + stride] = v1; b[i + 2 * stride] = v2; b[i + 3 * stride] = v3; }
        else acc += (v0.x + v0.y) + (v1.x + v1.y) + (v2.x + v2.y) + (v3.x + v3.y);
    }
    for (; i < n2; i += stride) {
        const d2_t v0 = stream_load<NT>(a + i);
        if (MODE == 1) b[i] = v0; else acc += v0.x + v0.y;
    }
    if (MODE == 0 && acc == 1.2345e300) sink[0] = acc; // keep the loads alive
}

avs_status stream_probe(int mode, const double *a, double *b, int64_t n, double *sink, int grid, hipStream_t st)
{
    const int64_t n2 = n / 2;
    switch (mode) {
    case 0: hipLaunchKernelGGL((k_stream_probe<false, 0>), dim3(grid), dim3(kBlock), 0, st, (const d2_t *)a, (d2_t *)b, n2, sink); break;
    case 1: hipLaunchKernelGGL((k_stream_probe<true, 0>), dim3(grid), dim3(kBlock), 0, st, (const d2_t *)a, (d2_t *)b, n2, sink); break;
    case 2: hipLaunchKernelGGL((k_stream_probe<false, 1>), dim3(grid), dim3(kBlock), 0, st, (const d2_t *)a, (d2_t *)b, n2, sink); break;
    default: set_error("unknown stream probe mode %d", mode); return AVS_EINVAL;
    }
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

// ---------------------------------------------------------------------------------------------
// SELL-C-sigma (C = 64 = one wavefront per slice) -- measurement only (BASELINE configs[4] asks for a blocked-ELL / SELL
// experiment; tools/sell_experiment.py builds the layout and reports padding + time, profiles/r02_sell_experiment.md).
// Slice s holds rows [64 s, 64 s + 64) of the sigma-sorted matrix, column-major: entry j of lane l at slice_ptr[s] + 64 j + l,
// padded to the slice's longest row with (col 0, val 0.0).  Every load is a full coalesced 512-B / 256-B wave access.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_spmv_sell(int64_t nslices, const int64_t *__restrict__ slice_ptr, const int32_t *__restrict__ col,
                                                   const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nslices) return;
    const int64_t b = slice_ptr[s], e = slice_ptr[s + 1];
    double sum = 0.;
    int64_t k = b + lane;
    for (; k + 192 < e; k += 256) { // 4 entries per lane in flight
        const double v0 = __builtin_nontemporal_load(val + k), v1 = __builtin_nontemporal_load(val + k + 64),
                     v2 = __builtin_nontemporal_load(val + k + 128), v3 = __builtin_nontemporal_load(val + k + 192);
        const int c0 = __builtin_nontemporal_load(col + k), c1 = __builtin_nontemporal_load(col + k + 64),
                  c2 = __builtin_nontemporal_load(col + k + 128), c3 = __builtin_nontemporal_load(col + k + 192);
        const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
        sum += v0 * x0;
        sum += v1 * x1;
        sum += v2 * x2;
        sum += v3 * x3;
    }
    for (; k < e; k += 64) sum += __builtin_nontemporal_load(val + k) * x[__builtin_nontemporal_load(col + k)];
    y[s * 64 + lane] = sum;
}

#endif // AVS_PROBES

static inline int stream_grid(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

int spmv_default_variant(const CsrView &) { return 24; } // 512 rows/WG, 32 KiB LDS, 16-B vector + non-temporal stream (profiles/r01_spmv_variants.md)

template <bool DOT>
static avs_status spmv_dispatch(const CsrView &A, const double *x, double *y, double *partial,
                                const PcgScalars *sc, int variant, hipStream_t stream, int *nblocks)
{
    if (A.n <= 0) { if (nblocks) *nblocks = 0; return AVS_OK; }
    if (A.brick && A.brick->ntiles > 0 && (variant == 0 || variant == spmv_default_variant(A))) { // brick-structured form (avs_brick.hip)
        if (nblocks) *nblocks = brick_partial_count(*A.brick, 8);
        return spmv_brick_launch(*A.brick, x, y, DOT ? partial : nullptr, (DOT && sc) ? &sc->done : nullptr, stream);
    }
    if (A.codes && (variant == 0 || variant == spmv_default_variant(A))) { // value-indexed matrix: 6 or 4 B per non-zero
        const int nt = (int)((A.n + kTileRows - 1) / kTileRows);
        if (nblocks) *nblocks = nt * (kTileRows / 64);
        return spmv_vi2_launch<kTileRows, kTileCap, DOT, kTileWin>(A, x, y, partial, sc, nullptr, nt, stream);
    }
#ifdef AVS_PROBES // geometry sweeps of the value-indexed kernels (profiles/r01_spmv_variants.md, r02_varvisc.md)
    if (A.tab_ptr && variant >= 51 && variant <= 54) { // geometry sweep of the tile-table kernel (profiles/r02_varvisc.md)
        const int nt = (int)((A.n + kTileRows - 1) / kTileRows);
        if (nblocks) *nblocks = nt * (kTileRows / 64);
        switch (variant) {
        case 51: return spmv_tlt_launch<4096, DOT, 1024>(A, x, y, partial, sc, nullptr, nt, stream);
        case 52: return spmv_tlt_launch<2048, DOT, 2048>(A, x, y, partial, sc, nullptr, nt, stream);
        case 53: return spmv_tlt_launch<4096, DOT, 2048>(A, x, y, partial, sc, nullptr, nt, stream);
        case 54: return spmv_tlt_launch<2048, DOT, 1024>(A, x, y, partial, sc, nullptr, nt, stream);
        }
    }
    if (A.codes && !A.tab_ptr && !A.cbase && variant >= 31 && variant <= 46) {
#define AVS_VI2_CASE(ID, BLK, CAP, ...)                                                                         \
    case ID: {                                                                                                  \
        const int nt = (int)((A.n + BLK - 1) / BLK);                                                            \
        if (nblocks) *nblocks = nt * (BLK / 64);                                                                \
        return spmv_vi2_launch<BLK, CAP, DOT, ##__VA_ARGS__>(A, x, y, partial, sc, nullptr, nt, stream);        \
    }
        switch (variant) { // geometry sweep of the value-indexed kernel (profiles/r01_spmv_variants.md)
            AVS_VI2_CASE(31, 256, 4096)
            AVS_VI2_CASE(32, 512, 8192)
            AVS_VI2_CASE(33, 512, 4096)
            AVS_VI2_CASE(34, 128, 2048)
            AVS_VI2_CASE(36, 1024, 8192)
            AVS_VI2_CASE(37, 256, 2048)
            AVS_VI2_CASE(38, 512, 2048)
            AVS_VI2_CASE(39, 256, 1024)
            AVS_VI2_CASE(40, 128, 1024)
            AVS_VI2_CASE(41, 256, 2048, 256)
            AVS_VI2_CASE(42, 512, 4096, 512)
            AVS_VI2_CASE(43, 256, 2048, 512)
            AVS_VI2_CASE(44, 128, 1024, 128)
            AVS_VI2_CASE(45, 128, 1024, 256)
            AVS_VI2_CASE(46, 512, 4096, 1024)
        }
#undef AVS_VI2_CASE
    }
#endif
    if (variant == 0) variant = spmv_default_variant(A);
    int g;
    switch (variant) { // the plain kernels below read A.val / A.col only
#define AVS_TILE_CASE(ID, BLK, CAP, VEC, XCD, CHUNK, NT)                                                                  \
    case ID:                                                                                                   \
        g = (int)((A.n + BLK - 1) / BLK);                                                                      \
        hipLaunchKernelGGL((k_spmv_tile<BLK, CAP, DOT, VEC, XCD, NT>), dim3(g), dim3(BLK), 0, stream, A, x, y, partial, sc, CHUNK); \
        break;
#ifdef AVS_PROBES // the round-1 kernel sweep (14 = the plain reference kernel avs_bench_spmv compares against)
    case 1:
        g = stream_grid(A.n);
        hipLaunchKernelGGL((k_spmv_stream<DOT>), dim3(g), dim3(kBlock), 0, stream, A, x, y, partial, sc);
        break;
    case 2:
        g = kVecGrid * 4;
        hipLaunchKernelGGL((k_spmv_vec<4, DOT>), dim3(g), dim3(kBlock), 0, stream, A, x, y, partial, sc);
        break;
    case 3:
        g = kVecGrid * 4;
        hipLaunchKernelGGL((k_spmv_vec<8, DOT>), dim3(g), dim3(kBlock), 0, stream, A, x, y, partial, sc);
        break;
    case 4:
        g = kVecGrid * 4;
        hipLaunchKernelGGL((k_spmv_vec<16, DOT>), dim3(g), dim3(kBlock), 0, stream, A, x, y, partial, sc);
        break;
        AVS_TILE_CASE(5, 256, 4096, false, true, 0, false)
        AVS_TILE_CASE(6, 256, 4096, true, false, 0, false)
        AVS_TILE_CASE(7, 256, 2048, false, false, 0, false)
        AVS_TILE_CASE(8, 128, 2048, false, false, 0, false)
        AVS_TILE_CASE(9, 512, 8192, false, false, 0, false)
        AVS_TILE_CASE(10, 256, 2048, true, false, 0, false)
        AVS_TILE_CASE(11, 256, 2048, true, true, 2, false)
        AVS_TILE_CASE(12, 256, 2048, true, true, 4, false)
        AVS_TILE_CASE(13, 256, 2048, true, true, 16, false)
        AVS_TILE_CASE(14, 256, 2048, true, false, 0, true)
        AVS_TILE_CASE(15, 256, 2048, true, true, 4, true)
        AVS_TILE_CASE(16, 256, 2048, true, true, 16, true)
        AVS_TILE_CASE(17, 256, 2048, true, true, 64, true)
        AVS_TILE_CASE(18, 256, 4096, true, true, 4, true)
        AVS_TILE_CASE(19, 256, 2048, false, false, 0, true)
        AVS_TILE_CASE(20, 256, 2048, true, true, 2, true)
        AVS_TILE_CASE(21, 256, 2048, true, true, 8, true)
        AVS_TILE_CASE(22, 128, 1024, true, false, 0, true)
        AVS_TILE_CASE(23, 128, 2048, true, false, 0, true)
#endif
        AVS_TILE_CASE(24, 512, 4096, true, false, 0, true)
#undef AVS_TILE_CASE
    default:
        set_error("unknown SpMV variant %d", variant);
        return AVS_EINVAL;
    }
    if (nblocks) *nblocks = g;
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

static size_t max_partials(int64_t n);

avs_status spmv_launch(const CsrView &A, const double *x, double *y, int variant, hipStream_t stream)
{
    return spmv_dispatch<false>(A, x, y, nullptr, nullptr, variant, stream, nullptr);
}

// default kernel restricted to a list of kTileRows-row tiles (multi-GPU overlap: interior tiles run while the halo
// travels); partial[tile] receives the tile's share of x.y, so several launches fill one partial array
int spmv_tile_rows() { return kTileRows; }
avs_status spmv_dot_tiles(const CsrView &A, const double *x, double *y, double *partial, const PcgScalars *sc,
                          const int32_t *tiles, int ntiles, hipStream_t stream)
{
    if (ntiles <= 0) return AVS_OK;
    if (A.codes) return spmv_vi2_launch<kTileRows, kTileCap, true, kTileWin>(A, x, y, partial, sc, tiles, ntiles, stream);
    hipLaunchKernelGGL((k_spmv_tile<kTileRows, 4096, true, true, false, true>), dim3(ntiles), dim3(kTileRows), 0, stream, A, x, y,
                       partial, sc, 0, tiles);
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

// the SpMV launch of the direct transport: all tiles (those flagged in hv.tile_bnd wait for the peers' entries and gather halo
// columns from the comm block) + the finalizer block; launch_blocks = ntiles + 1
avs_status spmv_dot_tiles_halo(const CsrView &A, const double *x, double *y, double *partial, const PcgScalars *sc,
                               const int32_t *tiles, int launch_blocks, const HaloView &hv, hipStream_t stream)
{
    if (A.codes) {
        if (A.tab_ptr) {
            const size_t lds = (size_t)(kTileCap + 4 + kTileWin + kTltLds + (A.cbase ? kCwinSlots / 2 : 0)) * sizeof(double);
            if (A.cbase)
                return spmv_vi2_launch_t<kTileRows, kTileCap, true, false, false, kTileWin, kTltLds, true, true>(A, x, y, partial, sc, tiles,
                                                                                                             launch_blocks, lds, stream, hv);
            return spmv_vi2_launch_t<kTileRows, kTileCap, true, false, false, kTileWin, kTltLds, true>(A, x, y, partial, sc, tiles, launch_blocks,
                                                                                                   lds, stream, hv);
        }
        const bool ltab = A.table_size <= kViLdsTable;
        if (A.cbase) {
            const size_t ldsw = (size_t)(kTileCap + 4 + kTileWin + kCwinSlots / 2 + ((A.table_size + 1) & ~1)) * sizeof(double);
            return spmv_vi2_launch_t<kTileRows, kTileCap, true, true, false, kTileWin, 0, true, true>(A, x, y, partial, sc, tiles, launch_blocks,
                                                                                                  ldsw, stream, hv);
        }
        const bool pack = A.packed != nullptr;
        const size_t lds = (size_t)(kTileCap + 4 + kTileWin + (ltab ? ((A.table_size + 1) & ~1) : 0)) * sizeof(double);
        if (ltab && pack) return spmv_vi2_launch_t<kTileRows, kTileCap, true, true, true, kTileWin, 0, true>(A, x, y, partial, sc, tiles, launch_blocks, lds, stream, hv);
        if (ltab) return spmv_vi2_launch_t<kTileRows, kTileCap, true, true, false, kTileWin, 0, true>(A, x, y, partial, sc, tiles, launch_blocks, lds, stream, hv);
        if (pack) return spmv_vi2_launch_t<kTileRows, kTileCap, true, false, true, kTileWin, 0, true>(A, x, y, partial, sc, tiles, launch_blocks, lds, stream, hv);
        return spmv_vi2_launch_t<kTileRows, kTileCap, true, false, false, kTileWin, 0, true>(A, x, y, partial, sc, tiles, launch_blocks, lds, stream, hv);
    }
    hipLaunchKernelGGL((k_spmv_tile<kTileRows, 4096, true, true, false, true, true>), dim3(launch_blocks), dim3(kTileRows), 0, stream, A, x, y,
                       partial, sc, 0, tiles, hv);
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

// the form used inside the PCG loop: y = A x and per-block partials of x.y
avs_status spmv_dot_launch(const CsrView &A, const double *x, double *y, double *partial, int variant, hipStream_t stream)
{
    return spmv_dispatch<true>(A, x, y, partial, nullptr, variant, stream, nullptr);
}
size_t spmv_partial_elems(int64_t n) { return max_partials(n); }

static size_t max_partials(int64_t n);
static size_t max_partials(int64_t n)
{
    size_t a = (size_t)((n + 63) / 64) + 16, b = (size_t)kVecGrid * 4; // up to one partial per wave of rows
    return 2 * (a > b ? a : b) + 4 * (size_t)kVecGrid + 16;
}

// ---------------------------------------------------------------------------------------------
// vector kernels
// ---------------------------------------------------------------------------------------------
// DiagonalPreconditioner::factorize: invdiag(j) = A(j,j) != 0 ? 1/A(j,j) : 1
__global__ __launch_bounds__(kBlock) void k_inv_diag(CsrView A, double *__restrict__ invd)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= A.n) return;
    double d = 0.;
    for (int k = A.row_ptr[i]; k < A.row_ptr[i + 1]; ++k)
        if (A.col[k] == (int32_t)i) d = A.val ? A.val[k] : A.table[(A.tab_ptr ? A.tab_ptr[i / kTileRows] : 0) + A.codes[k]];
    invd[i] = (d != 0. && !A.no_precond) ? 1. / d : 1.; // (no_precond: plain CG, z = r)
}

// r = b - t ; partials: [0..g) b.b, [g..2g) r.r
__global__ __launch_bounds__(kBlock) void k_init_residual(int64_t n, const double *__restrict__ b,
                                                          const double *__restrict__ t,
                                                          double *__restrict__ r, double *__restrict__ partial)
{
    __shared__ double red[4];
    double bb = 0., rr = 0.;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double bi = b[i];
        const double ri = bi - t[i];
        r[i] = ri;
        bb += bi * bi;
        rr += ri * ri;
    }
    bb = block_sum(bb, red);
    rr = block_sum(rr, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = bb;
        partial[gridDim.x + blockIdx.x] = rr;
    }
}

// p = invd * r ; partial r.p
__global__ __launch_bounds__(kBlock) void k_init_p(int64_t n, const double *__restrict__ r,
                                                   const double *__restrict__ invd, double *__restrict__ p,
                                                   double *__restrict__ x, double *__restrict__ partial,
                                                   const PcgScalars *sc)
{
    __shared__ double red[4];
    const int done = sc->done;
    double rz = 0.;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        if (done == 3) { x[i] = 0.; continue; } // rhsNorm2 == 0 -> x.setZero()
        if (done) continue;
        const double ri = r[i];
        const double zi = invd[i] * ri;
        p[i] = zi;
        rz += ri * zi;
    }
    rz = block_sum(rz, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = rz;
}

// value-indexed matrix: invd[i] == invtab[dcode[i]], where invtab[c] = 1 / table[c] (1 for a zero) and the extra
// entry invtab[table_size] = 1 stands for "no diagonal entry" -- the same doubles k_inv_diag writes
__global__ __launch_bounds__(kBlock) void k_inv_diag_coded(CsrView A, uint16_t *__restrict__ dcode, double *__restrict__ invtab)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i <= A.table_size) {
        const double d = i < A.table_size ? A.table[i] : 0.;
        invtab[i] = (d != 0. && !A.no_precond) ? 1. / d : 1.;
    }
    if (i >= A.n) return;
    unsigned code = (unsigned)A.table_size;
    for (int k = A.row_ptr[i]; k < A.row_ptr[i + 1]; ++k)
        if (A.col[k] == (int32_t)i) code = A.codes[k];
    dcode[i] = (uint16_t)code;
}

// r -= alpha t ; partials: r.r and r.(invd r).  (x += alpha p rides along with the p update below:
// one vector pass less per iteration -- 10 n instead of 11 n doubles of traffic.)
// FUSED (single-GPU loop, small systems): the alpha step -- fold of the SpMV's `nb` partial sums, alpha = r.z / p.Ap -- is done
// here by every workgroup for itself, as k_update_xp does with the beta step; only worth it while the SpMV leaves few
// partials (one per wave of 64 rows): the loop uses it up to kFuseAlphaMax of them.  The kernel's own partial sums then go
// to a second array (`partial`), because other workgroups are still reading the SpMV's.
template <bool CODED, bool FUSED, bool KEEP>
__global__ __launch_bounds__(kBlock) void k_update_r(int64_t n, double *__restrict__ r, const double *__restrict__ t,
                                                     const double *__restrict__ invd, const uint16_t *__restrict__ dcode,
                                                     PcgScalars *sc, double *__restrict__ partial,
                                                     const double *__restrict__ spmv_partial = nullptr, int nb = 0, int parity = 0)
{
    if (sc->done) {
        if (FUSED && blockIdx.x == 0 && threadIdx.x == 0 && sc->done == 2) sc->done = 1; // the pending x update has run (OP_ALPHA)
        return;
    }
    __shared__ double red[4];
    double alpha;
    if (FUSED) {
        __shared__ double tot;
        double pap = 0.;
        for (int k = threadIdx.x; k < nb; k += kBlock) pap += spmv_partial[k];
        pap = block_sum(pap, red);
        if (threadIdx.x == 0) tot = pap;
        __syncthreads();
        pap = tot;
        alpha = (parity ? sc->rho_alt : sc->rho) / pap;
        if (blockIdx.x == 0 && threadIdx.x == 0) { // what OP_ALPHA does
            sc->red[0] = pap;
            sc->pAp = pap;
            sc->alpha = alpha;
        }
    } else alpha = sc->alpha;
    double rr = 0., rz = 0.;
    // t = A p is read for the last time here and r is next read one kernel later: the streams that nobody reads again before they
    // are overwritten are loaded / stored NON-TEMPORALLY, so that they do not push the matrix out of the caches between two products
    // (`keep`: matrix and vectors together fit the Infinity Cache -- then everything is left to it)
    const int64_t n2 = n >> 1; // two rows per thread (16-B accesses), partial sums in the order (even row, odd row)
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n2; j += (int64_t)gridDim.x * kBlock) {
        const int64_t i = 2 * j;
#ifndef AVS_EXP_NO_RNT
        const d2_t rv = stream_load_k<KEEP>(reinterpret_cast<const d2_t *>(r + i));
#else
        const d2_t rv = *reinterpret_cast<const d2_t *>(r + i);
#endif
        const d2_t tv = stream_load_k<KEEP>(reinterpret_cast<const d2_t *>(t + i));
        double id0, id1;
        if (CODED) {
            const unsigned cc = stream_load_k<KEEP>(reinterpret_cast<const unsigned *>(dcode + i));
            id0 = invd[cc & 0xffffu];
            id1 = invd[cc >> 16];
        } else {
            const d2_t iv = *reinterpret_cast<const d2_t *>(invd + i);
            id0 = iv.x;
            id1 = iv.y;
        }
        d2_t rn;
        rn.x = rv.x - alpha * tv.x;
        rn.y = rv.y - alpha * tv.y;
#ifndef AVS_EXP_NO_RNT
        stream_store_k<KEEP>(rn, reinterpret_cast<d2_t *>(r + i));
#else
        *reinterpret_cast<d2_t *>(r + i) = rn;
#endif
        rr += rn.x * rn.x;
        rz += rn.x * (id0 * rn.x);
        rr += rn.y * rn.y;
        rz += rn.y * (id1 * rn.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double ri = r[i] - alpha * t[i];
        r[i] = ri;
        rr += ri * ri;
        rz += ri * ((CODED ? invd[dcode[i]] : invd[i]) * ri);
    }
    rr = block_sum(rr, red);
    rz = block_sum(rz, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = rr;
        partial[gridDim.x + blockIdx.x] = rz;
    }
}

// x += alpha p (Eigen does this before the convergence test, so it also runs in the iteration that
// converges: done == 2 = "converged, x update pending"); then p = invd r + beta p unless converged.
//
// FUSED (single-GPU loop): the beta step -- sum of k_update_r's 2 g partial sums, convergence test, beta = r.z / old r.z --
// is done HERE, by every workgroup for itself (8 + 8 loads per thread from L2, the same fixed order everywhere => the same
// bits everywhere), instead of in a k_reduce launch of its own between the two vector kernels (4.9 us + a launch gap per
// iteration).  Workgroup 0 publishes the scalars; the old r.z is read from the slot of this iteration's parity and the new
// one written to the other slot, so a workgroup that starts late still reads what the early ones read.
template <bool CODED, bool FUSED, bool KEEP>
__global__ __launch_bounds__(kBlock) void k_update_xp(int64_t n, double *__restrict__ x, double *__restrict__ p,
                                                      const double *__restrict__ r, const double *__restrict__ invd,
                                                      const uint16_t *__restrict__ dcode, PcgScalars *sc,
                                                      const double *__restrict__ partial = nullptr, int g = 0, int parity = 0)
{
    int done = sc->done;
    if (done == 1 || done == 3) return;
    const double alpha = sc->alpha;
    double beta = FUSED ? 0. : sc->beta;
    if (FUSED && done == 0) {
        __shared__ double red[4], tot[2];
        double rr = 0., rz = 0.;
        for (int i = threadIdx.x; i < g; i += kBlock) {
            rr += partial[i];
            rz += partial[g + i];
        }
        rr = block_sum(rr, red);
        rz = block_sum(rz, red);
        if (threadIdx.x == 0) { tot[0] = rr; tot[1] = rz; }
        __syncthreads();
        rr = tot[0];
        rz = tot[1];
        const double absOld = parity ? sc->rho_alt : sc->rho;
        if (rr < sc->threshold) done = 2; // Eigen: break before i++ (x += alpha p still pending)
        else beta = rz / absOld;
        if (blockIdx.x == 0 && threadIdx.x == 0) { // what OP_BETA does
            sc->red[0] = rr;
            sc->red[1] = rz;
            sc->rr = rr;
            if (done == 2) sc->done = 2;
            else {
                if (parity) sc->rho = rz;
                else sc->rho_alt = rz;
                sc->beta = beta;
                sc->iter += 1;
            }
        }
    }
    if (done == 2) {
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
            x[i] += alpha * p[i];
        return;
    }
    // two rows per thread: 16-B loads / stores (the arrays are 16-B aligned; the odd last row goes alone)
    const int64_t n2 = n >> 1;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n2; j += (int64_t)gridDim.x * kBlock) {
        const int64_t i = 2 * j;
        const d2_t pv = *reinterpret_cast<const d2_t *>(p + i);
        const d2_t xv = stream_load_k<KEEP>(reinterpret_cast<const d2_t *>(x + i));
        const d2_t rv = stream_load_k<KEEP>(reinterpret_cast<const d2_t *>(r + i));
        double id0, id1;
        if (CODED) {
            const unsigned cc = stream_load_k<KEEP>(reinterpret_cast<const unsigned *>(dcode + i));
            id0 = invd[cc & 0xffffu];
            id1 = invd[cc >> 16];
        } else {
            const d2_t iv = *reinterpret_cast<const d2_t *>(invd + i);
            id0 = iv.x;
            id1 = iv.y;
        }
        d2_t xn, pn;
        xn.x = xv.x + alpha * pv.x;       xn.y = xv.y + alpha * pv.y;
        pn.x = id0 * rv.x + beta * pv.x;  pn.y = id1 * rv.y + beta * pv.y;
        stream_store_k<KEEP>(xn, reinterpret_cast<d2_t *>(x + i)); // (x is touched once per iteration)
        *reinterpret_cast<d2_t *>(p + i) = pn;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double pi = p[i];
        x[i] += alpha * pi;
        p[i] = (CODED ? invd[dcode[i]] : invd[i]) * r[i] + beta * pi;
    }
}

// ---------------------------------------------------------------------------------------------
// Round 6 (review r05 #3): k_update_r and k_update_xp as ONE launch for HBM-sized systems -- r -= alpha t with its two sums, a GRID
// BARRIER, then x += alpha p ; p = invd r + beta p with the new r (and the diagonal codes) still in REGISTERS: 7.25 n instead of
// 8.5 n doubles of traffic per iteration and one launch boundary less.  kFusedGrid workgroups of 1024 threads, one per CU (128 registers
// per lane: 16 rows of r per lane x 2); every sum is formed in exactly the order of the two kernels it replaces -- thread T of
// workgroup B plays the threads (T & 255) of the 256-thread workgroups 4 B + (T >> 8) and 1024 + 4 B + (T >> 8) of the
// kVecGrid-workgroup launches (same rows, same order, the same wave / four-wave folds, the same partial-sum arrays, the same 256-lane
// folds of them) -- so the iteration, its scalars and the solution are bit-identical with the option on or off.
// The barrier: the partial sums leave with write-through stores (agent scope), one monotonic 64-bit counter (never reset: a launch's
// target is the next multiple of the grid above its own ticket), relaxed polls with s_sleep, bounded by wall_clock64 -- a grid that is
// not co-resident in time (the GPU shared with other work) sets sc->fault instead of hanging.  No L2 write-back fence anywhere: what
// crosses the barrier is read with agent-scope loads.  Plain launch (a graph node like any other); needs every CU free, which holds behind the persistent SpMV.
// ---------------------------------------------------------------------------------------------
static constexpr size_t kFusedTabBytes = (size_t)(2048 + 1) * sizeof(double) + 8; // CODED: the inverted dictionary (kViLdsTable + 1 entries)
static constexpr size_t kFusedLds = (size_t)8 * 1024 * 16; // the second emulated workgroup's new r: kFusedPairs x kFusedBlock pairs of doubles
static constexpr unsigned kFusedStride = (unsigned)kVecGrid * kBlock * 16u; // bytes between two pairs of an emulated thread
static constexpr int kFusedBlock = 1024, kFusedPairs = 8, kFusedBatch = 4; // (loads in flight per lane: kFusedBatch pairs of rows of each stream)
// (kFusedPairs pairs of rows per emulated thread: n <= 2 * kFusedPairs * kVecGrid * kBlock rows)
// element at a 32-bit BYTE offset from a uniform base (global_load ... saddr: no 64-bit address per lane and stream kept in registers)
template <class T> __device__ __forceinline__ T *at_off(T *base, unsigned byte_off)
{
    return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + (size_t)byte_off);
}
template <class T> __device__ __forceinline__ const T *at_off(const T *base, unsigned byte_off)
{
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)byte_off);
}
__device__ __forceinline__ double quarter_sum(double v, double *lds4, int t) // block_sum of the 256-thread workgroup this quarter plays (valid where t == 0)
{
    v = wave_sum(v);
    __syncthreads();
    if ((t & 63) == 0) lds4[t >> 6] = v;
    __syncthreads();
    double s = 0.;
    if (t == 0) s = ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
    return s;
}
template <bool CODED, bool KEEP>
__global__ __launch_bounds__(kFusedBlock) void k_update_fused(int64_t n, double *__restrict__ x, double *__restrict__ p, double *__restrict__ r,
                                                              const double *__restrict__ t, const double *__restrict__ invd,
                                                              const uint16_t *__restrict__ dcode, PcgScalars *sc, double *__restrict__ vpart,
                                                              const double *__restrict__ spmv_partial, int nb, int parity,
                                                              unsigned long long *__restrict__ bar, long long timeout_ticks, int tab_n)
{
    {
        const int d = sc->done;
        if (d) {
            if (blockIdx.x == 0 && threadIdx.x == 0 && d == 2) sc->done = 1; // the pending x update has run
            return;
        }
    }
    __shared__ double red[4][4];
    __shared__ double bc[4];
    __shared__ int sh_fail;
    const int T = threadIdx.x, tq = T & 255, q = T >> 8;
    constexpr int g = kVecGrid;
    if (CODED) { // the inverted dictionary into LDS (a look-up per row and phase: from L1 it was a second, dependent round trip per batch)
        extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds0[];
        double *tw = reinterpret_cast<double *>(fused_lds0 + kFusedLds);
        for (int i = T; i < tab_n; i += kFusedBlock) tw[i] = invd[i];
    }
    // ---- alpha: the fold of the SpMV's partial sums as every workgroup of k_update_r<FUSED> does it (the barriers in it publish the table)
    double pap = 0.;
    if (q == 0)
        for (int k = tq; k < nb; k += kBlock) pap += spmv_partial[k];
    pap = quarter_sum(pap, red[q], tq);
    if (T == 0) { bc[0] = pap; sh_fail = 0; }
    __syncthreads();
    pap = bc[0];
    const double rho_old = parity ? sc->rho_alt : sc->rho;
    const double alpha = rho_old / pap;
    // ---- phase 1: r -= alpha t   (loads of a lane beyond the last pair are clamped to pair 0 and their results dropped: no branches
    //      around the loads, one predicate per store and per sum)
 const int64_t n2 = n >> 1;
    const unsigned lim = (unsigned)n2 * 16u;               // byte offset behind the last pair
    d2_t rn[kFusedPairs];                                  // the new r of the first emulated workgroup's rows: registers ...
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
    d2_t *rl = reinterpret_cast<d2_t *>(fused_lds);        // ... of the second one's: LDS, rl[m * kFusedBlock + T] (128 KiB)
    const double *tab = reinterpret_cast<const double *>(fused_lds + kFusedLds); // CODED: the inverted dictionary (tab_n entries), staged below
    double rr[2] = {0., 0.}, rz[2] = {0., 0.};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int b = (int)blockIdx.x * 4 + q + (g / 2) * h;
        const unsigned o0 = ((unsigned)b * kBlock + (unsigned)tq) * 16u; // byte offset of this lane's first pair (n < 2^28 rows: checked by the host)
#pragma unroll
        for (int m0 = 0; m0 < kFusedPairs; m0 += kFusedBatch) {
            d2_t rv[kFusedBatch], tv[kFusedBatch], idv[kFusedBatch];
            unsigned cc[kFusedBatch];
#pragma unroll
            for (int u = 0; u < kFusedBatch; ++u) {
                const unsigned oj = o0 + (unsigned)(m0 + u) * kFusedStride;
                const unsigned o = oj < lim ? oj : 0u;
                rv[u] = *reinterpret_cast<const d2_t *>(at_off(r, o));
                tv[u] = stream_load_k<KEEP>(reinterpret_cast<const d2_t *>(at_off(t, o)));
                if (CODED) cc[u] = *reinterpret_cast<const unsigned *>(at_off(dcode, o >> 2)); // (read again in phase 2: 0.25 n)
                else idv[u] = *reinterpret_cast<const d2_t *>(at_off(invd, o));
            }
#pragma unroll
            for (int u = 0; u < kFusedBatch; ++u) {
                const int m = m0 + u;
                const unsigned oj = o0 + (unsigned)m * kFusedStride;
                double id0, id1;
                if (CODED) { id0 = tab[cc[u] & 0xffffu]; id1 = tab[cc[u] >> 16]; }
                else { id0 = idv[u].x; id1 = idv[u].y; }
                d2_t v;
                v.x = rv[u].x - alpha * tv[u].x;
                v.y = rv[u].y - alpha * tv[u].y;
                if (h == 0) rn[m] = v;
                else rl[m * kFusedBlock + T] = v;
                if (oj < lim) {
                    *reinterpret_cast<d2_t *>(at_off(r, oj)) = v;
                    rr[h] += v.x * v.x;
                    rz[h] += v.x * (id0 * v.x);
                    rr[h] += v.y * v.y;
                    rz[h] += v.y * (id1 * v.y);
                }
            }
            asm volatile("" ::: "memory"); // (the next batch's loads stay behind this batch: registers)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double r_last = 0.;
    if ((n & 1) && blockIdx.x == 0 && T == 0) { // (the odd last row: thread 0 of workgroup 0 of k_update_r)
        const int64_t i = n - 1;
        r_last = r[i] - alpha * t[i];
        r[i] = r_last;
        rr[0] += r_last * r_last;
        rz[0] += r_last * ((CODED ? invd[dcode[i]] : invd[i]) * r_last);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double srr = quarter_sum(rr[h], red[q], tq);
        const double srz = quarter_sum(rz[h], red[q], tq);
        if (tq == 0) { // write-through: the other workgroups read these behind the barrier
            const int b = (int)blockIdx.x * 4 + q + (g / 2) * h;
            __hip_atomic_store(vpart + b, srr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(vpart + g + b, srz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- grid barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the partial sums are acknowledged before the ticket is drawn
    __syncthreads();
    if (T == 0) {
        const unsigned long long ticket = __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = (ticket / gridDim.x + 1ull) * gridDim.x;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > timeout_ticks) { sh_fail = 1; break; }
        }
    }
    __syncthreads();
    if (sh_fail) { // not every workgroup arrived in time: the solve is void (the host reads sc->fault and redoes it with the two launches)
        if (T == 0) {
            __hip_atomic_store(&sc->fault, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sc->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // ---- beta: the fold of the 2 g partial sums as every workgroup of k_update_xp<FUSED> does it
    double frr = 0., frz = 0.;
    if (q == 0)
        for (int i = tq; i < g; i += kBlock) {
            frr += __hip_atomic_load(vpart + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            frz += __hip_atomic_load(vpart + g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    frr = quarter_sum(frr, red[q], tq);
    frz = quarter_sum(frz, red[q], tq);
    if (T == 0) { bc[1] = frr; bc[2] = frz; }
    __syncthreads();
    frr = bc[1];
    frz = bc[2];
    int done = 0;
    double beta = 0.;
    if (frr < sc->threshold) done = 2; // Eigen: break before i++ (x += alpha p still runs)
    else beta = frz / rho_old;
    if (blockIdx.x == 0 && T == 0) { // what OP_ALPHA and OP_BETA do
        sc->pAp = pap;
        sc->alpha = alpha;
        sc->red[0] = frr;
        sc->red[1] = frz;
        sc->rr = frr;
        if (done == 2) sc->done = 2;
        else {
            if (parity) sc->rho = frz;
            else sc->rho_alt = frz;
            sc->beta = beta;
            sc->iter += 1;
        }
    }
    // ---- phase 2: x += alpha p ; p = invd r + beta p
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int b = (int)blockIdx.x * 4 + q + (g / 2) * h;
        unsigned o0 = ((unsigned)b * kBlock + (unsigned)tq) * 16u;
        asm volatile("" : "+v"(o0)); // (opaque: the addresses are formed again here instead of being carried, spilled, across the barrier from phase 1)
#pragma unroll
        for (int m0 = 0; m0 < kFusedPairs; m0 += kFusedBatch) {
            d2_t pv[kFusedBatch], xv[kFusedBatch], idv[kFusedBatch]; // (the diagonal codes / the inverse diagonal are read again, r stays in registers)
            unsigned cc[kFusedBatch];
#pragma unroll
            for (int u = 0; u < kFusedBatch; ++u) {
                const unsigned oj = o0 + (unsigned)(m0 + u) * kFusedStride;
                const unsigned o = oj < lim ? oj : 0u;
                pv[u] = *reinterpret_cast<const d2_t *>(at_off(p, o));
                xv[u] = stream_load_k<KEEP>(reinterpret_cast<const d2_t *>(at_off(x, o)));
                if (CODED) cc[u] = stream_load_k<KEEP>(reinterpret_cast<const unsigned *>(at_off(dcode, o >> 2)));
                else idv[u] = *reinterpret_cast<const d2_t *>(at_off(invd, o));
            }
#pragma unroll
            for (int u = 0; u < kFusedBatch; ++u) {
                const int m = m0 + u;
                const unsigned oj = o0 + (unsigned)m * kFusedStride;
                double id0, id1;
                if (CODED) { id0 = tab[cc[u] & 0xffffu]; id1 = tab[cc[u] >> 16]; }
                else { id0 = idv[u].x; id1 = idv[u].y; }
                d2_t xn, pn;
                const d2_t rv = h == 0 ? rn[m] : rl[m * kFusedBlock + T];
                xn.x = xv[u].x + alpha * pv[u].x;
                xn.y = xv[u].y + alpha * pv[u].y;
                pn.x = id0 * rv.x + beta * pv[u].x;
                pn.y = id1 * rv.y + beta * pv[u].y;
                if (oj < lim) {
                    stream_store_k<KEEP>(xn, reinterpret_cast<d2_t *>(at_off(x, oj)));
                    if (done != 2) *reinterpret_cast<d2_t *>(at_off(p, oj)) = pn;
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if ((n & 1) && blockIdx.x == 0 && T == 0) {
        const int64_t i = n - 1;
        const double pi = p[i];
        x[i] += alpha * pi;
        if (done != 2) p[i] = (CODED ? invd[dcode[i]] : invd[i]) * r_last + beta * pi;
    }
}

// ---------------------------------------------------------------------------------------------
// scalar stages.  One 256-thread block sums `nb` partials of `nred` interleaved arrays in a fixed
// order (deterministic), then (optionally) applies the scalar update.
// ---------------------------------------------------------------------------------------------
static constexpr int kRedBlock = 1024;
__global__ __launch_bounds__(kRedBlock) void k_reduce(const double *__restrict__ partial, int nb, int nred,
                                                      PcgScalars *sc, int op, double tol, int skip_if_done, int red_off = 0)
{
    if (skip_if_done && sc->done) {
        if (threadIdx.x == 0 && (op == OP_ALPHA || op == OP_ALPHA_ODD) && sc->done == 2) sc->done = 1;
        return;
    }
    __shared__ double red[kRedBlock / 64];
    for (int q = 0; q < nred; ++q) {
        const double *src = partial + (size_t)q * nb;
        double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
        int i = threadIdx.x;
        for (; i + 3 * kRedBlock < nb; i += 4 * kRedBlock) { // 4 independent loads in flight
            const double a = src[i], b = src[i + kRedBlock], c = src[i + 2 * kRedBlock], d = src[i + 3 * kRedBlock];
            s0 += a; s1 += b; s2 += c; s3 += d;
        }
        for (; i < nb; i += kRedBlock) s0 += src[i];
        double s = wave_sum((s0 + s1) + (s2 + s3));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.;
#pragma unroll
            for (int w = 0; w < kRedBlock / 64; ++w) t += red[w];
            sc->red[red_off + q] = t;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && op != OP_NONE) apply_scalar_op(sc, op, tol);
}

// Same job with kRedBlocks workgroups for long partial arrays (the value-indexed SpMV leaves one partial per wave:
// 116 k at 512^3): block b sums a fixed contiguous share (two loads per thread, one round trip), the block that arrives last
// folds the kRedBlocks sums with a fixed tree and applies the scalar update -- the result does not depend on the arrival order.
static constexpr int kRedBlocks = 64; // <= 64: the last block folds them with one wave
__global__ __launch_bounds__(kRedBlock) void k_reduce_mb(const double *__restrict__ partial, int nb, int nred, PcgScalars *sc, int op,
                                                        double tol, int skip_if_done, int red_off, double *__restrict__ stage,
                                                        unsigned *__restrict__ ticket)
{
    if (skip_if_done && sc->done) { // every block sees a non-zero flag whether or not block 0 has already stepped it
        if (blockIdx.x == 0 && threadIdx.x == 0 && (op == OP_ALPHA || op == OP_ALPHA_ODD) && sc->done == 2) sc->done = 1;
        return;
    }
    __shared__ double red[kRedBlock / 64];
    __shared__ int last;
    const int share = (nb + kRedBlocks - 1) / kRedBlocks;
    const int lo = blockIdx.x * share, hi = (lo + share < nb) ? lo + share : nb;
    for (int q = 0; q < nred; ++q) {
        const double *src = partial + (size_t)q * nb;
        double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
        int i = lo + threadIdx.x;
        for (; i + 3 * kRedBlock < hi; i += 4 * kRedBlock) {
            const double a = src[i], b = src[i + kRedBlock], c = src[i + 2 * kRedBlock], d = src[i + 3 * kRedBlock];
            s0 += a; s1 += b; s2 += c; s3 += d;
        }
        for (; i < hi; i += kRedBlock) s0 += src[i];
        double s = wave_sum((s0 + s1) + (s2 + s3));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.;
#pragma unroll
            for (int w = 0; w < kRedBlock / 64; ++w) t += red[w];
            __hip_atomic_store(stage + q * kRedBlocks + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (threadIdx.x == 0) {
        // the block sums went out as write-through (agent-scope) atomic stores: waiting for their acknowledgement orders them
        // before the ticket -- an agent-scope release fence would also write back the XCD's dirty L2 (the y just computed)
        wait_own_stores();
        last = atomicAdd(ticket, 1u) == (unsigned)(kRedBlocks - 1);
    }
    __syncthreads();
    if (!last || threadIdx.x >= 64) return;
    for (int q = 0; q < nred; ++q) { // the workgroups ran on different XCDs (L2s): agent-scope atomic loads bypass the caches
        const double v = threadIdx.x < kRedBlocks ? __hip_atomic_load(stage + q * kRedBlocks + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.;
        const double t = wave_sum(v); // one load per lane, fixed summation tree: independent of the arrival order
        if (threadIdx.x == 0) sc->red[red_off + q] = t;
    }
    if (threadIdx.x != 0) return;
    *ticket = 0u;
    if (op != OP_NONE) apply_scalar_op(sc, op, tol);
}

__global__ void k_scalar(PcgScalars *sc, int op, double tol)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) apply_scalar_op(sc, op, tol);
}

// two partial sets in one launch (multi-GPU loop: its iterations are launch-bound, every launch saved counts):
// red[0 .. nredA) from A, red[nredA .. nredA + nredB) from B
__global__ __launch_bounds__(kRedBlock) void k_reduce_pair(const double *__restrict__ pa, int nba, int nreda, const double *__restrict__ pb,
                                                          int nbb, int nredb, PcgScalars *sc)
{
    __shared__ double red[kRedBlock / 64];
    for (int q = 0; q < nreda + nredb; ++q) {
        const bool first = q < nreda;
        const int nb = first ? nba : nbb;
        const double *src = first ? pa + (size_t)q * nba : pb + (size_t)(q - nreda) * nbb;
        double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
        int i = threadIdx.x;
        for (; i + 3 * kRedBlock < nb; i += 4 * kRedBlock) {
            const double a = src[i], b = src[i + kRedBlock], c = src[i + 2 * kRedBlock], d = src[i + 3 * kRedBlock];
            s0 += a; s1 += b; s2 += c; s3 += d;
        }
        for (; i < nb; i += kRedBlock) s0 += src[i];
        double s = wave_sum((s0 + s1) + (s2 + s3));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.;
#pragma unroll
            for (int w = 0; w < kRedBlock / 64; ++w) t += red[w];
            sc->red[q] = t;
        }
    }
}

static void reduce_launch(PcgWork *w, const double *partial, int nb, int nred, PcgScalars *sc, int op, double tol, int skip_if_done,
                          int red_off, hipStream_t stream)
{
    if (nb >= 16384 && w->stage.p && w->ticket.p)
        hipLaunchKernelGGL(k_reduce_mb, dim3(kRedBlocks), dim3(kRedBlock), 0, stream, partial, nb, nred, sc, op, tol, skip_if_done, red_off,
                           w->stage.p, w->ticket.p);
    else
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kRedBlock), 0, stream, partial, nb, nred, sc, op, tol, skip_if_done, red_off);
}


// ---------------------------------------------------------------------------------------------
// Single-reduction PCG (Chronopoulos & Gear 1989) -- used when the solve is partitioned over several
// GPUs: the same Krylov iterates in exact arithmetic, but gamma = r.u, delta = w.u and |r|^2 are
// all-reduced together, so an iteration has ONE all-reduce and ONE halo exchange (of u = M^-1 r)
// instead of two all-reduces + one exchange.  Costs one more vector (s = A p by recurrence): 12 n
// doubles of vector traffic instead of 10 n -- irrelevant once the solve is latency-bound.
//   u = M^-1 r ; w = A u ; p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_sr_init(int64_t n, const double *__restrict__ b, const double *__restrict__ t,
                                                    const double *__restrict__ invd, double *__restrict__ r,
                                                    double *__restrict__ u, double *__restrict__ partial)
{
    __shared__ double red[4];
    double bb = 0., ru = 0., rr = 0.;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double bi = b[i];
        const double ri = bi - t[i];
        const double ui = invd[i] * ri;
        r[i] = ri;
        u[i] = ui;
        bb += bi * bi;
        ru += ri * ui;
        rr += ri * ri;
    }
    bb = block_sum(bb, red);
    ru = block_sum(ru, red);
    rr = block_sum(rr, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = bb;
        partial[gridDim.x + blockIdx.x] = ru;
        partial[2 * gridDim.x + blockIdx.x] = rr;
    }
}

// The scalar step of the previous iteration (OP_SR_STEP on the all-reduced sums) is folded into the start of this
// kernel when `step` is set: every workgroup computes the same new scalars from state `in`, workgroup 0 stores them as
// state `out` (a different PcgScalars: nobody reads what is being written) -- one launch less per iteration of a loop
// that is launch-bound at 8 GPUs.  in == out and step == 0: scalars are used as they are.
__global__ __launch_bounds__(kBlock) void k_sr_update(int64_t n, double *__restrict__ x, double *__restrict__ r,
                                                      double *__restrict__ p, double *__restrict__ s,
                                                      double *__restrict__ u, const double *__restrict__ w,
                                                      const double *__restrict__ invd, const PcgScalars *in, PcgScalars *out,
                                                      int step, double *__restrict__ partial)
{
    int done = in->done;
    double alpha = in->alpha, beta = in->beta;
    if (step) {
        double rr = in->rr, rho = in->rho;
        int iter = in->iter;
        if (!done) { // OP_SR_STEP, red = [r.u, r.r, w.u]
            rr = in->red[1];
            if (in->red[1] < in->threshold) done = 1;
            else {
                const double gamma = in->red[0], delta = in->red[2];
                const double b2 = gamma / rho;
                alpha = gamma / (delta - b2 * gamma / alpha);
                beta = b2;
                rho = gamma;
                iter += 1;
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            PcgScalars o = *in;
            o.rr = rr; o.rho = rho; o.alpha = alpha; o.beta = beta; o.iter = iter; o.done = done;
            *out = o;
        }
    }
    if (done == 3) {
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) x[i] = 0.;
        return;
    }
    if (done) return;
    __shared__ double red[4];
    double ru = 0., rr = 0.;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double pi = u[i] + beta * p[i];
        const double si = w[i] + beta * s[i];
        p[i] = pi;
        s[i] = si;
        x[i] += alpha * pi;
        const double ri = r[i] - alpha * si;
        r[i] = ri;
        const double ui = invd[i] * ri;
        u[i] = ui;
        ru += ri * ui;
        rr += ri * ri;
    }
    ru = block_sum(ru, red);
    rr = block_sum(rr, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = ru;
        partial[gridDim.x + blockIdx.x] = rr;
    }
}

static avs_status pcg_solve_single_reduction(PcgWork *w, const CsrView &A, const double *b, double *x, double tol,
                                             int max_iters, hipStream_t stream, avs_solve_info *info, PcgDist *dist)
{
    const int64_t n = A.n;
    const int g = (int)((n + kBlock - 1) / kBlock < kVecGrid ? ((n + kBlock - 1) / kBlock > 0 ? (n + kBlock - 1) / kBlock : 1) : kVecGrid);
    const int rowgrid = (int)((n + kBlock - 1) / kBlock) > 0 ? (int)((n + kBlock - 1) / kBlock) : 1;
    const int variant = spmv_default_variant(A);
    AVS_TRY(w->s.alloc((size_t)n));
    AVS_TRY(w->u.alloc((size_t)w->n_ext));
    double *p = w->p.p, *r = w->r.p, *wv = w->t.p, *sv = w->s.p, *u = w->u.p, *invd = w->invd.p;
    double *pvec = w->partial.p;                        // 3 * g vector-kernel partials
    double *pspmv = w->partial.p + 4 * (size_t)kVecGrid; // SpMV partials behind them
    PcgScalars *sc = w->sc.p; // two states, ping-pong: sc[cur] is current, k_sr_update writes sc[cur ^ 1]
    int cur = 0;
    auto red_of = [&](int k) { return reinterpret_cast<double *>(reinterpret_cast<char *>(sc + cur) + offsetof(PcgScalars, red)) + k; };

    AVS_HIP(hipMemsetAsync(sc, 0, 2 * sizeof(PcgScalars), stream));
    AVS_HIP(hipMemsetAsync(p, 0, (size_t)w->n_ext * sizeof(double), stream));
    AVS_HIP(hipMemsetAsync(sv, 0, (size_t)n * sizeof(double), stream));
    hipLaunchKernelGGL(k_inv_diag, dim3(rowgrid), dim3(kBlock), 0, stream, A, invd);
    AVS_HIP(hipEventRecord(w->ev0, stream));
    // r = b - A x (x staged through u for the halo), u = M^-1 r, w = A u
    AVS_HIP(hipMemcpyAsync(u, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    AVS_TRY(dist_halo_exchange(dist, u, stream));
    AVS_TRY(spmv_dispatch<false>(A, u, wv, nullptr, nullptr, variant, stream, nullptr));
    hipLaunchKernelGGL(k_sr_init, dim3(g), dim3(kBlock), 0, stream, n, b, wv, invd, r, u, pvec);
    AVS_TRY(dist_halo_exchange(dist, u, stream));
    int nb = 0;
    AVS_TRY(spmv_dispatch<true>(A, u, wv, pspmv, nullptr, variant, stream, &nb));
    reduce_launch(w, pvec, g, 3, sc, (int)OP_NONE, tol, 0, 0, stream);
    reduce_launch(w, pspmv, nb, 1, sc, (int)OP_NONE, tol, 0, 3, stream);
    AVS_TRY(dist_allreduce(dist, red_of(0), 4, stream));
    hipLaunchKernelGGL(k_scalar, dim3(1), dim3(64), 0, stream, sc, (int)OP_SR_INIT, tol);
    AVS_HIP(hipGetLastError());

    int enqueued = 0, last_chunk = 0, spmv_samples = 0;
    double spmv_ms_sum = 0.;
    bool cancelled = false;
    AVS_TRY(w->cancel_word.alloc(2));
    while (true) {
        // avs_cancel: every rank contributes its request (0 / 1), the sum is the verdict of ALL ranks for this chunk boundary
        double stop_h[2] = {cancel_requested() ? 1. : 0., 0.};
        AVS_HIP(hipMemcpyAsync(w->cancel_word.p, stop_h, sizeof(double), hipMemcpyHostToDevice, stream));
        AVS_TRY(dist_allreduce(dist, w->cancel_word.p, 1, stream));
        AVS_HIP(hipMemcpyAsync(stop_h + 1, w->cancel_word.p, sizeof(double), hipMemcpyDeviceToHost, stream));
        AVS_HIP(hipMemcpyAsync(w->host_sc, sc + cur, sizeof(PcgScalars), hipMemcpyDeviceToHost, stream));
        AVS_HIP(hipStreamSynchronize(stream));
        if (info && last_chunk > 0) {
            const int ran = w->host_sc->iter + (w->host_sc->done ? 1 : 0);
            const int first = enqueued - last_chunk;
            for (int c2 = 0; c2 < last_chunk && first + c2 < ran; c2 += kSampleEvery) {
                float ems = 0.f;
                if (hipEventElapsedTime(&ems, w->evA[c2], w->evB[c2]) == hipSuccess) { spmv_ms_sum += ems; ++spmv_samples; }
            }
        }
        if (w->host_sc->done || enqueued >= max_iters) break;
        if (stop_h[1] != 0.) { (void)cancel_consume(); cancelled = true; break; }
        const int chunk = (max_iters - enqueued) < kChunk ? (max_iters - enqueued) : kChunk;
        for (int c = 0; c < chunk; ++c) {
            // first iteration of a chunk: scalars are final (SR_INIT or the explicit step below); afterwards the step of
            // the previous iteration rides in k_sr_update, which moves the state to the other slot
            const int step = c > 0 ? 1 : 0;
            const bool timed = info && (c % kSampleEvery == 0); // SpMV timing samples: every 4th iteration
            hipLaunchKernelGGL(k_sr_update, dim3(g), dim3(kBlock), 0, stream, n, x, r, p, sv, u, wv, invd, (const PcgScalars *)(sc + cur),
                               sc + (step ? (cur ^ 1) : cur), step, pvec);
            if (step) cur ^= 1;
            const PcgScalars *now = sc + cur;
            const int32_t *t_int = nullptr, *t_bnd = nullptr;
            int n_int = 0, n_bnd = 0;
            if (variant == 24 && !(A.brick && A.brick->ntiles > 0) && dist_tile_lists(dist, &t_int, &n_int, &t_bnd, &n_bnd)) {
                // overlap: the exchange runs on the communication stream while the tiles that touch no halo
                // column are multiplied; the halo-touching tiles follow once the halo has landed
                AVS_TRY(dist_halo_begin(dist, u, stream));
                if (timed) AVS_HIP(hipEventRecord(w->evA[c], stream));
                AVS_TRY(spmv_dot_tiles(A, u, wv, pspmv, now, t_int, n_int, stream));
                AVS_TRY(dist_halo_end(dist, stream));
                AVS_TRY(spmv_dot_tiles(A, u, wv, pspmv + (size_t)n_int * (A.codes ? kTileRows / 64 : 1), now, t_bnd, n_bnd, stream));
                if (timed) AVS_HIP(hipEventRecord(w->evB[c], stream));
                nb = (n_int + n_bnd) * (A.codes ? kTileRows / 64 : 1); // value-indexed kernel: one partial per wave
            } else {
                AVS_TRY(dist_halo_exchange(dist, u, stream));
                if (timed) AVS_HIP(hipEventRecord(w->evA[c], stream));
                AVS_TRY(spmv_dispatch<true>(A, u, wv, pspmv, now, variant, stream, &nb));
                if (timed) AVS_HIP(hipEventRecord(w->evB[c], stream));
            }
            if (nb < 16384) hipLaunchKernelGGL(k_reduce_pair, dim3(1), dim3(kRedBlock), 0, stream, pvec, g, 2, pspmv, nb, 1, sc + cur);
            else {
                reduce_launch(w, pvec, g, 2, sc + cur, (int)OP_NONE, tol, 0, 0, stream);
                reduce_launch(w, pspmv, nb, 1, sc + cur, (int)OP_NONE, tol, 0, 2, stream);
            }
            AVS_TRY(dist_allreduce(dist, red_of(0), 3, stream));
        }
        // the last iteration's step, explicitly: the host polls a final state
        hipLaunchKernelGGL(k_scalar, dim3(1), dim3(64), 0, stream, sc + cur, (int)OP_SR_STEP, tol);
        AVS_HIP(hipGetLastError());
        enqueued += chunk;
        last_chunk = chunk;
    }
    if (w->host_sc->done == 3) { // rhs == 0: x := 0
        hipLaunchKernelGGL(k_sr_update, dim3(g), dim3(kBlock), 0, stream, n, x, r, p, sv, u, wv, invd, (const PcgScalars *)(sc + cur), sc + cur, 0,
                           pvec);
    }
    AVS_HIP(hipEventRecord(w->ev1, stream));
    AVS_HIP(hipEventSynchronize(w->ev1));
    float ms = 0.f;
    AVS_HIP(hipEventElapsedTime(&ms, w->ev0, w->ev1));
    if (info) {
        const PcgScalars &h = *w->host_sc;
        info->iterations = h.iter;
        info->converged = (h.done != 0) ? 1 : 0;
        info->rhs_norm2 = h.rhs_norm2;
        info->error = (h.done == 3 || h.rhs_norm2 == 0.) ? 0. : sqrt(h.rr / h.rhs_norm2);
        info->n = n;
        info->nnz = A.nnz;
        info->solve_ms = ms;
        info->spmv_ms = spmv_samples ? spmv_ms_sum / spmv_samples : 0.;
        info->resident = 0;
        info->cancelled = cancelled ? 1 : 0;
    }
    return AVS_OK;
}

// ---------------------------------------------------------------------------------------------
// Direct-transport loop (world >= 1): the single-reduction iteration above with NO RCCL call and no host work inside:
//   k_sr_update_push  p, s, x, r, u + partials of r.u, |r|^2; then the boundary entries of u -> the peers' halo areas and the
//                     epoch flag (last pushing block).  (k_push does the same for the two set-up rounds.)
//   SpMV, all tiles   interior tiles run while the peers' entries travel; the tiles that read halo columns (the last ones) wait
//                     for the flags; every wave drops its x.Ax partial into a stage slot (fire and forget); the finalizer blocks
//                     (dispatched last) watch the slots, fold them, and the last one all-gathers the 3 sums with every rank and
//                     applies the scalar step
// = 2 launches per iteration, replayed from one hipGraph per chunk of kChunk iterations.
// ---------------------------------------------------------------------------------------------
void sr_update_geometry(long long n, int *grid, int *chunk)
{
    long long g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kVecGrid) g = kVecGrid;
    long long c = (n + g - 1) / g;
    c = (c + kBlock - 1) / kBlock * kBlock;
    if (c < kBlock) c = kBlock;
    *chunk = (int)c;
    *grid = (int)((n + c - 1) / c > 0 ? (n + c - 1) / c : 1);
}

// One thread of every pushing workgroup, AFTER the workgroup's stores have been acknowledged (wait_own_stores + barrier): take a
// ticket; the last one raises this rank's flag in the blocks of the peers it feeds.  Paranoid mode: the round's checksums first,
// acknowledged, then the flags.
__device__ __forceinline__ void push_raise_flags(const DistDev *dd, const unsigned long long *epoch, unsigned *ticket, unsigned nblocks)
{
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t != nblocks - 1u) return;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int np = dd->npeers;
    const unsigned long long E = *epoch + 1ull;
    if (dd->paranoid) {
        for (int i = 0; i < np; ++i)
            if (dd->send_off[i + 1] > dd->send_off[i])
                st_sys(dd->peer_hsum_dst[i], __hip_atomic_exchange(dd->psum + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        wait_own_stores();
    }
    for (int i = 0; i < np; ++i)
        if (dd->send_off[i + 1] > dd->send_off[i]) st_sys(dd->peer_hflag_dst[i], E);
}

// k_sr_update + k_push in one launch: workgroup b owns a CONTIGUOUS range of rows, updates them, and then stores those of its
// new u entries that a peer reads straight into that peer's halo area; the last pushing workgroup raises the flags.
// CODED (round 3): the diagonal's 2-B value code + the table of inverted values instead of the 8-B inverse (11.25 n instead of
// 12 n doubles of vector traffic; the same doubles, so nothing changes numerically); u is recomputed from r instead of read (10.25 n).  Rows are taken two at a time with 16-B
// loads / stores (ranges start at multiples of the block size, so the pairs are aligned).
// KEEP == false (round 5: the brick-structured form is small enough to stay in the Infinity Cache between two products -- but only
// if the ~0.5 GB this kernel moves do not push it out): p, s, x, r -- read and written once per iteration, by this kernel only -- are
// loaded and stored non-temporally; u (the next product's input) and w (the product's output, read here) stay cacheable.
template <bool CODED, bool KEEP = true>
__global__ __launch_bounds__(kBlock) void k_sr_update_push(int64_t n, double *__restrict__ x, double *__restrict__ r, double *__restrict__ p,
                                                           double *__restrict__ s, double *__restrict__ u, const double *__restrict__ w,
                                                           const double *__restrict__ invd, const uint16_t *__restrict__ dcode,
                                                           const PcgScalars *sc, double *__restrict__ partial,
                                                           const DistDev *__restrict__ dd, const unsigned long long *__restrict__ epoch,
                                                           unsigned *__restrict__ ticket)
{
    const int done = sc->done;
    const int64_t lo = (int64_t)blockIdx.x * dd->push_chunk;
    const int64_t hi = (lo + dd->push_chunk < n) ? lo + dd->push_chunk : n;
    if (done == 3) {
        for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) x[i] = 0.;
        return;
    }
    if (done) return;
    const double alpha = sc->alpha, beta = sc->beta;
    __shared__ double red[4];
    double ru = 0., rr = 0.;
    int64_t i = lo + 2 * (int64_t)threadIdx.x;
    for (; i + 1 < hi; i += 2 * kBlock) { // (lo is a multiple of kBlock: i is even, the 16-B accesses are aligned)
        const d2_t pv = stream_load_k<KEEP>(reinterpret_cast<const d2_t *>(p + i));
        const d2_t wv = *reinterpret_cast<const d2_t *>(w + i), sv = stream_load_k<KEEP>(reinterpret_cast<const d2_t *>(s + i));
        const d2_t xv = stream_load_k<KEEP>(reinterpret_cast<const d2_t *>(x + i)), rv = stream_load_k<KEEP>(reinterpret_cast<const d2_t *>(r + i));
        double id0, id1;
        if (CODED) {
            const unsigned cc = stream_load_k<KEEP>(reinterpret_cast<const unsigned *>(dcode + i));
            id0 = invd[cc & 0xffffu];
            id1 = invd[cc >> 16];
        } else {
            const d2_t iv = *reinterpret_cast<const d2_t *>(invd + i);
            id0 = iv.x;
            id1 = iv.y;
        }
        // u is not read: what memory holds is inv(d) * r of the r just loaded (this kernel, or the set-up round, wrote exactly that product)
        d2_t uv;
        uv.x = id0 * rv.x;
        uv.y = id1 * rv.y;
        d2_t pn, sn, xn, rn, un;
        pn.x = uv.x + beta * pv.x;  pn.y = uv.y + beta * pv.y;
        sn.x = wv.x + beta * sv.x;  sn.y = wv.y + beta * sv.y;
        xn.x = xv.x + alpha * pn.x; xn.y = xv.y + alpha * pn.y;
        rn.x = rv.x - alpha * sn.x; rn.y = rv.y - alpha * sn.y;
        un.x = id0 * rn.x;          un.y = id1 * rn.y;
        stream_store_k<KEEP>(pn, reinterpret_cast<d2_t *>(p + i));
        stream_store_k<KEEP>(sn, reinterpret_cast<d2_t *>(s + i));
        stream_store_k<KEEP>(xn, reinterpret_cast<d2_t *>(x + i));
        stream_store_k<KEEP>(rn, reinterpret_cast<d2_t *>(r + i));
        *reinterpret_cast<d2_t *>(u + i) = un;
        ru += rn.x * un.x;
        rr += rn.x * rn.x;
        ru += rn.y * un.y;
        rr += rn.y * rn.y;
    }
    if (i < hi) { // odd tail of the last range
        const double idi = CODED ? invd[dcode[i]] : invd[i];
        const double pi = idi * r[i] + beta * p[i]; // (u[i] == inv(d) r[i], see above)
        const double si = w[i] + beta * s[i];
        p[i] = pi;
        s[i] = si;
        x[i] += alpha * pi;
        const double ri = r[i] - alpha * si;
        r[i] = ri;
        const double ui = idi * ri;
        u[i] = ui;
        ru += ri * ui;
        rr += ri * ri;
    }
    ru = block_sum(ru, red); // (barriers inside: every u of this range is written before the push below reads it)
    rr = block_sum(rr, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = ru;
        partial[gridDim.x + blockIdx.x] = rr;
    }
    const int np = dd->npeers, G = (int)gridDim.x, b = (int)blockIdx.x;
    const int paranoid = dd->paranoid;
    const bool inject = paranoid && dd->inject_stale_round == (long long)(*epoch + 1ull); // test hook: one entry is NOT stored
    bool any = false;
    for (int i = 0; i < np; ++i) {
        const int a = dd->push_seg[i * (G + 1) + b], e = dd->push_seg[i * (G + 1) + b + 1];
        double *dst = dd->peer_halo_dst[i] - dd->send_off[i];
        unsigned long long cs = 0ull;
        for (int j = a + (int)threadIdx.x; j < e; j += kBlock) {
            const double v = u[dd->send_idx[j]];
            if (!(inject && j == 0)) __hip_atomic_store(dst + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            cs += (unsigned long long)__double_as_longlong(v);
        }
        if (paranoid && e > a) { // (block-uniform condition)
            cs = wave_sum_u64(cs);
            if ((threadIdx.x & 63) == 0 && cs) __hip_atomic_fetch_add(dd->psum + i, cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        any = any || e > a;
    }
    if (!any) return; // block-uniform
    wait_own_stores();
    __syncthreads();
    if (threadIdx.x == 0) push_raise_flags(dd, epoch, ticket, (unsigned)dd->n_send_blocks);
}

__global__ __launch_bounds__(256) void k_push(const DistDev *__restrict__ dd, const double *__restrict__ v,
                                              const unsigned long long *__restrict__ epoch, unsigned *__restrict__ ticket,
                                              const PcgScalars *sc)
{
    if (sc->done) return;
    const int np = dd->npeers;
    const int n_send = dd->send_off[np];
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int paranoid = dd->paranoid;
    const unsigned long long E = *epoch + 1ull;
    int mine = -1;
    unsigned long long bits = 0ull;
    if (j < n_send) {
        int i = 0;
        while (j >= dd->send_off[i + 1]) ++i;
        mine = i;
        // v == nullptr: the transport self-test's pattern of this round instead of vector entries
        bits = v ? (unsigned long long)__double_as_longlong(v[dd->send_idx[j]]) : selftest_pattern(E, dd->rank, j - dd->send_off[i]);
        // a write-through store over xGMI / into the peer process's block
        if (!(paranoid && dd->inject_stale_round == (long long)E && j == 0)) // (test hook: one entry is NOT stored)
            st_sys(reinterpret_cast<unsigned long long *>(dd->peer_halo_dst[i] + (j - dd->send_off[i])), bits);
    }
    if (paranoid) {
        for (int i = 0; i < np; ++i) { // a wave's lanes may feed different peers: one pass per peer
            if (dd->send_off[i + 1] <= (int)blockIdx.x * 256 || dd->send_off[i] >= (int)(blockIdx.x + 1) * 256) continue; // block-uniform
            const unsigned long long cs = wave_sum_u64(mine == i ? bits : 0ull);
            if ((threadIdx.x & 63) == 0 && cs) __hip_atomic_fetch_add(dd->psum + i, cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    wait_own_stores(); // acknowledged by the destination before this wave reaches the barrier
    __syncthreads();
    if (threadIdx.x == 0) push_raise_flags(dd, epoch, ticket, gridDim.x); // every block's stores are out: raise my flag in the blocks of the peers I feed
}

// One round of the transport self-test (direct_selftest): a single workgroup waits for the peers' flags, checks the checksums and
// that every halo entry IS this round's pattern (a stale entry holds the previous round's), then all-gathers the cumulative
// count of bad entries through the same finalisation the solve uses (which is also the barrier between rounds).
__global__ __launch_bounds__(256) void k_direct_selftest(HaloView hv, unsigned long long *bad_total)
{
    if (hv.sc->done) return; // a wait timed out in an earlier round: the peers find out through their own time limit
    const unsigned long long bad = paranoid_check<256>(hv, true);
    __shared__ double cum;
    if (threadIdx.x == 0) {
        *bad_total += (bad & 0xFFFFFFFFull) + (bad >> 32);
        cum = (double)*bad_total;
    }
    __syncthreads();
    double acc[4] = {0., 0., 0., 0.};
    if (threadIdx.x == 0) acc[0] = cum;
    dist_finalize<256>(hv, acc);
}

avs_status spmv_dot_tiles(const CsrView &A, const double *x, double *y, double *partial, const PcgScalars *sc, const int32_t *tiles,
                          int ntiles, hipStream_t stream);

// The finalizer of the direct transport as a launch of its own, behind the brick-structured form's SpMV (whose persistent workgroups
// keep three per CU with the LDS they have: the finalizer's staging would cost the third): folds the round's stage slots and the
// vector kernel's partials, all-gathers the sums with the other ranks and applies the scalar step (halo_finalizer, avs_halo.hpp).
__global__ __launch_bounds__(512) void k_halo_finalize(HaloView hv)
{
    if (hv.sc->done) return;
    halo_finalizer<512>(hv, (int)blockIdx.x);
}
// ... and in FRONT of it: wait for the peers' entries of this round (their epoch flags), then copy the comm block's halo area behind the
// owned entries of the vector the product reads (system-scope loads: the entries were written by other GPUs), so that the brick kernel
// itself is the plain one -- [owned | halo] is one array, as with the RCCL transport.
__global__ __launch_bounds__(256) void k_halo_gather(HaloView hv, double *__restrict__ vec)
{
    if (hv.sc->done) return;
    halo_wait(hv);
    const DistDev *dd = hv.dd;
    int n_halo = 0;
    for (int i = 0; i < dd->npeers; ++i) n_halo += dd->recv_cnt[i];
    const long long n_own = dd->n_own;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n_halo; j += gridDim.x * 256) vec[n_own + j] = ld_sys_f64(dd->my_halo + j);
}

#include "avs_pcg_resident.inl"

// Transport self-test, run once per plan right after the blocks are connected and before the direct transport is trusted with a
// solve (round-2 review: "never run on more than one GPU"): `rounds` full-size halo rounds over the real links with a
// round-dependent pattern, checksum + per-entry verification by the reader while the next flag is already being polled, and the
// bad-entry count all-gathered through the solve's own finalisation.  Every rank gets the same total => the same decision.
avs_status direct_selftest(const DirectArgs &da, int rounds, hipStream_t stream, long long *bad_entries, int *fault)
{
    DevBuf<PcgScalars> sc;
    DevBuf<unsigned long long> bad;
    AVS_TRY(sc.alloc(1));
    AVS_TRY(bad.alloc(1));
    AVS_HIP(hipMemsetAsync(sc.p, 0, sizeof(PcgScalars), stream));
    AVS_HIP(hipMemsetAsync(bad.p, 0, sizeof(unsigned long long), stream));
    const int push_blocks = da.n_send > 0 ? (da.n_send + 255) / 256 : 0;
    HaloView hv;
    hv.dd = da.dd;
    hv.epoch = da.epoch;
    hv.epoch_w = da.epoch;
    hv.fin_ticket = da.fin_ticket;
    hv.sc = sc.p;
    hv.nfin = 1;
    hv.nred_vec = 0;
    hv.op = 0;
    hv.selftest = 1;
    for (int r = 0; r < rounds; ++r) {
        if (push_blocks)
            hipLaunchKernelGGL(k_push, dim3(push_blocks), dim3(256), 0, stream, da.dd, (const double *)nullptr, (const unsigned long long *)da.epoch,
                               da.push_ticket, (const PcgScalars *)sc.p);
        hipLaunchKernelGGL(k_direct_selftest, dim3(1), dim3(256), 0, stream, hv, bad.p);
    }
    AVS_HIP(hipGetLastError());
    PcgScalars h;
    AVS_HIP(hipMemcpyAsync(&h, sc.p, sizeof(h), hipMemcpyDeviceToHost, stream));
    AVS_HIP(hipStreamSynchronize(stream));
    *fault = h.fault;
    *bad_entries = (long long)h.red[0]; // the LAST round's all-gathered cumulative count (all ranks hold the same number)
    return AVS_OK;
}

static avs_status pcg_solve_direct(PcgWork *w, const CsrView &A, const double *b, double *x, double tol, int max_iters,
                                   hipStream_t stream, avs_solve_info *info, const DirectArgs &da)
{
    const int64_t n = A.n;
    int g = 1, chunk_rows = kBlock; // every vector kernel of this loop uses the fused kernel's geometry (same partial layout)
    sr_update_geometry((long long)n, &g, &chunk_rows);
    AVS_REQUIRE(g == da.push_grid && chunk_rows == da.push_chunk, AVS_EINTERNAL, "push segments were built for another geometry");
    const int rowgrid = (int)((n + kBlock - 1) / kBlock) > 0 ? (int)((n + kBlock - 1) / kBlock) : 1;
    AVS_TRY(w->s.alloc((size_t)n));
    AVS_TRY(w->u.alloc((size_t)w->n_ext));
    double *p = w->p.p, *r = w->r.p, *wv = w->t.p, *sv = w->s.p, *u = w->u.p, *invd = w->invd.p;
    double *pvec = w->partial.p;                         // up to 3 * g vector-kernel partials
    PcgScalars *sc = w->sc.p;
    const bool brick = A.brick && A.brick->ntiles > 0;   // brick-structured form of the local rows: one partial per persistent workgroup
    const int ntiles = brick ? brick_partial_count(*A.brick, 8) : da.n_tiles_int + da.n_tiles_bnd;  // (word stream: == ceil(n / kTileRows))
    const int ppt = brick ? 1 : (A.codes ? kTileRows / 64 : 1); // value-indexed kernel: one partial per wave
    const int slots = ntiles * ppt;
    const int nfin = slots > 0 ? (slots + kFinShare - 1) / kFinShare : 1;
    AVS_TRY(w->stage2.alloc((size_t)(slots > 0 ? slots : 1) + (size_t)nfin));
    AVS_HIP(hipMemsetAsync(w->stage2.p, 0xFF, (size_t)(slots > 0 ? slots : 1) * sizeof(double), stream)); // arm: kSentinel in every slot
    const int push_blocks = da.n_send > 0 ? (da.n_send + 255) / 256 : 0;
    const int n_halo_cols = (int)(w->n_ext - n);

    AVS_TRY(w->cancel_dev.alloc(1));
    AVS_HIP(hipMemsetAsync(w->cancel_dev.p, 0, sizeof(int), stream));
    AVS_HIP(hipMemsetAsync(sc, 0, 2 * sizeof(PcgScalars), stream));
    AVS_HIP(hipMemsetAsync(p, 0, (size_t)w->n_ext * sizeof(double), stream));
    AVS_HIP(hipMemsetAsync(sv, 0, (size_t)n * sizeof(double), stream));
    hipLaunchKernelGGL(k_inv_diag, dim3(rowgrid), dim3(kBlock), 0, stream, A, invd);
    // one dictionary of few values: the loop's vector kernel reads a 2-B diagonal code (k_inv_diag_coded; never with tile-local tables)
    const bool coded = A.codes && !A.tab_ptr && A.table_size <= kViLdsTable;
    if (coded) {
        if (!w->dcode.p) AVS_TRY(w->dcode.alloc((size_t)n + 2));
        if (!w->invtab.p) AVS_TRY(w->invtab.alloc((size_t)kViLdsTable + 1));
        const int cg = (int)(((n > A.table_size + 1 ? n : A.table_size + 1) + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_inv_diag_coded, dim3(cg), dim3(kBlock), 0, stream, A, w->dcode.p, w->invtab.p);
    }
    AVS_HIP(hipEventRecord(w->ev0, stream));

    // one round: exchange `vec`, wv = A vec (+ partials of vec.wv), fold `nred_vec` vector partial arrays + that one, step `op`
    auto round = [&](const double *vec, int nred_vec, int op, hipEvent_t ea, hipEvent_t eb, bool push = true) -> avs_status {
        if (push && push_blocks) hipLaunchKernelGGL(k_push, dim3(push_blocks), dim3(256), 0, stream, da.dd, vec, (const unsigned long long *)da.epoch,
                                            da.push_ticket, (const PcgScalars *)sc);
        if (ea) AVS_HIP(hipEventRecord(ea, stream));
        // ONE launch over all tiles + the finalizer block: tiles that read halo columns (flagged; the last ones of the [interior |
        // halo-reading] row order) wait for the peers' flags, everything else multiplies while the halo travels
        HaloView hv;
        hv.dd = da.dd;
        hv.epoch = da.epoch;
        hv.epoch_w = da.epoch;
        hv.fin_ticket = da.fin_ticket;
        hv.sc = sc;
        hv.pvec = pvec;
        hv.stage = w->stage2.p;
        hv.stage2 = w->stage2.p + (slots > 0 ? slots : 1);
        hv.tile_bnd = da.tile_flags;
        hv.ntiles = ntiles;
        hv.ppt = ppt;
        hv.nfin = nfin;
        hv.g = g;
        hv.nred_vec = nred_vec;
        hv.op = op;
        hv.tol = tol;
        hv.cancel = w->cancel_dev.p;
        if (brick) { // halo into the vector's tail, the plain persistent grid (partials into the stage slots), the finalizer: three launches
            if (da.npeers > 0) {
                const int hg = n_halo_cols > 0 ? (n_halo_cols + 255) / 256 : 1;
                hipLaunchKernelGGL(k_halo_gather, dim3(hg < 64 ? hg : 64), dim3(256), 0, stream, hv, const_cast<double *>(vec));
            }
            AVS_TRY(spmv_brick_launch(*A.brick, vec, wv, w->stage2.p, &sc->done, stream));
            hipLaunchKernelGGL(k_halo_finalize, dim3(nfin), dim3(512), 0, stream, hv);
        } else {
            AVS_TRY(spmv_dot_tiles_halo(A, vec, wv, nullptr, sc, nullptr, ntiles + nfin, hv, stream));
        }
        if (eb) AVS_HIP(hipEventRecord(eb, stream));
        return AVS_OK;
    };
    // r = b - A x (x staged through u for the exchange), u = M^-1 r, w = A u
    AVS_HIP(hipMemcpyAsync(u, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    AVS_TRY(round(u, 0, (int)OP_NONE, nullptr, nullptr));
    hipLaunchKernelGGL(k_sr_init, dim3(g), dim3(kBlock), 0, stream, n, b, wv, invd, r, u, pvec);
    AVS_TRY(round(u, 3, (int)OP_SR_INIT, nullptr, nullptr));
    AVS_HIP(hipGetLastError());

    // Systems that fit on the chip (<= ~1 M rows, packed single-dictionary form): the rest of the solve in ONE cooperative launch,
    // matrix words in the register files, vector slices in LDS (avs_pcg_resident.inl).  Needs the GPU for itself.
    w->resident_used = 0;
    if (coded && (da.exclusive_device || cur_opt().resident_cus > 0) && resident_wanted(true)) {
        if (!w->resident) w->resident = new (std::nothrow) ResidentPlan();
        if (w->resident && resident_prepare(w->resident, A, w->n_ext, &da, stream)) {
            bool launched = false;
            AVS_TRY(resident_run(w->resident, A, x, r, p, sv, u, wv, w->dcode.p, w->invtab.p, sc, max_iters, &da, stream, &launched));
            w->resident_used = launched ? 1 : 0; // (refused: the loop below takes over from the same state)
        }
    }

    auto enqueue_iteration = [&](int c, bool timed) -> avs_status {
        // update and push in one launch (3 launches per iteration)
        if (coded && brick) // (the brick form serves single-dictionary matrices: coded)
            hipLaunchKernelGGL((k_sr_update_push<true, false>), dim3(g), dim3(kBlock), 0, stream, n, x, r, p, sv, u, wv, (const double *)w->invtab.p,
                               (const uint16_t *)w->dcode.p, (const PcgScalars *)sc, pvec, da.dd, (const unsigned long long *)da.epoch, da.push_ticket);
        else if (coded)
            hipLaunchKernelGGL((k_sr_update_push<true, true>), dim3(g), dim3(kBlock), 0, stream, n, x, r, p, sv, u, wv, (const double *)w->invtab.p,
                               (const uint16_t *)w->dcode.p, (const PcgScalars *)sc, pvec, da.dd, (const unsigned long long *)da.epoch, da.push_ticket);
        else
            hipLaunchKernelGGL((k_sr_update_push<false, true>), dim3(g), dim3(kBlock), 0, stream, n, x, r, p, sv, u, wv, (const double *)invd,
                               (const uint16_t *)nullptr, (const PcgScalars *)sc, pvec, da.dd, (const unsigned long long *)da.epoch, da.push_ticket);
        return round(u, 2, (int)OP_SR_STEP, timed ? w->evA[c] : nullptr, timed ? w->evB[c] : nullptr, false);
    };
    bool use_graph = cur_opt().graph != 0;
    int enqueued = 0, last_chunk = 0, spmv_samples = 0;
    double spmv_ms_sum = 0.;
    bool timed_chunk = true, cancel_sent = false;
    while (true) {
        AVS_HIP(hipMemcpyAsync(w->host_sc, sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, stream));
        AVS_HIP(hipStreamSynchronize(stream));
        if (w->host_sc->fault) {
            if (w->resident_used && w->resident) w->resident->ok = false; // the next distributed solve takes the launch-per-phase loop
            if (w->host_sc->fault == 4)
                set_error("direct transport (paranoid mode): a halo segment does not add up to the checksum its sender left ahead of the flag "
                          "-- stale or torn halo entries (iteration ~%d)", w->host_sc->iter);
            else
                set_error("direct transport: %s did not arrive within the time limit (rank stalled or dead?)",
                          w->host_sc->fault == 1 ? "a peer's halo entries" : (w->host_sc->fault == 2 ? "a peer's partial sums" : "a workgroup's partial sums"));
            return AVS_ERCCL;
        }
        if (info && last_chunk > 0 && timed_chunk) {
            const int ran = w->host_sc->iter + (w->host_sc->done ? 1 : 0);
            const int first = enqueued - last_chunk;
            for (int c2 = 0; c2 < last_chunk && first + c2 < ran; c2 += kSampleEvery) {
                float ems = 0.f;
                if (hipEventElapsedTime(&ems, w->evA[c2], w->evB[c2]) == hipSuccess) { spmv_ms_sum += ems; ++spmv_samples; }
            }
        }
        if (w->host_sc->done || enqueued >= max_iters || w->resident_used) break;
        if (cancel_requested() && !cancel_sent) { // avs_cancel: the request word the finalizer adds to the round's sums -- every rank stops in the same round
            static const int one = 1;
            AVS_HIP(hipMemcpyAsync(w->cancel_dev.p, &one, sizeof(int), hipMemcpyHostToDevice, stream));
            cancel_sent = true;
        }
        const int chunk = (max_iters - enqueued) < kChunk ? (max_iters - enqueued) : kChunk;
        const bool replay = use_graph && (enqueued / kChunk) % kTimedChunkEvery != 0 && chunk == kChunk && !w->graph_broken;
        timed_chunk = !replay;
        if (replay) {
            const void *key[10] = {A.row_ptr, A.col, A.codes, A.packed, A.table, x, b, (const void *)da.dd, (const void *)(intptr_t)A.n,
                                   (const void *)(intptr_t)(((int64_t)A.table_size << 8) + A.col_bits + 1000003ll * ntiles + 1000000007ll * (int64_t)A.epoch + (brick ? 7 : 0))};
            (void)da.tiles_int;
            if (w->graph && (memcmp(key, w->graph_key, sizeof(key)) != 0 || w->graph_tol != tol)) {
                (void)hipGraphExecDestroy(w->graph);
                w->graph = nullptr;
            }
            if (!w->graph) {
                hipGraph_t gr = nullptr;
                bool ok = hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed) == hipSuccess;
                if (ok) {
                    for (int c = 0; c < kChunk && ok; ++c) ok = enqueue_iteration(c, false) == AVS_OK;
                    ok = (hipStreamEndCapture(stream, &gr) == hipSuccess) && ok && gr;
                }
                if (ok) ok = hipGraphInstantiate(&w->graph, gr, nullptr, nullptr, 0) == hipSuccess;
                if (gr) (void)hipGraphDestroy(gr);
                if (!ok) {
                    (void)hipGetLastError();
                    w->graph = nullptr;
                    w->graph_broken = true;
                } else {
                    memcpy(w->graph_key, key, sizeof(key));
                    w->graph_tol = tol;
                }
            }
        }
        if (replay && w->graph) {
            AVS_HIP(hipGraphLaunch(w->graph, stream));
        } else {
            timed_chunk = true;
            for (int c = 0; c < chunk; ++c) AVS_TRY(enqueue_iteration(c, info && (c % kSampleEvery == 0)));
        }
        AVS_HIP(hipGetLastError());
        enqueued += chunk;
        last_chunk = chunk;
    }
    if (w->host_sc->done == 3) // rhs == 0: x := 0
        hipLaunchKernelGGL(k_sr_update, dim3(g), dim3(kBlock), 0, stream, n, x, r, p, sv, u, wv, invd, (const PcgScalars *)sc, sc, 0, pvec);
    AVS_HIP(hipEventRecord(w->ev1, stream));
    AVS_HIP(hipEventSynchronize(w->ev1));
    float ms = 0.f;
    AVS_HIP(hipEventElapsedTime(&ms, w->ev0, w->ev1));
    if (info) {
        const PcgScalars &h = *w->host_sc;
        info->iterations = h.iter;
        info->converged = (h.done != 0 && !h.cancelled) ? 1 : 0;
        info->rhs_norm2 = h.rhs_norm2;
        info->error = (h.done == 3 || h.rhs_norm2 == 0.) ? 0. : sqrt(h.rr / h.rhs_norm2);
        info->n = n;
        info->nnz = A.nnz;
        info->solve_ms = ms;
        info->spmv_ms = spmv_samples ? spmv_ms_sum / spmv_samples : 0.;
        info->resident = w->resident_used;
        info->cancelled = h.cancelled ? 1 : 0;
    }
    if (w->host_sc->cancelled) (void)cancel_consume();
    return AVS_OK;
}

// ---------------------------------------------------------------------------------------------
avs_status pcg_create(PcgWork **out, int64_t n, int64_t n_ext, hipStream_t)
{
    PcgWork *w = new (std::nothrow) PcgWork();
    AVS_REQUIRE(w, AVS_ENOMEM, "out of host memory");
    w->n = n;
    w->n_ext = n_ext;
    w->npartial = max_partials(n);
    avs_status s;
    if ((s = w->r.alloc((size_t)n)) || (s = w->p.alloc((size_t)n_ext)) || (s = w->t.alloc((size_t)n)) ||
        (s = w->invd.alloc((size_t)n)) || (s = w->partial.alloc(w->npartial)) || (s = w->sc.alloc(2)) ||
        (s = w->stage.alloc((size_t)kRedBlocks * 4)) || (s = w->ticket.alloc(1))) {
        delete w;
        return s;
    }
    if (hipMemset(w->ticket.p, 0, sizeof(unsigned)) != hipSuccess) {
        set_error("hipMemset failed");
        pcg_destroy(w);
        return AVS_EHIP;
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&w->host_sc), sizeof(PcgScalars)) != hipSuccess ||
        hipEventCreate(&w->ev0) != hipSuccess || hipEventCreate(&w->ev1) != hipSuccess) {
        set_error("pinned host / event allocation failed");
        pcg_destroy(w);
        return AVS_EHIP;
    }
    for (int i = 0; i < kChunk; ++i)
        if (hipEventCreate(&w->evA[i]) != hipSuccess || hipEventCreate(&w->evB[i]) != hipSuccess) {
            set_error("event allocation failed");
            pcg_destroy(w);
            return AVS_EHIP;
        }
    *out = w;
    return AVS_OK;
}

int64_t pcg_rows(const PcgWork *w) { return w ? w->n : -1; }
void pcg_fused_state(const PcgWork *w, int *used, int *faults)
{
    if (used) *used = w ? w->fused_used : 0;
    if (faults) *faults = w ? w->fused_faults : 0;
}

void pcg_destroy(PcgWork *w)
{
    if (!w) return;
    delete w->resident;
    if (w->graph) (void)hipGraphExecDestroy(w->graph);
    if (w->host_sc) (void)hipHostFree(w->host_sc);
    if (w->ev0) (void)hipEventDestroy(w->ev0);
    if (w->ev1) (void)hipEventDestroy(w->ev1);
    for (int i = 0; i < kChunk; ++i) {
        if (w->evA[i]) (void)hipEventDestroy(w->evA[i]);
        if (w->evB[i]) (void)hipEventDestroy(w->evB[i]);
    }
    delete w;
}

static avs_status reduce_stage(PcgWork *w, int nb, int nred, int op, double tol, int skip_if_done,
                               hipStream_t stream, PcgDist *dist)
{
    if (!dist) {
        reduce_launch(w, w->partial.p, nb, nred, w->sc.p, op, tol, skip_if_done, 0, stream);
    } else {
        // local sums -> RCCL all-reduce of sc->red[0..nred) -> scalar update
        reduce_launch(w, w->partial.p, nb, nred, w->sc.p, (int)OP_NONE, tol, 0, 0, stream);
        AVS_TRY(dist_allreduce(dist, reinterpret_cast<double *>(reinterpret_cast<char *>(w->sc.p) + offsetof(PcgScalars, red)), nred, stream));
        hipLaunchKernelGGL(k_scalar, dim3(1), dim3(64), 0, stream, w->sc.p, op, tol);
    }
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

// Single GPU, no partition: the single-reduction iteration on the chip when the system qualifies (*ran = false otherwise, nothing
// touched).  Set-up (r = b - A x, u = M^-1 r, w = A u, the three sums) with the launch-per-phase kernels, the loop resident.
static avs_status pcg_solve_resident_single(PcgWork *w, const CsrView &A, const double *b, double *x, double tol, int max_iters,
                                            hipStream_t stream, avs_solve_info *info, bool *ran)
{
    *ran = false;
    const int64_t n = A.n;
    const bool coded = A.codes && !A.tab_ptr && A.table_size <= kViLdsTable;
    if (!coded) return AVS_OK;
    if (!w->resident) w->resident = new (std::nothrow) ResidentPlan();
    if (cancel_requested()) return AVS_OK; // (the launch-per-phase loop consumes the request at its first poll: 0 iterations, cancelled = 1)
    if (!w->resident || !resident_prepare(w->resident, A, A.n, nullptr, stream)) return AVS_OK;
    const int g = (int)((n + kBlock - 1) / kBlock < kVecGrid ? ((n + kBlock - 1) / kBlock > 0 ? (n + kBlock - 1) / kBlock : 1) : kVecGrid);
    const int rowgrid = (int)((n + kBlock - 1) / kBlock) > 0 ? (int)((n + kBlock - 1) / kBlock) : 1;
    const int variant = spmv_default_variant(A);
    AVS_TRY(w->s.alloc((size_t)n));
    AVS_TRY(w->u.alloc((size_t)w->n_ext));
    double *p = w->p.p, *r = w->r.p, *wv = w->t.p, *sv = w->s.p, *u = w->u.p, *invd = w->invd.p;
    double *pvec = w->partial.p, *pspmv = w->partial.p + 4 * (size_t)kVecGrid;
    PcgScalars *sc = w->sc.p;
    AVS_HIP(hipMemsetAsync(sc, 0, 2 * sizeof(PcgScalars), stream));
    AVS_HIP(hipMemsetAsync(p, 0, (size_t)w->n_ext * sizeof(double), stream));
    AVS_HIP(hipMemsetAsync(sv, 0, (size_t)n * sizeof(double), stream));
    hipLaunchKernelGGL(k_inv_diag, dim3(rowgrid), dim3(kBlock), 0, stream, A, invd);
    if (!w->dcode.p) AVS_TRY(w->dcode.alloc((size_t)n + 2));
    if (!w->invtab.p) AVS_TRY(w->invtab.alloc((size_t)kViLdsTable + 1));
    const int cg = (int)(((n > A.table_size + 1 ? n : A.table_size + 1) + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_inv_diag_coded, dim3(cg), dim3(kBlock), 0, stream, A, w->dcode.p, w->invtab.p);
    AVS_HIP(hipEventRecord(w->ev0, stream));
    AVS_HIP(hipMemcpyAsync(u, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    AVS_TRY(spmv_dispatch<false>(A, u, wv, nullptr, nullptr, variant, stream, nullptr));
    hipLaunchKernelGGL(k_sr_init, dim3(g), dim3(kBlock), 0, stream, n, b, wv, invd, r, u, pvec);
    int nb = 0;
    AVS_TRY(spmv_dispatch<true>(A, u, wv, pspmv, nullptr, variant, stream, &nb));
    reduce_launch(w, pvec, g, 3, sc, (int)OP_NONE, tol, 0, 0, stream);
    reduce_launch(w, pspmv, nb, 1, sc, (int)OP_NONE, tol, 0, 3, stream);
    hipLaunchKernelGGL(k_scalar, dim3(1), dim3(64), 0, stream, sc, (int)OP_SR_INIT, tol);
    AVS_HIP(hipGetLastError());
    // the initial guess is kept: if a bounded wait inside the cooperative launch times out (the grid was not co-resident in time: a GPU
    // shared with a viewport or OpenCL work) the solve is redone from it by the launch-per-phase loop IN THIS CALL
    AVS_TRY(w->x_save.alloc((size_t)n));
    AVS_HIP(hipMemcpyAsync(w->x_save.p, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    bool launched = false;
    AVS_TRY(resident_run(w->resident, A, x, r, p, sv, u, wv, w->dcode.p, w->invtab.p, sc, max_iters, nullptr, stream, &launched));
    if (!launched) return AVS_OK; // (x is untouched: the launch-per-phase loop starts over from it)
    AVS_HIP(hipMemcpyAsync(w->host_sc, sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, stream));
    AVS_HIP(hipStreamSynchronize(stream));
    bool faulted = w->host_sc->fault != 0;
#ifdef AVS_PROBES
    if (getenv("AVS_CG_RESIDENT_FAKE_FAULT")) faulted = true; // test hook of exactly this path (probe build only)
#endif
    if (faulted) {
        w->resident->ok = false; // not again on this context
        w->resident_faults++;
        AVS_HIP(hipMemcpyAsync(x, w->x_save.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        return AVS_OK;           // *ran stays false: pcg_solve carries on with the launch-per-phase loop
    }
    *ran = true;
    w->resident_used = 1;
    if (w->host_sc->done == 3) AVS_HIP(hipMemsetAsync(x, 0, (size_t)n * sizeof(double), stream)); // rhs == 0: x := 0
    AVS_HIP(hipEventRecord(w->ev1, stream));
    AVS_HIP(hipEventSynchronize(w->ev1));
    float ms = 0.f;
    AVS_HIP(hipEventElapsedTime(&ms, w->ev0, w->ev1));
    if (info) {
        const PcgScalars &h = *w->host_sc;
        info->iterations = h.iter;
        info->converged = (h.done != 0) ? 1 : 0;
        info->rhs_norm2 = h.rhs_norm2;
        info->error = (h.done == 3 || h.rhs_norm2 == 0.) ? 0. : sqrt(h.rr / h.rhs_norm2);
        info->n = n;
        info->nnz = A.nnz;
        info->solve_ms = ms;
        info->spmv_ms = 0.; // no separate SpMV launch to time
        info->resident = 1;
        info->cancelled = 0;
    }
    // avs_cancel during the cooperative launch: the launch cannot be interrupted, but the request ends HERE -- consumed, and reported when the
    // loop stopped at max_iterations without converging (a converged solve is a converged solve; the request is consumed either way, so that
    // it cannot hit the next solve on this context)
    if (cancel_consume() && info && w->host_sc->done == 0) {
        info->cancelled = 1;
        info->converged = 0;
    }
    return AVS_OK;
}

#include "avs_pcg_f32.inl"

// (KEEP is a template parameter of the vector kernels: see stream_load_k)
#define AVS_VEC_LAUNCH(KERNEL, C, F, ...)                                                                             \
    do {                                                                                                              \
        if (keep) hipLaunchKernelGGL((KERNEL<C, F, true>), dim3(g), dim3(kBlock), 0, stream, __VA_ARGS__);            \
        else hipLaunchKernelGGL((KERNEL<C, F, false>), dim3(g), dim3(kBlock), 0, stream, __VA_ARGS__);                \
    } while (0)
avs_status pcg_solve(PcgWork *w, const CsrView &A, const double *b, double *x, double tol, int max_iters,
                     hipStream_t stream, avs_solve_info *info, PcgDist *dist)
{
    const int64_t n = A.n;
    AVS_REQUIRE(w && w->n == n, AVS_EINVAL, "pcg workspace does not match the system size");
    if (dist) {
        DirectArgs da;
        if (dist_direct_args(dist, &da)) return pcg_solve_direct(w, A, b, x, tol, max_iters, stream, info, da);
    }
    if (dist && dist_wants_single_reduction(dist))
        return pcg_solve_single_reduction(w, A, b, x, tol, max_iters, stream, info, dist);
    if (!dist && A.f32_vectors > 0) return pcg_solve_f32(w, A, b, x, tol, max_iters, stream, info); // AVS_PRECISION_F32: float vectors and scalars
    if (!dist && resident_wanted(false)) { // systems that fit on the chip (<= ~1 M rows, packed form): one cooperative launch
        bool ran = false;
        const avs_status rs = pcg_solve_resident_single(w, A, b, x, tol, max_iters, stream, info, &ran);
        if (ran || rs != AVS_OK) return rs;
    }
    if (!dist && A.f32_vectors < 0) return pcg_solve_f32(w, A, b, x, tol, max_iters, stream, info); // (auto: the resident loop did not take it)
    if (A.brick && A.brick->ntiles > 0) { // one partial per wave of every tile: tiles may be smaller than 512 rows
        const size_t need = 2 * ((size_t)A.brick->ntiles * 8 + 16) + 4 * (size_t)kVecGrid + 16;
        if (need > w->npartial) {
            AVS_TRY(w->partial.alloc(need));
            w->npartial = need;
            if (w->graph) { (void)hipGraphExecDestroy(w->graph); w->graph = nullptr; }
        }
    }
    const int vgrid = (int)((n + kBlock - 1) / kBlock < kVecGrid ? (n + kBlock - 1) / kBlock : kVecGrid);
    const int g = vgrid > 0 ? vgrid : 1;
    const int rowgrid = (int)((n + kBlock - 1) / kBlock) > 0 ? (int)((n + kBlock - 1) / kBlock) : 1;
    const int variant = spmv_default_variant(A);
    double *p = w->p.p, *r = w->r.p, *t = w->t.p, *invd = w->invd.p, *partial = w->partial.p;
    PcgScalars *sc = w->sc.p;

    AVS_HIP(hipMemsetAsync(sc, 0, sizeof(PcgScalars), stream));
    hipLaunchKernelGGL(k_inv_diag, dim3(rowgrid), dim3(kBlock), 0, stream, A, invd);
    // few distinct values: the two vector kernels of the loop read a 2-B diagonal code instead of the 8-B inverse
    const bool coded = A.codes && !A.tab_ptr && A.table_size <= kViLdsTable; // (tile-local codes are not indices into one table)
    if (coded) {
        if (!w->dcode.p) AVS_TRY(w->dcode.alloc((size_t)n + 2));
        if (!w->invtab.p) AVS_TRY(w->invtab.alloc((size_t)kViLdsTable + 1));
        const int cg = (int)(((n > A.table_size + 1 ? n : A.table_size + 1) + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_inv_diag_coded, dim3(cg), dim3(kBlock), 0, stream, A, w->dcode.p, w->invtab.p);
    }
    AVS_HIP(hipEventRecord(w->ev0, stream));

    // residual = rhs - mat * x
    if (dist) {
        // x must be visible in its extended form for the product: stage it through p
        AVS_HIP(hipMemcpyAsync(p, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        AVS_TRY(dist_halo_exchange(dist, p, stream));
        AVS_TRY(spmv_dispatch<false>(A, p, t, nullptr, nullptr, variant, stream, nullptr));
    } else {
        AVS_TRY(spmv_dispatch<false>(A, x, t, nullptr, nullptr, variant, stream, nullptr));
    }
    hipLaunchKernelGGL(k_init_residual, dim3(g), dim3(kBlock), 0, stream, n, b, t, r, partial);
    AVS_TRY(reduce_stage(w, g, 2, OP_INIT, tol, 0, stream, dist));
    hipLaunchKernelGGL(k_init_p, dim3(g), dim3(kBlock), 0, stream, n, r, invd, p, x, partial, sc);
    AVS_TRY(reduce_stage(w, g, 1, OP_RHO0, tol, 0, stream, dist));
    AVS_HIP(hipGetLastError());

    int enqueued = 0, last_chunk = 0;
    double spmv_ms_sum = 0.;
    int spmv_samples = 0;
    const bool sample = (info != nullptr);
    bool finished = false, cancelled = false;
    bool timed_chunk = true;
    bool use_graph = !dist;
    use_graph = use_graph && cur_opt().graph != 0;
    bool fuse_beta = !dist; // multi-GPU (RCCL transport): the sums are all-reduced between the two vector kernels
    fuse_beta = fuse_beta && cur_opt().fuse_beta != 0;
    static_assert(kChunk % 2 == 0, "the parity of an iteration is taken from its position in the chunk");
    const int keep = A.keep_cached ? 1 : 0; // matrix + vectors fit the Infinity Cache: no non-temporal hints in the vector kernels
    // one fused launch for the two vector kernels (k_update_fused): HBM-sized single-GPU systems whose rows fit its registers, a device
    // with one CU per workgroup of its grid; the initial guess is kept so that a timed-out barrier costs a redo, not a wrong answer
    bool fuse_vec = false;
    if (!dist && fuse_beta && cur_opt().fuse_vectors != 0 && !w->fused_off && g == kVecGrid &&
        n <= (int64_t)2 * kFusedPairs * kVecGrid * kBlock && n < ((int64_t)1 << 28) && (cur_opt().fuse_vectors > 0 || !keep)) {
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        fuse_vec = cus >= kVecGrid / 8;
    }
    if (fuse_vec) {
        static std::atomic<unsigned long long> raised{0}; // the kernel's dynamic LDS limit: once per device
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && !((raised.load() >> dev) & 1ull)) {
            AVS_HIP(hipFuncSetAttribute((const void *)k_update_fused<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kFusedLds + kFusedTabBytes)));
            AVS_HIP(hipFuncSetAttribute((const void *)k_update_fused<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kFusedLds + kFusedTabBytes)));
            AVS_HIP(hipFuncSetAttribute((const void *)k_update_fused<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kFusedLds + kFusedTabBytes)));
            AVS_HIP(hipFuncSetAttribute((const void *)k_update_fused<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kFusedLds + kFusedTabBytes)));
            raised.fetch_or(1ull << dev);
        }
        if (!w->fused_bar.p) {
            AVS_TRY(w->fused_bar.alloc(2));
            AVS_HIP(hipMemsetAsync(w->fused_bar.p, 0, 2 * sizeof(unsigned long long), stream));
        }
        AVS_TRY(w->x_save.alloc((size_t)n));
        AVS_HIP(hipMemcpyAsync(w->x_save.p, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    }
    w->fused_used = fuse_vec ? 1 : 0;
    const long long fused_timeout = (long long)cur_opt().fused_timeout_ms * 100000ll; // wall_clock64: 100 MHz
    auto enqueue_iteration = [&](int c, bool timed) -> avs_status {
        int nb = 0;
        if (dist) AVS_TRY(dist_halo_exchange(dist, p, stream));
        if (timed) AVS_HIP(hipEventRecord(w->evA[c], stream));
        AVS_TRY(spmv_dispatch<true>(A, p, t, partial, sc, variant, stream, &nb)); // tmp = A p ; p.tmp
        if (timed) AVS_HIP(hipEventRecord(w->evB[c], stream));
        // single GPU: the beta step rides in k_update_xp (4 launches per iteration); r.z alternates between two slots by the
        // parity of the iteration -- every chunk starts at a multiple of kChunk (even), so c & 1 IS that parity
        const int parity = fuse_beta ? (c & 1) : 0;
        const bool fuse_alpha = fuse_beta && nb <= kFuseAlphaMax; // few SpMV partials: every workgroup of k_update_r folds them itself
        double *vpart = fuse_alpha ? partial + (w->npartial / 2) : partial; // (the SpMV's are still being read)
        if (fuse_vec && fuse_alpha) {
            const int fg = kVecGrid / 8;
#define AVS_FUSED_LAUNCH(C, K, ...) hipLaunchKernelGGL((k_update_fused<C, K>), dim3(fg), dim3(kFusedBlock), kFusedLds + (C ? kFusedTabBytes : 0), stream, __VA_ARGS__, A.table_size + 1)
            if (coded) {
                if (keep) AVS_FUSED_LAUNCH(true, true, n, x, p, r, t, w->invtab.p, w->dcode.p, sc, vpart, partial, nb, parity, w->fused_bar.p, fused_timeout);
                else AVS_FUSED_LAUNCH(true, false, n, x, p, r, t, w->invtab.p, w->dcode.p, sc, vpart, partial, nb, parity, w->fused_bar.p, fused_timeout);
            } else {
                if (keep) AVS_FUSED_LAUNCH(false, true, n, x, p, r, t, invd, nullptr, sc, vpart, partial, nb, parity, w->fused_bar.p, fused_timeout);
                else AVS_FUSED_LAUNCH(false, false, n, x, p, r, t, invd, nullptr, sc, vpart, partial, nb, parity, w->fused_bar.p, fused_timeout);
            }
#undef AVS_FUSED_LAUNCH
            return AVS_OK;
        }
        if (fuse_alpha) {
            if (coded) AVS_VEC_LAUNCH(k_update_r, true, true, n, r, t, w->invtab.p, w->dcode.p, sc, vpart, partial, nb, parity);
            else AVS_VEC_LAUNCH(k_update_r, false, true, n, r, t, invd, nullptr, sc, vpart, partial, nb, parity);
        } else {
            AVS_TRY(reduce_stage(w, nb, 1, parity ? OP_ALPHA_ODD : OP_ALPHA, tol, 1, stream, dist));
            if (coded) AVS_VEC_LAUNCH(k_update_r, true, false, n, r, t, w->invtab.p, w->dcode.p, sc, partial, nullptr, 0, 0);
            else AVS_VEC_LAUNCH(k_update_r, false, false, n, r, t, invd, nullptr, sc, partial, nullptr, 0, 0);
        }
        if (fuse_beta) {
            if (coded) AVS_VEC_LAUNCH(k_update_xp, true, true, n, x, p, r, w->invtab.p, w->dcode.p, sc, vpart, g, parity);
            else AVS_VEC_LAUNCH(k_update_xp, false, true, n, x, p, r, invd, nullptr, sc, vpart, g, parity);
            return AVS_OK;
        }
        AVS_TRY(reduce_stage(w, g, 2, OP_BETA, tol, 1, stream, dist));
        if (coded) AVS_VEC_LAUNCH(k_update_xp, true, false, n, x, p, r, w->invtab.p, w->dcode.p, sc, nullptr, 0, 0);
        else AVS_VEC_LAUNCH(k_update_xp, false, false, n, x, p, r, invd, nullptr, sc, nullptr, 0, 0);
        return AVS_OK;
    };
    while (!finished) {
        AVS_HIP(hipMemcpyAsync(w->host_sc, sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, stream));
        AVS_HIP(hipStreamSynchronize(stream));
        if (sample && last_chunk > 0 && timed_chunk) {
            // SpMV launches that really ran: iterations 0..iter (the one that detected convergence included)
            const int ran = w->host_sc->iter + ((w->host_sc->done == 1 || w->host_sc->done == 2) ? 1 : 0);
            const int first = enqueued - last_chunk;
            for (int c2 = 0; c2 < last_chunk && first + c2 < ran; c2 += kSampleEvery) {
                float ems = 0.f;
                if (hipEventElapsedTime(&ems, w->evA[c2], w->evB[c2]) == hipSuccess) {
                    spmv_ms_sum += ems;
                    ++spmv_samples;
                }
            }
        }
        if (w->host_sc->done || enqueued >= max_iters) break;
        if (!dist && cancel_consume()) { cancelled = true; break; } // avs_cancel (single GPU: the host's poll is the only party to agree with)
        const int chunk = (max_iters - enqueued) < kChunk ? (max_iters - enqueued) : kChunk;
        // some chunks are enqueued launch by launch with the SpMV timing events; the other full chunks replay one captured
        // hipGraph (4 kernel nodes per iteration): no per-launch host work, smaller gaps between the short kernels of small
        // systems.  Kernels past convergence exit at once, so replaying a whole chunk is always safe.
        // every kTimedChunkEvery-th chunk stays a plain, timed one so that the SpMV samples cover the whole solve
        const bool replay = use_graph && (enqueued / kChunk) % kTimedChunkEvery != 0 && chunk == kChunk && !w->graph_broken;
        timed_chunk = !replay;
        if (replay) {
            const void *key[10] = {A.row_ptr, A.col, A.val, A.codes, A.packed, A.table, x, (const void *)(intptr_t)A.n,
                                   (const void *)(intptr_t)(A.table_size * 64 + A.col_bits), (const void *)(intptr_t)((coded ? 1 : 0) | (fuse_beta ? 2 : 0) | (A.brick ? 4 : 0) | (int64_t)(A.epoch << 4) | (fuse_vec ? 8 : 0))};
            if (w->graph && (memcmp(key, w->graph_key, sizeof(key)) != 0 || w->graph_tol != tol)) {
                (void)hipGraphExecDestroy(w->graph);
                w->graph = nullptr;
            }
            if (!w->graph) {
                hipGraph_t gr = nullptr;
                bool ok = hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed) == hipSuccess;
                if (ok) {
                    for (int c = 0; c < kChunk && ok; ++c) ok = enqueue_iteration(c, false) == AVS_OK;
                    ok = (hipStreamEndCapture(stream, &gr) == hipSuccess) && ok && gr;
                }
                if (ok) ok = hipGraphInstantiate(&w->graph, gr, nullptr, nullptr, 0) == hipSuccess;
                if (gr) (void)hipGraphDestroy(gr);
                if (!ok) { // fall back to plain launches for the rest of this workspace's life
                    (void)hipGetLastError();
                    w->graph = nullptr;
                    w->graph_broken = true;
                } else {
                    memcpy(w->graph_key, key, sizeof(key));
                    w->graph_tol = tol;
                }
            }
        }
        if (replay && w->graph) {
            AVS_HIP(hipGraphLaunch(w->graph, stream));
        } else {
            timed_chunk = true;
            for (int c = 0; c < chunk; ++c) AVS_TRY(enqueue_iteration(c, sample && (c % kSampleEvery == 0)));
        }
        AVS_HIP(hipGetLastError());
        enqueued += chunk;
        last_chunk = chunk;
    }
#ifdef AVS_PROBES
    if (fuse_vec && getenv("AVS_PCG_FUSED_FAKE_FAULT")) w->host_sc->fault = 4; // test hook of exactly the path below (probe build only)
#endif
    if (fuse_vec && w->host_sc->fault == 4) { // the fused launch's grid barrier timed out (GPU shared with other work): the solve again, from
        w->fused_off = true;                    // the initial guess, with the two vector launches -- on this workspace from now on
        w->fused_faults++;
        if (w->graph) { (void)hipGraphExecDestroy(w->graph); w->graph = nullptr; }
        AVS_HIP(hipMemcpyAsync(x, w->x_save.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        return pcg_solve(w, A, b, x, tol, max_iters, stream, info, dist);
    }
    AVS_HIP(hipEventRecord(w->ev1, stream));
    AVS_HIP(hipEventSynchronize(w->ev1));
    float ms = 0.f;
    AVS_HIP(hipEventElapsedTime(&ms, w->ev0, w->ev1));
    if (info) {
        const PcgScalars &h = *w->host_sc;
        info->iterations = h.iter;
        info->converged = (h.done != 0) ? 1 : 0;
        info->rhs_norm2 = h.rhs_norm2;
        info->error = (h.done == 3 || h.rhs_norm2 == 0.) ? 0. : sqrt(h.rr / h.rhs_norm2);
        info->n = n;
        info->nnz = A.nnz;
        info->solve_ms = ms;
        info->spmv_ms = spmv_samples ? spmv_ms_sum / spmv_samples : 0.;
        info->resident = 0;
        info->cancelled = cancelled ? 1 : 0;
    }
    return AVS_OK;
}

#ifdef AVS_PROBES
avs_status spmv_sell_launch(int64_t nslices, const int64_t *slice_ptr, const int32_t *col, const double *val, const double *x, double *y,
                            hipStream_t stream)
{
    if (nslices <= 0) return AVS_OK;
    hipLaunchKernelGGL(k_spmv_sell, dim3((unsigned)((nslices + 3) / 4)), dim3(256), 0, stream, nslices, slice_ptr, col, val, x, y);
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

#endif

} // namespace avs

#ifdef AVS_PROBES
extern "C" avs_status avs_spmv_sell(int64_t nslices, const int64_t *slice_ptr, const int32_t *col, const double *val, const double *x,
                                    double *y, int32_t repeats, void *stream, double *ms_per_launch)
{
    AVS_REQUIRE(slice_ptr && col && val && x && y && repeats > 0, AVS_EINVAL, "bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    AVS_TRY(avs::spmv_sell_launch(nslices, slice_ptr, col, val, x, y, st)); // warm-up
    avs::Timer t(st);
    t.start();
    for (int i = 0; i < repeats; ++i) AVS_TRY(avs::spmv_sell_launch(nslices, slice_ptr, col, val, x, y, st));
    const double ms = t.stop() / repeats;
    if (ms_per_launch) *ms_per_launch = ms;
    return AVS_OK;
}
#endif // AVS_PROBES
