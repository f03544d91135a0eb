// Internal declarations shared by the HIP translation units of libavs_hip.so.
// Nothing in here crosses the C ABI (include/avs.h).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "avs.h"
#include "avs_probe.h"

struct avs_ctx;
namespace avs {

// ---------------------------------------------------------------------------------------------
// error plumbing: thread-local message + status codes, never exceptions across the ABI
// ---------------------------------------------------------------------------------------------
void set_error(const char *fmt, ...);

#define AVS_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) {                                                               \
            ::avs::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return AVS_EHIP;                                                                   \
        }                                                                                      \
    } while (0)

#define AVS_TRY(call)                       \
    do {                                    \
        avs_status s__ = (call);            \
        if (s__ != AVS_OK) return s__;      \
    } while (0)

#define AVS_REQUIRE(cond, status, ...)      \
    do {                                    \
        if (!(cond)) {                      \
            ::avs::set_error(__VA_ARGS__);  \
            return (status);                \
        }                                   \
    } while (0)

// ---------------------------------------------------------------------------------------------
// roctx ranges named after the reference's perf-monitor scopes (UT_PerfMonAutoSolveEvent, cpp:306-874), so that a rocprofv3
// --marker-trace timeline reads like Houdini's performance monitor.  The roctx library is looked up at run time (no link
// dependency; AVS_ROCTX=0 disables).
// ---------------------------------------------------------------------------------------------
void roctx_push(const char *name);
void roctx_pop();
struct Scope {
    explicit Scope(const char *name) { roctx_push(name); }
    ~Scope() { roctx_pop(); }
    Scope(const Scope &) = delete;
    Scope &operator=(const Scope &) = delete;
};
struct PhaseScope { // consecutive phases of one function: next() closes the previous range, the destructor the last
    bool open = false;
    void next(const char *name)
    {
        if (open) roctx_pop();
        roctx_push(name);
        open = true;
    }
    ~PhaseScope()
    {
        if (open) roctx_pop();
    }
};

// ---------------------------------------------------------------------------------------------
// Options: every switch of the library.  The defaults come from the environment ONCE, when a context is created (options_from_env is
// the only function that calls getenv); the switches a host may legitimately set are then reachable through avs_set_solver_option.
// Code deep inside a solve reads the options of the context whose entry point is running on this thread (cur_opt()).
// ---------------------------------------------------------------------------------------------
// what a tile costs the persistent kernel, in microseconds, as far as the descriptors show it: fitted to the per-tile phase stamps of the
// probe build (tools/probes/brick_cost_fit.py; profiles/r05_notes.md)
struct BrickCost {
    double tile = 2.69, row = 0.0015, run = 0.0001, word = 0.0019, etile = 2.96, quad = 0.0027; // (round 6, four-lane fill runs: fit of 67 k tiles -- 512^3 beam, 512^3 tank, 1024^3 sheet; residual 13 % of a tile)
};
struct Options {
    // user-facing (avs_set_solver_option)
    int resident = 1;            // AVS_CG_RESIDENT: the CU-resident PCG loop where a system qualifies
    int transport = 0;           // AVS_DIST_TRANSPORT: 0 auto (direct after its self-test, else RCCL), 1 rccl, 2 direct
    int paranoid = 0;            // AVS_DIST_PARANOID: per-round halo checksums
    int graph = 1;               // AVS_PCG_GRAPH: hipGraph replay of iteration chunks
    int brick = -1;              // AVS_BRICK: brick-structured SpMV form: -1 auto (systems >= kBrickMinSystemRows; decided by the fill rule), 0 never,
                                 // 1 always, 2 tune (auto, but decided by timing both forms at the assembly: not reproducible from run to run)
    // storage forms of the solve matrix
    int brick_interleave = 1, brick_shift = 3, value_index = 1, value_pack = 1, tile_tables = 1, column_windows = 1;
    double brick_min_regular = 0.6;
    int brick_timing = 0;
    int brick_value_codes = 1;   // AVS_BRICK_VALUE_CODES: the value-code variant of the brick form for matrices without one small dictionary (variable viscosity)
    int brick_plan = 1;          // AVS_BRICK_PLAN: cost-balanced planned walk of the brick kernel (0: the static strided walks)
    BrickCost brick_cost;        // AVS_BRICK_COST=tile,row,run,word,etile,quad
    // single-GPU loop
    int fuse_beta = 1;
    int fuse_vectors = 0;        // AVS_PCG_FUSE_VECTORS: the two vector kernels of the single-GPU loop as one launch with a grid barrier (k_update_fused,
                                 // avs_pcg.hip): 0 never, 1 wherever the system qualifies, -1 only systems larger than the Infinity Cache
    int fused_timeout_ms = 2000; // AVS_PCG_FUSED_TIMEOUT_MS: bound of that barrier's wait
    int post_dof_sample = -1;    // AVS_POST_DOF_SAMPLE: the transfer samples its nodes from the velocity DOFs (1) / by a sweep over the node lattices (0);
                                 // -1: from the DOFs when the level-0 node lattice has more than 32 x as many nodes as there are DOFs
    int prepass_temporal = 1;    // AVS_PREPASS_TEMPORAL: the device pre-pass skips what its allocations already hold from their last filling (0: every run fills everything)
    int f32_vectors = -1;        // AVS_F32_VECTORS: AVS_PRECISION_F32 contexts iterate on float vectors with float scalars (what Eigen's float CG does):
                                 // 1 always, 0 never (fp64 iteration on the float system), -1 auto: where the system is too large for the
                                 // CU-resident loop (there the vectors' bandwidth is what an iteration costs; a system that fits the chip is
                                 // solved faster by the resident fp64 loop: 128^3 beam 54 k against 29 k it/s)
    // CU-resident loop: tuning and test switches
    int resident_cus = 0, resident_equal_lanes = 0, resident_max_global = 3, resident_max_quads = 0, resident_no_stream = 0;
    long long resident_remap_chunk = 0;
    double resident_lane_fill = 0.90, resident_remote_cost = 3.0, resident_stream_cost = 1.5;
    int resident_coherent_fill = 1, resident_timers = 0, resident_verbose = 0;
    // multi-GPU
    int dist_standard_cg = 0, dist_overlap = 1, dist_loopback = 0, dist_host_plan = 0, dist_selftest_rounds = 64, dist_split_rows = 1, dist_plane_shift = -1;
    int trace_phases = 0;        // AVS_TRACE_PHASES: wall time of the steps of the pre-pass' numbering and of avs_dist_assemble on stderr (diagnosis: synchronises)
    long long dist_timeout_ms = 0; // 0: the defaults (20 s between ranks, 2 s inside one device)
};
Options options_from_env();
const Options &cur_opt();
struct OptScope { // RAII: the options of `c` are the current ones on this thread while an entry point of the C ABI runs
    explicit OptScope(const struct ::avs_ctx *c);
    ~OptScope();
    OptScope(const OptScope &) = delete;
    OptScope &operator=(const OptScope &) = delete;
    const Options *prev;
};

// avs_cancel: the request word of the context whose solve runs on this thread; the PCG loops poll it where they poll the device state
struct CancelScope {
    explicit CancelScope(std::atomic<int> *flag);
    ~CancelScope();
    CancelScope(const CancelScope &) = delete;
    CancelScope &operator=(const CancelScope &) = delete;
    std::atomic<int> *prev;
};
bool cancel_requested();   // a request is pending for the running solve (not consumed)
bool cancel_consume();     // ... and taken: the loop that sees true ends now

// ---------------------------------------------------------------------------------------------
// device buffer with explicit lifetime (hipMalloc/hipFree), sized in elements
// ---------------------------------------------------------------------------------------------
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    // scratch that lives across calls: grows, never shrinks (n stays the capacity)
    avs_status reserve(size_t count)
    {
        if (p && n >= count) return AVS_OK;
        return alloc(count + count / 16);
    }
    avs_status alloc(size_t count)
    {
        if (count == n && p) return AVS_OK;
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
            return AVS_ENOMEM;
        }
        n = count;
        return AVS_OK;
    }
};

// ---------------------------------------------------------------------------------------------
// Lattices that change hands without a copy (round 5).  The device pre-pass (avs_prepass.hip) OWNS the label / index / weight lattices it
// computes; avs_prepass_apply LENDS them to a solver context (a reference on the allocation) instead of copying ~12 full-size lattices
// per level set and frame (78 GB at 1024^3: as long as the pre-pass itself).  A lent lattice is never written again while somebody else
// holds it: the pre-pass keeps TWO allocations per lattice and writes the one nobody else references (frame k fills set A while the
// context still reads set B of frame k-1; apply swaps the context over and B is free for frame k+1) -- the memory of today's copy, none
// of its traffic.  Every allocation carries a unique id: what the pre-pass remembers about a lattice's contents from the last time it
// filled it (tile occupancy, far-brick constants: avs_prepass.hip "temporal reuse") is keyed on that id.
// ---------------------------------------------------------------------------------------------
uint64_t next_buffer_id();
template <typename T>
struct SharedBuf { // producer side
    std::shared_ptr<DevBuf<T>> h[2];
    uint64_t ids[2] = {0, 0};
    int cur = 0;
    T *p = nullptr;   // the allocation being filled / last filled
    size_t n = 0;
    uint64_t id = 0;  // ... and its identity (changes whenever p names other memory or fresh memory)
    // an allocation of `count` elements that nobody else references: the current one if it is exclusive, else the other slot, else new
    avs_status alloc(size_t count)
    {
        int pick = -1;
        for (int k = 0; k < 2 && pick < 0; ++k) {
            const int s = (cur + k) & 1;
            if (h[s] && h[s].use_count() == 1 && h[s]->n == count && h[s]->p) pick = s;
        }
        if (pick < 0) {
            for (int k = 0; k < 2 && pick < 0; ++k) {
                const int s = (cur + k) & 1;
                if (!h[s] || h[s].use_count() == 1) pick = s; // empty, or exclusive with another size
            }
            if (pick < 0) pick = cur ^ 1;                    // both lent out: drop our reference to the older loan
            h[pick] = std::make_shared<DevBuf<T>>();
            const avs_status st = h[pick]->alloc(count);
            if (st != AVS_OK) { h[pick].reset(); p = nullptr; n = 0; id = 0; return st; }
            ids[pick] = next_buffer_id();
        }
        cur = pick;
        p = h[cur]->p;
        n = h[cur]->n;
        id = ids[cur];
        return AVS_OK;
    }
    void release()
    {
        h[0].reset();
        h[1].reset();
        p = nullptr;
        n = 0;
        id = 0;
    }
    std::shared_ptr<DevBuf<T>> handle() const { return h[cur]; }
};
template <typename T>
struct LatBuf { // consumer side: owns its memory (alloc) or holds a reference on a lent allocation (adopt)
    DevBuf<T> own;
    std::shared_ptr<DevBuf<T>> lent;
    T *p = nullptr;
    size_t n = 0;
    avs_status alloc(size_t count)
    {
        lent.reset();
        const avs_status st = own.alloc(count);
        p = own.p;
        n = own.n;
        return st;
    }
    void adopt(std::shared_ptr<DevBuf<T>> hnd)
    {
        own.release();
        lent = std::move(hnd);
        p = lent ? lent->p : nullptr;
        n = lent ? lent->n : 0;
    }
    void release()
    {
        own.release();
        lent.reset();
        p = nullptr;
        n = 0;
    }
};

// copy helper honouring avs_memspace on the "outside" end
inline hipError_t copy_in(void *dst_dev, const void *src, size_t bytes, avs_memspace where, hipStream_t s)
{
    return hipMemcpyAsync(dst_dev, src, bytes,
                          where == AVS_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s);
}
inline hipError_t copy_out(void *dst, const void *src_dev, size_t bytes, avs_memspace where, hipStream_t s)
{
    return hipMemcpyAsync(dst, src_dev, bytes,
                          where == AVS_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s);
}

// ---------------------------------------------------------------------------------------------
// device-side view of the inputs (passed to kernels by value: < 1 KiB of kernarg)
// ---------------------------------------------------------------------------------------------
struct FieldView {
    const float *data;
    float cval;
    int is_const;
};

struct PyramidView {
    int levels;
    int n[3];
    int enhanced;
    int f32; // avs_desc::precision == AVS_PRECISION_F32: the narrowing points of SolveType = fpreal32
    double dx, dt;
    const int8_t *labels[AVS_MAX_LEVELS];
    const int32_t *vidx[AVS_MAX_LEVELS][3];
    const int32_t *eidx[AVS_MAX_LEVELS][3];
    const int32_t *cidx[AVS_MAX_LEVELS];
    FieldView centerw, edgew[3], facew[3], visc, dens, vel[3], solidvel[3];
};

struct StencilView { // SoA, entry k of stencil s at [k*count + s]
    int64_t count;
    int32_t *cnt, *idx, *bcnt;
    double *coef, *bval;
    double *weight; // edge: per stencil; centre: per cell (count/3)
};

// ---------------------------------------------------------------------------------------------
// CSR system resident in HBM
// ---------------------------------------------------------------------------------------------
// HIP-event stopwatch on one stream
struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    explicit Timer(hipStream_t st) : s(st)
    {
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
    }
    ~Timer()
    {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
    void start() { (void)hipEventRecord(a, s); }
    double stop()
    {
        (void)hipEventRecord(b, s);
        (void)hipEventSynchronize(b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, a, b);
        return ms;
    }
};

struct CsrView {
    int64_t n = 0, nnz = 0;
    const int32_t *row_ptr = nullptr;
    const int32_t *col = nullptr;
    const double *val = nullptr;
    // value-indexed form (lossless): val[k] == table[codes[k]]; used by the SpMV when codes != nullptr
    const uint16_t *codes = nullptr;
    const double *table = nullptr;
    int table_size = 0;
    // packed form: packed[k] = codes[k] << col_bits | col[k] (4 B per non-zero) when bits(n) + bits(table) <= 32
    const uint32_t *packed = nullptr;
    int col_bits = 0;
    // tile-local dictionaries: the codes of the rows of SpMV tile t (spmv_tile_rows() rows) index
    // table[tab_ptr[t] .. tab_ptr[t + 1]); table_size is then the total over all tiles
    const int32_t *tab_ptr = nullptr;
    // windowed columns: packed[k] = code << 20 | slot << 14 | offset, column = cbase[64 * tile + slot] + offset -- the columns a
    // 512-row tile reads lie in a few 16384-wide windows of the brick-major numbering (col_bits is 0 in this form)
    const int32_t *cbase = nullptr;
    // generation of the value index these pointers belong to (build_matrix_index): plans derived from the WORDS (the CU-resident loop's
    // re-encoding) are keyed on it -- a re-assembly may rewrite the same buffers
    uint64_t epoch = 0;
    int no_precond = 0; // AVS_PRECONDITIONER_NONE: the inverse diagonal the loops multiply with is 1 everywhere
    bool keep_cached = false; // the matrix words are small enough to stay in the Infinity Cache between two products: plain loads
    const struct BrickView *brick = nullptr; // host pointer: the brick-structured form of this matrix (single-GPU launch-per-phase loop)
    int f32_vectors = 0; // AVS_PRECISION_F32: the values are floats; the single-GPU solve iterates on float vectors (avs_pcg_f32.inl): 1 always,
                         // -1 where the CU-resident loop does not take the system, 0 never
};
constexpr int kCwinOffBits = 14, kCwinSlotBits = 6, kCwinSlots = 1 << kCwinSlotBits, kCwinCodeBits = 32 - kCwinOffBits - kCwinSlotBits;

// ---------------------------------------------------------------------------------------------
// brick-structured form (avs_brick.hip): tile-local lattice slots + global row patterns + streamed rows
// ---------------------------------------------------------------------------------------------
struct ValueIndex;
struct BrickForm;
constexpr int kBlkHdrWords = 48;         // header of a tile's descriptor block (avs_brick.hip)
constexpr int kBrickLoff[5] = {0, 3000, 3648, 3840, 3921}; // slot offsets of the level-0..3 lattices: (8>>l)+2 cells per axis, 3 faces per cell
constexpr int kBrickSlots = 3921, kBrickSlotsPad = 3936;
constexpr int kBrickMaxRows = 1024;   // rows per tile, two per thread (a fuller brick is cut into two tiles with one lattice origin)
#ifndef AVS_BRICK_RUNLEN
#define AVS_BRICK_RUNLEN 4
#endif
// entries per halo fill run = lanes that fill it (16, 8, 4 or 2).  Rounds 4-5 used 16: but 85 % of the natural runs are 1-3 entries long (512^3
// beam: 136 runs per tile, lengths 1: 53 %, 2: 10 %, 3: 22 %, 8: 5 %, 16: 5 %), so five batches of 32 sixteen-lane groups did the work of two
// batches of 128 four-lane groups (175 per tile).  In-loop A/B (profiles/r06_notes.md): SpMV 104.0 -> 95.2 us, headline + 5.4 %.
constexpr int kBrickRunLen = AVS_BRICK_RUNLEN;
constexpr int kBrickMaxRuns = kBrickRunLen == 16 ? 320 : 512;    // halo fill runs per tile (and the descriptor block must fit kBlockStride words)
constexpr int kBrickXSlots = 160;     // extra x slots per tile behind the lattice (slots 3936 .. 4095: off-lattice columns in the 27 neighbour bricks)
constexpr int kBrickPatWords = 2560;  // pattern words staged in LDS per tile (10 KiB)
constexpr int kBrickPatWordsVc = 1536; // ... in the value-code variant (geometry-only patterns: fewer per tile), which needs the LDS for the tile's value table
constexpr int kBrickTileStride = 448;  // doubles reserved per tile in BrickForm::ttab
constexpr int kBrickTileVals = 446;   // value-code variant: distinct values of a tile's pattern rows (one LDS entry each + the 0.0 of the padding words)
constexpr int kBrickPark = 1024;      // products of streamed rows parked per pass in a G tile (8 KiB): 52.7 KiB per workgroup, three per CU
constexpr int kBrickPatMax = 384;     // patterns per tile
constexpr int kBrickRowBase = 120;    // LDS table of row bases: 3 axes x 4 levels x 10 coordinates
constexpr int kBrickPatLen = 64;      // longest row stored as a pattern
constexpr int kBrickTableMax = 2048;  // value dictionary entries (LDS resident)
constexpr int kBrickMinRows = 64;     // bricks with fewer rows are merged into E tiles
constexpr int64_t kBrickMinSystemRows = 2000000; // smaller systems run the CU-resident loop (or are launch-bound): the form is not built
constexpr int64_t kBrickMinSystemRowsVc = 1000000; // ... matrices the resident loop cannot take (tile-local value tables): the form from here on
// rows per tile: the form beats the word stream / the contiguous-eighths walk beats the dealt chunks.  Round 6 (four-lane fill runs, priorities): the
// form wins by 23-57 % at every fill measured, 311 .. 622 rows per tile (tools/probes/fill_rule.py, profiles/r06_notes.md): 330 -> 300
constexpr double kBrickMinFill = 300., kBrickEighthsFill = 420.;
constexpr double kBrickMinRegularVc = 0.9; // value-code variant: pattern rows / rows below which AUTO keeps the word stream
constexpr int kBrickETileRows = 256;   // rows per E tile: its words fit one pass of the lattice's LDS (3936) unless the rows are long
struct BrickView {
    int ntiles = 0;
    const uint2 *tile_blk = nullptr;  // per tile: first 16-B unit of its descriptor block in `blocks`, units (<= 512)
    const uint32_t *blocks = nullptr; // descriptor blocks (layout: avs_brick.hip), 16-B aligned
    const uint2 *rdesc = nullptr;     // per pattern row, execution order: {pattern id << 20 | level << 18 | axis << 16 | (cz+1) << 8 | (cy+1) << 4 | (cx+1), position in the tile}
    const uint16_t *ownslot = nullptr; // per row: its own lattice slot in its tile (0xffff: none); the fill runs cover the halo only
    const uint32_t *pwords = nullptr; // pattern words: delta << 19 (signed 13 bits) | lattice level << 14 | value code << 3; a pattern is
                                      // padded to quads with its first entry's slot and the code of 0.0 (= table_size)
    const uint2 *sdesc = nullptr;     // streamed rows: local row | len << 16, first word relative to the tile's sword0
    int walk = 1;                     // tile -> workgroup: 0 one contiguous eighth of the tiles per XCD, 1 chunks dealt to the XCDs in turn (avs_brick.hip)
    const uint32_t *swords = nullptr; // code << col_bits | column, CSR order; col_bits == 0: 64-bit words column | code << 32
    const double *table = nullptr;
    int table_size = 0, col_bits = 0;
    int debug = 0; // measurement only: 1 no fill, 2 no pattern rows, 4 no streamed rows (wrong results), 16 phase stamps, 512 / 2048 without the
                   // load phase's / the longest rows' s_setprio
    int n_rows = 0; // partitioned systems: columns >= n_rows are halo entries ([owned | halo] local numbering)
    // planned walk (BrickForm::plan_walk): workgroup b of a grid of wgrid workgroups takes the tiles wlist[wptr[b] .. wptr[b + 1]) (tile_blk entries)
    const uint2 *wlist = nullptr;
    const int32_t *wptr = nullptr;
    int wgrid = 0;
    // value-code variant (variable viscosity, round 5): the patterns carry the geometry only; every pattern row streams 2-B codes into its
    // tile's value table (table_size is then the LDS capacity of a tile's table)
    int vc = 0;
    const uint2 *vcodes = nullptr;   // per row quad four 16-bit byte offsets into the tile's table, wave-interleaved in execution order
    const double *ttab = nullptr;    // the tiles' value tables, one after the other
    // float-vector loop (AVS_PRECISION_F32, avs_pcg_f32.inl): the pattern table with byte offsets for 4-B elements; f32 = the planned walk
    // and the partial-sum count are those of the float kernel's grid
    const uint32_t *pwords32 = nullptr;
    int f32 = 0;
};
// the form's arrays, owned by the context next to the CSR / value index of the solve matrix (avs_brick_build.hip)
struct BrickScratch { // build-time buffers, kept across frames
    void release();
    DevBuf<uint64_t> geo, row_hash;
    DevBuf<int32_t> first, bidx, scan_tmp, bstart, bbrick, run_first, run_id, run_start, tcount, tile0, rep, slot_id, pat_rep, pat_off, row_pid, slen, sstart, tile_nprow;
    DevBuf<uint32_t> ewords, rgeo;
    DevBuf<uint16_t> eslot;
    DevBuf<unsigned long long> keys, total;
    DevBuf<int> counters;
    DevBuf<char> tiles;
    DevBuf<uint8_t> force_e;
};
struct BrickForm {
    DevBuf<uint2> tile_blk, rdesc, sdesc;
    DevBuf<uint32_t> blocks, pwords, pwords32, swords;
    DevBuf<uint16_t> ownslot;
    DevBuf<uint8_t> tile_flags;       // per tile (walk order): 1 = its rows read halo columns (partitioned systems)
    DevBuf<uint2> vcodes;             // value-code variant: the pattern rows' value codes (BrickView::vcodes)
    DevBuf<double> ttab;              // ... and the tiles' value tables, kBrickTileStride doubles apart
    bool vc = false;
    int64_t code_quads = 0, table_values = 0;
    DevBuf<uint2> wlist;              // planned walk: the workgroups' tile sequences, one after the other
    DevBuf<int32_t> wptr;
    int wgrid = 0;                    // the grid the plan was laid out for (0: no plan)
    double plan_makespan = 0., plan_mean = 0.; // the model's estimate for the slowest / the average workgroup (microseconds)
    BrickScratch scratch;
    int ntiles = 0, patterns = 0, halo_tiles = 0;
    int64_t n_rows = 0;               // rows of the system the form was built for (columns >= n_rows: halo)
    int64_t regular_rows = 0, streamed_words = 0, block_words = 0, pattern_words = 0, streamed_rows = 0;
    bool ready = false;
    bool wide = false;    // streamed words are 64 bits (column | code << 32)
    void clear();   // not ready; buffers stay for the next build
    void release(); // + every buffer and the build scratch freed (a matrix the form does not serve)
    void view(BrickView &B, const ValueIndex &vi) const;
    // lays the tiles out over a persistent grid of `grid` workgroups (a multiple of 8: workgroup b runs on XCD b mod 8) from the cost
    // model; xcd_mode 0: one contiguous, cost-equal range of tiles per XCD, 1: chunks dealt to the XCDs in turn
    avs_status plan_walk(int grid, int xcd_mode, const BrickCost &cost, hipStream_t st);
    int64_t stored_bytes(int64_t n) const; // what one SpMV launch reads of the matrix
};
// what the form is built from: a CSR in the solver's numbering whose rows may read columns beyond the rows (the halo of a partitioned
// system: columns [n_rows, n_cols) in the rank's local numbering), its value index, and the face behind every column
struct BrickSource {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    const int32_t *row_ptr = nullptr, *col = nullptr;
    const ValueIndex *vi = nullptr;
    const double *val = nullptr;      // the values (CSR order): read by the value-code variant, whose tiles build their own tables
    const int32_t *vdof = nullptr;    // dof table, reference numbering (level | axis << 8, i, j, k)
    const int32_t *ref_id = nullptr;  // n_cols: reference DOF id of every column (single GPU: the brick-major permutation)
    int nx = 0, ny = 0, nz = 0, levels = 0, brick_shift = 3;
};
avs_status build_brick_form(BrickForm &bf, const BrickSource &src, const Options &opt, hipStream_t st); // avs_brick_build.hip
size_t brick_lds_bytes(const BrickView &B);                 // of the kernel the view is laid out for (B.f32)
size_t brick_lds_bytes(const BrickView &B, int elem_bytes); // vectors of doubles (8) / floats (4)
bool brick_lds_fits(const BrickView &B);          // the workgroup's LDS (lattice + value table + pattern image) within the device's limit
avs_status spmv_brick_launch(const BrickView &B, const double *x, double *y, double *partial, const int *done_flag, hipStream_t stream);
int brick_partial_count(const BrickView &B);   // partial sums the fused-dot launch writes (one per persistent workgroup), kernel of B.f32
int brick_partial_count(const BrickView &B, int elem_bytes);
avs_status spmv_brick_launch_f32(const BrickView &B, const float *x, float *y, double *partial, const int *done_flag, hipStream_t stream);

// the lossless storage forms of one matrix's values (avs_reorder.hip), owned next to the CSR arrays
constexpr int64_t kKeepCachedBytes = 200ll << 20; // what may stay in the 256 MB Infinity Cache across a PCG iteration
struct ValueIndex {
    DevBuf<uint16_t> codes;
    DevBuf<double> table;
    DevBuf<uint32_t> packed;
    DevBuf<int32_t> tab_ptr;
    DevBuf<int32_t> cbase;   // windowed columns: 64 window bases per tile
    int table_size = 0;  // 0 = plain CSR
    int col_bits = 0;    // > 0 = packed words
    bool tile_tables = false;
    bool col_windows = false;
    uint64_t epoch = 0;  // set by build_matrix_index: a new number for every matrix it indexes
    void clear() { table_size = 0; col_bits = 0; tile_tables = false; col_windows = false; }
    int bytes_per_nonzero() const { return table_size <= 0 ? 12 : ((col_bits > 0 || col_windows) ? 4 : 6); }
    void apply(CsrView &A) const
    {
        A.codes = nullptr; A.table = nullptr; A.table_size = 0; A.packed = nullptr; A.col_bits = 0; A.tab_ptr = nullptr; A.cbase = nullptr;
        A.epoch = epoch;
        if (table_size <= 0) return;
        A.codes = codes.p; A.table = table.p; A.table_size = table_size;
        if (col_bits > 0) { A.packed = packed.p; A.col_bits = col_bits; }
        if (tile_tables) A.tab_ptr = tab_ptr.p;
        if (col_windows) { A.packed = packed.p; A.cbase = cbase.p; A.col_bits = 0; }
        A.keep_cached = (int64_t)bytes_per_nonzero() * A.nnz + 40 * A.n <= kKeepCachedBytes; // words + the five vectors of an iteration
    }
};
// picks the most compact lossless form that fits: one dictionary of <= 2048 values (LDS-resident; packed 4-B words when the
// bits allow), else tile-local dictionaries (values repeat inside a tile even when the matrix has 10^4..10^5 distinct
// ones: smoothly varying viscosity), else one dictionary of <= 65536 values, else plain CSR
avs_status build_matrix_index(const int32_t *row_ptr, const int32_t *col, const double *val, int64_t n, int64_t nnz, int64_t n_cols,
                              ValueIndex &vi, hipStream_t st);

// Slab-local pre-pass (round 6, SURVEY 8(e) "assembles the rows it owns from its slab of the pyramids plus a halo"): what a rank's
// lattices hold.  Along the cut axis the lattices of level l are valid for the entries [win_lo[l], win_hi[l]) (level-l cells; win_hi ==
// the level's cell count means "to the end of every lattice", face and edge lattices' extra entry included); ids are the GLOBAL ones.
constexpr int kSlabIndexMargin = 12;  // level-l cells around the rank's slab whose index lattices are classified (then rounded out to 16-entry tiles)
constexpr int kSlabStencilMargin = 4; // level-l cells around the slab whose edge / centre stresses get a stencil (a stencil reads the lattices <= 2 cells of the coarser level further)
struct SlabWindow {
    bool on = false;
    int axis = 0, world = 1, rank = 0;
    int cuts[33] = {};  // fine cells along `axis`: rank r owns the faces at positions [cuts[r], cuts[r + 1])
    int win_lo[AVS_MAX_LEVELS] = {}, win_hi[AVS_MAX_LEVELS] = {};
};

// AVS_TRACE_PHASES=1: wall time between marks, the stream synchronised at every mark (diagnosis only)
struct PhaseTrace {
    hipStream_t s;
    const char *tag;
    bool on;
    std::chrono::steady_clock::time_point t;
    PhaseTrace(hipStream_t st, const char *tg, bool enabled) : s(st), tag(tg), on(enabled)
    {
        if (on) { (void)hipStreamSynchronize(s); t = std::chrono::steady_clock::now(); }
    }
    void mark(const char *what)
    {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[avs %s] %-28s %8.3f ms\n", tag, what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// what avs_prepass_apply hands to a context: references on the pre-pass's own allocations (no copy; see SharedBuf)
struct PrepassLoan {
    SlabWindow slab;
    // slab-local pre-pass: the ids (ascending) of the velocity / edge / centre DOFs inside the window
    std::shared_ptr<DevBuf<int32_t>> wlist[3];
    int64_t n_window[3] = {0, 0, 0};
    int levels = 0;
    std::shared_ptr<DevBuf<int8_t>> labels[AVS_MAX_LEVELS];
    std::shared_ptr<DevBuf<int32_t>> vidx[AVS_MAX_LEVELS][3], eidx[AVS_MAX_LEVELS][3], cidx[AVS_MAX_LEVELS], ridx[3];
    std::shared_ptr<DevBuf<float>> centerw, edgew[3], facew[3];
    int64_t counts[3] = {0, 0, 0};
    // dof tables (4 x int32 per dof: level | axis << 8, i, j, k) written by the numbering pass itself; null: the context builds them from the lattices
    std::shared_ptr<DevBuf<int32_t>> dof[3];
    // the 16^3-tile occupancy every regular-grid index lattice was classified with (device pointers valid during the call; null: unknown)
    const uint8_t *ridx_occ[3] = {nullptr, nullptr, nullptr};
    int ridx_occ_tiles[3][3] = {};
};
avs_status adopt_prepass_lattices(avs_ctx *c, const PrepassLoan &loan);                                     // avs_api.hip
avs_status adopt_regular_index_lattice(avs_ctx *c, int32_t axis, std::shared_ptr<DevBuf<int32_t>> handle, const uint8_t *occ16,
                                       const int occ16_tiles[3]);                                          // avs_post.hip

// avs_desc::precision == AVS_PRECISION_F32: the solution as the reference's Eigen::VectorXf holds it (avs_api.hip)
void narrow_solution_if_f32(avs_ctx *c, double *x, int64_t n);

// PCG work space + device scalars (see avs_pcg.hip)
struct PcgWork;

avs_status pcg_create(PcgWork **w, int64_t n, int64_t n_ext, hipStream_t stream);
void pcg_destroy(PcgWork *w);
int64_t pcg_rows(const PcgWork *w); // rows the workspace was sized for (-1: none)
void pcg_fused_state(const PcgWork *w, int *used, int *faults); // k_update_fused: the last solve ran it / launches whose barrier timed out

// Jacobi-PCG in Eigen's operation order; x holds the initial guess, returns the solution.
// Optional halo hook (multi-GPU) is wired by avs_dist.hip through PcgDist.
struct PcgDist; // opaque: halo exchange + all-reduce
avs_status pcg_solve(PcgWork *w, const CsrView &A, const double *b, double *x, double tol, int max_iters,
                     hipStream_t stream, avs_solve_info *info, PcgDist *dist);

// SpMV launchers (variant 0 = default)
avs_status spmv_launch(const CsrView &A, const double *x, double *y, int variant, hipStream_t stream);
int spmv_default_variant(const CsrView &A);
avs_status spmv_dot_launch(const CsrView &A, const double *x, double *y, double *partial, int variant, hipStream_t stream);
size_t spmv_partial_elems(int64_t n);
avs_status stream_probe(int mode, const double *a, double *b, int64_t n, double *sink, int grid, hipStream_t st);

// scan (exclusive, int32 -> int32, n+1 outputs: out[n] = total)
avs_status exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, int32_t *block_tmp, size_t block_tmp_elems,
                              hipStream_t stream);
size_t scan_tmp_elems(int64_t n);
bool dist_matrix_format(avs_ctx *c, avs_matrix_format *fmt); // avs_dist.hip: local rows of a partitioned / distributed system
avs_status build_stencils(avs_ctx *c);
avs_status pad_stencils(avs_ctx *c, bool edge); // unused stencil slots := (-1, 0.0), before a read-back
avs_status build_initial_guess(avs_ctx *c);
avs_status build_initial_guess_rows(avs_ctx *c, const int32_t *ids, int64_t m);
avs_status assemble_rows(avs_ctx *c, const int32_t *ids, int64_t m, DevBuf<int32_t> &row_ptr, DevBuf<int32_t> &col, DevBuf<double> &val,
                         DevBuf<double> &rhs, int64_t *nnz_out, int64_t *nraw_out);
avs_status count_raw_rows(avs_ctx *c, DevBuf<int32_t> &counts);
avs_status build_brick_permutation(avs_ctx *c, int brick_shift); // c->perm / c->inv from the dof table alone
void plane_owners_from_weights(const int64_t *weight, int nplanes, int world_size, int *plane_owner); // avs_partition.cpp

// ---------------------------------------------------------------------------------------------
// Direct transport (avs_dist.hip): every rank owns one fine-grained "comm block" that its peers map (IPC handle across
// processes, plain pointer inside one process): [flags + scalar staging | halo area].  A round of the distributed PCG =
// push boundary entries into the peers' halo areas + epoch flag, SpMV (halo-touching tiles wait for the flags), then a
// flag-based all-gather of the CG partial sums by the last workgroup to finish.  No RCCL call, no host involvement.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxRanks = 32;

struct CommHeader {                               // start of every comm block (device memory, written by peers)
    unsigned long long hflag[kMaxRanks];          // hflag[q]: epoch of the last halo rank q completed in my halo area
    unsigned long long hsum[kMaxRanks];           // paranoid mode (AVS_DIST_PARANOID=1, the transport self-test): checksum (sum of the bit
                                                  // patterns mod 2^64) of the entries rank q stored in this round, written BEFORE hflag[q]
    double red[2][kMaxRanks][4];                  // the partial sums, double-buffered by epoch parity; armed with an all-ones NaN,
                                                  // a slot is "there" as soon as it holds anything else (re-armed by its reader)
};

struct DistDev {                                  // device-resident, read-only for the kernels of one plan
    int rank, world, npeers, n_send_blocks;
    long long n_own;
    CommHeader *mine;                             // my block (local address)
    const double *my_halo;                        // my halo area (behind the header)
    int peer_rank[kMaxRanks];
    int send_off[kMaxRanks + 1];                  // segments of send_idx, peer after peer
    int recv_cnt[kMaxRanks];
    double *peer_halo_dst[kMaxRanks];             // where my entries for peer i start inside ITS halo area
    unsigned long long *peer_hflag_dst[kMaxRanks]; // &peer i's hflag[my rank]
    double *all_red_dst[kMaxRanks];               // &rank q's red[0][my rank][0] (q = 0 .. world-1, me included)
    // paranoid mode: every pushing workgroup adds the bit patterns of what it stored for peer i to psum[i] (device-local
    // accumulators) before it takes its ticket; the last one moves the totals into the peers' hsum[my rank], waits for the
    // acknowledgement, then raises the flags.  The reader (dist_finalize, once per round, before it contributes its partial sums --
    // i.e. before any peer can start the next round) re-adds its halo segments and compares: a stale or torn halo entry is a
    // fault (code 4), not a silently wrong product.
    int paranoid;
    int recv_off[kMaxRanks];                      // where peer i's entries start inside my halo area
    unsigned long long *peer_hsum_dst[kMaxRanks]; // &peer i's hsum[my rank]
    unsigned long long *psum;                     // kMaxRanks accumulators (zero between rounds)
    long long inject_stale_round;                 // test hook (AVS_DIST_INJECT_STALE=k): in round k the first entry for peer 0 is NOT stored
    const int32_t *send_idx;
    long long timeout_ticks;                      // wall_clock64() ticks a flag wait may take before it reports a fault
    // fused update + push (k_sr_update_push): workgroup b of push_grid owns rows [b push_chunk, (b+1) push_chunk); the entries
    // of peer i's send list inside that range are push_seg[i (push_grid + 1) + b .. + b + 1)
    const int32_t *push_seg;
    int push_grid, push_chunk;
};
// row ranges of the fused vector kernel: grid and rows per workgroup (a multiple of the block size) for n rows
void sr_update_geometry(long long n, int *grid, int *chunk);

struct DirectArgs {                               // what pcg_solve_direct needs from the plan
    const DistDev *dd = nullptr;                  // device pointer
    unsigned long long *epoch = nullptr;          // device: completed rounds of this comm block
    unsigned *push_ticket = nullptr, *fin_ticket = nullptr;
    int n_send = 0, npeers = 0;
    int push_grid = 0, push_chunk = 0; // > 0: the DistDev carries push segments (fused update + push available)
    const int32_t *tiles_int = nullptr, *tiles_bnd = nullptr;
    int n_tiles_int = 0, n_tiles_bnd = 0;
    const uint8_t *tile_flags = nullptr; // per tile: 1 = reads halo columns
    bool exclusive_device = false;       // no other rank of the group runs on this physical GPU (the CU-resident loop needs every CU)
};
bool dist_direct_args(PcgDist *d, DirectArgs *out); // false: the direct transport is not connected
// `rounds` pattern rounds over the connected blocks (avs_pcg.hip); *bad_entries = all-gathered count of wrong entries / checksums
avs_status direct_selftest(const DirectArgs &da, int rounds, hipStream_t stream, long long *bad_entries, int *fault);

// inputs already laid out on the PADDED octree lattice (what the device pre-pass produces): no crop / pad step (avs_api.hip, avs_post.hip)
avs_status set_scalar_field_lattice(::avs_ctx *c, avs_field_kind kind, int32_t axis, const float *data, float constant, avs_memspace where,
                                    bool padded_lattice);
avs_status set_regular_index_lattice(::avs_ctx *c, int32_t axis, const int32_t *idx, avs_memspace where, bool padded_lattice);
// field on a smaller lattice (sx, sy, sz) -> (rx, ry, rz): outside, the border value (replicate) or `fill`
avs_status pad_lattice_f32(const float *src, int sx, int sy, int sz, float *dst, int rx, int ry, int rz, bool replicate, float fill, hipStream_t st);
avs_status pad_lattice_i32(const int32_t *src, int sx, int sy, int sz, int32_t *dst, int rx, int ry, int rz, int32_t fill, hipStream_t st);
avs_status crop_lattice_f32(const float *src, int rx, int ry, int rz, float *dst, int sx, int sy, int sz, hipStream_t st);

// multi-GPU hooks called from pcg_solve (implemented in avs_dist.hip)
avs_status dist_halo_exchange(PcgDist *d, double *p_ext, hipStream_t stream);
avs_status dist_allreduce(PcgDist *d, double *dev_scalars, int count, hipStream_t stream);
void dist_release(struct ::avs_ctx *c);
bool dist_wants_single_reduction(PcgDist *d);
// overlap support: tile lists (interior / halo-touching spmv_tile_rows()-row tiles) and a split exchange
bool dist_tile_lists(PcgDist *d, const int32_t **t_int, int *n_int, const int32_t **t_bnd, int *n_bnd);
avs_status dist_halo_begin(PcgDist *d, double *p_ext, hipStream_t main_stream);
avs_status dist_halo_end(PcgDist *d, hipStream_t main_stream);
int spmv_tile_rows();
avs_status build_reordered_system(struct ::avs_ctx *c, int brick_shift);
avs_status build_brick_form(struct ::avs_ctx *c); // avs_brick_build.hip
#ifdef AVS_PROBES
// y = A x through the form the loops launch (+ the folded partial sums of x.y of the fused-dot instantiation); avs_api.hip
avs_status probe_spmv_form(const CsrView &A, const double *x, double *y, bool fused, double *dot_out, hipStream_t st);
avs_status spmv_f32_probe(const CsrView &A, const double *x, double *y, bool fused, double *dot_out, hipStream_t st); // avs_pcg_f32.inl
#endif
avs_status unpermute(struct ::avs_ctx *c, const double *xp, double *x);
// builds the value dictionary of `val` (nnz entries); *table_size = 0 when there are more than 65536 distinct values
avs_status build_value_index(const double *val, int64_t nnz, DevBuf<uint16_t> &codes, DevBuf<double> &table, int *table_size, hipStream_t st);
// packs (code, column) into one 32-bit word per non-zero when both fit; *col_bits = 0 when they do not (or AVS_VALUE_PACK=0)
avs_status build_packed_index(const uint16_t *codes, const int32_t *col, int64_t nnz, int64_t n_cols, int table_size,
                              DevBuf<uint32_t> &packed, int *col_bits, hipStream_t st);

} // namespace avs

// ---------------------------------------------------------------------------------------------
// the context behind the opaque avs_ctx
// ---------------------------------------------------------------------------------------------
struct avs_ctx {
    avs_desc desc{};
    hipStream_t stream = nullptr;
    bool own_stream = false;

    // inputs
    avs::LatBuf<int8_t> labels[AVS_MAX_LEVELS];   // (LatBuf: set by copy through the avs_set_* entries, or lent by avs_prepass_apply)
    avs::LatBuf<int32_t> vidx[AVS_MAX_LEVELS][3], eidx[AVS_MAX_LEVELS][3], cidx[AVS_MAX_LEVELS];
    bool have_labels[AVS_MAX_LEVELS] = {};
    bool have_vidx[AVS_MAX_LEVELS][3] = {}, have_eidx[AVS_MAX_LEVELS][3] = {}, have_cidx[AVS_MAX_LEVELS] = {};
    struct Field {
        avs::LatBuf<float> buf;
        float cval = 0.f;
        bool is_const = true;
    };
    Field centerw, edgew[3], facew[3], visc, dens, vel[3], solidvel[3];
    avs::Options opt;   // switches: environment defaults taken at avs_create, avs_set_solver_option afterwards
    std::atomic<int> cancel{0}; // avs_cancel: set by any thread, consumed by the solve loop that sees it
    int64_t n_vel = -1, n_edge = -1, n_center = -1;
    int no_precond = 0; // avs_set_solver_option(AVS_OPTION_PRECONDITIONER, AVS_PRECONDITIONER_NONE)

    // dof tables: 4 x int32 per dof (level | axis << 8, i, j, k)
    avs::LatBuf<int32_t> vdof, edof, cdof; // (built from the index lattices, or lent by the pre-pass, which writes them while it numbers the DOFs)
    bool tables_ready = false;
    // slab-local pre-pass (avs_prepass_set_slab + avs_prepass_apply): the lattices and dof tables hold this rank's window only; the ids
    // of the DOFs inside it (velocity, edge, centre; ascending).  Only avs_dist_assemble works on such a context.
    avs::SlabWindow slab;
    avs::LatBuf<int32_t> wlist[3];
    int64_t n_window[3] = {0, 0, 0};

    // stencils
    avs::DevBuf<int32_t> e_cnt, e_idx, e_bcnt, c_cnt, c_idx, c_bcnt;
    avs::DevBuf<double> e_coef, e_bval, e_w, c_coef, c_bval, c_w;
    bool stencils_ready = false;

    // system
    avs::DevBuf<double> x0, rhs, val, x;
    avs::DevBuf<int32_t> row_ptr, col;
    int64_t nnz = 0, nraw = 0;
    bool guess_ready = false, system_ready = false, solved = false;
    bool guess_partial = false; // x0 holds the restriction of SOME rows only (avs_dist_assemble): never handed out

    // post-solve transfer (avs_post.hip): regular-grid classification + interpolator work fields
    avs::LatBuf<int32_t> ridx[3];
    bool have_ridx[3] = {};
    avs::DevBuf<uint8_t> ridx_tiles[3]; // per 64 x 8 x 8 tile of the face lattice: any face the transfer writes (avs_post.hip)
    avs::DevBuf<float> post_vel[AVS_MAX_LEVELS][3], post_nval[AVS_MAX_LEVELS][3], post_nw[AVS_MAX_LEVELS][3];
    avs::DevBuf<int32_t> post_nf[AVS_MAX_LEVELS];
    avs::DevBuf<int8_t> post_nlab[AVS_MAX_LEVELS];
    avs::DevBuf<float> post_out[3]; // staging of the regular-grid output (host destination or padded octree grid), kept across calls
    bool post_ready = false;
    // temporal reuse (round 5): what the transfer's staging grids hold since the last transfer -- post_vel[l][a] all zero again (the scattered
    // entries are zeroed at the end of a transfer), node labels / values non-zero only where post_nlab is: the next transfer clears those
    // nodes instead of zero-filling ~27 B per node of every level.  The pointers name the allocations the claim was made for.
    const void *post_vel_zero[AVS_MAX_LEVELS][3] = {};
    const void *post_nodes_sparse[AVS_MAX_LEVELS] = {};
    // the nodes the last transfer labelled, per level (k_nodes_sample_dofs): walked by the node passes and by the next transfer's clear
    avs::DevBuf<uint32_t> post_list[AVS_MAX_LEVELS];
    avs::DevBuf<unsigned> post_list_count;
    unsigned post_list_n[AVS_MAX_LEVELS] = {};
    bool post_lists_valid = false;

    // brick-major copy of the system used by the solve (avs_reorder.hip); perm: new -> old
    avs::DevBuf<int32_t> perm, inv, p_row_ptr, p_col;
    avs::DevBuf<double> p_val, p_rhs, p_x0, p_x;
    // value dictionary of the solve matrix (avs_reorder.hip): at most 65536 distinct doubles
    avs::ValueIndex vi;
    avs::BrickForm brick;   // brick-structured form of the solve matrix (large single-dictionary systems)
    avs::BrickView brick_view;
    // AVS_BRICK_AUTO: brick form or word stream, whichever multiplied faster when a matrix of (about) this size was first seen
    int brick_verdict = 0, brick_walk = 1;
    int64_t brick_verdict_rows = 0;
    int brick_verdict_reuse = 0;    // frames a cached "no" of the fill rule has been reused (re-examined every 16th)
    double brick_tune_ms[2] = {0., 0.};   // stream, brick (ms per launch at the measurement)
    avs::DevBuf<double> brick_tune_y;
    bool reordered = false;
    int brick_shift = 3; // 8^3 fine cells per brick; < 0 disables the renumbering
    avs_assembly_info ainfo{};

    // scratch of the assembly kept between frames (raw triplets: 1.5 GB at 512^3 -- allocating and freeing it every call cost
    // several ms of host time; with 288 GB of HBM it simply stays)
    struct AsmScratch {
        avs::DevBuf<int32_t> row_count, rawptr, scan_tmp, raw_col, len_new, ids_in, wave_slots, ucount, coarse_list, long_rows;
        avs::DevBuf<double> raw_val;
        avs::DevBuf<uint32_t> keys_in, keys_out;
        avs::DevBuf<char> sort_tmp;
        avs::DevBuf<int> err;
    } scratch;

    avs::PcgWork *pcg = nullptr;
    avs::PcgDist *dist = nullptr;

    avs::PyramidView view() const;
};
