// avs_pcg_resident.inl -- CU-resident single-reduction PCG for systems that fit ON the chip (included by avs_pcg.hip).
//
// At 0.93 M rows per rank (the 8-way partition of the 512^3 headline system) an iteration of the launch-per-phase loop is two
// latency-bound kernels (k_sr_update_push 19 us + SpMV with its finalizer 42 us, profiles/r03_notes.md): neither the HBM nor the
// CUs are busy, the time goes into ramps, tails and dependent round trips.  MI355X has 512 KiB of VGPRs and 160 KiB of LDS per CU
// -- 168 MiB on the chip.  A rank's 14 M packed matrix words are 56 MB: they FIT IN THE REGISTER FILES.  So:
//
//   * ONE persistent workgroup of 1024 threads per CU (cooperative launch: all of them are resident), for the whole solve;
//   * every lane keeps the packed words (value code | column, 4 B) of its <= 6 consecutive rows in 64 VGPRs -- loaded once;
//   * the workgroup's slices of u, r, p, s live in LDS (4 x 8 B x <= 4.6 k rows); x and w = A u are touched by their owner lane
//     only and stay in global memory (L2-resident, 4 n doubles of traffic per iteration instead of 12 n + the matrix);
//   * an iteration = update (LDS) -> u to global (write-through) + boundary entries into the peers' halo areas -> grid barrier
//     (+ L1/L2 invalidate) -> SpMV from registers / LDS (columns of other workgroups: plain loads of the global u; of other
//     ranks: the comm block's halo area) -> ONE reduction: slot per workgroup, the last to arrive folds them in slot order,
//     all-gathers with the other ranks through the comm blocks (same sentinel-armed slots as dist_finalize), applies the scalar
//     step and publishes (alpha, beta, done, |r|^2) in sentinel-armed broadcast slots everybody polls.
//
// Same recurrences as k_sr_update_push + OP_SR_STEP (Chronopoulos-Gear), same left-to-right row sums (one lane per row), so the
// iterates agree with the launch-per-phase loop up to the order of additions inside the three dot products.
// Every wait is bounded (wall_clock64): a missing workgroup / peer ends the kernel with sc->fault set, never a hung GPU.
// Not used when two ranks share one physical GPU (two kernels that each need every CU cannot wait for each other).

#include <time.h>

static constexpr int kResThreads = 1024;
static constexpr int kResQuads = 15;     // 128-bit register quads of matrix words per lane (60 VGPRs)
static constexpr int kResQuadWords = 5;  // 25-bit words per quad; a row takes ceil(len / 5) consecutive quads of ONE lane
static constexpr int kResRowsMax = 6;    // rows per lane
static constexpr int kResWordBits = 25;  // value code | workgroup-local column
#ifndef AVS_RES_UPD
#define AVS_RES_UPD 4
#endif
#ifndef AVS_RES_FILL
#define AVS_RES_FILL 4
#endif
static constexpr int kResFill = AVS_RES_FILL; // remote columns per thread in flight in the cache fill
#ifndef AVS_RES_STREAM_DB
#define AVS_RES_STREAM_DB 0
#endif
#ifndef AVS_RES_STREAM
#define AVS_RES_STREAM 2
#endif
static constexpr int kResStream = AVS_RES_STREAM; // streamed quads per batch (AVS_RES_STREAM_DB: the next batch in flight while this one is multiplied --
                                                  // measured on the 4-way partition: 2 / 4 per batch, with and without: 73.8-76.2 us, all within noise)
static constexpr int kResUpd = AVS_RES_UPD; // rows per thread in flight in the vector update
                                                            // (update 10.3 -> 8.5 us, SpMV 12.1 -> 13.0 us: the live registers push matrix quads to scratch), off
static constexpr int kResTimers = 8;   // phase time stamps per iteration (AVS_CG_RESIDENT_TIMERS=n)
static constexpr int kResGens = 4;     // generations of the broadcast slots (a ring: re-armed two iterations ahead)

struct ResidentArgs {
    // the local system (row pointers of the packed CSR; the words come re-encoded, see rwords)
    const int32_t *row_ptr;
    const double *table;
    int table_size;
    int n;     // own rows
    int G;     // workgroups (= CUs used)
    // lane plan
    const int32_t *lane_row0;
    const uint32_t *lane_meta;            // register rows (3 bits) | streamed rows (bits 3..9) | words of a long row left in memory (bits 10..31)
    const int32_t *wg_lane0, *wg_row0;    // G + 1 entries each
    // workgroup-local re-encoding of the words (k_resident_remap): code << lc_bits | local column; local column < rows of the
    // workgroup = one of its own rows (LDS slice of u), >= : slot of the workgroup's remote-column cache behind the slice
    const uint32_t *rwords;
    // STREAMED rows (a slab larger than the register files: the 4-way partition): after its register rows a lane owns `m` more
    // consecutive rows (lane_meta bits 3..9) whose quads stay in memory -- same 5 x 25-bit quads, a row = whole quads, bit 127 = the
    // row's last quad -- laid out per wave lane-interleaved (quad j of lane l at swords[16 B x (wave_soff[wave] + 64 j + l)]: one
    // 1-KiB run per wave load), padded to the wave's longest lane with quads of zero words
    const uint32_t *swords;
    const int32_t *wave_soff;             // G x 16 + 1, in quads
    int lc_bits;
    int max_quads;                        // quads a lane uses (kResQuads; tests lower it to send ordinary rows down the long-row path)
    const int32_t *rem_list;              // G x rem_stride: source of every remote slot (< n: global u, >= n: the halo area)
    int rem_stride;
    const int32_t *rem_count;             // G
    // vectors
    double *x, *r, *p, *s, *u, *w;
    const uint16_t *dcode;                // diagonal's value code per row; invtab[code] = 1 / table[code]
    const double *invtab;
    // synchronisation (device memory, agent scope)
    unsigned *bar_count;                  // [0]: pushing workgroups that have stored their boundary entries (the last raises the halo flags)
    unsigned long long *bar_flags;        // G: update phases workgroup g has completed (its u is in memory)
    const unsigned *dep_mask;             // G x 32: the workgroups whose u entries workgroup b reads
    int n_push_wgs;                       // workgroups with boundary entries to push
    double *slots;                        // G x 4 partial sums, sentinel-armed (the value is its own arrival flag)
    double *bcast;                        // kResGens x 4: (alpha, beta, rho, done), sentinel-armed ring
    PcgScalars *sc;
    int max_iters;
    long long timeout_ticks;
    // other ranks (direct transport); dd == nullptr: a single-GPU solve
    const DistDev *dd;
    unsigned long long *epoch;            // rounds completed on this comm block
    const uint8_t *wg_halo;               // per workgroup: reads halo columns
    const int32_t *push_seg;              // npeers x (G + 1): segments of send_idx per workgroup
    long long *timers;                    // optional: max_timed x kResTimers wall-clock stamps of workgroup 0
    int max_timed;
    int coherent_fill;                    // 1: the remote columns are read with agent / system scope loads and L2 is NOT invalidated (see phase B)
    long long *wg_times;                  // optional: G x 4 stamps of every workgroup in iteration 20 (start, update done, fill done, SpMV done)
};

// Loop-body reads of the kernel arguments go through these: one scalar load from the kernarg segment AT THE USE.  Left to itself the
// compiler loads all ~40 fields of ResidentArgs at entry, keeps the ~25 the loop touches live across it, runs out of SGPRs and spills
// them to VGPR lanes: 817 v_readlane restores in a 4,000-instruction iteration (20 % of the VALU issue slots of a VALU-bound loop).
// (the struct is the kernel's only parameter: it sits at offset 0 of the kernarg segment)
template <int OFF> __device__ __forceinline__ unsigned long long res_karg64()
{
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(OFF));
    return v;
}
template <int OFF> __device__ __forceinline__ unsigned res_karg32()
{
    unsigned v;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(OFF));
    return v;
}
#define RES_P(f) (reinterpret_cast<decltype(ResidentArgs::f)>(res_karg64<(int)offsetof(ResidentArgs, f)>()))
#define RES_I(f) ((int)res_karg32<(int)offsetof(ResidentArgs, f)>())
// The same pointer, typed as GLOBAL memory.  The value comes out of an asm statement, so the compiler only knows a generic pointer and
// emits flat_load / flat_store -- which tick the LDS counter (lgkmcnt) as well as vmcnt: every LDS wait then also waits for the
// outstanding global loads, and nothing can be kept in flight across LDS work (the streamed quads, the update's x / w / s loads).
template <class T> using res_gptr = T __attribute__((address_space(1))) *;
#define RES_G(f) (reinterpret_cast<res_gptr<std::remove_pointer_t<decltype(ResidentArgs::f)>>>(res_karg64<(int)offsetof(ResidentArgs, f)>()))

__device__ __forceinline__ bool res_spin_u64(const unsigned long long *f, unsigned long long want, long long timeout)
{
    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (wall_clock64() - t0 > timeout) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}
// 16-B write-through store at agent scope (global_store_dwordx4 ... sc1): what two agent-scope atomic double stores would do, in one
// fabric write.  The caller orders it with wait_own_stores() (s_waitcnt vmcnt(0)) like every other write-through store here.
__device__ __forceinline__ void res_store_wt16(double *p, d2_t v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// the same at system scope (sc0 sc1): boundary entries into a peer's halo area
__device__ __forceinline__ void res_store_sys16(double *p, d2_t v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// a sentinel-armed slot: spin until it holds a value (false: timed out)
__device__ __forceinline__ bool res_take_slot(const double *slot, long long timeout, double *out)
{
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(slot);
    unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == kSentinel) {
        const long long t0 = wall_clock64();
        while ((v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kSentinel) {
            if (wall_clock64() - t0 > timeout) { *out = 0.; return false; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    *out = __longlong_as_double((long long)v);
    return true;
}

// three sums at once over the 1024 threads: one LDS round (fixed order: lanes by shuffle tree, then the 16 waves ascending);
// valid in thread 0
__device__ __forceinline__ void res_block_fold3(double &v0, double &v1, double &v2, double *lds48)
{
    const double a0 = wave_sum(v0), a1 = wave_sum(v1), a2 = wave_sum(v2);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        lds48[w] = a0;
        lds48[16 + w] = a1;
        lds48[32 + w] = a2;
    }
    __syncthreads();
    double t0 = 0., t1 = 0., t2 = 0.;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kResThreads / 64; ++w) {
            t0 += lds48[w];
            t1 += lds48[16 + w];
            t2 += lds48[32 + w];
        }
    }
    v0 = t0; v1 = t1; v2 = t2;
}

// Plan kernel, once per matrix: workgroup b re-encodes the words of ITS rows with workgroup-local columns.  Columns outside its row
// range get slots of a remote cache, numbered in ASCENDING column order (a bitmap of all local columns in LDS: pass 1 sets the bits,
// pass 2 counts them per 512-column block, pass 3 rewrites the words -- the slot of a column is the number of set bits below it), so
// the per-iteration fill of the cache reads ascending addresses (runs of neighbouring entries coalesce; a hash order cost 10 us
// per iteration on the workgroups that read 8 k halo entries) and the numbering is deterministic.
static constexpr int kRemapBlock = 16; // bitmap words per prefix block
static constexpr int kRemapChunk = 1 << 20; // columns per bitmap pass (128 KiB of LDS + 8 KiB of block prefixes): larger slabs take several passes
__global__ __launch_bounds__(kResThreads) void k_resident_remap(const uint32_t *__restrict__ packed, const int32_t *__restrict__ row_ptr, int col_bits,
                                                                int lc_bits, const int32_t *__restrict__ wg_row0, int rem_cap, int n_ext,
                                                                uint32_t *__restrict__ rwords, int32_t *__restrict__ rem_list,
                                                                int32_t *__restrict__ rem_count, int *__restrict__ fail, int n_own,
                                                                unsigned *__restrict__ dep_mask, int chunk)
{
    extern __shared__ unsigned bm[]; // bitmap[nw], prefix[nb + 1] of ONE chunk of columns
    __shared__ unsigned deps[32];    // bit g: a remote column of this workgroup belongs to workgroup g (G <= 1024)
    const int b = blockIdx.x, tid = threadIdx.x;
    const int r0 = wg_row0[b], r1 = wg_row0[b + 1];
    const int k0 = row_ptr[r0], k1 = row_ptr[r1];
    const unsigned cmask = (1u << col_bits) - 1u;
    const int wrows = r1 - r0;
    if (tid < 32) deps[tid] = 0u;
    int base = 0; // remote slots handed out by the chunks below this one (ascending columns overall)
    for (int c0 = 0; c0 < n_ext; c0 += chunk) {
        const int c1 = (c0 + chunk < n_ext) ? c0 + chunk : n_ext;
        const int nw = (c1 - c0 + 31) >> 5, nb = (nw + kRemapBlock - 1) / kRemapBlock;
        unsigned *prefix = bm + nb * kRemapBlock;
        for (int i = tid; i < nb * kRemapBlock; i += kResThreads) bm[i] = 0u;
        __syncthreads();
        for (int k = k0 + tid; k < k1; k += kResThreads) {
            const int col = (int)(packed[k] & cmask);
            if ((col < r0 || col >= r1) && col >= c0 && col < c1) atomicOr(&bm[(col - c0) >> 5], 1u << ((col - c0) & 31));
        }
        __syncthreads();
        for (int i = tid; i < nb; i += kResThreads) {
            unsigned c = 0u;
#pragma unroll
            for (int w = 0; w < kRemapBlock; ++w) c += (unsigned)__popc(bm[i * kRemapBlock + w]);
            prefix[i + 1] = c;
        }
        __syncthreads();
        if (tid == 0) { // exclusive scan over a few thousand block counts
            unsigned run = 0u;
            prefix[0] = 0u;
            for (int i = 1; i <= nb; ++i) {
                run += prefix[i];
                prefix[i] = run;
            }
        }
        __syncthreads();
        const int total = (int)prefix[nb];
        auto slot_of = [&](int col) { // (col relative to c0)
            const int w = col >> 5, blk = w / kRemapBlock;
            unsigned sl = prefix[blk];
            for (int j = blk * kRemapBlock; j < w; ++j) sl += (unsigned)__popc(bm[j]);
            return (int)(sl + (unsigned)__popc(bm[w] & ((1u << (col & 31)) - 1u)));
        };
        // the list of sources, ascending: every set bit
        if (base + total <= rem_cap)
            for (int w = tid; w < nw; w += kResThreads) {
                unsigned bits = bm[w];
                if (!bits) continue;
                int sl = base + slot_of(w << 5);
                while (bits) {
                    const int bit = __ffs((int)bits) - 1;
                    const int col = c0 + (w << 5) + bit;
                    rem_list[(size_t)b * rem_cap + sl++] = col;
                    if (col < n_own) { // the workgroup that owns (writes) this entry of u: binary search in the row boundaries
                        int lo = 0, hi = (int)gridDim.x;
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (wg_row0[mid] <= col) lo = mid;
                            else hi = mid;
                        }
                        atomicOr(&deps[lo >> 5], 1u << (lo & 31));
                    }
                    bits &= bits - 1u;
                }
            }
        for (int k = k0 + tid; k < k1; k += kResThreads) {
            const uint32_t wd = packed[k];
            const int col = (int)(wd & cmask);
            if ((col < r0 || col >= r1) && col >= c0 && col < c1)
                rwords[k] = ((wd >> col_bits) << lc_bits) | (uint32_t)(wrows + base + slot_of(col - c0));
        }
        base += total;
        __syncthreads(); // the bitmap is cleared for the next chunk
    }
    if (tid == 0) {
        rem_count[b] = (int32_t)base;
        if (base > rem_cap) atomicExch(fail, 1);
    }
    if (tid < 32) dep_mask[(size_t)b * 32 + tid] = deps[tid];
    for (int k = k0 + tid; k < k1; k += kResThreads) { // the workgroup's own rows (and anything past the local columns: never read)
        const uint32_t wd = packed[k];
        const int col = (int)(wd & cmask);
        if (col >= r0 && col < r1) rwords[k] = ((wd >> col_bits) << lc_bits) | (uint32_t)(col - r0);
        else if (col >= n_ext) rwords[k] = (wd >> col_bits) << lc_bits;
    }
}

// Plan kernel for STREAMED rows: thread = lane; packs the re-encoded words of the lane's streamed rows into quads (five 25-bit words, a
// row = whole quads, bit 127 of its last one set) in the wave's lane-interleaved stream (see ResidentArgs::swords), padded to the wave's
// longest lane with quads of zero words.
__global__ __launch_bounds__(kResThreads) void k_resident_stream_layout(const int32_t *__restrict__ row_ptr, const uint32_t *__restrict__ rwords,
                                                                        const int32_t *__restrict__ lane_row0, const uint32_t *__restrict__ lane_meta,
                                                                        const int32_t *__restrict__ wg_lane0, const int32_t *__restrict__ wave_soff,
                                                                        uint32_t padword, u4_t *__restrict__ squads)
{
    const int b = blockIdx.x, tid = threadIdx.x, wv = b * (kResThreads / 64) + (tid >> 6);
    const int off0 = wave_soff[wv], wcnt = (wave_soff[wv + 1] - off0) >> 6;
    u4_t *dst = squads + off0 + (tid & 63);
    const int lane = wg_lane0[b] + tid;
    auto pack = [&](const unsigned long long *wv5, bool last) {
        const unsigned long long lo = wv5[0] | (wv5[1] << 25) | (wv5[2] << 50);
        unsigned long long hi = (wv5[2] >> 14) | (wv5[3] << 11) | (wv5[4] << 36);
        if (last) hi |= 1ull << 63;
        return u4_t{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
    };
    int j = 0;
    if (lane < wg_lane0[b + 1]) {
        const uint32_t meta = lane_meta[lane];
        const int m = (int)((meta >> 3) & 127u);
        const int r = lane_row0[lane] + (int)(meta & 7u);
        for (int rr = r; rr < r + m; ++rr) {
            const int ks = row_ptr[rr], ke = row_ptr[rr + 1];
            for (int k = ks; k < ke && j < wcnt; k += kResQuadWords, ++j) {
                unsigned long long w5[kResQuadWords];
                for (int t = 0; t < kResQuadWords; ++t) w5[t] = k + t < ke ? rwords[k + t] : padword;
                dst[(size_t)64 * (size_t)j] = pack(w5, k + kResQuadWords >= ke);
            }
        }
    }
    const unsigned long long p5[kResQuadWords] = {padword, padword, padword, padword, padword};
    for (; j < wcnt; ++j) dst[(size_t)64 * (size_t)j] = pack(p5, false);
}

// NG: how many of the row-local vectors (s, then p, then r) stay in global memory (owner-only accesses) instead of LDS -- what is
// left of the LDS then holds a larger remote-column cache (workgroups in coarse regions read 2-3x as many remote columns as rows).
// Matrix words: 25 bits (value code | local column), five per 128-bit register quad, 15 quads per lane (60 VGPRs).  A row takes
// ceil(len / 5) consecutive quads of one lane, padded with words that address a zero of the dictionary, so rows end at quad
// boundaries: the inner loop is decode, two LDS reads and one FMA per word, and one end-of-row test per quad.  (Measured on the
// 8-way partition of the 512^3 system, tools/probes/rowlen_local.py: rows of 2 / 12 / 15 / 17 / 18 / 20 / 26 words make up 97 %;
// quads + <= 6 rows per lane need 0.89-0.92 of the chip's 262,144 lanes; slots of 15 or 18 words would need 1.06.)
template <int NG, bool STREAM>
__global__ __launch_bounds__(kResThreads) void k_cg_resident(ResidentArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double rlds[];
    // u: the workgroup's slice, then its remote-column cache (one index space: a word's local column addresses both)
    // (the split is per workgroup -- its own row and remote-column counts: workgroups of coarse regions have few rows and many remote
    // columns, those of fine regions the opposite, and the sum is what has to fit)
    const int wrows_al = (a.wg_row0[blockIdx.x + 1] - a.wg_row0[blockIdx.x] + 1) & ~1, nrem_al = (a.rem_count[blockIdx.x] + 1) & ~1;
    double *u_l = rlds;
    double *r_l = u_l + wrows_al + nrem_al;
    double *p_l = r_l + (NG < 3 ? wrows_al : 0);
    double *s_l = p_l + (NG < 2 ? wrows_al : 0);
    double *tbl = s_l + (NG < 1 ? wrows_al : 0); // table_size values + one zero (what the padding words multiply with)
    double *itab = tbl + a.table_size + 1;          // table_size + 1 inverted values
    double *fold = itab + a.table_size + 1;         // 3 x 16 wave sums
    double *bc = fold + 48;                         // 4 rank sums + 4 broadcast scalars
    __shared__ int sh_fail;
    __shared__ double rank_all[kMaxRanks * 4];
    const int tid = threadIdx.x, b = blockIdx.x, G = a.G;
    const DistDev *dd = a.dd;
    const long long timeout = a.timeout_ticks;
    const int wrow0 = a.wg_row0[b], wrows = a.wg_row0[b + 1] - wrow0;
    const int lane = a.wg_lane0[b] + tid;
    const bool have = lane < a.wg_lane0[b + 1];
    int row0_c = 0, nrows = 0, tail = 0;
    if (have) {
        row0_c = a.lane_row0[lane];
        const uint32_t meta = a.lane_meta[lane];
        nrows = (int)(meta & 7u);
        tail = (int)(meta >> 10);
    }
    // ---- one-time loads: matrix words -> registers, vectors -> LDS, tables -> LDS -------------------------------------------
    u4_t m[kResQuads];
    unsigned endmask = 0u; // bit q: quad q holds the last words of a row
    int nquads = 0;        // quads of the lane that hold words
    const unsigned padword = (unsigned)a.table_size << a.lc_bits; // code = table_size (the zero), column 0
    {
        const int row0 = row0_c;
        int rcur = row0, off = 0; // row being laid out, words of it already placed
        int len = (have && nrows > 0) ? a.row_ptr[row0 + 1] - a.row_ptr[row0] : 0;
#pragma unroll
        for (int q = 0; q < kResQuads; ++q) {
            int cnt = 0, src = 0;
            if (rcur < row0 + nrows && q < a.max_quads) {
                cnt = len - off < kResQuadWords ? len - off : kResQuadWords;
                src = a.row_ptr[rcur] + off;
                nquads = q + 1;
                off += kResQuadWords;
                if (off >= len && tail == 0) { // the row is complete (a lane with a tail closes its single row after the tail)
                    endmask |= 1u << q;
                    ++rcur;
                    off = 0;
                    len = rcur < row0 + nrows ? a.row_ptr[rcur + 1] - a.row_ptr[rcur] : 0;
                }
            }
            unsigned long long wv[kResQuadWords];
#pragma unroll
            for (int t = 0; t < kResQuadWords; ++t) wv[t] = t < cnt ? a.rwords[src + t] : padword;
            const unsigned long long lo = wv[0] | (wv[1] << 25) | (wv[2] << 50);         // five 25-bit words -> 125 bits
            const unsigned long long hi = (wv[2] >> 14) | (wv[3] << 11) | (wv[4] << 36);
            m[q] = u4_t{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
        }
    }
    // quads the WAVE has to walk: the longest lane's (a scalar loop bound: shorter lanes multiply their padding words by the zero)
    int wave_nq = nquads;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(wave_nq, o, 64);
        wave_nq = other > wave_nq ? other : wave_nq;
    }
    wave_nq = __builtin_amdgcn_readfirstlane(wave_nq);
    for (int i = tid; i < wrows; i += kResThreads) {
        u_l[i] = a.u[wrow0 + i];
        if (NG < 3) r_l[i] = a.r[wrow0 + i];
        if (NG < 2) p_l[i] = a.p[wrow0 + i];
        if (NG < 1) s_l[i] = a.s[wrow0 + i];
    }
    for (int i = tid; i <= a.table_size; i += kResThreads) {
        tbl[i] = i < a.table_size ? a.table[i] : 0.;
        itab[i] = a.invtab[i];
    }
    if (tid == 0) sh_fail = 0;
    __syncthreads();
    const unsigned cmask = (1u << a.lc_bits) - 1u;
    const int cbits = a.lc_bits;
    const int nrem = a.rem_count[b];
    const int32_t *rem = a.rem_list + (size_t)b * a.rem_stride;
    const bool mydep = tid < G && tid != b && ((a.dep_mask[(size_t)b * 32 + (tid >> 5)] >> (tid & 31)) & 1u) != 0u; // producer of this workgroup
    bool pushes = false;
    if (dd)
        for (int i = 0; i < dd->npeers; ++i) pushes = pushes || a.push_seg[i * (G + 1) + b + 1] > a.push_seg[i * (G + 1) + b];
    const unsigned long long E0 = dd ? *a.epoch : 0ull; // every workgroup reads the same value: it is only written at the very end
    const double *halo = dd ? dd->my_halo : nullptr;
    double alpha = a.sc->alpha, beta = a.sc->beta;
    int done = a.sc->done, iter = a.sc->iter;
    double rho = a.sc->rho;
    const double threshold = a.sc->threshold;
    const int max_iters = a.max_iters, max_timed = a.max_timed;
    double alpha_pend = 0.;
    bool x_pending = false;
    const bool coherent = a.coherent_fill != 0;
    const int tsize = a.table_size;
    const bool timing = a.timers && b == 0 && tid == 0, stamps = a.wg_times && tid == 0;
    int it = 0;
    for (; it < max_iters && !done; ++it) {
        const bool timed = timing && it < max_timed;
        long long *ts = timed ? RES_P(timers) + (size_t)it * kResTimers : nullptr;
        if (timed) ts[0] = wall_clock64();
        if (stamps && it == 20) RES_P(wg_times)[4 * b + 0] = wall_clock64();
        const unsigned long long E = E0 + (unsigned long long)it + 1ull;
        // (opaque per iteration: otherwise the 64-bit addresses of x, w, s for all six rows -- 36 registers -- are hoisted out of the
        // loop and kept live through the SpMV walk, and the allocator parks the matrix quads in scratch instead)
        int row0 = row0_c;
        asm volatile("" : "+v"(row0));
        // ---- A: vector update of the workgroup's rows (k_sr_update_push's arithmetic), u to global, boundary entries to the peers.
        // Row i of the slice belongs to thread i mod 1024 HERE (not to the lane that sums it in the SpMV): the vectors in LDS do not
        // care, and the ones in global memory (x, w, and the tiers) are read and written as whole 512-B runs per wave instead of
        // 8 B every ~32 B (the lane-owned order cost the L1 four times the tag look-ups: the update was 11 us of a 40 us iteration) ----
        // x is touched every SECOND iteration: the update that computes p_new still holds p_old, which is all the skipped
        // x += alpha_prev p_old needs (16 of the 56-82 B per row of this phase, every other time)
        const bool skip_x = (it & 1) == 0;
        double ru = 0., rr = 0.;
        {
            const res_gptr<double> gx = RES_G(x) + wrow0, gp = NG >= 2 ? RES_G(p) + wrow0 : nullptr, gr = NG >= 3 ? RES_G(r) + wrow0 : nullptr;
            const res_gptr<double> gs = NG >= 1 ? RES_G(s) + wrow0 : nullptr;
            const res_gptr<const double> gw = RES_G(w) + wrow0;
            const res_gptr<const uint16_t> gd = RES_G(dcode) + wrow0;
            for (int i0 = tid; i0 < wrows; i0 += kResUpd * kResThreads) {
                double xv[kResUpd], wv[kResUpd], sv[kResUpd], pv[kResUpd], rv[kResUpd];
                unsigned dv[kResUpd];
#pragma unroll
                for (int j = 0; j < kResUpd; ++j) {
                    const int i = i0 + j * kResThreads;
                    if (i < wrows) {
                        if (!skip_x) xv[j] = gx[i];
                        wv[j] = gw[i];
                        dv[j] = gd[i];
                        if (NG >= 1) sv[j] = gs[i];
                        if (NG >= 2) pv[j] = gp[i];
                        if (NG >= 3) rv[j] = gr[i];
                    }
                }
#pragma unroll
                for (int j = 0; j < kResUpd; ++j) {
                    const int i = i0 + j * kResThreads;
                    if (i < wrows) {
                        const double p_old = NG >= 2 ? pv[j] : p_l[i];
                        const double pi = u_l[i] + beta * p_old;
                        const double si = wv[j] + beta * (NG >= 1 ? sv[j] : s_l[i]);
                        if (NG >= 2) gp[i] = pi;
                        else p_l[i] = pi;
                        if (NG >= 1) gs[i] = si;
                        else s_l[i] = si;
                        if (!skip_x) gx[i] = (xv[j] + alpha_pend * p_old) + alpha * pi; // = the two sequential updates, bit for bit
                        const double ri = (NG >= 3 ? rv[j] : r_l[i]) - alpha * si;
                        if (NG >= 3) gr[i] = ri;
                        else r_l[i] = ri;
                        const double ui = itab[dv[j]] * ri;
                        u_l[i] = ui;
                        ru += ri * ui;
                        rr += ri * ri;
                    }
                }
            }
        }
        x_pending = skip_x;
        if (skip_x) alpha_pend = alpha;
        __syncthreads(); // the workgroup's u is complete in LDS
        // u to global for the other workgroups: write-through (other XCDs read it), coalesced, 16 B per lane where the slice allows
        // (8-B sc1 stores cost 2.7x per byte, MI355X_MICROARCH.md)
        {
            double *const gu = RES_P(u);
            const int head = (wrow0 & 1) && wrows > 0 ? 1 : 0; // 16-B alignment of the global address
            if (tid == 0 && head) __hip_atomic_store(gu + wrow0, u_l[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int pairs = (wrows - head) >> 1;
            for (int i = tid; i < pairs; i += kResThreads) {
                d2_t v;
                v.x = u_l[head + 2 * i];
                v.y = u_l[head + 2 * i + 1];
                res_store_wt16(gu + wrow0 + head + 2 * i, v);
            }
            if (tid == 0 && ((wrows - head) & 1)) __hip_atomic_store(gu + wrow0 + wrows - 1, u_l[wrows - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (dd && dd->npeers) {
            const int32_t *const pseg = RES_P(push_seg);
            for (int i = 0; i < dd->npeers; ++i) {
                const int sa = pseg[i * (G + 1) + b], se = pseg[i * (G + 1) + b + 1];
                double *dst = dd->peer_halo_dst[i] - dd->send_off[i];
                const int32_t *const sidx = dd->send_idx;
                // consecutive entries of the peer's halo area: two per fabric write where the address allows (an 8-B write-through store
                // costs 2.7x per byte, MI355X_MICROARCH.md); the flag that orders them is raised after wait_own_stores()
                const int head = (sa < se && (reinterpret_cast<uintptr_t>(dst + sa) & 15u)) ? 1 : 0;
                if (tid == 0 && head) __hip_atomic_store(dst + sa, u_l[sidx[sa] - wrow0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const int pairs = (se - sa - head) >> 1;
                for (int q = tid; q < pairs; q += kResThreads) {
                    const int j = sa + head + 2 * q;
                    d2_t v;
                    v.x = u_l[sidx[j] - wrow0];
                    v.y = u_l[sidx[j + 1] - wrow0];
                    res_store_sys16(dst + j, v);
                }
                if (tid == 0 && ((se - sa - head) & 1)) __hip_atomic_store(dst + se - 1, u_l[sidx[se - 1] - wrow0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        wait_own_stores(); // u (agent scope) and the peers' entries (system scope) acknowledged before this wave reaches the barrier
        if (timed) ts[1] = wall_clock64();
        if (stamps && it == 20) RES_P(wg_times)[4 * b + 1] = wall_clock64();
        // ---- B: no grid barrier here: a workgroup only needs the u entries of the workgroups it reads from (10-30 of 256: the
        // neighbours in the brick order).  It publishes "my update k is in memory" (one write-through flag, after its stores were
        // acknowledged) and polls the flags of its producers, one per thread.  The reduction at the end of the iteration is the
        // grid-wide synchronisation that keeps iteration k + 1 from overwriting what iteration k still reads.  The workgroups that
        // push boundary entries to other ranks take a ticket; the last one raises this rank's halo flags.  Then drop the stale lines
        // of u from L1 / L2 ----
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(RES_P(bar_flags) + b, (unsigned long long)it + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (dd && pushes) { // (a monotone ticket, zeroed by the host before the launch: no reset to order)
                const unsigned t = __hip_atomic_fetch_add(RES_P(bar_count), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((t + 1u) % (unsigned)RES_I(n_push_wgs) == 0u)
                    for (int i = 0; i < dd->npeers; ++i)
                        if (dd->send_off[i + 1] > dd->send_off[i]) st_sys(dd->peer_hflag_dst[i], E);
            }
        }
        if (mydep && !res_spin_u64(RES_P(bar_flags) + tid, (unsigned long long)it + 1ull, timeout)) sh_fail = 1;
        if (dd && RES_P(wg_halo)[b] && tid >= 960 && tid < 960 + dd->npeers && dd->recv_cnt[tid - 960] > 0)
            if (!wait_flag(&dd->mine->hflag[dd->peer_rank[tid - 960]], E, timeout, RES_P(sc), 1)) sh_fail = 1;
        __syncthreads(); // every producer's u and the peers' halo entries are in memory
        if (!coherent) {
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // buffer_inv sc1: this CU's L1 and the XCD's L2 drop other writers' lines
            __syncthreads();
        }
        if (sh_fail) break; // (block-uniform)
        if (timed) ts[2] = wall_clock64();
        // ---- C: w = A u for the lane's rows.  First the workgroup's remote columns -> LDS, ONE round trip for all of them (plain loads
        // of the global u: this CU's L1 / the XCD's L2 were invalidated behind the barrier; other ranks' entries: the halo area of the
        // comm block, fine-grained memory first touched after the flag); then every gather is an LDS read.
        const res_gptr<const double> fu = RES_G(u);
        const int fn = RES_I(n);
        for (int k0 = tid; k0 < nrem; k0 += kResFill * kResThreads) { // kResFill loads in flight per lane: a halo-reading workgroup fills 8-10 k slots
            double v[kResFill];
#pragma unroll
            for (int j = 0; j < kResFill; ++j) {
                const int k = k0 + j * kResThreads;
                if (k < nrem) {
                    const int src = rem[k];
                    if (!coherent) v[j] = (src < fn) ? fu[src] : halo[src - fn];
                    else if (src < fn) v[j] = __hip_atomic_load(fu + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else v[j] = __hip_atomic_load(halo + (src - fn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
#pragma unroll
            for (int j = 0; j < kResFill; ++j) {
                const int k = k0 + j * kResThreads;
                if (k < nrem) u_l[wrows + k] = v[j];
            }
        }
        __syncthreads();
        if (timed) ts[3] = wall_clock64();
        if (stamps && it == 20) RES_P(wg_times)[4 * b + 2] = wall_clock64();
        double wu = 0.;
        {
            const res_gptr<double> gw = RES_G(w);
            double acc = 0.;
            int rk = 0; // row of the lane being summed
            unsigned em = endmask;
            int wnq = wave_nq;
            asm volatile("" : "+v"(em)); // (loop-invariant: keep the compiler from turning them into 15 masks held in spilled SGPRs)
            asm volatile("" : "+s"(wnq));
#pragma unroll
            for (int q = 0; q < kResQuads; ++q) {
                if (q < wnq) { // (scalar branch)
                    u4_t mm = m[q];
                    // the words are loop-invariant: without this the compiler hoists every decode (code, column, LDS addresses) out of
                    // the iteration loop and spills hundreds of registers
                    asm volatile("" : "+v"(mm));
                    // five 25-bit words at bit offsets 0, 25, 50, 75, 100 of the quad: column = low lc_bits, code = the bits above (bit-field
                    // extracts straight from the shifted dwords: no intermediate 25-bit mask)
                    const unsigned kb = (unsigned)(kResWordBits - cbits);
                    { // words 0, 1 (three groups with scheduling barriers between them: ten reads in flight at once need 20 more
                      // registers than the file has next to the 60 of the matrix, and the allocator then parks the MATRIX in scratch)
                        const unsigned t1 = __builtin_amdgcn_alignbit(mm.y, mm.x, 25);
                        const double v0 = tbl[__builtin_amdgcn_ubfe(mm.x, (unsigned)cbits, kb)], x0 = u_l[mm.x & cmask];
                        const double v1 = tbl[__builtin_amdgcn_ubfe(t1, (unsigned)cbits, kb)], x1 = u_l[t1 & cmask];
                        acc += v0 * x0; // left to right inside the row, multiply then add (no FMA): the oracle's order and rounding, the same
                                        // row sums as the launch-per-phase kernels bit for bit (padding words add +0)
                        acc += v1 * x1;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    { // words 2, 3
                        const unsigned t2 = __builtin_amdgcn_alignbit(mm.z, mm.y, 18);
                        const unsigned t3 = __builtin_amdgcn_alignbit(mm.w, mm.z, 11);
                        const double v2 = tbl[__builtin_amdgcn_ubfe(t2, (unsigned)cbits, kb)], x2 = u_l[t2 & cmask];
                        const double v3 = tbl[__builtin_amdgcn_ubfe(t3, (unsigned)cbits, kb)], x3 = u_l[t3 & cmask];
                        acc += v2 * x2;
                        acc += v3 * x3;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    { // word 4
                        const double v4 = tbl[__builtin_amdgcn_ubfe(mm.w, 4u + (unsigned)cbits, kb)], x4 = u_l[__builtin_amdgcn_ubfe(mm.w, 4u, (unsigned)cbits)];
                        acc += v4 * x4;
                    }
                    if ((em >> q) & 1u) {
                        gw[row0 + rk] = acc;
                        wu += acc * u_l[row0 + rk - wrow0];
                        acc = 0.;
                        ++rk;
                    }
                }
                __builtin_amdgcn_sched_barrier(0); // keep the scheduler from hoisting later quads' reads: their live ranges push the matrix to scratch
            }
            if (tail > 0) { // a row of more words than the registers hold (a coarse face ringed by fine ones; a handful per scene)
                const int kt = RES_P(row_ptr)[row0] + RES_I(max_quads) * kResQuadWords;
                const res_gptr<const uint32_t> twords = RES_G(rwords);
                for (int k = kt; k < kt + tail; ++k) {
                    const uint32_t wd = twords[k];
                    acc += tbl[wd >> cbits] * u_l[wd & cmask];
                }
                gw[row0] = acc;
                wu += acc * u_l[row0 - wrow0];
            }
            if (STREAM) { // the lane's streamed rows: the same quads, from memory -- one 1-KiB run per wave load
                const int32_t *const so = RES_P(wave_soff) + b * (kResThreads / 64) + (tid >> 6);
                const int off0 = __builtin_amdgcn_readfirstlane(so[0]);
                const int wcnt = (__builtin_amdgcn_readfirstlane(so[1]) - off0) >> 6; // quads of the wave's longest lane
                const res_gptr<const u4_t> sq = reinterpret_cast<res_gptr<const u4_t>>(RES_G(swords)) + off0 + (tid & 63);
                const unsigned kb = (unsigned)(kResWordBits - cbits);
                const unsigned pw = (unsigned)tsize << cbits; // a word that multiplies the dictionary's zero
                const unsigned long long plo = (unsigned long long)pw | ((unsigned long long)pw << 25) | ((unsigned long long)pw << 50);
                const unsigned long long phi = ((unsigned long long)pw >> 14) | ((unsigned long long)pw << 11) | ((unsigned long long)pw << 36);
                const u4_t padq = u4_t{(unsigned)plo, (unsigned)(plo >> 32), (unsigned)phi, (unsigned)(phi >> 32)};
                int row = row0 + nrows;
                double sacc = 0.;
#if AVS_RES_STREAM_DB
                u4_t qn[kResStream]; // the batch in flight while the previous one is multiplied
#pragma unroll
                for (int i = 0; i < kResStream; ++i) qn[i] = (i < wcnt) ? sq[(size_t)64 * (size_t)i] : padq;
#endif
                for (int j0 = 0; j0 < wcnt; j0 += kResStream) {
                    u4_t qv[kResStream];
#if AVS_RES_STREAM_DB
#pragma unroll
                    for (int i = 0; i < kResStream; ++i) qv[i] = qn[i];
#pragma unroll
                    for (int i = 0; i < kResStream; ++i) qn[i] = (j0 + kResStream + i < wcnt) ? sq[(size_t)64 * (size_t)(j0 + kResStream + i)] : padq;
#else
#pragma unroll
                    for (int i = 0; i < kResStream; ++i) qv[i] = (j0 + i < wcnt) ? sq[(size_t)64 * (size_t)(j0 + i)] : padq;
#endif
#pragma unroll
                    for (int i = 0; i < kResStream; ++i) {
                        const u4_t mm = qv[i];
                        const unsigned t1 = __builtin_amdgcn_alignbit(mm.y, mm.x, 25);
                        const unsigned t2 = __builtin_amdgcn_alignbit(mm.z, mm.y, 18);
                        const unsigned t3 = __builtin_amdgcn_alignbit(mm.w, mm.z, 11);
                        const double v0 = tbl[__builtin_amdgcn_ubfe(mm.x, (unsigned)cbits, kb)], x0 = u_l[mm.x & cmask];
                        const double v1 = tbl[__builtin_amdgcn_ubfe(t1, (unsigned)cbits, kb)], x1 = u_l[t1 & cmask];
                        const double v2 = tbl[__builtin_amdgcn_ubfe(t2, (unsigned)cbits, kb)], x2 = u_l[t2 & cmask];
                        const double v3 = tbl[__builtin_amdgcn_ubfe(t3, (unsigned)cbits, kb)], x3 = u_l[t3 & cmask];
                        const double v4 = tbl[__builtin_amdgcn_ubfe(mm.w, 4u + (unsigned)cbits, kb)], x4 = u_l[__builtin_amdgcn_ubfe(mm.w, 4u, (unsigned)cbits)];
                        sacc += v0 * x0;
                        sacc += v1 * x1;
                        sacc += v2 * x2;
                        sacc += v3 * x3;
                        sacc += v4 * x4;
                        if (mm.w >> 31) { // (bit 127) the row's last quad
                            gw[row] = sacc;
                            wu += sacc * u_l[row - wrow0];
                            sacc = 0.;
                            ++row;
                        }
                    }
                }
            }
        }
        // ---- D: one reduction of (r.u, |r|^2, w.u): every workgroup drops its three sums into sentinel-armed slots (fire and forget:
        // the value is its own arrival); workgroup 0 takes them in slot order, exchanges with the other ranks, applies the scalar step
        // and publishes it in the broadcast ring everybody polls ----
        double s0 = ru, s1 = rr, s2 = wu;
        res_block_fold3(s0, s1, s2, fold);
        if (tid == 0) {
            double *const sl = RES_P(slots) + 4 * b;
            __hip_atomic_store(sl + 0, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sl + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sl + 2, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (timed) ts[4] = wall_clock64();
        if (stamps && it == 20) { // every workgroup's own phase stamps of one iteration (imbalance diagnostics)
            RES_P(wg_times)[4 * b + 3] = wall_clock64();
        }
        const int gen = it & (kResGens - 1);
        if (b == 0) {
            double v0 = 0., v1 = 0., v2 = 0.;
            if (tid < G) {
                double *const sl = RES_P(slots) + 4 * tid;
                bool ok = res_take_slot(sl + 0, timeout, &v0);
                ok = res_take_slot(sl + 1, timeout, &v1) && ok;
                ok = res_take_slot(sl + 2, timeout, &v2) && ok;
                if (!ok) sh_fail = 1;
                // re-arm, and wait for the sentinel stores to be acknowledged: the broadcast below releases the workgroups into the next
                // iteration, whose partial sums land in these very slots -- a sentinel still in flight then would overwrite one (a
                // workgroup barrier does not wait for vmcnt)
                for (int k = 0; k < 3; ++k)
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(sl + k), kSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wait_own_stores();
            }
            res_block_fold3(v0, v1, v2, fold);
            if (tid == 0) { bc[0] = v0; bc[1] = v1; bc[2] = v2; bc[3] = 0.; }
            __syncthreads();
            if (dd && dd->world > 1) {
                // all-gather of the rank sums through the comm blocks: dist_finalize's sentinel-armed slots (value = its own arrival)
                const int epar = (int)(E & 1ull);
                if (tid < dd->world * 4) {
                    const int q = tid >> 2, k = tid & 3;
                    unsigned long long *dst = reinterpret_cast<unsigned long long *>(dd->all_red_dst[q] + (size_t)epar * kMaxRanks * 4 + k);
                    st_sys(dst, (unsigned long long)__double_as_longlong(bc[k]));
                    unsigned long long *src = reinterpret_cast<unsigned long long *>(&dd->mine->red[epar][q][k]);
                    unsigned long long v = ld_sys(src);
                    if (v == kSentinel) {
                        const long long tw = wall_clock64();
                        while ((v = ld_sys(src)) == kSentinel) {
                            if (wall_clock64() - tw > timeout) {
                                __hip_atomic_store(&RES_P(sc)->fault, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                sh_fail = 1;
                                v = 0ull;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
                    rank_all[tid] = __longlong_as_double((long long)v);
                    st_sys(src, kSentinel);
                }
                __syncthreads();
                if (tid == 0)
                    for (int k = 0; k < 3; ++k) {
                        double t = 0.;
                        for (int q = 0; q < dd->world; ++q) t += rank_all[q * 4 + k]; // rank order: identical on every rank
                        bc[k] = t;
                    }
                __syncthreads();
            }
            if (tid == 0) {
                // OP_SR_STEP on (gamma, |r|^2, delta)
                const double gamma = bc[0], rr_all = bc[1], delta = bc[2];
                int nd = 0;
                double na = alpha, nb = beta, nrho = rho;
                int niter = iter;
                if (sh_fail) nd = 1;
                else if (rr_all < threshold) nd = 1;
                else {
                    nb = gamma / rho;
                    na = gamma / (delta - nb * gamma / alpha);
                    nrho = gamma;
                    niter = iter + 1;
                }
                // the ring: re-arm the generation two iterations ahead, publish this one
                double *const ring = RES_P(bcast);
                double *ahead = ring + 4 * ((it + 2) & (kResGens - 1)), *me = ring + 4 * gen;
                for (int k = 0; k < 4; ++k)
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(ahead + k), kSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(me + 0, na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(me + 1, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(me + 2, nrho, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(me + 3, (double)(nd * 1048576 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // done | 1 (never the sentinel)
                if (timed) { // debug record (AVS_CG_RESIDENT_TIMERS): the sums this step was taken from
                    ts[6] = __double_as_longlong(rr_all);
                    ts[7] = __double_as_longlong(delta - (gamma / rho) * gamma / alpha);
                }
                // the host's copy of the state (always written by this one thread: plain stores)
                PcgScalars *const hs = RES_P(sc);
                hs->red[0] = gamma; hs->red[1] = rr_all; hs->red[2] = delta;
                hs->rr = rr_all; hs->alpha = na; hs->beta = nb; hs->rho = nrho; hs->iter = niter; hs->done = nd;
                if (sh_fail && !hs->fault) hs->fault = 3;
            }
        }
        // everybody (the publisher included) picks the step up from the ring
        if (tid < 4)
            if (!res_take_slot(RES_P(bcast) + 4 * gen + tid, timeout, &bc[4 + tid])) sh_fail = 1;
        __syncthreads();
        if (sh_fail) break;
        const int nd = (int)bc[7] >> 20;
        if (!nd) {
            alpha = bc[4];
            beta = bc[5];
            rho = bc[6];
            iter += 1;
        }
        done = nd;
        if (timed) ts[5] = wall_clock64();
        __syncthreads(); // bc is rewritten next iteration
    }
    // ---- write the vectors back (a later solve / the host reads them), close the round counter -----------------------------------
    if (x_pending) // the last update skipped x: x += alpha p with the p it left behind
        for (int i = tid; i < wrows; i += kResThreads) a.x[wrow0 + i] += alpha_pend * (NG >= 2 ? a.p[wrow0 + i] : p_l[i]);
    for (int i = tid; i < wrows; i += kResThreads) {
        if (NG < 3) a.r[wrow0 + i] = r_l[i];
        if (NG < 2) a.p[wrow0 + i] = p_l[i];
        if (NG < 1) a.s[wrow0 + i] = s_l[i];
    }
    if (sh_fail && tid == 0) {
        if (!__hip_atomic_load(&a.sc->fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) __hip_atomic_store(&a.sc->fault, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.sc->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (dd && b == 0 && tid == 0) *a.epoch = E0 + (unsigned long long)it;
}

// ---------------------------------------------------------------------------------------------
// host side: the lane plan (which rows / words every lane of every workgroup keeps), built once per matrix
// ---------------------------------------------------------------------------------------------
struct ResidentPlan {
    DevBuf<int32_t> lane_row0, wg_lane0, wg_row0, push_seg, rem_list, rem_count;
    DevBuf<uint32_t> lane_meta, rwords, swords;
    DevBuf<int32_t> wave_soff;
    bool streams = false;
    DevBuf<unsigned long long> bar_flags;
    DevBuf<unsigned> dep_mask;
    int n_push_wgs = 0;
    DevBuf<uint8_t> wg_halo;
    DevBuf<unsigned> bar_count;
    DevBuf<double> slots, bcast;
    DevBuf<long long> timers;
    int G = 0, max_timed = 0, lc_bits = 0, ng = 0, max_quads = kResQuads;
    size_t lds = 0;
    const void *key[4] = {};
    int64_t key_n = -1;
    uint64_t key_epoch = 0;
    bool ok = false, tried = false;
    std::string why;
};

static bool resident_wanted(bool distributed)
{
    (void)distributed;
    return cur_opt().resident != 0; // default for every system that qualifies (plan: 1.5-2 ms per new matrix; AVS_CG_RESIDENT=0 keeps the launch-per-phase loops)
}

static const void *resident_kernel(int ng, bool streams)
{
    // (streamed rows are a template parameter: their code costs the plain kernels 15 more spilled registers otherwise)
    switch (ng) {
    case 0: return streams ? (const void *)k_cg_resident<0, true> : (const void *)k_cg_resident<0, false>;
    case 1: return streams ? (const void *)k_cg_resident<1, true> : (const void *)k_cg_resident<1, false>;
    case 2: return streams ? (const void *)k_cg_resident<2, true> : (const void *)k_cg_resident<2, false>;
    default: return streams ? (const void *)k_cg_resident<3, true> : (const void *)k_cg_resident<3, false>;
    }
}

// Builds (or re-uses) the plan for A; returns false (with plan->why) when the system does not qualify.
static bool resident_prepare(ResidentPlan *pl, const CsrView &A, int64_t n_cols, const DirectArgs *da, hipStream_t stream)
{
    const void *key[4] = {A.row_ptr, A.packed, A.table, da ? (const void *)da->dd : nullptr};
    if (pl->tried && memcmp(key, pl->key, sizeof(key)) == 0 && pl->key_n == A.n && pl->key_epoch == A.epoch) return pl->ok;
    pl->key_epoch = A.epoch; // (a re-assembly with the same DOF count rewrites the same buffers: the words of the plan would be stale)
    pl->tried = true;
    pl->ok = false;
    memcpy(pl->key, key, sizeof(key));
    pl->key_n = A.n;
    const bool verbose = cur_opt().resident_verbose > 0;
    timespec plan_t0{};
    clock_gettime(CLOCK_MONOTONIC, &plan_t0);
    timespec stage_t = plan_t0;
    auto stage = [&](const char *what) { // (verbose: where the plan's milliseconds go)
        if (!verbose) return;
        timespec t{};
        clock_gettime(CLOCK_MONOTONIC, &t);
        fprintf(stderr, "[avs resident]   plan stage %-28s %.2f ms\n", what, (t.tv_sec - stage_t.tv_sec) * 1e3 + (t.tv_nsec - stage_t.tv_nsec) * 1e-6);
        stage_t = t;
    };
    auto no = [&](const char *why) {
        pl->why = why;
        if (verbose) fprintf(stderr, "[avs resident] not used: %s (n = %lld)\n", why, (long long)A.n);
        return false;
    };
    if (!A.packed || !A.codes || A.tab_ptr || A.cbase || A.col_bits <= 0) return no("needs the packed single-dictionary form");
    if (A.table_size > 1023 || A.n < 1 || A.n >= (1ll << 31)) return no("dictionary too large");
    int dev = 0, cus = 0, coop = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !coop || cus < 1) {
        (void)hipGetLastError();
        return no("no cooperative launch");
    }
    const int64_t n = A.n;
    int G = cus;
    if (cur_opt().resident_cus > 0 && cur_opt().resident_cus < cus) G = cur_opt().resident_cus; // tests: two ranks on ONE GPU, each on a share of the CUs
    // Cheap refusals first (before the row pointers cross PCIe and the host walks them: 20-100 ms at 4-7 M rows, per new matrix):
    // every workgroup keeps its slice of u in LDS next to its remote columns (never less than about half as much again)
    if ((size_t)(n / G) * sizeof(double) > (size_t)(160 * 1024) * 65 / 100) return no("the workgroups' slices of u leave no room for their remote columns in the LDS");
    std::vector<int32_t> rp((size_t)n + 1);
    if (hipMemcpyAsync(rp.data(), A.row_ptr, rp.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
        (void)hipGetLastError();
        return no("row pointer download failed");
    }
    // ---- lanes: consecutive rows, each in ceil(len / 5) of the lane's 15 quads, at most 6 rows; a row of more than 75 words sits
    // alone, the rest of it is read from memory ----
    int max_quads = kResQuads;
    if (cur_opt().resident_max_quads >= 1 && cur_opt().resident_max_quads <= kResQuads) max_quads = cur_opt().resident_max_quads; // tests: long-row path
    const int W = max_quads * kResQuadWords;
    std::vector<int32_t> lrow;
    std::vector<uint32_t> lmeta;
    lrow.reserve((size_t)n / 4 + 16);
    lmeta.reserve((size_t)n / 4 + 16);
    // T: streamed QUADS per lane.  0 while the slab fits the register files; otherwise every lane takes, after its register rows, rows
    // worth about T quads (error diffusion keeps the average; equal quads per lane, not equal rows: a wave walks its longest lane's
    // stream, and with rows of 3-6 quads an equal-rows split padded the streams by 64 %); at most 127 rows per lane
    const char *lane_fail = nullptr;
    auto quads_of = [&](int64_t row) { return (rp[(size_t)row + 1] - rp[(size_t)row] + kResQuadWords - 1) / kResQuadWords; };
    auto form_lanes = [&](double T) {
        lrow.clear();
        lmeta.clear();
        double debt = 0.;
        for (int64_t i = 0; i < n;) {
            const int Lr = rp[(size_t)i + 1] - rp[(size_t)i];
            if (Lr <= 0) { lane_fail = "empty row"; return; } // (every row of this system carries its diagonal, cpp:2768)
            if (Lr > W) {
                if (Lr - W >= (1 << 22)) { lane_fail = "row too long"; return; }
                lrow.push_back((int32_t)i);
                lmeta.push_back(1u | ((unsigned)(Lr - W) << 10));
                ++i;
                continue;
            }
            int rows = 0, used = 0;
            const int64_t first = i;
            while (i < n && rows < kResRowsMax) {
                const int Li = rp[(size_t)i + 1] - rp[(size_t)i];
                if (Li <= 0 || Li > W) break;
                const int k = (Li + kResQuadWords - 1) / kResQuadWords;
                if (used + k > max_quads) break;
                used += k;
                ++rows;
                ++i;
            }
            debt += T;
            int m = 0;
            while (i < n && m < 127) {
                if (rp[(size_t)i + 1] - rp[(size_t)i] <= 0) { lane_fail = "empty row"; return; }
                const int k = quads_of(i);
                if ((double)k > debt + 0.5 * (double)k) break; // (take the row when at least half of it is owed)
                debt -= (double)k;
                ++m;
                ++i;
            }
            lrow.push_back((int32_t)first);
            lmeta.push_back((unsigned)rows | ((unsigned)m << 3));
        }
    };
    stage("row pointers to the host");
    const int64_t lane_cap = (int64_t)G * kResThreads;
    int64_t q_total = 0;
    for (int64_t i = 0; i < n; ++i) q_total += quads_of(i);
    // (a slab whose quads exceed what the lanes could hold even at 14 of 15 quads each needs streamed rows for certain: the
    // registers-only pass -- 2 ns per row on the host -- is skipped)
    const bool surely_streams = (double)q_total > 14. * 0.93 * (double)lane_cap;
    if (!surely_streams) form_lanes(0.);
    if (lane_fail) return no(lane_fail);
    stage("lanes (registers only)");
    double stream_T = 0.;
    if (surely_streams || (int64_t)lrow.size() > lane_cap * 93 / 100) {
        if (cur_opt().resident_no_stream) return no("too many rows for the register files of this GPU");
        // register quads an average lane holds: measured when the registers-only pass ran, else 12.8 (4-way slab 12.9, 256^3 beam 12.7)
        const double q_lane = surely_streams ? 12.8 : (double)q_total / (double)lrow.size();
        double Lt = cur_opt().resident_lane_fill * (double)lane_cap;
        for (int attempt = 0; attempt < 8; ++attempt, Lt *= 0.97) {
            stream_T = ((double)q_total - Lt * q_lane) / Lt;
            form_lanes(stream_T);
            if (lane_fail) return no(lane_fail);
            if ((int64_t)lrow.size() <= lane_cap * 96 / 100) break;
        }
        if ((int64_t)lrow.size() > lane_cap * 97 / 100) return no("too many rows for the register files of this GPU, even with streamed rows");
    }
    const int64_t L = (int64_t)lrow.size();
    int64_t lpw = (L + G - 1) / G;
    if (lpw > kResThreads) return no("too many rows for the register files of this GPU");
    stage("lanes with streamed rows");
    // quads a lane streams per iteration (cost model, stream layout)
    std::vector<int32_t> lane_sw((size_t)L, 0);
    int64_t stream_words = 0;
    for (int64_t l = 0; l < L; ++l) {
        const int m = (int)((lmeta[(size_t)l] >> 3) & 127u);
        if (m) {
            const int64_t r = (int64_t)lrow[(size_t)l] + (int64_t)(lmeta[(size_t)l] & 7u);
            for (int64_t q = r; q < r + m; ++q) lane_sw[(size_t)l] += (rp[(size_t)q + 1] - rp[(size_t)q] + kResQuadWords - 1) / kResQuadWords; // quads
            stream_words += rp[(size_t)(r + m)] - rp[(size_t)r];
        }
    }
    std::vector<int32_t> wl((size_t)G + 1), wr((size_t)G + 1), rc((size_t)G);
    int max_rows = 0;
    int code_bits = 1;
    while ((1 << code_bits) < A.table_size + 1) ++code_bits; // + the zero the padding words address
    if (code_bits >= kResWordBits - 8) return no("dictionary needs too many bits");
    const size_t lds_max = 160 * 1024 - 4096 - 1024;
    const int cap = stream_T > 0. ? 32768 : 16384; // stride of the per-workgroup source lists (a workgroup with more remote columns does not qualify)
    const int64_t n_ext = n_cols > n ? n_cols : n;
    int64_t chunk_cols = std::min<int64_t>(n_ext, kRemapChunk);
    if (cur_opt().resident_remap_chunk >= 512) // tests: several bitmap passes on a small system
        chunk_cols = std::min<int64_t>(chunk_cols, (cur_opt().resident_remap_chunk + 511) / 512 * 512);
    const size_t remap_lds = ((size_t)(((chunk_cols + 31) / 32 + kRemapBlock - 1) / kRemapBlock) * (kRemapBlock + 1) + 2) * sizeof(unsigned);
    DevBuf<int> fail;
    if (pl->wg_row0.alloc((size_t)G + 1) != AVS_OK || pl->rwords.alloc((size_t)A.nnz) != AVS_OK || pl->rem_count.alloc((size_t)G) != AVS_OK ||
        pl->rem_list.alloc((size_t)G * cap) != AVS_OK || fail.alloc(1) != AVS_OK || pl->dep_mask.alloc((size_t)G * 32) != AVS_OK || G > 1024 ||
        hipFuncSetAttribute((const void *)k_resident_remap, hipFuncAttributeMaxDynamicSharedMemorySize, (int)remap_lds) != hipSuccess) {
        (void)hipGetLastError();
        return no("plan allocation failed");
    }
    // Tiers: the fewest row-local vectors in global memory that fit.  Tier 1 (s) is tried with re-balancing first; tiers 2 and 3 (p, r
    // too) only when that fails -- larger slabs (streamed rows: the 2- and 4-way partitions) -- since their vector traffic is back in
    // the update phase (coalesced since the second pass of round 3, which is what makes them worth it).
    int max_ng_limit = 3;
    max_ng_limit = cur_opt().resident_max_global;
    // Workgroup boundaries by estimated time, not by lanes: the SpMV phase costs per lane (every lane walks its quads), the vector
    // update per row (measured: ~12.7 ns per lane, ~3.6 ns per row of a workgroup) -- workgroups of fine regions have 2x the rows
    // of those in coarse regions at equal lanes.  Then the words are re-encoded (k_resident_remap) and the LDS footprints checked:
    // a workgroup whose slices + remote columns do not fit (the ones that read the halo: up to 10 k remote columns) gets its lanes
    // re-weighted and the split is redone -- a few rounds.
    std::vector<double> lane_w((size_t)L, 1.0), cum((size_t)L + 1, 0.), lane_extra((size_t)L, 0.);
    // remote columns cost their workgroup a fill (the halo-reading workgroups fill 8-10 k and finished 5 us after the
    // median one).  Known only after a first split: round 0 measures them, round 1 splits with them spread over the workgroup's lanes.
    // Per slot, in the units of c_lane / c_row: 0 / 1.1 / 2 / 3 / 4.5 -> 8-way loop-back (ranks 0 / 3) 33.8 / 35.9, 32.6 / 35.0, 31.6 / 33.0,
    // 30.0 / 32.1, 30.4 / 32.3 us per iteration.
    std::vector<int32_t> rc_round0;
    bool round0_done = false, reuse_round0 = false;
    const double c_rem = cur_opt().resident_remote_cost;
    const double c_lane = 12.7, c_row = 3.6;
    const double kStreamCost = cur_opt().resident_stream_cost;
    int ng = -1, lc_bits = 0, max_cols = 0;
    size_t lds = 0;
    const char *last_reason = "the vector slices + remote columns of a workgroup do not fit the LDS";
    for (int max_ng = std::min(1, max_ng_limit); max_ng <= max_ng_limit && ng < 0; ++max_ng) {
    if ((size_t)(4 - max_ng) * (size_t)(n / G) * sizeof(double) > lds_max) continue; // (even the average workgroup's slices would not fit)
    std::fill(lane_w.begin(), lane_w.end(), 1.0);
    std::fill(lane_extra.begin(), lane_extra.end(), 0.);
    bool give_up = false, reweighted = false, extras_active = false, extras_off = false;
    for (int round = 0; round < (stream_T > 0. ? 9 : 5) && ng < 0 && !give_up; ++round) { // (large slabs: a lower tier is worth more rounds)
        for (int64_t l = 0; l < L; ++l) // (a streamed quad costs what a register quad does plus its load; 15 quads = one lane's walk)
            cum[(size_t)l + 1] = cum[(size_t)l] + lane_w[(size_t)l] * (c_lane * (1. + kStreamCost * (double)lane_sw[(size_t)l] / (double)kResQuads) +
                                                                      c_row * (double)((lmeta[(size_t)l] & 7u) + ((lmeta[(size_t)l] >> 3) & 127u)) +
                                                                      lane_extra[(size_t)l]);
        int64_t l0 = 0;
        wl[0] = 0;
        for (int b = 1; b <= G; ++b) {
            // equal shares of what is LEFT (a workgroup clipped at 1024 lanes hands its surplus to the following ones)
            const double target = cum[(size_t)l0] + (cum[(size_t)L] - cum[(size_t)l0]) / (double)(G - b + 1);
            int64_t l1 = std::lower_bound(cum.begin() + l0, cum.end(), target) - cum.begin();
            if (b == G) l1 = L;
            l1 = std::min<int64_t>(std::max(l1, l0), std::min<int64_t>(L, l0 + kResThreads));
            wl[(size_t)b] = (int32_t)l1;
            l0 = l1;
        }
        if (l0 != L && extras_active && !reweighted) { // the remote-column term alone pushed a workgroup past 1024 lanes: split without it
            std::fill(lane_extra.begin(), lane_extra.end(), 0.);
            extras_active = false;
            extras_off = true;
            --round;
            continue;
        }
        if (l0 != L || cur_opt().resident_equal_lanes) {
            if (reweighted) { give_up = true; break; } // (re-weighting pushed a workgroup past 1024 lanes: give up)
            for (int b = 0; b <= G; ++b) wl[(size_t)b] = (int32_t)std::min<int64_t>((int64_t)b * lpw, L);
        }
        max_rows = 0;
        for (int b = 0; b <= G; ++b) wr[(size_t)b] = wl[(size_t)b] < L ? lrow[(size_t)wl[(size_t)b]] : (int32_t)n;
        for (int b = 0; b < G; ++b) max_rows = std::max(max_rows, wr[(size_t)b + 1] - wr[(size_t)b]);
        lc_bits = 1;
        while ((1 << lc_bits) < max_rows + cap + 2) ++lc_bits;
        if (lc_bits + code_bits > kResWordBits) lc_bits = kResWordBits - code_bits; // (checked against the real counts below)
        if (hipMemcpy(pl->wg_row0.p, wr.data(), wr.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemsetAsync(fail.p, 0, sizeof(int), stream) != hipSuccess) {
            (void)hipGetLastError();
            return no("plan upload failed");
        }
        // re-encode the words for this split (the plan kernel) and fetch the remote-column counts
        auto remap = [&]() -> int { // 0 ok, 1 a source list overflowed, -1 failure
            hipLaunchKernelGGL(k_resident_remap, dim3((unsigned)G), dim3(kResThreads), remap_lds, stream, A.packed, A.row_ptr, A.col_bits, lc_bits,
                               (const int32_t *)pl->wg_row0.p, cap, (int)n_ext, pl->rwords.p, pl->rem_list.p, pl->rem_count.p, fail.p, (int)n,
                               pl->dep_mask.p, (int)chunk_cols);
            int f = 0;
            if (hipMemcpyAsync(&f, fail.p, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipMemcpyAsync(rc.data(), pl->rem_count.p, (size_t)G * sizeof(int32_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess) {
                (void)hipGetLastError();
                return -1;
            }
            return f ? 1 : 0;
        };
        // (round 0 of a later tier is the split of the first tier's round 0: its counts are re-used, the words re-encoded only if it is accepted)
        reuse_round0 = round == 0 && round0_done && !cur_opt().resident_equal_lanes;
        if (reuse_round0) rc = rc_round0;
        else {
            const int rm = remap();
            if (rm < 0) return no("remap failed");
            if (rm > 0) { last_reason = "a workgroup reads more remote columns than its source list holds"; give_up = true; break; }
        }
        if (round == 0) {
            if (!reuse_round0) { rc_round0 = rc; round0_done = true; }
            // a tier whose TOTAL demand is close to the chip's LDS never fits (lanes, not LDS, bound the split; every further round is a
            // re-encoding pass: 10 ms of plan time on a 1.3 M-row system): next tier
            const int t = max_ng < 3 ? (max_ng < 0 ? 0 : max_ng) : 3;
            const double limit0 = (double)(lds_max - (2 * ((size_t)A.table_size + 1) + 48 + 8) * sizeof(double)) / sizeof(double);
            double demand = 0.;
            for (int b = 0; b < G; ++b) demand += (double)(4 - t) * (double)(wr[(size_t)b + 1] - wr[(size_t)b]) + (double)rc[(size_t)b];
            if (demand > 0.88 * limit0 * (double)G) { give_up = true; break; }
        }
        if (round == 0 && c_rem > 0. && !extras_off) { // the remote columns are known now: one more split that counts them
            bool any = false;
            for (int b = 0; b < G; ++b) {
                const int64_t lanes_b = wl[(size_t)b + 1] - wl[(size_t)b];
                for (int64_t l = wl[(size_t)b]; l < wl[(size_t)b + 1]; ++l) {
                    any = any || lane_extra[(size_t)l] == 0.;
                    lane_extra[(size_t)l] = c_rem * (double)rc[(size_t)b] / (double)(lanes_b > 0 ? lanes_b : 1);
                }
            }
            if (any) { extras_active = true; continue; }
        }
        // LDS split: every workgroup holds its slice of u + its remote-column cache, and as many of r, p, s as still fit (tiers: NG =
        // 0 .. 3 of them in global memory instead).  Footprint of workgroup b: (4 - NG) rows_b + remote_b doubles; the largest decides.
        const size_t extra = (2 * ((size_t)A.table_size + 1) + 48 + 8) * sizeof(double);
        max_cols = 0;
        for (int t = 0; t <= 3 && t <= max_ng && ng < 0; ++t) {
            size_t worst = 0;
            for (int b = 0; b < G; ++b) {
                const size_t rows_b = (size_t)((wr[(size_t)b + 1] - wr[(size_t)b] + 1) & ~1), rem_b = (size_t)((rc[(size_t)b] + 1) & ~1);
                worst = std::max(worst, (size_t)(4 - t) * rows_b + rem_b);
                max_cols = std::max(max_cols, (int)(rows_b + rem_b));
            }
            const size_t need = worst * sizeof(double) + extra;
            if (verbose) fprintf(stderr, "[avs resident] round %d, LDS tier %d: largest workgroup footprint %zu B (limit %zu)\n", round, t, need, lds_max);
            if (need <= lds_max) { ng = t; lds = need; }
        }
        if (ng >= 0 && reuse_round0) { // accepted on re-used counts: the words still hold another split's encoding
            const int rm = remap();
            if (rm != 0) return no("remap failed");
        }
        if (ng < 0) { // shrink the offenders (at the largest tier allowed) and split again
            const int t = max_ng < 3 ? (max_ng < 0 ? 0 : max_ng) : 3;
            const double limit = (double)(lds_max - extra) / sizeof(double);
            { // ... unless the MEDIAN workgroup does not fit either: no re-split helps, go to the next tier
                std::vector<double> fps((size_t)G);
                for (int b = 0; b < G; ++b) fps[(size_t)b] = (double)(4 - t) * (double)(wr[(size_t)b + 1] - wr[(size_t)b]) + (double)rc[(size_t)b];
                std::nth_element(fps.begin(), fps.begin() + G / 2, fps.end());
                if (fps[(size_t)G / 2] > 0.98 * limit) { give_up = true; break; }
            }
            for (int b = 0; b < G; ++b) {
                const double fp = (double)(4 - t) * (double)(wr[(size_t)b + 1] - wr[(size_t)b]) + (double)rc[(size_t)b];
                if (fp > 0.97 * limit) {
                    reweighted = true;
                    for (int64_t l = wl[(size_t)b]; l < wl[(size_t)b + 1]; ++l) lane_w[(size_t)l] *= 1.12 * fp / limit;
                }
            }
        }
    }
    }
    stage("split + re-encoding rounds");
    if (verbose) {
        std::vector<int32_t> srt(rc);
        std::sort(srt.begin(), srt.end());
        fprintf(stderr, "[avs resident] remote columns per workgroup: min %d / median %d / 90 %% %d / max %d; rows per workgroup <= %d\n", srt[0],
                srt[(size_t)G / 2], srt[(size_t)G * 9 / 10], srt[(size_t)G - 1], max_rows);
    }
    if (ng < 0) return no(last_reason);
    if ((1 << lc_bits) < max_cols) return no("rows + remote columns exceed the word's column bits");
    lpw = 0;
    for (int b = 0; b < G; ++b) lpw = std::max<int64_t>(lpw, wl[(size_t)b + 1] - wl[(size_t)b]);
    const void *kern = resident_kernel(ng, stream_words > 0);
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096) != hipSuccess) {
        (void)hipGetLastError();
        return no("LDS opt-in refused");
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kResThreads, lds) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        return no("kernel does not fit a CU (registers / LDS)");
    }
    std::vector<uint8_t> whalo((size_t)G, 0);
    std::vector<int32_t> seg(1, 0);
    if (da && da->dd) {
        DistDev h;
        if (hipMemcpy(&h, da->dd, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return no("DistDev download failed"); }
        if (h.paranoid) return no("paranoid mode keeps the launch-per-phase loop");
        const int np = h.npeers;
        std::vector<int32_t> sidx((size_t)(da->n_send > 0 ? da->n_send : 1));
        if (da->n_send && hipMemcpy(sidx.data(), h.send_idx, (size_t)da->n_send * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            return no("send list download failed");
        }
        seg.assign((size_t)(np > 0 ? np : 1) * (size_t)(G + 1), 0);
        for (int i = 0; i < np; ++i) {
            const int32_t *lo = sidx.data() + h.send_off[i], *hi = sidx.data() + h.send_off[i + 1];
            for (int b = 0; b <= G; ++b) {
                const int32_t *itp = std::lower_bound(lo, hi, wr[(size_t)b]);
                seg[(size_t)i * (G + 1) + b] = (int32_t)(itp - sidx.data());
            }
        }
        pl->n_push_wgs = 0;
        for (int b = 0; b < G; ++b) {
            bool any = false;
            for (int i = 0; i < np; ++i) any = any || seg[(size_t)i * (G + 1) + b + 1] > seg[(size_t)i * (G + 1) + b];
            pl->n_push_wgs += any ? 1 : 0;
        }
        // workgroups that read halo columns wait for the peers' flags: every workgroup overlapping a halo-reading 512-row tile of the plan
        std::vector<int32_t> tb((size_t)(da->n_tiles_bnd > 0 ? da->n_tiles_bnd : 1));
        if (da->n_tiles_bnd && hipMemcpy(tb.data(), da->tiles_bnd, (size_t)da->n_tiles_bnd * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            return no("tile list download failed");
        }
        for (int t = 0; t < da->n_tiles_bnd; ++t) {
            const int64_t r0 = (int64_t)tb[(size_t)t] * kTileRows, r1 = std::min<int64_t>(r0 + kTileRows, n);
            for (int b = 0; b < G; ++b)
                if (wr[(size_t)b] < r1 && wr[(size_t)b + 1] > r0) whalo[(size_t)b] = 1;
        }
    }
    bool up = pl->lane_row0.alloc((size_t)L) == AVS_OK && pl->lane_meta.alloc((size_t)L) == AVS_OK && pl->wg_lane0.alloc((size_t)G + 1) == AVS_OK &&
              pl->push_seg.alloc(seg.size()) == AVS_OK && pl->wg_halo.alloc((size_t)G) == AVS_OK && pl->bar_count.alloc(2) == AVS_OK &&
              pl->bar_flags.alloc((size_t)G) == AVS_OK && pl->slots.alloc((size_t)G * 4) == AVS_OK && pl->bcast.alloc(4 * kResGens) == AVS_OK;
    if (!up) return no("plan allocation failed");
    up = hipMemcpy(pl->lane_row0.p, lrow.data(), (size_t)L * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(pl->lane_meta.p, lmeta.data(), (size_t)L * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(pl->wg_lane0.p, wl.data(), wl.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(pl->push_seg.p, seg.data(), seg.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(pl->wg_halo.p, whalo.data(), whalo.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (!up) { (void)hipGetLastError(); return no("plan upload failed"); }
    stage("push segments, uploads");
    pl->streams = false;
    if (stream_words > 0) { // the waves' lane-interleaved streams of the rows that do not fit the registers
        const int wpg = kResThreads / 64;
        std::vector<int32_t> soff((size_t)G * wpg + 1, 0);
        int64_t run = 0;
        for (int b = 0; b < G; ++b)
            for (int w = 0; w < wpg; ++w) {
                int mx = 0;
                for (int64_t l = (int64_t)wl[(size_t)b] + 64 * w; l < std::min<int64_t>((int64_t)wl[(size_t)b] + 64 * (w + 1), wl[(size_t)b + 1]); ++l)
                    mx = std::max(mx, lane_sw[(size_t)l]);
                soff[(size_t)b * wpg + w] = (int32_t)run;
                run += 64 * (int64_t)mx;
            }
        soff[(size_t)G * wpg] = (int32_t)run;
        if (run >= (1ll << 29)) return no("streamed quads exceed 32-bit offsets");
        if (pl->swords.alloc((size_t)run * 4) != AVS_OK || pl->wave_soff.alloc(soff.size()) != AVS_OK ||
            hipMemcpy(pl->wave_soff.p, soff.data(), soff.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            return no("stream allocation failed");
        }
        hipLaunchKernelGGL(k_resident_stream_layout, dim3((unsigned)G), dim3(kResThreads), 0, stream, A.row_ptr, (const uint32_t *)pl->rwords.p,
                           (const int32_t *)pl->lane_row0.p, (const uint32_t *)pl->lane_meta.p, (const int32_t *)pl->wg_lane0.p,
                           (const int32_t *)pl->wave_soff.p, (uint32_t)A.table_size << lc_bits, reinterpret_cast<u4_t *>(pl->swords.p));
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { (void)hipGetLastError(); return no("stream layout failed"); }
        pl->streams = true;
        if (verbose)
            fprintf(stderr, "[avs resident] streamed rows: %.1f %% of the words (%lld of %lld), %.1f MB per iteration incl. padding, %.1f streamed quads per lane\n",
                    100. * (double)stream_words / (double)A.nnz, (long long)stream_words, (long long)A.nnz, (double)run * 16e-6, stream_T);
    }
    if (verbose) {
        timespec t1{};
        clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "[avs resident] plan built in %.2f ms (host: lanes + split; device: re-encoding)\n",
                (t1.tv_sec - plan_t0.tv_sec) * 1e3 + (t1.tv_nsec - plan_t0.tv_nsec) * 1e-6);
    }
    if (verbose)
        fprintf(stderr, "[avs resident] plan: n = %lld, %lld lanes (%lld per workgroup), %d workgroups, <= %d rows per workgroup, %d-bit local columns, "
                        "%d row-local vectors in global memory, LDS %zu B\n", (long long)n, (long long)L, (long long)lpw, G, max_rows, lc_bits, ng, lds);
    pl->G = G;
    pl->lc_bits = lc_bits;
    pl->max_quads = max_quads;
    pl->ng = ng;
    pl->lds = lds;
    pl->ok = true;
    pl->why.clear();
    return true;
}

// Runs the rest of the solve (state in sc / the vectors, as the set-up rounds left it) in ONE cooperative launch.
// *launched = false: the cooperative launch was refused (the grid is not co-resident on this device right now); nothing was
// touched, the plan is retired and the caller carries on with the launch-per-phase loop.
static avs_status resident_run(ResidentPlan *pl, const CsrView &A, double *x, double *r, double *p, double *s, double *u, double *wv,
                               const uint16_t *dcode, const double *invtab, PcgScalars *sc, int max_iters, const DirectArgs *da,
                               hipStream_t stream, bool *launched)
{
    *launched = false;
    ResidentArgs a{};
    a.row_ptr = A.row_ptr;
    a.table = A.table;
    a.table_size = A.table_size;
    a.n = (int)A.n;
    a.G = pl->G;
    a.lane_row0 = pl->lane_row0.p;
    a.lane_meta = pl->lane_meta.p;
    a.wg_lane0 = pl->wg_lane0.p;
    a.wg_row0 = pl->wg_row0.p;
    a.rwords = pl->rwords.p;
    a.swords = pl->streams ? pl->swords.p : nullptr;
    a.wave_soff = pl->streams ? pl->wave_soff.p : nullptr;
    a.lc_bits = pl->lc_bits;
    a.max_quads = pl->max_quads;
    a.rem_list = pl->rem_list.p;
    a.rem_stride = (int)(pl->rem_list.n / (size_t)pl->G);
    a.rem_count = pl->rem_count.p;
    a.x = x; a.r = r; a.p = p; a.s = s; a.u = u; a.w = wv;
    a.dcode = dcode;
    a.invtab = invtab;
    a.bar_count = pl->bar_count.p;
    a.bar_flags = pl->bar_flags.p;
    a.dep_mask = pl->dep_mask.p;
    a.n_push_wgs = pl->n_push_wgs > 0 ? pl->n_push_wgs : 1;
    a.slots = pl->slots.p;
    a.bcast = pl->bcast.p;
    a.sc = sc;
    a.max_iters = max_iters;
    int dev = 0, khz = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) {
        (void)hipGetLastError();
        khz = 100000;
    }
    // a wait between workgroups of ONE device ends in microseconds unless the cooperative grid is not co-resident in time (a shared
    // GPU): 2 s, then the solve is redone by the launch-per-phase loop in the same call; waits on peers keep the transport's 20 s
    long long ms = da ? 20000 : 2000;
    if (cur_opt().dist_timeout_ms > 0) ms = cur_opt().dist_timeout_ms;
    a.timeout_ticks = (long long)khz * ms;
    a.dd = da ? da->dd : nullptr;
    a.epoch = da ? da->epoch : nullptr;
    a.wg_halo = pl->wg_halo.p;
    a.push_seg = pl->push_seg.p;
    a.timers = nullptr;
    a.max_timed = 0;
    a.coherent_fill = cur_opt().resident_coherent_fill;
    if (cur_opt().resident_timers > 0) {
            pl->max_timed = cur_opt().resident_timers > 4096 ? 4096 : cur_opt().resident_timers;
            AVS_TRY(pl->timers.alloc((size_t)pl->max_timed * kResTimers + (size_t)pl->G * 4));
            AVS_HIP(hipMemsetAsync(pl->timers.p, 0, ((size_t)pl->max_timed * kResTimers + (size_t)pl->G * 4) * sizeof(long long), stream));
            a.timers = pl->timers.p;
            a.max_timed = pl->max_timed;
            a.wg_times = pl->timers.p + (size_t)pl->max_timed * kResTimers;
        }
    AVS_HIP(hipMemsetAsync(pl->bar_count.p, 0, 2 * sizeof(unsigned), stream));
    AVS_HIP(hipMemsetAsync(pl->bar_flags.p, 0, (size_t)pl->G * sizeof(unsigned long long), stream));
    AVS_HIP(hipMemsetAsync(pl->slots.p, 0xFF, (size_t)pl->G * 4 * sizeof(double), stream));   // armed: kSentinel in every slot
    AVS_HIP(hipMemsetAsync(pl->bcast.p, 0xFF, 4 * kResGens * sizeof(double), stream));
    void *args[] = {&a};
    const hipError_t le = hipLaunchCooperativeKernel(resident_kernel(pl->ng, pl->streams), dim3((unsigned)pl->G), dim3(kResThreads), args, (unsigned)pl->lds, stream);
    if (le != hipSuccess) {
        (void)hipGetLastError();
        pl->ok = false;
        pl->why = std::string("cooperative launch refused: ") + hipGetErrorString(le);
        if (cur_opt().resident_verbose > 0) fprintf(stderr, "[avs resident] not used: %s\n", pl->why.c_str());
        return AVS_OK;
    }
    *launched = true;
    if (a.timers) { // per-phase averages of workgroup 0 (tuning aid)
        std::vector<long long> t((size_t)pl->max_timed * kResTimers + (size_t)pl->G * 4);
        AVS_HIP(hipMemcpyAsync(t.data(), pl->timers.p, t.size() * sizeof(long long), hipMemcpyDeviceToHost, stream));
        AVS_HIP(hipStreamSynchronize(stream));
        {
            const long long *wt = t.data() + (size_t)pl->max_timed * kResTimers;
            std::vector<double> ua, sp, tot;
            long long first = 0;
            for (int b = 0; b < pl->G; ++b)
                if (wt[4 * b + 3]) {
                    if (!first || wt[4 * b] < first) first = wt[4 * b];
                }
            for (int b = 0; b < pl->G; ++b)
                if (wt[4 * b + 3]) {
                    ua.push_back((double)(wt[4 * b + 1] - wt[4 * b]) * 1e3 / khz);
                    sp.push_back((double)(wt[4 * b + 3] - wt[4 * b + 2]) * 1e3 / khz);
                    tot.push_back((double)(wt[4 * b + 3] - first) * 1e3 / khz);
                }
            if (!tot.empty() && cur_opt().resident_verbose > 0) {
                std::vector<int32_t> wl2((size_t)pl->G + 1), wr2((size_t)pl->G + 1), rc2((size_t)pl->G);
                (void)hipMemcpy(wl2.data(), pl->wg_lane0.p, wl2.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(wr2.data(), pl->wg_row0.p, wr2.size() * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(rc2.data(), pl->rem_count.p, rc2.size() * 4, hipMemcpyDeviceToHost);
                std::vector<int> order((size_t)pl->G);
                for (int b = 0; b < pl->G; ++b) order[(size_t)b] = b;
                std::sort(order.begin(), order.end(), [&](int x, int y) { return wt[4 * x + 3] > wt[4 * y + 3]; });
                for (int i = 0; i < 4; ++i) {
                    const int b = order[(size_t)i];
                    fprintf(stderr, "[avs resident]   slow workgroup %d: %d lanes, %d rows, %d remote; update %.2f, barrier+fill %.2f, spmv %.2f us\n", b,
                            wl2[(size_t)b + 1] - wl2[(size_t)b], wr2[(size_t)b + 1] - wr2[(size_t)b], rc2[(size_t)b], (double)(wt[4 * b + 1] - wt[4 * b]) * 1e3 / khz,
                            (double)(wt[4 * b + 2] - wt[4 * b + 1]) * 1e3 / khz, (double)(wt[4 * b + 3] - wt[4 * b + 2]) * 1e3 / khz);
                }
                const int b = order[(size_t)pl->G / 2];
                fprintf(stderr, "[avs resident]   median workgroup %d: %d lanes, %d rows, %d remote; update %.2f, barrier+fill %.2f, spmv %.2f us\n", b,
                        wl2[(size_t)b + 1] - wl2[(size_t)b], wr2[(size_t)b + 1] - wr2[(size_t)b], rc2[(size_t)b], (double)(wt[4 * b + 1] - wt[4 * b]) * 1e3 / khz,
                        (double)(wt[4 * b + 2] - wt[4 * b + 1]) * 1e3 / khz, (double)(wt[4 * b + 3] - wt[4 * b + 2]) * 1e3 / khz);
            }
            if (!ua.empty()) {
                auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)((v.size() - 1) * f)]; };
                fprintf(stderr, "[avs resident] iteration 20, all workgroups, us (min / median / max): update %.2f / %.2f / %.2f | spmv %.2f / %.2f / %.2f | "
                                "SpMV done after the first workgroup started %.2f / %.2f / %.2f\n", q(ua, 0), q(ua, .5), q(ua, 1), q(sp, 0), q(sp, .5), q(sp, 1),
                        q(tot, 0), q(tot, .5), q(tot, 1));
            }
        }
        double sum[5] = {0, 0, 0, 0, 0};
        int cnt = 0;
        for (int i = 0; i < pl->max_timed; ++i) {
            const long long *q = t.data() + (size_t)i * kResTimers;
            if (!q[5]) break;
            for (int k = 0; k < 5; ++k) sum[k] += (double)(q[k + 1] - q[k]);
            ++cnt;
        }
        if (cur_opt().resident_verbose >= 2)
            for (int i = 0; i < pl->max_timed && i < 80; ++i) {
                const long long *q = t.data() + (size_t)i * kResTimers;
                double rr, den;
                memcpy(&rr, q + 6, 8);
                memcpy(&den, q + 7, 8);
                fprintf(stderr, "[avs resident]   it %d: |r|^2 %.6e  alpha denominator %.6e\n", i, rr, den);
            }
        if (cnt)
            fprintf(stderr, "[avs resident] %d iterations of workgroup 0, us: update+push %.2f | barrier+inv+flags %.2f | remote fill %.2f | spmv+fold %.2f | "
                            "reduce+broadcast %.2f\n", cnt, sum[0] / cnt * 1e3 / khz, sum[1] / cnt * 1e3 / khz, sum[2] / cnt * 1e3 / khz,
                    sum[3] / cnt * 1e3 / khz, sum[4] / cnt * 1e3 / khz);
    }
    return AVS_OK;
}
