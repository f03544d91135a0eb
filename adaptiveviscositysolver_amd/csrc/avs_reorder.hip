// avs_reorder.hip -- brick-major renumbering of the assembled system for the solve.
//
// The reference numbers DOFs (level, axis, 16^3 tile, voxel) (cpp:1566-1593): the u, v and w rows of
// one spatial tile are millions of ids apart, so the x entries a row-tile gathers are shared with row
// tiles that run at a completely different time (and on another XCD) -- PMC shows the x vector being
// pulled ~6x through the L2s (profiles/spmv_traffic.json).  For the solve only, rows/columns are
// renumbered brick-major: all DOFs (every level, every axis) whose face position falls in one B^3 brick
// of fine cells become contiguous, in reference-id order inside the brick (stable sort).  The in-row
// entry order is NOT changed, so every row sum is still the reference's left-to-right sum
// (bit-identical y); inputs and outputs of the C ABI stay in the reference's numbering.
#include <atomic>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "avs_internal.hpp"

namespace avs {

static constexpr int kBlock = 256;

// INTERLEAVE (round 3): the key also carries the fine cell inside the brick (z, y, x), so that inside a brick the stable sort leaves
// the DOFs CELL by cell -- u, v, w of one cell next to each other -- instead of all u faces, then all v, then all w.  A 512-row
// SpMV tile is then a compact 8 x 8 x 2.7-cell slab with all three components: 8 of a row's 15 entries are faces of the OTHER two
// components of the same / adjacent cells, which the axis-major order kept 512 and 1024 rows away -- outside the tile's LDS window
// of x (its own 512 rows) -- and the cell-major order brings inside it.
__global__ __launch_bounds__(kBlock) void k_brick_keys(const int32_t *__restrict__ vdof, int64_t n, int nx, int ny, int nz,
                                                       int shift, int interleave, uint32_t *__restrict__ keys, int32_t *__restrict__ ids)
{
    const int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (d >= n) return;
    const int4 rec = reinterpret_cast<const int4 *>(vdof)[d];
    const int level = rec.x & 0xff;
    int px = rec.y << level, py = rec.z << level, pz = rec.w << level;
    px = px < nx ? px : nx - 1;
    py = py < ny ? py : ny - 1;
    pz = pz < nz ? pz : nz - 1;
    const uint32_t bx = (uint32_t)(px >> shift), by = (uint32_t)(py >> shift), bz = (uint32_t)(pz >> shift);
    const uint32_t nbx = (uint32_t)((nx + (1 << shift) - 1) >> shift), nby = (uint32_t)((ny + (1 << shift) - 1) >> shift);
    uint32_t key = (bz * nby + by) * nbx + bx;
    if (interleave) {
        const uint32_t m = (1u << shift) - 1u;
        key = (key << (3 * shift)) | ((((uint32_t)pz & m) << (2 * shift)) | (((uint32_t)py & m) << shift) | ((uint32_t)px & m));
    }
    keys[d] = key;
    ids[d] = (int32_t)d;
}

__global__ __launch_bounds__(kBlock) void k_invert(const int32_t *__restrict__ perm, int64_t n, int32_t *__restrict__ inv,
                                                   const int32_t *__restrict__ row_ptr, int32_t *__restrict__ len_new)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t old = perm[i];
    inv[old] = (int32_t)i;
    if (row_ptr) len_new[i] = row_ptr[old + 1] - row_ptr[old];
}

__global__ __launch_bounds__(kBlock) void k_permute_rows(int64_t n, const int32_t *__restrict__ perm, const int32_t *__restrict__ inv,
                                                         const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                         const double *__restrict__ val, const int32_t *__restrict__ row_ptr_new,
                                                         int32_t *__restrict__ col_new, double *__restrict__ val_new)
{
    const int sub = threadIdx.x & 15;
    const int64_t group = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 4;
    const int64_t ngroups = ((int64_t)gridDim.x * kBlock) >> 4;
    for (int64_t row = group; row < n; row += ngroups) {
        const int src = row_ptr[perm[row]], dst = row_ptr_new[row], len = row_ptr_new[row + 1] - dst;
        for (int k = sub; k < len; k += 16) {
            col_new[dst + k] = inv[col[src + k]];
            val_new[dst + k] = val[src + k];
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_gather_d(const double *__restrict__ src, const int32_t *__restrict__ idx,
                                                     double *__restrict__ dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock > 0 ? (n + kBlock - 1) / kBlock : 1); }

// c->perm (new -> reference id) / c->inv from the dof table alone: stable radix sort of the brick keys
avs_status build_brick_permutation(avs_ctx *c, int brick_shift)
{
    hipStream_t st = c->stream;
    const int64_t n = c->n_vel;
    DevBuf<uint32_t> &keys_in = c->scratch.keys_in, &keys_out = c->scratch.keys_out;
    DevBuf<int32_t> &ids_in = c->scratch.ids_in;
    AVS_TRY(keys_in.reserve((size_t)n));
    AVS_TRY(keys_out.reserve((size_t)n));
    AVS_TRY(ids_in.reserve((size_t)n));
    AVS_TRY(c->perm.reserve((size_t)(n > 0 ? n : 1)));
    AVS_TRY(c->inv.reserve((size_t)(n > 0 ? n : 1)));
    if (n == 0) return AVS_OK;
    // cell-major order inside the bricks while brick id + cell bits fit the 32-bit sort key (AVS_BRICK_INTERLEAVE=0: axis-major, round 2)
    int interleave = cur_opt().brick_interleave;
    {
        const uint64_t nb = (uint64_t)((c->desc.nx >> brick_shift) + 1) * ((c->desc.ny >> brick_shift) + 1) * ((c->desc.nz >> brick_shift) + 1);
        if ((nb << (3 * brick_shift)) >= (1ull << 32)) interleave = 0;
    }
    hipLaunchKernelGGL(k_brick_keys, dim3(grid_for(n)), dim3(kBlock), 0, st, c->vdof.p, n, c->desc.nx, c->desc.ny, c->desc.nz,
                       brick_shift, interleave, keys_in.p, ids_in.p);
    size_t tmp_bytes = 0;
    AVS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in.p, keys_out.p, ids_in.p, c->perm.p, (size_t)n, 0, 32, st));
    DevBuf<char> &tmp = c->scratch.sort_tmp;
    AVS_TRY(tmp.reserve(tmp_bytes));
    AVS_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys_in.p, keys_out.p, ids_in.p, c->perm.p, (size_t)n, 0, 32, st)); // stable
    hipLaunchKernelGGL(k_invert, dim3(grid_for(n)), dim3(kBlock), 0, st, c->perm.p, n, c->inv.p, (const int32_t *)nullptr,
                       (int32_t *)nullptr);
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

// builds c->perm / c->inv and the permuted system (c->p_*)
avs_status build_reordered_system(avs_ctx *c, int brick_shift)
{
    hipStream_t st = c->stream;
    const int64_t n = c->n_vel, nnz = c->nnz;
    AVS_REQUIRE(c->system_ready, AVS_ESTATE, "system not assembled");
    DevBuf<int32_t> &len_new = c->scratch.len_new, &scan_tmp = c->scratch.scan_tmp;
    AVS_TRY(len_new.reserve((size_t)n + 1));
    AVS_TRY(scan_tmp.reserve(scan_tmp_elems(n)));
    AVS_TRY(build_brick_permutation(c, brick_shift));
    // (reserve: a simulation's sizes change from frame to frame; an exact allocation was a hipFree + hipMalloc of 1.3 GB per frame at 512^3)
    AVS_TRY(c->p_row_ptr.reserve((size_t)n + 1));
    AVS_TRY(c->p_col.reserve((size_t)nnz));
    AVS_TRY(c->p_val.reserve((size_t)nnz));
    AVS_TRY(c->p_rhs.reserve((size_t)n));
    AVS_TRY(c->p_x0.reserve((size_t)n));
    if (n == 0) { c->reordered = true; return AVS_OK; }
    hipLaunchKernelGGL(k_invert, dim3(grid_for(n)), dim3(kBlock), 0, st, c->perm.p, n, c->inv.p, c->row_ptr.p, len_new.p);
    AVS_TRY(exclusive_scan_i32(len_new.p, c->p_row_ptr.p, n, scan_tmp.p, scan_tmp.n, st));
    hipLaunchKernelGGL(k_permute_rows, dim3(8192), dim3(kBlock), 0, st, n, c->perm.p, c->inv.p, c->row_ptr.p, c->col.p, c->val.p,
                       c->p_row_ptr.p, c->p_col.p, c->p_val.p);
    hipLaunchKernelGGL(k_gather_d, dim3(grid_for(n)), dim3(kBlock), 0, st, c->rhs.p, c->perm.p, c->p_rhs.p, n);
    hipLaunchKernelGGL(k_gather_d, dim3(grid_for(n)), dim3(kBlock), 0, st, c->x0.p, c->perm.p, c->p_x0.p, n);
    AVS_HIP(hipGetLastError());
    AVS_TRY(build_matrix_index(c->p_row_ptr.p, c->p_col.p, c->p_val.p, n, nnz, n, c->vi, st));
    c->reordered = true;
    // systems too large for the CU-resident loop: the brick-structured form of the matrix (avs_brick.hip); AVS_BRICK=0 / 1 never / always
    c->brick.clear();
    const int mode = c->opt.brick; // -1 auto, 0 never, 1 always, 2 tune
    // (matrices without one small dictionary -- variable viscosity: tile-local tables or > kViLdsTable values -- never run the CU-resident loop:
    //  the form serves them from kBrickMinSystemRowsVc rows on.  BASELINE configs[2], 256^3 mu(x), 1.27 M rows: 14.5 k -> 15.7 k it/s, round 6)
    const bool resident_kind = c->vi.table_size > 0 && !c->vi.tile_tables && c->vi.table_size <= kBrickTableMax; // (= kViLdsTable of avs_pcg.hip: what the resident loop stages)
    const int64_t min_rows = resident_kind ? kBrickMinSystemRows : kBrickMinSystemRowsVc;
    int want = (mode == 0) ? 0 : (mode == 1 ? 1 : (n >= min_rows ? 1 : 0));
    // AUTO keeps the form where it multiplies faster than the word stream, and picks its tile walk (BrickView::walk).  Both follow from
    // a STRUCTURAL rule -- rows per tile -- so that the same input gives the same kernel, the same fold order of p.Ap and therefore the
    // same iteration count and solution bits in every run, with or without a profiler attached (round-4 review: the choice used to be
    // a three-launch stopwatch).  The rule is what the undisturbed timings of round 4 correlate with (profiles/r04_brick_spmv.md,
    // section 7): the cost is per tile, so the form wins where tiles are full (fat volume, ~490 rows per shell brick: 0.7x the
    // stream's time; thin sheets, ~330-340: 0.87x) and the contiguous-eighths walk wins where every eighth holds the same mix of tiles
    // (fat volumes).  AVS_BRICK_TUNE (mode 2) restores the measurement for experiments.  A simulation's next frames reuse the verdict
    // (and do not build a form that lost) until the row count has moved by more than 10 %.
    const bool autosel = (mode < 0 || mode == 2) && want;
    const bool cached = c->brick_verdict_rows > 0 && std::llabs(n - c->brick_verdict_rows) * 10 <= c->brick_verdict_rows;
    // (a cached "no" is re-examined every 16th frame: the rule depends on the rows per TILE, which can cross the threshold while the row
    //  count stays within 10 % -- round-5 advisor finding)
    if (autosel && cached && !c->brick_verdict && (++c->brick_verdict_reuse & 15) != 0) want = 0;
    if (want) AVS_TRY(build_brick_form(c));
    c->brick.view(c->brick_view, c->vi);
    if (want && c->brick.ready && mode == 2 && cached) {
        // AVS_BRICK_TUNE: the MEASURED verdict and walk of the first frame stand until the row count moves (the rule does not overwrite them)
    } else if (want && c->brick.ready) {
        const double fill = (double)n / (double)c->brick.ntiles;
        if (autosel) c->brick_verdict = fill >= kBrickMinFill ? 1 : 0;
        else c->brick_verdict = 1;
        c->brick_walk = fill >= kBrickEighthsFill ? 0 : 1;
        c->brick_verdict_rows = n;
    } else if (want && autosel) { // not regular enough / too many values / a limit: remembered like a lost comparison
        c->brick_verdict = 0;
        c->brick_verdict_rows = n;
    }
    if (mode == 2 && want && c->brick.ready && !cached) { // measurement instead of the rule (three launches each; host-synchronous)
        AVS_TRY(c->brick_tune_y.reserve((size_t)n));
        CsrView A;
        A.n = n;
        A.nnz = nnz;
        A.row_ptr = c->p_row_ptr.p;
        A.col = c->p_col.p;
        A.val = c->p_val.p;
        c->vi.apply(A);
        double ms[3] = {0., 0., 0.}; // word stream, brick form walk 0, walk 1
        bool launched = true;
        for (int form = 0; form < 3 && launched; ++form) {
            BrickView V = c->brick_view;
            V.walk = form - 1;
            auto launch = [&]() -> avs_status {
                return form ? spmv_brick_launch(V, c->p_x0.p, c->brick_tune_y.p, nullptr, nullptr, st)
                            : spmv_launch(A, c->p_x0.p, c->brick_tune_y.p, 0, st);
            };
            if (launch() != AVS_OK) { launched = false; break; } // (first touch, kernel load)
            Timer t(st);
            t.start();
            for (int r = 0; r < 3 && launched; ++r) launched = launch() == AVS_OK;
            ms[form] = t.stop() / 3;
        }
        if (launched) {
            c->brick_walk = ms[2] < 0.97 * ms[1] ? 1 : 0; // (the contiguous eighths win the loop by more than they win a stand-alone launch)
            const double best = c->brick_walk ? ms[2] : ms[1];
            c->brick_verdict = best < 0.92 * ms[0] ? 1 : 0;
            c->brick_tune_ms[0] = ms[0];
            c->brick_tune_ms[1] = best;
        } else {
            (void)hipGetLastError();
            c->brick_verdict = 0; // a form that cannot be launched here loses; the word stream serves the matrix
        }
    }
    if (autosel && !c->brick_verdict) { // the word stream is the form for this matrix: nothing of the brick form stays allocated
        c->brick.release();
        c->brick.view(c->brick_view, c->vi);
    }
    // AVS_PRECISION_F32 with float vectors: the loop launches the float kernel: the walk is laid out for ITS grid
    const int view_f32 = (c->desc.precision == AVS_PRECISION_F32 && c->opt.f32_vectors != 0) ? 1 : 0;
    c->brick_view.f32 = view_f32;
    if (c->brick.ready && c->opt.brick_plan) { // the persistent grid's walk, laid out from the tiles' estimated costs (BrickForm::plan_walk)
        AVS_TRY(c->brick.plan_walk(brick_partial_count(c->brick_view), c->brick_walk, c->opt.brick_cost, st));
        c->brick.view(c->brick_view, c->vi);
        c->brick_view.f32 = view_f32;
    }
    c->brick_view.walk = c->brick_walk;
    return AVS_OK;
}

// ---------------------------------------------------------------------------------------------
// Value-indexed CSR ("CSR-VI"): the matrix of this problem is made of very few distinct numbers --
// products of a handful of weights, spacings and control-volume fractions (110 distinct values on the
// 512^3 uniform-viscosity beam, 7.7 k with the variable-viscosity field) -- so the 8-byte value stream
// of the SpMV is replaced by a 2-byte code stream into a table of doubles.  Lossless: val[k] ==
// table[codes[k]] bit for bit, products and row sums are unchanged.  More than 65536 distinct values
// => no dictionary (plain CSR).
// ---------------------------------------------------------------------------------------------
static constexpr int kHashBits = 18; // 262144 slots for <= 65536 keys
static constexpr unsigned long long kEmpty = 0xFFFFFFFFFFFFFFFFull; // a NaN pattern: never a matrix value

__device__ __forceinline__ unsigned hash64(unsigned long long k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    return (unsigned)k & ((1u << kHashBits) - 1);
}

static constexpr int kInsertBlock = 1024; // one LDS filter serves 16 waves: first sightings (global look-ups) per wave stay few

__global__ __launch_bounds__(kInsertBlock) void k_vi_insert(const double *__restrict__ val, int64_t nnz, unsigned long long *__restrict__ slots,
                                                      int *__restrict__ count)
{
    // Keys this workgroup has already seen (LDS, four probes): the matrix holds few distinct values, so almost every look-up ends
    // here.  What has to stay rare is the GLOBAL look-up of a hot key: the 110 values of the headline matrix live in 110 lines
    // of one L2 each, and atomics (or L2 reads) on one line serialize -- 8192 short-lived workgroups x 110 first sightings
    // cost 1.5 ms.  Hence few, persistent workgroups (the launch), and an L1 invalidate after a compare-and-swap so that the
    // CU's later first sightings of that key are plain L1 hits instead of another atomic on a stale "empty" line.
    __shared__ unsigned long long seen[1024];
    for (int i = threadIdx.x; i < 1024; i += kInsertBlock) seen[i] = kEmpty;
    __syncthreads();
    constexpr int kU = 4; // values in flight per thread
    for (int64_t k0 = (int64_t)blockIdx.x * kInsertBlock * kU + threadIdx.x; k0 < nnz; k0 += (int64_t)gridDim.x * kInsertBlock * kU) {
        unsigned long long keys[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t k = k0 + (int64_t)u * kInsertBlock;
            keys[u] = k < nnz ? (unsigned long long)__double_as_longlong(val[k]) : kEmpty;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const unsigned long long key = keys[u];
            if (key == kEmpty) continue;
            unsigned h = hash64(key);
            bool hit = false;
#pragma unroll
            for (unsigned t = 0; t < 4u; ++t) {
                const unsigned s = (h + t) & 1023u;
                const unsigned long long cur = seen[s];
                if (cur == key) { hit = true; break; }
                if (cur == kEmpty) { seen[s] = key; break; } // it is (about to be) in the global table; losing a race only costs a repeated look-up
            }
            if (hit) continue;
            for (int probe = 0; probe < (1 << kHashBits); ++probe) {
                const unsigned long long cur = slots[h]; // plain read: an L1 hit for a key this CU has met before
                if (cur == key) break;
                if (cur == kEmpty) {
                    const unsigned long long old = atomicCAS(&slots[h], kEmpty, key);
                    if (old == kEmpty) atomicAdd(count, 1);
                    if (old == kEmpty || old == key) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // drop the stale line from this CU's L1
                        break;
                    }
                }
                if (*count > 65536) return; // too many distinct values: give up early
                h = (h + 1) & ((1u << kHashBits) - 1);
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_vi_collect(const unsigned long long *__restrict__ slots, unsigned long long *__restrict__ keys,
                                                       int *__restrict__ cursor)
{
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= (1u << kHashBits)) return;
    const unsigned long long k = slots[i];
    if (k != kEmpty) {
        const int at = atomicAdd(cursor, 1);
        if (at < 65536) keys[at] = k;
    }
}

// slot -> code: after the host sorted the keys, every table entry finds its slot again
__global__ __launch_bounds__(kBlock) void k_vi_assign(const unsigned long long *__restrict__ slots, const double *__restrict__ table, int n,
                                                      uint16_t *__restrict__ slot_code)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = (unsigned long long)__double_as_longlong(table[i]);
    unsigned h = hash64(key);
    while (slots[h] != key) h = (h + 1) & ((1u << kHashBits) - 1);
    slot_code[h] = (uint16_t)i;
}

__global__ __launch_bounds__(kBlock) void k_vi_encode(const double *__restrict__ val, int64_t nnz, const unsigned long long *__restrict__ slots,
                                                      const uint16_t *__restrict__ slot_code, uint16_t *__restrict__ codes)
{
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * kBlock) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(val[k]);
        unsigned h = hash64(key);
        while (slots[h] != key) h = (h + 1) & ((1u << kHashBits) - 1);
        codes[k] = slot_code[h];
    }
}

avs_status build_value_index(const double *val, int64_t nnz, DevBuf<uint16_t> &codes, DevBuf<double> &table, int *table_size,
                             hipStream_t st)
{
    *table_size = 0;
    if (nnz == 0) return AVS_OK;
    if (!cur_opt().value_index) return AVS_OK;
    DevBuf<unsigned long long> slots, keys;
    DevBuf<uint16_t> slot_code;
    DevBuf<int> counters;
    AVS_TRY(slots.alloc(1u << kHashBits));
    AVS_TRY(keys.alloc(65536));
    AVS_TRY(slot_code.alloc(1u << kHashBits));
    AVS_TRY(counters.alloc(2));
    AVS_HIP(hipMemsetAsync(slots.p, 0xFF, sizeof(unsigned long long) << kHashBits, st));
    AVS_HIP(hipMemsetAsync(counters.p, 0, 2 * sizeof(int), st));
    hipLaunchKernelGGL(k_vi_insert, dim3(256), dim3(kInsertBlock), 0, st, val, nnz, slots.p, counters.p); // persistent: see the kernel
    int h_count[2] = {0, 0};
    AVS_HIP(hipMemcpyAsync(h_count, counters.p, sizeof(h_count), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    if (h_count[0] > 65536) return AVS_OK; // plain CSR
    const int nkeys = h_count[0];
    hipLaunchKernelGGL(k_vi_collect, dim3((1u << kHashBits) / kBlock), dim3(kBlock), 0, st, slots.p, keys.p, counters.p + 1);
    std::vector<unsigned long long> h_keys((size_t)nkeys);
    AVS_HIP(hipMemcpyAsync(h_keys.data(), keys.p, (size_t)nkeys * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    std::sort(h_keys.begin(), h_keys.end()); // deterministic codes whatever the insertion order was
    AVS_TRY(table.alloc((size_t)nkeys));
    AVS_TRY(codes.alloc((size_t)nnz));
    AVS_HIP(hipMemcpyAsync(table.p, h_keys.data(), (size_t)nkeys * sizeof(double), hipMemcpyHostToDevice, st)); // same bit patterns
    hipLaunchKernelGGL(k_vi_assign, dim3(grid_for(nkeys)), dim3(kBlock), 0, st, slots.p, table.p, nkeys, slot_code.p);
    hipLaunchKernelGGL(k_vi_encode, dim3(8192), dim3(kBlock), 0, st, val, nnz, slots.p, slot_code.p, codes.p);
    AVS_HIP(hipGetLastError());
    AVS_HIP(hipStreamSynchronize(st)); // temporaries die here; h_keys must outlive the upload
    *table_size = nkeys;
    return AVS_OK;
}

// (code, column) pairs in one word: the headline system has 110 values (7 bits) and 7.4 M columns (23 bits)
__global__ __launch_bounds__(kBlock) void k_vi_pack(const uint16_t *__restrict__ codes, const int32_t *__restrict__ col, int64_t nnz,
                                                    int col_bits, uint32_t *__restrict__ packed)
{
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * kBlock)
        packed[k] = ((uint32_t)codes[k] << col_bits) | (uint32_t)col[k];
}

static int bits_for(int64_t count) // bits that hold 0 .. count-1
{
    int b = 1;
    while (((int64_t)1 << b) < count) ++b;
    return b;
}

avs_status build_packed_index(const uint16_t *codes, const int32_t *col, int64_t nnz, int64_t n_cols, int table_size,
                              DevBuf<uint32_t> &packed, int *col_bits, hipStream_t st)
{
    *col_bits = 0;
    if (nnz == 0 || table_size <= 0) return AVS_OK;
    if (!cur_opt().value_pack) return AVS_OK;
    const int cb = bits_for(n_cols), vb = bits_for(table_size);
    if (cb + vb > 32) return AVS_OK;
    AVS_TRY(packed.alloc((size_t)nnz));
    hipLaunchKernelGGL(k_vi_pack, dim3(8192), dim3(kBlock), 0, st, codes, col, nnz, cb, packed.p);
    AVS_HIP(hipGetLastError());
    *col_bits = cb;
    return AVS_OK;
}

// x (reference numbering) = xp[inv]
avs_status unpermute(avs_ctx *c, const double *xp, double *x)
{
    const int64_t n = c->n_vel;
    if (n) hipLaunchKernelGGL(k_gather_d, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, xp, c->inv.p, x, n);
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

// ---------------------------------------------------------------------------------------------
// Tile-local dictionaries.  With a smoothly varying viscosity every stress weight w_s differs, the matrix holds
// 10^4..10^5 distinct doubles and ONE dictionary no longer fits in LDS (its look-ups then cost an L1 access each, like
// the x gathers: 262 us per SpMV at 512^3 vs 276 us for plain 12-B CSR).  But inside one 512-row SpMV tile -- a third
// of an 8^3 brick -- the same few hundred values repeat (mu varies little across 8 cells, the geometry factors are
// the usual handful).  So every tile gets its own sorted-by-slot dictionary: 2-B tile-local codes + the tile's table
// (8 B per distinct value, read once per launch into LDS).  Lossless like the global form: val[k] ==
// table[tab_ptr[tile] + codes[k]] bit for bit.
// One workgroup per tile builds an LDS hash set of the tile's values; pass 1 counts, pass 2 (after the scan of the
// counts) writes tables and codes.  Code assignment follows slot order (it may differ between runs when two keys
// collide; the VALUES behind the codes, and therefore every product and sum, do not).
// ---------------------------------------------------------------------------------------------
static constexpr int kTltRows = 512;       // == spmv_tile_rows()
static constexpr int kTltSlots = 4096;     // LDS hash slots per tile (32 KiB)
static constexpr int kTltMaxKeys = 3072;   // more distinct values in one tile: the form is pointless anyway (codes must stay < 4096)

__device__ __forceinline__ unsigned tlt_hash(unsigned long long k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    return (unsigned)k & (kTltSlots - 1);
}

template <bool WRITE>
__global__ __launch_bounds__(kTltRows) void k_tlt_build(int64_t n, const int32_t *__restrict__ row_ptr, const double *__restrict__ val,
                                                        int32_t *__restrict__ tab_len, const int32_t *__restrict__ tab_ptr,
                                                        double *__restrict__ table, uint16_t *__restrict__ codes, int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long tlt_smem[];
    unsigned long long *slots = tlt_smem;                                   // kTltSlots
    uint16_t *rank = reinterpret_cast<uint16_t *>(tlt_smem + kTltSlots);   // kTltSlots (WRITE only)
    __shared__ int count, wave_tot[kTltRows / 64];
    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t row0 = tile * kTltRows, rlast = (row0 + kTltRows < n) ? row0 + kTltRows : n;
    const int s = row_ptr[row0], e = row_ptr[rlast];
    for (int i = tid; i < kTltSlots; i += kTltRows) slots[i] = kEmpty;
    if (tid == 0) count = 0;
    __syncthreads();
    for (int k = s + tid; k < e; k += kTltRows) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(val[k]);
        unsigned h = tlt_hash(key);
        while (true) {
            const unsigned long long cur = slots[h];
            if (cur == key) break;
            if (cur == kEmpty) {
                const unsigned long long old = atomicCAS(&slots[h], kEmpty, key);
                if (old == kEmpty) { atomicAdd(&count, 1); break; }
                if (old == key) break;
            }
            if (*(volatile int *)&count > kTltMaxKeys) break; // give up: the caller drops the whole form
            h = (h + 1) & (kTltSlots - 1);
        }
    }
    __syncthreads();
    const int distinct = count;
    if (distinct > kTltMaxKeys) {
        if (tid == 0) { *overflow = 1; if (!WRITE) tab_len[tile] = 0; }
        return;
    }
    if (!WRITE) {
        if (tid == 0) tab_len[tile] = distinct;
        return;
    }
    // rank of every occupied slot in slot order: thread t owns slots [per t, per t + per)
    constexpr int per = kTltSlots / kTltRows;
    int mine = 0;
#pragma unroll
    for (int i = 0; i < per; ++i) mine += slots[tid * per + i] != kEmpty;
    int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if ((tid & 63) >= o) inc += u;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = inc;
    __syncthreads();
    int base = inc - mine;
    for (int w = 0; w < (tid >> 6); ++w) base += wave_tot[w];
    const int t0 = tab_ptr[tile];
#pragma unroll
    for (int i = 0; i < per; ++i) {
        const unsigned long long key = slots[tid * per + i];
        if (key != kEmpty) {
            rank[tid * per + i] = (uint16_t)base;
            table[t0 + base] = __longlong_as_double((long long)key);
            ++base;
        }
    }
    __syncthreads();
    for (int k = s + tid; k < e; k += kTltRows) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(val[k]);
        unsigned h = tlt_hash(key);
        while (slots[h] != key) h = (h + 1) & (kTltSlots - 1);
        codes[k] = rank[h];
    }
}

static avs_status build_tile_tables(const int32_t *row_ptr, const double *val, int64_t n, int64_t nnz, ValueIndex &vi, bool *ok,
                                    hipStream_t st)
{
    *ok = false;
    if (n == 0 || nnz == 0) return AVS_OK;
    if (!cur_opt().tile_tables) return AVS_OK;
    const int64_t ntiles = (n + kTltRows - 1) / kTltRows;
    DevBuf<int32_t> tab_len, scan_tmp;
    DevBuf<int> overflow;
    AVS_TRY(tab_len.alloc((size_t)ntiles + 1));
    AVS_TRY(vi.tab_ptr.alloc((size_t)ntiles + 1));
    AVS_TRY(scan_tmp.alloc(scan_tmp_elems(ntiles + 1)));
    AVS_TRY(overflow.alloc(1));
    AVS_HIP(hipMemsetAsync(overflow.p, 0, sizeof(int), st));
    const size_t lds = (size_t)kTltSlots * (sizeof(unsigned long long) + sizeof(uint16_t));
    AVS_HIP(hipFuncSetAttribute((const void *)k_tlt_build<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    AVS_HIP(hipFuncSetAttribute((const void *)k_tlt_build<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_tlt_build<false>, dim3((unsigned)ntiles), dim3(kTltRows), lds, st, n, row_ptr, val, tab_len.p,
                       (const int32_t *)nullptr, (double *)nullptr, (uint16_t *)nullptr, overflow.p);
    AVS_TRY(exclusive_scan_i32(tab_len.p, vi.tab_ptr.p, ntiles, scan_tmp.p, scan_tmp.n, st));
    int h_over = 0, total = 0;
    AVS_HIP(hipMemcpyAsync(&h_over, overflow.p, sizeof(int), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipMemcpyAsync(&total, vi.tab_ptr.p + ntiles, sizeof(int), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    // worth it only while the tables stay a small share of the stream: 8 B per entry against 6 B saved per non-zero
    if (h_over || total <= 0 || (int64_t)total * 8 > nnz * 2) return AVS_OK;
    AVS_TRY(vi.table.alloc((size_t)total));
    AVS_TRY(vi.codes.alloc((size_t)nnz));
    hipLaunchKernelGGL(k_tlt_build<true>, dim3((unsigned)ntiles), dim3(kTltRows), lds, st, n, row_ptr, val, (int32_t *)nullptr,
                       (const int32_t *)vi.tab_ptr.p, vi.table.p, vi.codes.p, overflow.p);
    AVS_HIP(hipGetLastError());
    AVS_HIP(hipStreamSynchronize(st)); // temporaries die here
    vi.table_size = total;
    vi.col_bits = 0;
    vi.tile_tables = true;
    *ok = true;
    return AVS_OK;
}

// ---------------------------------------------------------------------------------------------
// Windowed columns.  In the brick-major numbering the columns a 512-row tile reads belong to its own and the neighbouring
// bricks: a few runs of consecutive ids.  Cut the id range into aligned windows of 2^14 ids; a tile touches a handful of
// them (<= 64 or the form is dropped).  A non-zero is then ONE 32-bit word: value code (12 bits, tile-local or global
// dictionary) | window slot (6 bits) | offset in the window (14 bits), whatever the matrix size -- 25-bit columns
// (1024^3) or thousands of distinct values (variable viscosity) no longer cost 6 B per non-zero.  Lossless: the word
// decodes to exactly (col[k], val[k]).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTltRows) void k_cwin_build(int64_t n, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                         const uint16_t *__restrict__ codes, int32_t *__restrict__ cbase,
                                                         uint32_t *__restrict__ words, int *__restrict__ overflow)
{
    __shared__ int keys[256];      // open-addressing set of col >> 14
    __shared__ int sorted[kCwinSlots];
    __shared__ int count;
    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t row0 = tile * kTltRows, rlast = (row0 + kTltRows < n) ? row0 + kTltRows : n;
    const int s = row_ptr[row0], e = row_ptr[rlast];
    if (tid < 256) keys[tid] = -1;
    if (tid == 0) count = 0;
    __syncthreads();
    for (int k = s + tid; k < e; k += kTltRows) {
        const int key = col[k] >> kCwinOffBits;
        unsigned h = ((unsigned)key * 2654435761u) >> 24;
        for (int probe = 0; probe < 256; ++probe) {
            const int cur = keys[h];
            if (cur == key) break;
            if (cur == -1) {
                const int old = atomicCAS(&keys[h], -1, key);
                if (old == -1) { atomicAdd(&count, 1); break; }
                if (old == key) break;
            }
            h = (h + 1) & 255u;
        }
    }
    __syncthreads();
    const int m = count;
    if (m > kCwinSlots) {
        if (tid == 0) *overflow = 1;
        return;
    }
    if (tid == 0) { // <= 64 keys: insertion sort by one thread is plenty
        int c = 0;
        for (int i = 0; i < 256; ++i)
            if (keys[i] != -1) {
                int j = c++;
                const int v = keys[i];
                while (j > 0 && sorted[j - 1] > v) { sorted[j] = sorted[j - 1]; --j; }
                sorted[j] = v;
            }
        for (int i = c; i < kCwinSlots; ++i) sorted[i] = 0x7fffffff;
    }
    __syncthreads();
    if (tid < kCwinSlots) cbase[tile * kCwinSlots + tid] = tid < m ? (sorted[tid] << kCwinOffBits) : 0;
    for (int k = s + tid; k < e; k += kTltRows) {
        const int c = col[k], key = c >> kCwinOffBits;
        int lo = 0, hi = m - 1; // binary search in the sorted window list
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sorted[mid] < key) lo = mid + 1; else hi = mid;
        }
        const unsigned code = codes[k];
        if (code >= (1u << kCwinCodeBits)) { *overflow = 1; return; }
        words[k] = (code << (kCwinOffBits + kCwinSlotBits)) | ((unsigned)lo << kCwinOffBits) | ((unsigned)c & ((1u << kCwinOffBits) - 1u));
    }
}

static avs_status build_column_windows(const int32_t *row_ptr, const int32_t *col, int64_t n, int64_t nnz, ValueIndex &vi, hipStream_t st)
{
    vi.col_windows = false;
    if (n == 0 || nnz == 0 || vi.table_size <= 0) return AVS_OK;
    if (!cur_opt().column_windows) return AVS_OK;
    const int64_t ntiles = (n + kTltRows - 1) / kTltRows;
    DevBuf<int> overflow;
    AVS_TRY(overflow.alloc(1));
    AVS_HIP(hipMemsetAsync(overflow.p, 0, sizeof(int), st));
    AVS_TRY(vi.cbase.alloc((size_t)ntiles * kCwinSlots));
    AVS_TRY(vi.packed.alloc((size_t)nnz));
    hipLaunchKernelGGL(k_cwin_build, dim3((unsigned)ntiles), dim3(kTltRows), 0, st, n, row_ptr, col, (const uint16_t *)vi.codes.p, vi.cbase.p,
                       vi.packed.p, overflow.p);
    AVS_HIP(hipGetLastError());
    int h_over = 0;
    AVS_HIP(hipMemcpyAsync(&h_over, overflow.p, sizeof(int), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    if (!h_over) {
        vi.col_windows = true;
        vi.col_bits = 0;
    }
    return AVS_OK;
}

avs_status build_matrix_index(const int32_t *row_ptr, const int32_t *col, const double *val, int64_t n, int64_t nnz, int64_t n_cols,
                              ValueIndex &vi, hipStream_t st)
{
    vi.clear();
    static std::atomic<uint64_t> generation{0};
    vi.epoch = ++generation;
    if (nnz == 0) return AVS_OK;
    if (!cur_opt().value_index) return AVS_OK;
    int global_size = 0;
    AVS_TRY(build_value_index(val, nnz, vi.codes, vi.table, &global_size, st));
    if (global_size > 0 && global_size <= 2048) { // LDS-resident dictionary (+ packed words when the bits allow)
        vi.table_size = global_size;
        AVS_TRY(build_packed_index(vi.codes.p, col, nnz, n_cols, global_size, vi.packed, &vi.col_bits, st));
        if (vi.col_bits == 0) AVS_TRY(build_column_windows(row_ptr, col, n, nnz, vi, st)); // e.g. 25-bit columns + 8-bit codes (1024^3)
        return AVS_OK;
    }
    bool ok = false;
    ValueIndex tiled;
    AVS_TRY(build_tile_tables(row_ptr, val, n, nnz, tiled, &ok, st));
    if (ok) {
        std::swap(vi.codes.p, tiled.codes.p); std::swap(vi.codes.n, tiled.codes.n);
        std::swap(vi.table.p, tiled.table.p); std::swap(vi.table.n, tiled.table.n);
        std::swap(vi.tab_ptr.p, tiled.tab_ptr.p); std::swap(vi.tab_ptr.n, tiled.tab_ptr.n);
        vi.table_size = tiled.table_size;
        vi.col_bits = 0;
        vi.tile_tables = true;
        AVS_TRY(build_column_windows(row_ptr, col, n, nnz, vi, st)); // 2-B tile-local code + windowed column = one word
        return AVS_OK;
    }
    if (global_size > 0) { // one big dictionary read through L1: still 6 B instead of 12 B per non-zero
        vi.table_size = global_size;
        AVS_TRY(build_packed_index(vi.codes.p, col, nnz, n_cols, global_size, vi.packed, &vi.col_bits, st));
    }
    return AVS_OK;
}

} // namespace avs
