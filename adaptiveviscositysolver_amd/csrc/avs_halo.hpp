// avs_halo.hpp -- device code shared by the SpMV kernels of the partitioned solve (avs_pcg.hip: word-stream kernels, avs_brick.hip: the
// brick-structured form): the CG scalars, the direct transport's flag / slot protocol and the finalizer blocks that
// fold a round's partial sums, all-gather them with the other ranks through the comm blocks and apply the scalar step.
// Nothing here crosses the C ABI.  Reference: the all-reduce of the CG dot products north_star asks for, replacing Eigen's serial
// p.dot(tmp) / squaredNorm inside ConjugateGradient (cpp:618-630).
#pragma once

#include "avs_internal.hpp"

namespace avs {

struct PcgScalars {
    double rho;        // absNew = r.z
    double pAp;
    double rr;         // residualNorm2
    double alpha, beta;
    double threshold, rhs_norm2;
    double red[4];     // reduction staging (all-reduced in multi-GPU mode)
    int iter;          // completed iterations (Eigen's i)
    int done;          // 1: converged, 2: converged in this iteration (x update pending), 3: rhs == 0 (x := 0)
    int fault;         // direct transport: a peer's flag did not arrive in time (1: halo, 2: partial sums); done is set too
    int cancelled;     // avs_cancel reached the loop through the CG sums of a partitioned solve (every rank in the same round); done is set too
    double rho_alt;    // single-GPU loop with the beta step fused into k_update_xp: r.z of odd iterations (rho: even ones), so
                       // that the workgroups that still read the old value never race with the one that writes the new one
};

// ---------------------------------------------------------------------------------------------
// direct transport helpers (DistDev / CommHeader: avs_internal.hpp).  Flags and everything a peer writes are accessed
// with system-scope atomics / fences: the other end is another GPU (xGMI) or another process.
// ---------------------------------------------------------------------------------------------
struct HaloView {                        // by-value argument of the SpMV launch of the direct transport (all tiles + nfin finalizer blocks)
    const DistDev *dd = nullptr;
    const unsigned long long *epoch = nullptr; // completed rounds; this round's flags carry *epoch + 1
    unsigned long long *epoch_w = nullptr;
    unsigned *fin_ticket = nullptr;      // finalizer blocks that have folded their share
    PcgScalars *sc = nullptr;            // writable: the last finalizer applies the scalar step
    const double *pvec = nullptr;        // partial sums of the preceding vector kernel: nred_vec arrays of g
    double *stage = nullptr;             // ntiles * ppt slots: x.Ax partial of every wave / tile, armed with kSentinel between rounds
    double *stage2 = nullptr;            // nfin: the finalizer blocks' folded shares
    const uint8_t *tile_bnd = nullptr;   // per tile: 1 = its rows read halo columns (wait for the peers' flags first)
    int ntiles = 0, ppt = 1, nfin = 1;   // workgroups ntiles .. ntiles + nfin - 1 of the launch are the finalizers
    int g = 0, nred_vec = 0, op = 0;
    double tol = 0.;
    int selftest = 0;                    // the caller (k_direct_selftest) has run paranoid_check itself
    const int *cancel = nullptr;         // avs_cancel: this rank's request word (device); travels in the unused fourth sum of an iteration's round
};
constexpr unsigned long long kSentinel = 0xFFFFFFFFFFFFFFFFull; // an all-ones NaN: never a partial sum
constexpr int kFinShare = 16384;         // stage slots one finalizer block watches and folds

// Synchronisation recipe (no L2 write-back fences: a release fence at agent / system scope flushes every dirty line of the
// XCD's L2 -- the vectors just written -- and measured 15 us per round):
//   * everything a peer (or another XCD) must see is written with system- / agent-scope ATOMIC stores or exchanges: they
//     write through to memory; everything read back is read with atomic loads, which bypass the caches;
//   * "data before flag": the writer waits for its own stores to be acknowledged -- an explicit `s_waitcnt vmcnt(0)`
//     (wait_own_stores; on gfx950 stores count in vmcnt and a write-through store is acknowledged by the memory side it was
//     written through to: the peer's HBM over xGMI for sc0 sc1, this device's memory for sc1) -- or uses exchanges, whose
//     return value IS the acknowledgement, before the ticket / flag goes out.  A workgroup-scope release fence is NOT
//     enough: it lowers to `s_waitcnt lgkmcnt(0)` only (round-2 review, found in the shipped ISA).  tests/test_isa_ordering.py
//     disassembles the library and checks that the wait sits between the last halo store and the barrier / ticket.
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double ld_sys_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void wait_own_stores()
{
    // every VMEM operation this wave has issued (loads, stores, atomics) has completed: no cache write-back, no invalidate.
    // The "memory" clobber keeps the compiler from moving the stores below it or the ticket / flag above it.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// bounded wait: a missing peer must not hang the GPU (the host turns the fault into AVS_ERCCL)
__device__ __forceinline__ bool wait_flag(const unsigned long long *f, unsigned long long want, long long timeout, PcgScalars *sc, int code)
{
    if (ld_sys(f) >= want) return true;
    const long long t0 = wall_clock64();
    while (ld_sys(f) < want) {
        if (wall_clock64() - t0 > timeout) {
            __hip_atomic_store(&sc->fault, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sc->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    return true;
}

// paranoid mode / transport self-test: checksums are sums of the 64-bit patterns (mod 2^64: order-independent, exact)
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v; // lane 0
}
// what rank `from` stores as entry k of its segment in round E of the transport self-test (never the sentinel, never 0)
__device__ __forceinline__ unsigned long long selftest_pattern(unsigned long long E, int from, int k)
{
    return (E << 40) ^ ((unsigned long long)(from + 1) << 32) ^ (unsigned long long)(unsigned)k ^ 0x4000000000000000ull;
}

// wave64 sum by shuffles (fixed order)
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}


enum ScalarOp { OP_NONE = 0, OP_INIT = 1, OP_RHO0 = 2, OP_ALPHA = 3, OP_BETA = 4, OP_SR_INIT = 5, OP_SR_STEP = 6, OP_ALPHA_ODD = 7 };

__device__ inline void apply_scalar_op(PcgScalars *sc, int op, double tol)
{
    switch (op) {
    case OP_INIT: { // red[0] = b.b, red[1] = r.r
        sc->rhs_norm2 = sc->red[0];
        sc->rr = sc->red[1];
        sc->iter = 0;
        if (sc->red[0] == 0.) { sc->done = 3; sc->rr = 0.; break; }
        double thr = tol * tol * sc->red[0];
        const double considerAsZero = 2.2250738585072014e-308;
        if (thr < considerAsZero) thr = considerAsZero;
        sc->threshold = thr;
        sc->done = (sc->red[1] < thr) ? 1 : 0;
        break;
    }
    case OP_RHO0:
        if (!sc->done) sc->rho = sc->red[0];
        break;
    case OP_ALPHA:
    case OP_ALPHA_ODD: // odd iteration of the fused loop: r.z lives in rho_alt
        if (sc->done == 2) sc->done = 1; // the pending x update of the converged iteration has run
        else if (!sc->done) { sc->pAp = sc->red[0]; sc->alpha = (op == OP_ALPHA_ODD ? sc->rho_alt : sc->rho) / sc->red[0]; }
        break;
    case OP_BETA:
        if (!sc->done) {
            sc->rr = sc->red[0];
            if (sc->red[0] < sc->threshold) sc->done = 2; // Eigen: break before i++ (x += alpha p still pending)
            else {
                const double absOld = sc->rho;
                sc->rho = sc->red[1];
                sc->beta = sc->red[1] / absOld;
                sc->iter += 1;
            }
        }
        break;
    case OP_SR_INIT: { // red = [b.b, r.u, r.r, w.u]
        sc->rhs_norm2 = sc->red[0];
        sc->rr = sc->red[2];
        sc->iter = 0;
        if (sc->red[0] == 0.) { sc->done = 3; sc->rr = 0.; break; }
        double thr = tol * tol * sc->red[0];
        const double considerAsZero = 2.2250738585072014e-308;
        if (thr < considerAsZero) thr = considerAsZero;
        sc->threshold = thr;
        if (sc->red[2] < thr) { sc->done = 1; break; }
        sc->done = 0;
        sc->rho = sc->red[1];
        sc->alpha = sc->red[1] / sc->red[3];
        sc->beta = 0.;
        break;
    }
    case OP_SR_STEP: // red = [r.u, r.r, w.u] after x, r were updated with the current alpha
        if (!sc->done) {
            sc->rr = sc->red[1];
            if (sc->red[1] < sc->threshold) sc->done = 1; // x is already updated: plain "converged"
            else {
                const double gamma_old = sc->rho, gamma = sc->red[0], delta = sc->red[2];
                const double beta = gamma / gamma_old;
                sc->alpha = gamma / (delta - beta * gamma / sc->alpha);
                sc->beta = beta;
                sc->rho = gamma;
                sc->iter += 1;
            }
        }
        break;
    default: break;
    }
}



// halo-touching tiles: wait until every peer's entries of this round have landed in my halo area
__device__ __forceinline__ void halo_wait(const HaloView &hv)
{
    const DistDev *dd = hv.dd;
    const int tid = threadIdx.x;
    if (tid < dd->npeers && dd->recv_cnt[tid] > 0)
        wait_flag(&dd->mine->hflag[dd->peer_rank[tid]], *hv.epoch + 1ull, dd->timeout_ticks, hv.sc, 1);
    __syncthreads();
}

// Last workgroup of the halo-touching SpMV launch: fold the round's partial sums (fixed order), exchange them with every
// rank through the comm blocks (each rank stores its 4 sums into every block, parity-buffered, sentinel-armed slots),
// add the contributions in RANK order -- every rank computes bit-identical scalars, so all ranks take the same
// convergence decision in the same iteration -- and apply the scalar step.  Replaces k_reduce + ncclAllReduce + k_scalar.
template <int BLK>
__device__ __forceinline__ double block_fold(double v, double *lds) // fixed order: lanes (shuffle tree), then waves ascending; valid in thread 0
{
    const double sw = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = sw;
    __syncthreads();
    double t = 0.;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < BLK / 64; ++w) t += lds[w];
    }
    return t;
}

// per-thread shares of the preceding vector kernel's partial arrays (plain loads: written by the previous launch)
template <int BLK>
__device__ __forceinline__ void fold_vec_partials(const HaloView &hv, double acc[4])
{
    for (int q = 0; q < hv.nred_vec; ++q) {
        const double *src = hv.pvec + (size_t)q * hv.g;
        double s0 = 0., s1 = 0.;
        int i = threadIdx.x;
        for (; i + BLK < hv.g; i += 2 * BLK) {
            const double a = src[i], b = src[i + BLK];
            s0 += a; s1 += b;
        }
        for (; i < hv.g; i += BLK) s0 += src[i];
        acc[q] = s0 + s1;
    }
}

// Paranoid mode (AVS_DIST_PARANOID=1 and the transport self-test): once per round, BEFORE this rank contributes its partial sums
// (so before any peer can start the next round and overwrite the halo), re-add every halo segment and compare with the checksum
// the sender left in hsum[] ahead of its flag.  A stale / torn / missing entry => fault 4 (AVS_ERCCL on the host).
// pattern != 0: additionally every entry must BE the self-test pattern of this round (returns the number that are not).
template <int BLK>
__device__ unsigned long long paranoid_check(const HaloView &hv, bool pattern)
{
    __shared__ unsigned long long pc_red[BLK / 64];
    __shared__ unsigned long long pc_bad;
    const DistDev *dd = hv.dd;
    const int tid = threadIdx.x;
    const unsigned long long E = *hv.epoch + 1ull;
    if (tid == 0) pc_bad = 0ull;
    if (tid < dd->npeers && dd->recv_cnt[tid] > 0) wait_flag(&dd->mine->hflag[dd->peer_rank[tid]], E, dd->timeout_ticks, hv.sc, 1);
    __syncthreads();
    const unsigned long long *halo = reinterpret_cast<const unsigned long long *>(dd->my_halo);
    for (int i = 0; i < dd->npeers; ++i) {
        const int cnt = dd->recv_cnt[i], off = dd->recv_off[i], q = dd->peer_rank[i];
        if (cnt <= 0) continue;
        unsigned long long cs = 0ull, bad = 0ull;
        for (int j = tid; j < cnt; j += BLK) {
            const unsigned long long v = ld_sys(halo + off + j);
            cs += v;
            if (pattern && v != selftest_pattern(E, q, j)) ++bad;
        }
        cs = wave_sum_u64(cs);
        bad = wave_sum_u64(bad);
        __syncthreads();
        if ((tid & 63) == 0) pc_red[tid >> 6] = cs;
        if ((tid & 63) == 0 && bad) atomicAdd(&pc_bad, bad);
        __syncthreads();
        if (tid == 0) {
            unsigned long long t = 0ull;
#pragma unroll
            for (int w = 0; w < BLK / 64; ++w) t += pc_red[w];
            if (t != ld_sys(&dd->mine->hsum[q])) {
                pc_bad += 1ull << 32; // checksum mismatches in the high half
                if (!pattern) { // a solve stops here; the self-test only counts (every rank must still get every round's total)
                    __hip_atomic_store(&hv.sc->fault, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&hv.sc->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    __syncthreads();
    return pc_bad;
}

// acc[0 .. nred_vec) = this thread's share of the vector partials, acc[nred_vec] = its share of the SpMV's x.Ax
template <int BLK>
__device__ void dist_finalize(const HaloView &hv, double acc[4])
{
    __shared__ double fin_red[4][BLK / 64];
    __shared__ double fin_sum[4];
    __shared__ double fin_all[kMaxRanks * 4];
    const DistDev *dd = hv.dd;
    if (dd->paranoid && !hv.selftest) (void)paranoid_check<BLK>(hv, false); // block-uniform
    const int tid = threadIdx.x;
    const int nred = hv.nred_vec + 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = wave_sum(acc[q]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) fin_red[q][tid >> 6] = acc[q];
    }
    __syncthreads();
    if (tid < 4) { // fixed order: waves ascending
        double t = 0.;
#pragma unroll
        for (int w = 0; w < BLK / 64; ++w) t += fin_red[tid][w];
        fin_sum[tid] = tid < nred ? t : 0.;
        // the fourth sum is free in an iteration's round (r.u, |r|^2, w.u): it carries "this rank was asked to stop" -- summed over the
        // ranks like the others, so every rank leaves the loop in the same round
        if (tid == 3 && nred < 4 && hv.cancel && __hip_atomic_load(hv.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) fin_sum[3] = 1.;
    }
    __syncthreads();
    const unsigned long long E = *hv.epoch + 1ull;
    const int par = (int)(E & 1ull);
    if (tid < dd->world * 4) {
        // One lane per (rank, value).  The value IS the message: every slot of red[parity] is armed with the sentinel (an
        // all-ones NaN), a rank drops its 4 sums into its slots of every block with fire-and-forget write-through stores, and
        // every rank watches the world x 4 slots of its OWN block until none holds the sentinel -- one one-way trip after the
        // slowest rank, instead of exchange (round trip) + flag (one way) + read.  A consumed slot is re-armed at once; it is
        // written again two rounds later, after its writer has seen this rank's contribution to the round in between.
        const int q = tid >> 2, k = tid & 3;
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(dd->all_red_dst[q] + (size_t)par * kMaxRanks * 4 + k);
        st_sys(dst, (unsigned long long)__double_as_longlong(fin_sum[k]));
        unsigned long long *src = reinterpret_cast<unsigned long long *>(&dd->mine->red[par][q][k]);
        unsigned long long v = ld_sys(src);
        if (v == kSentinel) {
            const long long t0 = wall_clock64();
            while ((v = ld_sys(src)) == kSentinel) {
                if (wall_clock64() - t0 > dd->timeout_ticks) {
                    __hip_atomic_store(&hv.sc->fault, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&hv.sc->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = 0ull;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        fin_all[tid] = __longlong_as_double((long long)v);
        st_sys(src, kSentinel);
    }
    __syncthreads();
    if (tid == 0) {
        for (int k = 0; k < nred; ++k) {
            double t = 0.;
            for (int q = 0; q < dd->world; ++q) t += fin_all[q * 4 + k]; // rank order: identical on every rank
            hv.sc->red[k] = t;
        }
        if (hv.op != 0) apply_scalar_op(hv.sc, hv.op, hv.tol);
        if (nred < 4 && hv.cancel) {
            double stop = 0.;
            for (int q = 0; q < dd->world; ++q) stop += fin_all[q * 4 + 3];
            if (stop != 0. && !hv.sc->done) { hv.sc->cancelled = 1; hv.sc->done = 1; }
        }
        *hv.epoch_w = E;
        __hip_atomic_store(hv.fin_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Finalizer block f of the SpMV launch (dispatched behind the tiles).  The tiles' waves drop their x.Ax partials into the stage
// slots with fire-and-forget write-through stores -- no barrier, no ticket, nothing at the end of a tile's life.  A finalizer
// watches its share of the slots until none holds the sentinel any more, folds the share in slot order, re-arms the slots for
// the next round and takes a ticket; the last finalizer to arrive runs dist_finalize.
template <int BLK>
__device__ void halo_finalizer(const HaloView &hv, int f)
{
    __shared__ double share_red[BLK / 64];
    __shared__ int fin_is_last;
    const int tid = threadIdx.x;
    const int total = hv.ntiles * hv.ppt;
    const int lo = f * kFinShare, hi = (lo + kFinShare < total) ? lo + kFinShare : total;
    double acc[4] = {0., 0., 0., 0.};
    if (hv.nfin == 1) fold_vec_partials<BLK>(hv, acc); // ready since the previous launch: folded while the tiles still run
    // one pass checks and folds: lane-strided slots, 8 loads in flight per thread, fixed order => the sum is valid as soon as a pass
    // meets no sentinel.  The last tiles dispatched are usually the last to finish: peek at the final slot before scanning.
    const long long t0 = wall_clock64();
    const unsigned long long *slots = reinterpret_cast<const unsigned long long *>(hv.stage);
    double sum = 0.;
    bool timed_out = false;
    while (hi > lo) {
        // (block-uniform decisions only: thread 0 looks, the barrier broadcasts)
        const int ready = __syncthreads_or(tid == 0 && __hip_atomic_load(slots + (hi - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kSentinel);
        if (ready) {
            int missing = 0;
            double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
            int i = lo + tid;
            for (; i + 7 * BLK < hi; i += 8 * BLK) {
                unsigned long long v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __hip_atomic_load(slots + i + k * BLK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < 8; ++k) missing |= v[k] == kSentinel;
                a0 += __longlong_as_double((long long)v[0]); a1 += __longlong_as_double((long long)v[1]);
                a2 += __longlong_as_double((long long)v[2]); a3 += __longlong_as_double((long long)v[3]);
                a0 += __longlong_as_double((long long)v[4]); a1 += __longlong_as_double((long long)v[5]);
                a2 += __longlong_as_double((long long)v[6]); a3 += __longlong_as_double((long long)v[7]);
            }
            for (; i < hi; i += BLK) {
                const unsigned long long v = __hip_atomic_load(slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                missing |= v == kSentinel;
                a0 += __longlong_as_double((long long)v);
            }
            sum = (a0 + a1) + (a2 + a3);
            if (!__syncthreads_or(missing)) break;
        }
        if (__syncthreads_or(tid == 0 && wall_clock64() - t0 > hv.dd->timeout_ticks)) { timed_out = true; break; }
        __builtin_amdgcn_s_sleep(2);
    }
    if (timed_out && tid == 0) {
        __hip_atomic_store(&hv.sc->fault, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&hv.sc->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int k = lo + tid; k < hi; k += BLK) // re-arm for the next round (complete before the next launch starts)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(hv.stage) + k, kSentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (hv.nfin == 1) { // the usual case of a partitioned solve (<= 2048 tiles per rank): no hand-over between finalizers
        acc[hv.nred_vec] = sum;
        dist_finalize<BLK>(hv, acc);
        return;
    }
    const double t = block_fold<BLK>(sum, share_red);
    if (tid == 0) {
        __hip_atomic_store(hv.stage2 + f, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wait_own_stores();
        fin_is_last = __hip_atomic_fetch_add(hv.fin_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)hv.nfin - 1u;
    }
    __syncthreads();
    if (!fin_is_last) return;
    fold_vec_partials<BLK>(hv, acc);
    double s2 = 0.; // the finalizer blocks' shares (a few dozen at most), lanes ascending
    for (int i = tid; i < hv.nfin; i += BLK) s2 += __hip_atomic_load(hv.stage2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc[hv.nred_vec] = s2;
    dist_finalize<BLK>(hv, acc);
}


} // namespace avs
