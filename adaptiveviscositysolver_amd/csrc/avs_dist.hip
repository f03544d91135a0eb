// avs_dist.hip -- multi-GPU layer of the PCG solve (no reference counterpart).
//
// One rank = one GPU = one process.  Each rank keeps the rows of the faces in its spatial slab
// (avs_partition.cpp), numbered [owned | halo grouped by owner].  Per CG iteration:
//   C1 halo exchange  pack owned p entries (k_pack) -> ncclSend / ncclRecv inside one group, the
//                     receive lands directly in the halo tail of p (no unpack kernel);
//   C2 all-reduce     ncclAllReduce(sum) of 1-2 fp64 scalars, in place in the device scalar block.
// Both are enqueued on the solver's stream: no host synchronisation inside the iteration.
// Messages are tiny (tens of KB / 8-16 B): the cost is launch + link latency, not xGMI bandwidth.
//
// For single-GPU testing the same code path runs over an in-process transport: "virtual ranks"
// (one avs_ctx and one host thread each, sharing one device) exchange through device-to-device
// copies and a host-side rendezvous.  Slow, but it executes the identical partition / halo /
// reduction logic and is what tests/test_gpu_dist.py checks against the single-rank solve.
#include <rccl/rccl.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "avs_internal.hpp"

struct avs_local_group {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    std::vector<avs::PcgDist *> members;
    std::vector<double> red; // world x 4 staging for all-reduce
    std::vector<int32_t> red_i32; // dist_allreduce_i32
    std::vector<uint8_t> blobs; // world x AVS_DIST_BLOB_BYTES: comm-block descriptors of the direct transport
    int direct_votes = 0;       // members whose direct_connect succeeded (agreement: all or none)
    bool failed = false;

    void barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        const long gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen; });
        }
    }
};

namespace avs {

struct PcgDist {
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;     // all-reduces (solver stream)
    ncclComm_t comm_p2p = nullptr; // halo send/recv (communication stream): a communicator of its own, so that
                                   // operations in flight on two streams never share one communicator
    avs_local_group *group = nullptr;
    int device = 0;

    // plan
    int64_t n_global = 0, n_own = 0, n_halo = 0, n_send = 0, nnz_local = 0;
    std::vector<int32_t> peers, send_counts, recv_counts, send_offs, recv_offs;
    DevBuf<int32_t> send_idx, own_global;
    DevBuf<double> sendbuf;

    // local system
    DevBuf<int32_t> row_ptr, col;
    DevBuf<double> val, rhs, x0, x;
    ValueIndex vi;            // lossless storage form of the LOCAL rows (own dictionaries: the rank's rows hold a subset of the values)
    BrickForm brick;          // brick-structured form of the local rows (round 5; slabs of >= kBrickMinSystemRows rows), avs_brick_build.hip
    BrickView brick_view;
    DevBuf<int32_t> local_ref; // reference DOF id behind every local column [owned | halo] (what the brick builder reads the geometry from)
    PcgWork *pcg = nullptr;
    bool partitioned = false, solved = false, reordered = false;
    // slab cuts along cut_axis (fine cells, world + 1 entries): the ones this assembly used, and the ones its per-plane weights suggest
    // for the next frame (slab-local assembly: the pre-pass needs the cuts BEFORE anything is counted)
    std::vector<int32_t> cuts, next_cuts;
    int cut_axis = 0;
    // the second set of the local arrays the [interior | halo-reading] row order is written to (swapped in; kept across frames: the
    // allocation of 170 MB of columns and values cost 3.6 ms of a 9-ms slab assembly, 19 ms the first time)
    // work arrays of the assembly (flags, scans, lookups, sort keys ...), kept across frames: twenty hipMalloc / hipFree pairs per
    // assembly were 2 ms of a 7-ms slab assembly
    struct Scratch {
        DevBuf<uint32_t> keys_in, keys_out, needed_by, pm;
        DevBuf<int32_t> dom, flag, pos, g2l, scan_tmp, ids, halo_tmp, tile_bnd, tile_int, tile_pos, w32, plane_owner, raw_count;
        DevBuf<uint16_t> plane;
        DevBuf<uint8_t> owner, is_halo;
        DevBuf<char> sort_tmp;
        DevBuf<int> mark_err;
        DevBuf<unsigned long long> weight;
    } ws;
    DevBuf<int32_t> alt_row_ptr, alt_col, alt_own, alt_ids, alt_new_of, alt_len;
    DevBuf<double> alt_val, alt_rhs;
    DevBuf<uint32_t> alt_nb;

    // direct transport (peer-mapped comm blocks), see avs_internal.hpp
    int transport = AVS_TRANSPORT_RCCL;
    bool hosted = false;          // neither RCCL communicator nor in-process group: the host program carries the blobs
    void *comm_block = nullptr;
    size_t comm_bytes = 0;
    void *peer_block[kMaxRanks] = {};
    bool peer_ipc[kMaxRanks] = {}; // mapped with hipIpcOpenMemHandle (to be closed)
    DevBuf<DistDev> dd;
    DevBuf<int32_t> push_seg;
    DevBuf<uint8_t> tile_flags;
    int push_grid = 0, push_chunk = 0;
    DevBuf<unsigned long long> epoch;
    DevBuf<unsigned> tickets;
    DevBuf<uint8_t> blob_send, blob_recv; // RCCL groups: staging of the blob all-gather ...
    DevBuf<int> vote_word;                // ... and of the transport votes (allocated once, at avs_dist_init)
    DevBuf<unsigned long long> psum; // paranoid mode: per-peer checksum accumulators of the push
    DistDev dd_host{};               // what d->dd holds (the self-test toggles `paranoid` around its rounds)
    bool exclusive_device = true;    // no other rank of the group on this physical GPU
    long long selftest_bad = -1;     // transport self-test of this plan: -1 not run, else the all-gathered count of bad entries
    int selftest_rounds = 0;
    std::vector<uint8_t> blob;
    bool direct_prepared = false, direct_ready = false;
    bool direct_pending = false; // a new plan exists: the (collective) transport set-up runs at the next avs_dist_solve

    // overlap of the exchange with the interior rows
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_halo = nullptr;
    DevBuf<int32_t> tiles_int, tiles_bnd;
    int n_tiles_int = 0, n_tiles_bnd = 0;
    bool no_overlap = false;
};

#define AVS_NCCL(call)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r__ = (call);                                                                       \
        if (r__ != ncclSuccess) {                                                                        \
            ::avs::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, ncclGetErrorString(r__));      \
            return AVS_ERCCL;                                                                            \
        }                                                                                                \
    } while (0)

__global__ __launch_bounds__(256) void k_pack(const double *__restrict__ p, const int32_t *__restrict__ idx,
                                              double *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = p[idx[i]];
}

__global__ __launch_bounds__(256) void k_scatter_own(const double *__restrict__ x, const int32_t *__restrict__ own_global,
                                                     double *__restrict__ full, int64_t n_own)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_own) full[own_global[i]] = x[i];
}

template <typename T>
__global__ __launch_bounds__(256) void k_gather_i(const T *__restrict__ src, const int32_t *__restrict__ idx,
                                                  T *__restrict__ dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// C1: p_ext = [owned | halo]; fills the halo tail from the peers' owned entries
avs_status dist_halo_exchange(PcgDist *d, double *p_ext, hipStream_t stream)
{
    if (d->world == 1) return AVS_OK;
    if (d->n_send)
        hipLaunchKernelGGL(k_pack, dim3((unsigned)((d->n_send + 255) / 256)), dim3(256), 0, stream, p_ext, d->send_idx.p,
                           d->sendbuf.p, d->n_send);
    if (d->comm) {
        AVS_NCCL(ncclGroupStart());
        for (size_t i = 0; i < d->peers.size(); ++i) {
            if (d->send_counts[i])
                AVS_NCCL(ncclSend(d->sendbuf.p + d->send_offs[i], (size_t)d->send_counts[i], ncclDouble, d->peers[i], d->comm_p2p, stream));
            if (d->recv_counts[i])
                AVS_NCCL(ncclRecv(p_ext + d->n_own + d->recv_offs[i], (size_t)d->recv_counts[i], ncclDouble, d->peers[i], d->comm_p2p, stream));
        }
        AVS_NCCL(ncclGroupEnd());
        return AVS_OK;
    }
    // in-process transport
    avs_local_group *g = d->group;
    AVS_REQUIRE(g, AVS_ESTATE, "multi-GPU layer not initialised");
    AVS_HIP(hipStreamSynchronize(stream)); // my send buffer is packed
    g->barrier();
    for (size_t i = 0; i < d->peers.size(); ++i) {
        if (!d->recv_counts[i]) continue;
        PcgDist *peer = g->members[(size_t)d->peers[i]];
        // where does the peer keep what it sends to me?
        int64_t off = -1;
        for (size_t j = 0; j < peer->peers.size(); ++j)
            if (peer->peers[j] == d->rank) off = peer->send_offs[j];
        AVS_REQUIRE(off >= 0, AVS_EINTERNAL, "peer %d has no send list for rank %d", d->peers[i], d->rank);
        AVS_HIP(hipMemcpyAsync(p_ext + d->n_own + d->recv_offs[i], peer->sendbuf.p + off,
                               (size_t)d->recv_counts[i] * sizeof(double), hipMemcpyDeviceToDevice, stream));
    }
    AVS_HIP(hipStreamSynchronize(stream));
    g->barrier(); // nobody repacks before everybody has copied
    return AVS_OK;
}

bool dist_tile_lists(PcgDist *d, const int32_t **t_int, int *n_int, const int32_t **t_bnd, int *n_bnd)
{
    if (!d || d->world <= 1 || d->no_overlap || !d->comm_stream || d->n_tiles_int + d->n_tiles_bnd == 0) return false;
    if (!cur_opt().dist_overlap) return false;
    *t_int = d->tiles_int.p; *n_int = d->n_tiles_int;
    *t_bnd = d->tiles_bnd.p; *n_bnd = d->n_tiles_bnd;
    return true;
}

// exchange on the communication stream, ordered after everything enqueued on main_stream so far
avs_status dist_halo_begin(PcgDist *d, double *p_ext, hipStream_t main_stream)
{
    AVS_HIP(hipEventRecord(d->ev_ready, main_stream));
    AVS_HIP(hipStreamWaitEvent(d->comm_stream, d->ev_ready, 0));
    AVS_TRY(dist_halo_exchange(d, p_ext, d->comm_stream));
    AVS_HIP(hipEventRecord(d->ev_halo, d->comm_stream));
    return AVS_OK;
}
avs_status dist_halo_end(PcgDist *d, hipStream_t main_stream)
{
    AVS_HIP(hipStreamWaitEvent(main_stream, d->ev_halo, 0));
    return AVS_OK;
}

// C2: in-place sum of `count` doubles (count <= 4) over all ranks
avs_status dist_allreduce(PcgDist *d, double *dev, int count, hipStream_t stream)
{
    if (d->world == 1) return AVS_OK;
    if (d->comm) {
        AVS_NCCL(ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, d->comm, stream));
        return AVS_OK;
    }
    avs_local_group *g = d->group;
    AVS_REQUIRE(g && count <= 4, AVS_ESTATE, "multi-GPU layer not initialised");
    double h[4] = {0, 0, 0, 0};
    AVS_HIP(hipMemcpyAsync(h, dev, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, stream));
    AVS_HIP(hipStreamSynchronize(stream));
    for (int k = 0; k < count; ++k) g->red[(size_t)d->rank * 4 + k] = h[k];
    g->barrier();
    double s[4] = {0, 0, 0, 0};
    for (int r = 0; r < g->world; ++r) // fixed rank order: every rank computes identical sums
        for (int k = 0; k < count; ++k) s[k] += g->red[(size_t)r * 4 + k];
    g->barrier();
    AVS_HIP(hipMemcpyAsync(dev, s, (size_t)count * sizeof(double), hipMemcpyHostToDevice, stream));
    AVS_HIP(hipStreamSynchronize(stream));
    return AVS_OK;
}

// in-place sum of `count` int32 over all ranks (slab-local pre-pass: per-tile DOF counts; slab-local assembly: per-plane weights)
avs_status dist_allreduce_i32(PcgDist *d, int32_t *dev, int64_t count, hipStream_t stream)
{
    if (d->world == 1 || count <= 0) return AVS_OK;
    if (d->comm) {
        AVS_NCCL(ncclAllReduce(dev, dev, (size_t)count, ncclInt32, ncclSum, d->comm, stream));
        return AVS_OK;
    }
    avs_local_group *g = d->group;
    AVS_REQUIRE(g, AVS_ESTATE, "no RCCL communicator and no in-process group: a hosted group sums through the caller's callback");
    std::vector<int32_t> mine((size_t)count);
    AVS_HIP(hipMemcpyAsync(mine.data(), dev, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    AVS_HIP(hipStreamSynchronize(stream));
    g->barrier();
    if (d->rank == 0) g->red_i32.assign((size_t)count, 0);
    g->barrier();
    for (int r = 0; r < g->world; ++r) { // rank after rank: integer sums do not depend on the order, the vector is shared
        if (r == d->rank)
            for (int64_t i = 0; i < count; ++i) g->red_i32[(size_t)i] += mine[(size_t)i];
        g->barrier();
    }
    AVS_HIP(hipMemcpyAsync(dev, g->red_i32.data(), (size_t)count * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    AVS_HIP(hipStreamSynchronize(stream));
    g->barrier(); // nobody clears the vector before everybody has copied it
    return AVS_OK;
}

// world > 1: one all-reduce per iteration instead of two (AVS_DIST_CG=standard keeps Eigen's loop)
bool dist_wants_single_reduction(PcgDist *d)
{
    if (!d || d->world <= 1) return false;
    return !cur_opt().dist_standard_cg;
}

// ---------------------------------------------------------------------------------------------
// Direct transport: set-up.  Each rank allocates its comm block (fine-grained device memory: header + halo area),
// describes it in a blob (IPC handle, pid, raw pointer, where each peer's entries land), the blobs travel (RCCL
// all-gather / in-process group / the host program), every rank maps its peers' blocks and fills its DistDev.
// ---------------------------------------------------------------------------------------------
struct DistBlob {
    uint32_t magic;
    int32_t rank, pid, device;
    uint64_t raw_ptr, bytes;
    int64_t n_halo;
    uint64_t nonce;                 // of the exporting PROCESS: pids alone collide across pid namespaces (containers on one node)
    int32_t recv_off_of[kMaxRanks]; // offset inside my halo area where rank q's entries land (-1: none)
    int32_t recv_cnt_of[kMaxRanks];
    int32_t send_cnt_of[kMaxRanks]; // what I send to rank q: every rank can check every pair from the gathered blobs
    hipIpcMemHandle_t handle;
    int32_t have_handle;
    int32_t pci;                    // physical GPU: domain << 16 | bus << 8 | device (two ranks on one GPU cannot both run the CU-resident loop)
};
static_assert(sizeof(DistBlob) <= AVS_DIST_BLOB_BYTES, "blob does not fit");

// one random 64-bit number per process (not per context): "same process" = same nonce AND same pid
static uint64_t process_nonce()
{
    static const uint64_t nonce = [] {
        uint64_t v = 0;
        if (FILE *f = fopen("/dev/urandom", "rb")) {
            if (fread(&v, sizeof(v), 1, f) != 1) v = 0;
            fclose(f);
        }
        if (!v) { // no urandom: address-space layout + clock + pid
            timespec ts{};
            clock_gettime(CLOCK_REALTIME, &ts);
            v = (uint64_t)(uintptr_t)&v ^ ((uint64_t)ts.tv_nsec << 20) ^ ((uint64_t)ts.tv_sec << 44) ^ (uint64_t)getpid() * 0x9E3779B97F4A7C15ull;
        }
        return v | 1ull; // never 0: a zeroed blob is never "this process"
    }();
    return nonce;
}
static bool same_process(const DistBlob &a, const DistBlob &b) { return a.pid == b.pid && a.nonce == b.nonce && a.nonce != 0; }
constexpr uint32_t kBlobMagic = 0x41565342u; // "AVSB"
constexpr size_t kHeaderBytes = (sizeof(CommHeader) + 255) & ~(size_t)255;

static void direct_release(PcgDist *d)
{
    for (int q = 0; q < kMaxRanks; ++q) {
        if (d->peer_block[q] && d->peer_ipc[q]) (void)hipIpcCloseMemHandle(d->peer_block[q]);
        d->peer_block[q] = nullptr;
        d->peer_ipc[q] = false;
    }
    if (d->comm_block) (void)hipFree(d->comm_block);
    d->comm_block = nullptr;
    d->comm_bytes = 0;
    d->direct_prepared = d->direct_ready = false;
    d->transport = AVS_TRANSPORT_RCCL;
}

// allocates the comm block for the current plan and fills d->blob
static avs_status direct_prepare(avs_ctx *c, PcgDist *d)
{
    direct_release(d);
    AVS_HIP(hipSetDevice(c->desc.device));
    const size_t bytes = kHeaderBytes + (size_t)(d->n_halo > 0 ? d->n_halo : 1) * sizeof(double);
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess || !p) {
        (void)hipGetLastError();
        set_error("direct transport: fine-grained allocation of %zu bytes failed (%s)", bytes, hipGetErrorString(e));
        return AVS_ENOMEM;
    }
    d->comm_block = p;
    d->comm_bytes = bytes;
    AVS_HIP(hipMemset(p, 0, bytes));
    AVS_HIP(hipMemset((char *)p + offsetof(CommHeader, red), 0xFF, sizeof(CommHeader::red))); // partial-sum slots: armed (sentinel)
    DistBlob b;
    memset(&b, 0, sizeof(b));
    b.magic = kBlobMagic;
    b.rank = d->rank;
    b.pid = (int32_t)getpid();
    b.nonce = process_nonce();
    b.device = c->desc.device;
    {
        int dom = 0, bus = 0, dv = 0;
        (void)hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, c->desc.device);
        (void)hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, c->desc.device);
        (void)hipDeviceGetAttribute(&dv, hipDeviceAttributePciDeviceId, c->desc.device);
        (void)hipGetLastError();
        b.pci = (dom << 16) | ((bus & 0xff) << 8) | (dv & 0xff);
    }
    b.raw_ptr = (uint64_t)(uintptr_t)p;
    b.bytes = bytes;
    b.n_halo = d->n_halo;
    for (int q = 0; q < kMaxRanks; ++q) { b.recv_off_of[q] = -1; b.recv_cnt_of[q] = 0; b.send_cnt_of[q] = 0; }
    for (size_t i = 0; i < d->peers.size(); ++i) {
        if (d->recv_counts[i] > 0) {
            b.recv_off_of[d->peers[i]] = d->recv_offs[i];
            b.recv_cnt_of[d->peers[i]] = d->recv_counts[i];
        }
        b.send_cnt_of[d->peers[i]] = d->send_counts[i];
    }
    if (d->world > 1 && hipIpcGetMemHandle(&b.handle, p) == hipSuccess) b.have_handle = 1; // peers in this process need none
    else (void)hipGetLastError();
    d->blob.assign(AVS_DIST_BLOB_BYTES, 0);
    memcpy(d->blob.data(), &b, sizeof(b));
    AVS_TRY(d->dd.alloc(1));
    AVS_TRY(d->epoch.alloc(1));
    AVS_TRY(d->tickets.alloc(2));
    AVS_TRY(d->psum.alloc(kMaxRanks));
    AVS_HIP(hipMemset(d->epoch.p, 0, sizeof(unsigned long long)));
    AVS_HIP(hipMemset(d->tickets.p, 0, 2 * sizeof(unsigned)));
    AVS_HIP(hipMemset(d->psum.p, 0, kMaxRanks * sizeof(unsigned long long)));
    d->direct_prepared = true;
    return AVS_OK;
}

// what I send to q must be what q expects from me (both sides derived their lists independently from the symmetric pattern):
// a mismatch fails here, with a status, instead of hanging in the first exchange -- whatever the transport
static avs_status check_exchange_counts(PcgDist *d, const uint8_t *blobs)
{
    // EVERY pair, from the gathered blobs only: all ranks see the same blobs, so all ranks return the same status (a rank-local
    // verdict would leave the others waiting in the next collective)
    for (int a = 0; a < d->world; ++a)
        for (int q = 0; q < d->world; ++q) {
            if (q == a) continue;
            DistBlob ba, bq;
            memcpy(&ba, blobs + (size_t)a * AVS_DIST_BLOB_BYTES, sizeof(ba));
            memcpy(&bq, blobs + (size_t)q * AVS_DIST_BLOB_BYTES, sizeof(bq));
            if (ba.magic != kBlobMagic || bq.magic != kBlobMagic) continue; // a rank without a block: nothing to compare with
            AVS_REQUIRE(ba.send_cnt_of[q] == bq.recv_cnt_of[a], AVS_EINTERNAL, "rank %d sends %d halo entries to rank %d, which expects %d", a,
                        ba.send_cnt_of[q], q, bq.recv_cnt_of[a]);
        }
    return AVS_OK;
}

// maps the peers' blocks and fills the device-side descriptor; `blobs` = world x AVS_DIST_BLOB_BYTES in rank order
static avs_status direct_connect(avs_ctx *c, PcgDist *d, const uint8_t *blobs, bool check_counts = true)
{
    AVS_REQUIRE(d->direct_prepared, AVS_ESTATE, "direct transport: no comm block (assemble / partition first)");
    AVS_HIP(hipSetDevice(c->desc.device));
    std::vector<DistBlob> all((size_t)d->world);
    for (int q = 0; q < d->world; ++q) {
        memcpy(&all[(size_t)q], blobs + (size_t)q * AVS_DIST_BLOB_BYTES, sizeof(DistBlob));
        AVS_REQUIRE(all[(size_t)q].magic == kBlobMagic && all[(size_t)q].rank == q, AVS_EINVAL, "direct transport: blob %d is not rank %d's", q, q);
    }
    if (check_counts) AVS_TRY(check_exchange_counts(d, blobs)); // (the loop-back measurement aid hands in synthetic blobs)
    const DistBlob &me = all[(size_t)d->rank];
    for (int q = 0; q < d->world; ++q) {
        if (q == d->rank) { d->peer_block[q] = d->comm_block; continue; }
        const DistBlob &b = all[(size_t)q];
        if (same_process(b, me) && me.nonce == process_nonce()) { // same process (in-process group): the pointer itself, peer access when on another GPU
            if (b.device != c->desc.device) {
                const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                    set_error("direct transport: no peer access from device %d to %d (%s)", c->desc.device, b.device, hipGetErrorString(e));
                    return AVS_EHIP;
                }
                (void)hipGetLastError();
            }
            d->peer_block[q] = (void *)(uintptr_t)b.raw_ptr;
        } else {
            AVS_REQUIRE(b.have_handle, AVS_EHIP, "direct transport: rank %d could not export its comm block", q);
            void *m = nullptr;
            const hipError_t e = hipIpcOpenMemHandle(&m, b.handle, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess || !m) {
                (void)hipGetLastError();
                set_error("direct transport: hipIpcOpenMemHandle(rank %d) failed: %s", q, hipGetErrorString(e));
                return AVS_EHIP;
            }
            d->peer_block[q] = m;
            d->peer_ipc[q] = true;
        }
    }
    DistDev h;
    memset(&h, 0, sizeof(h));
    h.rank = d->rank;
    h.world = d->world;
    h.npeers = (int)d->peers.size();
    h.n_own = d->n_own;
    h.mine = (CommHeader *)d->comm_block;
    h.my_halo = (const double *)((char *)d->comm_block + kHeaderBytes);
    AVS_REQUIRE(h.npeers <= kMaxRanks, AVS_EINVAL, "too many peers");
    for (int i = 0; i < h.npeers; ++i) {
        const int q = d->peers[(size_t)i];
        h.peer_rank[i] = q;
        h.send_off[i] = d->send_offs[(size_t)i];
        h.recv_cnt[i] = d->recv_counts[(size_t)i];
        h.recv_off[i] = d->recv_offs[(size_t)i];
        const int off = all[(size_t)q].recv_off_of[d->rank];
        h.peer_halo_dst[i] = (double *)((char *)d->peer_block[q] + kHeaderBytes) + (off > 0 ? off : 0);
        h.peer_hflag_dst[i] = &((CommHeader *)d->peer_block[q])->hflag[d->rank];
        h.peer_hsum_dst[i] = &((CommHeader *)d->peer_block[q])->hsum[d->rank];
    }
    h.send_off[h.npeers] = (int)d->n_send;
    for (int q = 0; q < d->world; ++q) {
        CommHeader *hq = (CommHeader *)d->peer_block[q];
        h.all_red_dst[q] = &hq->red[0][d->rank][0];
    }
    h.send_idx = d->send_idx.p;
    h.psum = d->psum.p;
    h.inject_stale_round = -1;
    h.paranoid = cur_opt().paranoid != 0;
#ifdef AVS_PROBES
    if (const char *e = getenv("AVS_DIST_INJECT_STALE")) { // test hook: proves the paranoid check sees a stale entry
        h.inject_stale_round = atoll(e);
        h.paranoid = 1;
    }
#endif
    // segments of the send lists per workgroup of the fused update + push kernel (send lists are ascending local row ids)
    {
        sr_update_geometry((long long)d->n_own, &d->push_grid, &d->push_chunk);
        const int G = d->push_grid;
        std::vector<int32_t> sidx((size_t)d->n_send), seg((size_t)(h.npeers > 0 ? h.npeers : 1) * (size_t)(G + 1), 0);
        if (d->n_send) AVS_HIP(hipMemcpy(sidx.data(), d->send_idx.p, sidx.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        std::vector<char> has((size_t)G, 0);
        for (int i = 0; i < h.npeers; ++i) {
            const int32_t *lo = sidx.data() + h.send_off[i], *hi = sidx.data() + h.send_off[i + 1];
            for (int b = 0; b <= G; ++b) {
                const long long first_row = (long long)b * d->push_chunk;
                const int32_t *it = std::lower_bound(lo, hi, first_row, [](int32_t v, long long key) { return (long long)v < key; });
                seg[(size_t)i * (G + 1) + b] = (int32_t)(it - sidx.data());
            }
            for (int b = 0; b < G; ++b)
                if (seg[(size_t)i * (G + 1) + b + 1] > seg[(size_t)i * (G + 1) + b]) has[(size_t)b] = 1;
        }
        int nblocks = 0;
        for (int b = 0; b < G; ++b) nblocks += has[(size_t)b];
        h.n_send_blocks = nblocks;
        AVS_TRY(d->push_seg.alloc(seg.size()));
        AVS_HIP(hipMemcpy(d->push_seg.p, seg.data(), seg.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        h.push_seg = d->push_seg.p;
        h.push_grid = G;
        h.push_chunk = d->push_chunk;
    }
    { // per-tile flag: does the tile read halo columns? (from the plan's list of such tiles)
        const int nt = d->n_tiles_int + d->n_tiles_bnd;
        std::vector<uint8_t> f((size_t)(nt > 0 ? nt : 1), 0);
        std::vector<int32_t> tb((size_t)d->n_tiles_bnd);
        if (d->n_tiles_bnd) AVS_HIP(hipMemcpy(tb.data(), d->tiles_bnd.p, tb.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int32_t t : tb) f[(size_t)t] = 1;
        AVS_TRY(d->tile_flags.alloc(f.size()));
        AVS_HIP(hipMemcpy(d->tile_flags.p, f.data(), f.size(), hipMemcpyHostToDevice));
    }
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->desc.device) != hipSuccess || khz <= 0) {
        (void)hipGetLastError();
        khz = 100000; // 100 MHz constant clock
    }
    long long ms = 20000;
    if (cur_opt().dist_timeout_ms > 0) ms = cur_opt().dist_timeout_ms;
    h.timeout_ticks = (long long)khz * ms;
    AVS_HIP(hipMemcpy(d->dd.p, &h, sizeof(h), hipMemcpyHostToDevice));
    d->dd_host = h;
    d->exclusive_device = true;
    for (int q = 0; q < d->world; ++q)
        if (q != d->rank && all[(size_t)q].pci == me.pci && check_counts) d->exclusive_device = false; // (loop-back: the "peers" are this rank itself)
    d->direct_ready = true;
    d->transport = AVS_TRANSPORT_DIRECT;
    d->selftest_bad = -1;
    d->selftest_rounds = 0;
    return AVS_OK;
}

bool dist_direct_args(PcgDist *d, DirectArgs *out)
{
    if (!d || !d->direct_ready) return false;
    out->dd = d->dd.p;
    out->epoch = d->epoch.p;
    out->push_ticket = d->tickets.p;
    out->fin_ticket = d->tickets.p + 1;
    out->n_send = (int)d->n_send;
    out->npeers = (int)d->peers.size();
    out->push_grid = d->push_grid;
    out->push_chunk = d->push_chunk;
    out->tiles_int = d->tiles_int.p;
    out->tiles_bnd = d->tiles_bnd.p;
    out->n_tiles_int = d->n_tiles_int;
    out->n_tiles_bnd = d->n_tiles_bnd;
    out->tile_flags = d->tile_flags.p;
    out->exclusive_device = d->exclusive_device;
    return true;
}

// Transport self-test (avs_pcg.hip: direct_selftest).  Runs with `paranoid` forced on, then restores the plan's setting.  Returns
// AVS_OK when the rounds ran; *passed tells whether every entry of every round arrived intact and in time ON EVERY RANK (the
// count is all-gathered through the comm blocks, so all ranks agree without another collective).  AVS_DIST_SELFTEST_ROUNDS=0 skips.
static avs_status run_selftest(avs_ctx *c, PcgDist *d, bool *passed)
{
    *passed = true;
    int rounds = cur_opt().dist_selftest_rounds;
    if (cur_opt().dist_loopback) rounds = 0; // a looped-back rank never fills its halo area
    if (rounds <= 0 || !d->direct_ready) return AVS_OK;
    DirectArgs da;
    if (!dist_direct_args(d, &da)) return AVS_OK;
    DistDev h = d->dd_host;
    const int keep_paranoid = h.paranoid;
    h.paranoid = 1;
    AVS_HIP(hipMemcpyAsync(d->dd.p, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));
    long long bad = 0;
    int fault = 0;
    const avs_status st = direct_selftest(da, rounds, c->stream, &bad, &fault);
    h.paranoid = keep_paranoid;
    AVS_HIP(hipMemcpyAsync(d->dd.p, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));
    AVS_HIP(hipStreamSynchronize(c->stream));
    AVS_TRY(st);
    d->selftest_rounds = rounds;
    d->selftest_bad = fault ? -2 : bad;
    *passed = fault == 0 && bad == 0;
    if (!*passed)
        set_error("direct transport self-test failed on rank %d: %lld bad halo entries / checksums in %d rounds, fault %d", d->rank, bad, rounds, fault);
    return AVS_OK;
}

// after a plan exists: choose the transport, exchange the blobs, connect -- all ranks end up with the SAME transport
static avs_status direct_setup(avs_ctx *c, PcgDist *d)
{
    const bool forced = cur_opt().transport == 2;
    const bool off = cur_opt().transport == 1;
    direct_release(d);
    if (d->hosted) { // the host program finishes the set-up (avs_dist_export_blob / avs_dist_import_blobs)
        AVS_REQUIRE(!off, AVS_EINVAL, "a hosted group has no RCCL communicator: AVS_DIST_TRANSPORT=rccl is impossible");
        return direct_prepare(c, d);
    }
    avs_status st = direct_prepare(c, d); // the blobs also carry the send / receive counts every transport must agree on
    std::vector<uint8_t> all((size_t)d->world * AVS_DIST_BLOB_BYTES, 0);
    int ok = st == AVS_OK ? 1 : 0;
    if (d->world == 1) {
        if (ok) memcpy(all.data(), d->blob.data(), AVS_DIST_BLOB_BYTES);
    } else if (d->group) {
        avs_local_group *g = d->group;
        {
            std::lock_guard<std::mutex> lk(g->m);
            if (g->blobs.size() != all.size()) g->blobs.assign(all.size(), 0);
            if (ok) memcpy(g->blobs.data() + (size_t)d->rank * AVS_DIST_BLOB_BYTES, d->blob.data(), AVS_DIST_BLOB_BYTES);
            else memset(g->blobs.data() + (size_t)d->rank * AVS_DIST_BLOB_BYTES, 0, AVS_DIST_BLOB_BYTES);
        }
        g->barrier();
        {
            std::lock_guard<std::mutex> lk(g->m);
            all = g->blobs;
        }
        g->barrier();
    } else if (d->comm) {
        DevBuf<uint8_t> &send = d->blob_send, &recv = d->blob_recv; // (allocated in avs_dist_init: no allocation between collectives)
        AVS_REQUIRE(send.p && recv.n >= all.size(), AVS_EINTERNAL, "blob staging was not allocated at avs_dist_init");
        if (!ok) d->blob.assign(AVS_DIST_BLOB_BYTES, 0);
        AVS_HIP(hipMemcpyAsync(send.p, d->blob.data(), AVS_DIST_BLOB_BYTES, hipMemcpyHostToDevice, c->stream));
        AVS_NCCL(ncclAllGather(send.p, recv.p, AVS_DIST_BLOB_BYTES, ncclUint8, d->comm, c->stream));
        AVS_HIP(hipMemcpyAsync(all.data(), recv.p, all.size(), hipMemcpyDeviceToHost, c->stream));
        AVS_HIP(hipStreamSynchronize(c->stream));
    } else {
        return AVS_OK;
    }
    if (d->world > 1) AVS_TRY(check_exchange_counts(d, all.data())); // (decided from the gathered blobs: the same status on every rank)
    if (off) { // RCCL transport requested: the comm block is not needed
        direct_release(d);
        return AVS_OK;
    }
    // Every decision below is taken from the GATHERED blobs (identical on all ranks), never from a rank-local result, so that no
    // rank leaves while the others enter the vote: a blob without the magic = that rank could not prepare a block = nobody connects.
    bool every = true, shared_device = false;
    for (int a = 0; a < d->world; ++a) {
        DistBlob ba;
        memcpy(&ba, all.data() + (size_t)a * AVS_DIST_BLOB_BYTES, sizeof(ba));
        if (ba.magic != kBlobMagic) { every = false; continue; }
        // Ranks of ONE process on ONE device (virtual ranks, a test set-up) share that device's few hardware queues: the
        // waiting kernel of one rank can sit in the same queue in front of the kernel it waits for (measured:
        // tools/probes/spin_probe.hip, only as many streams as hardware queues make progress).  Such groups keep the
        // host-mediated transport unless AVS_DIST_TRANSPORT=direct insists.
        for (int b2 = a + 1; b2 < d->world; ++b2) {
            DistBlob bb;
            memcpy(&bb, all.data() + (size_t)b2 * AVS_DIST_BLOB_BYTES, sizeof(bb));
            if (bb.magic == kBlobMagic && same_process(ba, bb) && ba.device == bb.device) shared_device = true;
        }
    }
    if (!every || (shared_device && !forced)) {
        direct_release(d);
        if (!every && forced) {
            set_error("direct transport: a rank could not allocate / export its comm block");
            return st != AVS_OK ? st : AVS_ERCCL;
        }
        return AVS_OK;
    }
    st = direct_connect(c, d, all.data());
    ok = st == AVS_OK ? 1 : 0;
    // agreement: one rank that cannot connect sends everybody back to the RCCL transport.  (The vote's device word is allocated
    // before the collective is entered; an allocation failure there is reported as a "no" vote through a host-side 0.)
    auto vote = [&](int mine, int *all_ok) -> avs_status {
        *all_ok = mine;
        if (d->world > 1 && d->group) {
            avs_local_group *g = d->group;
            {
                std::lock_guard<std::mutex> lk(g->m);
                if (d->rank == 0) g->direct_votes = 0;
            }
            g->barrier();
            {
                std::lock_guard<std::mutex> lk(g->m);
                g->direct_votes += mine;
            }
            g->barrier();
            {
                std::lock_guard<std::mutex> lk(g->m);
                *all_ok = g->direct_votes == d->world;
            }
            g->barrier();
        } else if (d->world > 1 && d->comm) {
            AVS_HIP(hipMemcpyAsync(d->vote_word.p, &mine, sizeof(int), hipMemcpyHostToDevice, c->stream));
            AVS_NCCL(ncclAllReduce(d->vote_word.p, d->vote_word.p, 1, ncclInt32, ncclMin, d->comm, c->stream));
            AVS_HIP(hipMemcpyAsync(all_ok, d->vote_word.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            AVS_HIP(hipStreamSynchronize(c->stream));
        }
        return AVS_OK;
    };
    int all_ok = ok;
    AVS_TRY(vote(ok, &all_ok));
    if (all_ok) {
        // every rank is connected: prove the links before a solve depends on them.  The rounds rendezvous through the comm blocks
        // themselves; the bad-entry count is all-gathered there, and one more vote covers a rank whose rounds timed out.
        bool passed = true;
        const avs_status ts = run_selftest(c, d, &passed);
        int pass_all = (ts == AVS_OK && passed) ? 1 : 0;
        AVS_TRY(vote(pass_all, &pass_all));
        if (!pass_all) {
            all_ok = 0;
            if (st == AVS_OK) st = AVS_ERCCL;
        }
    }
    if (!all_ok) {
        if (forced) {
            if (ok && st == AVS_OK) set_error("direct transport: another rank could not connect");
            return st != AVS_OK ? st : AVS_ERCCL;
        }
        direct_release(d); // quietly: the RCCL / host-mediated transport takes over
    }
    return AVS_OK;
}

void dist_release(avs_ctx *c)
{
    PcgDist *d = c->dist;
    if (!d) return;
    direct_release(d);
    pcg_destroy(d->pcg);
    if (d->comm_p2p && d->comm_p2p != d->comm) (void)ncclCommDestroy(d->comm_p2p);
    if (d->comm) (void)ncclCommDestroy(d->comm);
    if (d->comm_stream) (void)hipStreamDestroy(d->comm_stream);
    if (d->ev_ready) (void)hipEventDestroy(d->ev_ready);
    if (d->ev_halo) (void)hipEventDestroy(d->ev_halo);
    delete d;
    c->dist = nullptr;
}


// ---------------------------------------------------------------------------------------------
// Device-side partition planner: the same plan avs_partition.cpp builds on the host (slabs on multiples of
// 2^(levels-1) fine cells balanced by non-zeros, local numbering [owned ascending | halo grouped by owner,
// ascending], ascending send lists), but from the CSR that already lives in HBM: flag + exclusive scan +
// scatter, one pass over the rows for the halo / "needed by" marks.  Only O(planes) + O(world) scalars visit
// the host.  AVS_DIST_PLAN=host selects the host planner (tests compare the two array for array).
// ---------------------------------------------------------------------------------------------
constexpr int kPlanLdsPlanes = 2048;

__device__ __forceinline__ int plane_of_dof(const int32_t *__restrict__ vdof, int64_t src, int axis, int extent, int shift)
{
    const int32_t *t = vdof + 4 * src;
    const int level = t[0] & 0xff;
    int64_t pos = (int64_t)t[1 + axis] << level;
    if (pos >= extent) pos = extent - 1;
    return (int)(pos >> shift);
}

// plane of every row + per-plane weights (non-zeros + 2 per row, as the host planner)
__global__ __launch_bounds__(256) void k_plan_planes(int64_t n, const int32_t *__restrict__ vdof, const int32_t *__restrict__ perm,
                                                     const int32_t *__restrict__ row_ptr, int axis, int extent, int shift,
                                                     int nplanes, uint16_t *__restrict__ plane, unsigned long long *__restrict__ weight)
{
    __shared__ unsigned long long h[kPlanLdsPlanes];
    const bool lds = nplanes <= kPlanLdsPlanes;
    if (lds) {
        for (int i = threadIdx.x; i < nplanes; i += 256) h[i] = 0ull;
        __syncthreads();
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int pl = plane_of_dof(vdof, perm ? perm[i] : i, axis, extent, shift);
        plane[i] = (uint16_t)pl;
        const unsigned long long w = (unsigned long long)(row_ptr[i + 1] - row_ptr[i]) + 2ull;
        atomicAdd(lds ? &h[pl] : &weight[pl], w);
    }
    if (lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < nplanes; i += 256)
            if (h[i]) atomicAdd(&weight[i], h[i]);
    }
}

__global__ __launch_bounds__(256) void k_plan_owner(int64_t n, const uint16_t *__restrict__ plane, const int32_t *__restrict__ plane_owner,
                                                    int rank, uint8_t *__restrict__ owner, int32_t *__restrict__ own_flag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int q = plane_owner[plane[i]];
    owner[i] = (uint8_t)q;
    own_flag[i] = q == rank;
}

// one pass over all rows: my rows mark the foreign columns they read (halo), foreign rows mark which of my
// columns their owner needs (bit q)
__global__ __launch_bounds__(256) void k_plan_mark(int64_t n, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                   const uint8_t *__restrict__ owner, int rank, uint8_t *__restrict__ is_halo,
                                                   uint32_t *__restrict__ needed_by)
{
    const int sub = threadIdx.x & 15;
    const int64_t group = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int64_t ngroups = ((int64_t)gridDim.x * 256) >> 4;
    for (int64_t r = group; r < n; r += ngroups) {
        const int q = owner[r];
        const int s = row_ptr[r], e = row_ptr[r + 1];
        for (int k = s + sub; k < e; k += 16) {
            const int32_t c = col[k];
            const int oc = owner[c];
            if (q == rank) {
                if (oc != rank) is_halo[c] = 1; // same value from every writer
            } else if (oc == rank) {
                const uint32_t bit = 1u << q;
                if (!(needed_by[c] & bit)) atomicOr(&needed_by[c], bit);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_plan_flag_halo(int64_t n, const uint8_t *__restrict__ is_halo, const uint8_t *__restrict__ owner,
                                                        int q, int32_t *__restrict__ f, const int32_t *__restrict__ dom = nullptr)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t id = dom ? dom[i] : i;
    f[i] = is_halo[id] && owner[id] == q;
}

__global__ __launch_bounds__(256) void k_plan_flag_send(int64_t n, const uint8_t *__restrict__ owner, int rank,
                                                        const uint32_t *__restrict__ needed_by, int q, int32_t *__restrict__ f)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) f[i] = owner[i] == rank && ((needed_by[i] >> q) & 1u);
}

// flagged i -> out[pos[i]] = (map ? map[i] : i); g2l[i] = base + pos[i] when g2l is given
__global__ __launch_bounds__(256) void k_plan_scatter(int64_t n, const int32_t *__restrict__ f, const int32_t *__restrict__ pos,
                                                      const int32_t *__restrict__ map, int32_t *__restrict__ out,
                                                      int32_t *__restrict__ g2l, int32_t base, const int32_t *__restrict__ dom = nullptr)
{
    // dom (slab-local assembly): entry i stands for the id dom[i] (the window's DOFs in brick-major order) instead of for i itself
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !f[i]) return;
    const int32_t id = dom ? dom[i] : (int32_t)i;
    out[pos[i]] = map ? map[i] : id;
    if (g2l) g2l[id] = base + pos[i];
}

__global__ __launch_bounds__(256) void k_plan_local_len(int64_t n_own, const int32_t *__restrict__ own_global,
                                                        const int32_t *__restrict__ row_ptr, int32_t *__restrict__ len)
{
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l > n_own) return;
    len[l] = l < n_own ? row_ptr[own_global[l] + 1] - row_ptr[own_global[l]] : 0;
}

// local rows: columns through global -> local, values (and value codes) copied, in-row order unchanged;
// tile_bnd[tile] = 1 when a row of the tile reads a halo column
__global__ __launch_bounds__(256) void k_plan_local_rows(int64_t n_own, const int32_t *__restrict__ own_global,
                                                         const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                         const double *__restrict__ val, const uint16_t *__restrict__ codes,
                                                         const int32_t *__restrict__ g2l, const int32_t *__restrict__ row_ptr_l,
                                                         int32_t *__restrict__ col_l, double *__restrict__ val_l,
                                                         uint16_t *__restrict__ codes_l, int tile_rows, int32_t *__restrict__ tile_bnd)
{
    const int sub = threadIdx.x & 15;
    const int64_t group = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int64_t ngroups = ((int64_t)gridDim.x * 256) >> 4;
    for (int64_t l = group; l < n_own; l += ngroups) {
        const int src = row_ptr[own_global[l]], dst = row_ptr_l[l], len = row_ptr_l[l + 1] - dst;
        bool touches = false;
        for (int k = sub; k < len; k += 16) {
            const int32_t lc = g2l[col[src + k]];
            col_l[dst + k] = lc;
            val_l[dst + k] = val[src + k];
            if (codes) codes_l[dst + k] = codes[src + k];
            touches |= lc >= n_own;
        }
        if (touches) tile_bnd[l / tile_rows] = 1;
    }
}

__global__ __launch_bounds__(256) void k_plan_not(int64_t n, const int32_t *__restrict__ f, int32_t *__restrict__ g)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) g[i] = !f[i];
}

static inline unsigned grid256(int64_t n) { return (unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1); }

// exclusive scan of 0/1 flags + their total (one small read-back)
static avs_status scan_flags(const int32_t *f, int32_t *pos, int64_t n, DevBuf<int32_t> &tmp, int64_t *total, hipStream_t st)
{
    *total = 0;
    if (n == 0) return AVS_OK;
    AVS_TRY(exclusive_scan_i32(f, pos, n, tmp.p, tmp.n, st));
    int32_t last[2] = {0, 0};
    AVS_HIP(hipMemcpyAsync(&last[0], pos + (n - 1), 4, hipMemcpyDeviceToHost, st));
    AVS_HIP(hipMemcpyAsync(&last[1], f + (n - 1), 4, hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    *total = (int64_t)last[0] + last[1];
    return AVS_OK;
}

// fills the plan and the local system of `d` from the context's (brick-major when `ro`) system
static avs_status plan_on_device(avs_ctx *c, PcgDist *d, int cut_axis, int extent, bool ro)
{
    hipStream_t st = c->stream;
    const int64_t n = c->n_vel;
    const int rank = d->rank, world = d->world;
    AVS_REQUIRE(world <= 32, AVS_EINVAL, "at most 32 ranks");
    const int32_t *g_rp = ro ? c->p_row_ptr.p : c->row_ptr.p, *g_col = ro ? c->p_col.p : c->col.p;
    const double *g_val = ro ? c->p_val.p : c->val.p;
    const bool vi = false; // the local rows get their own dictionaries afterwards (build_matrix_index)
    const int shift = c->desc.levels - 1;
    const int nplanes = (extent + (1 << shift) - 1) >> shift;
    AVS_REQUIRE(nplanes <= 65535, AVS_EINVAL, "too many cut planes");

    DevBuf<uint16_t> plane;
    DevBuf<unsigned long long> weight;
    DevBuf<int32_t> plane_owner, flag, pos, g2l, scan_tmp, halo_tmp;
    DevBuf<uint8_t> owner, is_halo;
    DevBuf<uint32_t> needed_by;
    AVS_TRY(plane.alloc((size_t)n));
    AVS_TRY(weight.alloc((size_t)nplanes));
    AVS_TRY(plane_owner.alloc((size_t)nplanes));
    AVS_TRY(flag.alloc((size_t)n + 1));
    AVS_TRY(pos.alloc((size_t)n + 1));
    AVS_TRY(g2l.alloc((size_t)n));
    AVS_TRY(scan_tmp.alloc(scan_tmp_elems(n + 1)));
    AVS_TRY(owner.alloc((size_t)n));
    AVS_TRY(is_halo.alloc((size_t)n));
    AVS_TRY(needed_by.alloc((size_t)n));
    AVS_HIP(hipMemsetAsync(weight.p, 0, (size_t)nplanes * sizeof(unsigned long long), st));
    AVS_HIP(hipMemsetAsync(is_halo.p, 0, (size_t)n, st));
    AVS_HIP(hipMemsetAsync(needed_by.p, 0, (size_t)n * sizeof(uint32_t), st));
    AVS_HIP(hipMemsetAsync(g2l.p, 0xFF, (size_t)n * sizeof(int32_t), st));

    // 1. slabs: per-plane weights -> (host, O(planes)) greedy cuts -> owner of every row
    hipLaunchKernelGGL(k_plan_planes, dim3(2048), dim3(256), 0, st, n, c->vdof.p, ro ? c->perm.p : nullptr, g_rp, cut_axis, extent,
                       shift, nplanes, plane.p, weight.p);
    std::vector<unsigned long long> h_w((size_t)nplanes);
    AVS_HIP(hipMemcpyAsync(h_w.data(), weight.p, h_w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> h_w64(h_w.begin(), h_w.end());
    std::vector<int> h_po((size_t)nplanes);
    plane_owners_from_weights(h_w64.data(), nplanes, world, h_po.data());
    AVS_HIP(hipMemcpyAsync(plane_owner.p, h_po.data(), h_po.size() * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_plan_owner, dim3(grid256(n)), dim3(256), 0, st, n, plane.p, plane_owner.p, rank, owner.p, flag.p);

    // 2. owned rows, ascending
    int64_t n_own = 0;
    AVS_TRY(scan_flags(flag.p, pos.p, n, scan_tmp, &n_own, st));
    AVS_TRY(d->own_global.alloc((size_t)n_own));
    hipLaunchKernelGGL(k_plan_scatter, dim3(grid256(n)), dim3(256), 0, st, n, flag.p, pos.p, (const int32_t *)nullptr, d->own_global.p,
                       g2l.p, 0);

    // 3. halo / needed-by marks
    hipLaunchKernelGGL(k_plan_mark, dim3(8192), dim3(256), 0, st, n, g_rp, g_col, owner.p, rank, is_halo.p, needed_by.p);

    // 4. halo numbering (grouped by owner) and send lists (ascending owned entries each peer reads)
    std::vector<int64_t> recv_cnt((size_t)world, 0), send_cnt((size_t)world, 0);
    AVS_TRY(halo_tmp.alloc((size_t)n)); // halo ids are only needed to number them; the list itself is not kept
    int64_t n_halo = 0;
    for (int q = 0; q < world; ++q) {
        if (q == rank) continue;
        hipLaunchKernelGGL(k_plan_flag_halo, dim3(grid256(n)), dim3(256), 0, st, n, is_halo.p, owner.p, q, flag.p);
        AVS_TRY(scan_flags(flag.p, pos.p, n, scan_tmp, &recv_cnt[(size_t)q], st));
        if (recv_cnt[(size_t)q])
            hipLaunchKernelGGL(k_plan_scatter, dim3(grid256(n)), dim3(256), 0, st, n, flag.p, pos.p, (const int32_t *)nullptr,
                               halo_tmp.p, g2l.p, (int32_t)(n_own + n_halo));
        n_halo += recv_cnt[(size_t)q];
    }
    int64_t n_send = 0;
    for (int q = 0; q < world; ++q) { // first pass: counts, so that send_idx can be allocated once
        if (q == rank) continue;
        hipLaunchKernelGGL(k_plan_flag_send, dim3(grid256(n)), dim3(256), 0, st, n, owner.p, rank, needed_by.p, q, flag.p);
        AVS_TRY(scan_flags(flag.p, pos.p, n, scan_tmp, &send_cnt[(size_t)q], st));
        n_send += send_cnt[(size_t)q];
    }
    AVS_TRY(d->send_idx.alloc((size_t)n_send));
    AVS_TRY(d->sendbuf.alloc((size_t)n_send));
    {
        int64_t off = 0;
        for (int q = 0; q < world; ++q) {
            if (q == rank || !send_cnt[(size_t)q]) continue;
            hipLaunchKernelGGL(k_plan_flag_send, dim3(grid256(n)), dim3(256), 0, st, n, owner.p, rank, needed_by.p, q, flag.p);
            AVS_TRY(exclusive_scan_i32(flag.p, pos.p, n, scan_tmp.p, scan_tmp.n, st));
            hipLaunchKernelGGL(k_plan_scatter, dim3(grid256(n)), dim3(256), 0, st, n, flag.p, pos.p, (const int32_t *)g2l.p,
                               d->send_idx.p + off, (int32_t *)nullptr, 0);
            off += send_cnt[(size_t)q];
        }
    }
    d->peers.clear();
    d->send_counts.clear();
    d->recv_counts.clear();
    for (int q = 0; q < world; ++q)
        if (q != rank && (send_cnt[(size_t)q] || recv_cnt[(size_t)q])) {
            d->peers.push_back(q);
            d->send_counts.push_back((int32_t)send_cnt[(size_t)q]);
            d->recv_counts.push_back((int32_t)recv_cnt[(size_t)q]);
        }

    // 5. local CSR
    AVS_TRY(d->row_ptr.alloc((size_t)n_own + 1));
    hipLaunchKernelGGL(k_plan_local_len, dim3(grid256(n_own + 1)), dim3(256), 0, st, n_own, d->own_global.p, g_rp, flag.p);
    AVS_TRY(exclusive_scan_i32(flag.p, d->row_ptr.p, n_own + 1, scan_tmp.p, scan_tmp.n, st));
    int32_t nnz_local = 0;
    AVS_HIP(hipMemcpyAsync(&nnz_local, d->row_ptr.p + n_own, 4, hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    AVS_TRY(d->col.alloc((size_t)nnz_local));
    AVS_TRY(d->val.alloc((size_t)nnz_local));
    const int T = spmv_tile_rows();
    const int64_t ntiles = (n_own + T - 1) / T;
    DevBuf<int32_t> tile_bnd, tile_int, tile_pos;
    AVS_TRY(tile_bnd.alloc((size_t)ntiles + 1));
    AVS_TRY(tile_int.alloc((size_t)ntiles + 1));
    AVS_TRY(tile_pos.alloc((size_t)ntiles + 1));
    AVS_HIP(hipMemsetAsync(tile_bnd.p, 0, ((size_t)ntiles + 1) * 4, st));
    if (n_own)
        hipLaunchKernelGGL(k_plan_local_rows, dim3(8192), dim3(256), 0, st, n_own, d->own_global.p, g_rp, g_col, g_val,
                           (const uint16_t *)nullptr, g2l.p, d->row_ptr.p, d->col.p, d->val.p, (uint16_t *)nullptr, T, tile_bnd.p);
    (void)vi;

    // 6. tile lists for the overlap of the exchange with the interior rows
    int64_t n_bnd = 0, n_int = 0;
    AVS_TRY(scan_flags(tile_bnd.p, tile_pos.p, ntiles, scan_tmp, &n_bnd, st));
    AVS_TRY(d->tiles_bnd.alloc((size_t)n_bnd));
    if (n_bnd)
        hipLaunchKernelGGL(k_plan_scatter, dim3(grid256(ntiles)), dim3(256), 0, st, ntiles, tile_bnd.p, tile_pos.p, (const int32_t *)nullptr,
                           d->tiles_bnd.p, (int32_t *)nullptr, 0);
    if (ntiles) hipLaunchKernelGGL(k_plan_not, dim3(grid256(ntiles)), dim3(256), 0, st, ntiles, tile_bnd.p, tile_int.p);
    AVS_TRY(scan_flags(tile_int.p, tile_pos.p, ntiles, scan_tmp, &n_int, st));
    AVS_TRY(d->tiles_int.alloc((size_t)n_int));
    if (n_int)
        hipLaunchKernelGGL(k_plan_scatter, dim3(grid256(ntiles)), dim3(256), 0, st, ntiles, tile_int.p, tile_pos.p, (const int32_t *)nullptr,
                           d->tiles_int.p, (int32_t *)nullptr, 0);
    AVS_HIP(hipGetLastError());
    AVS_HIP(hipStreamSynchronize(st)); // temporaries die here
    d->n_tiles_int = (int)n_int;
    d->n_tiles_bnd = (int)n_bnd;
    d->n_own = n_own;
    d->n_halo = n_halo;
    d->n_send = n_send;
    d->nnz_local = nnz_local;
    return AVS_OK;
}


// ---------------------------------------------------------------------------------------------
// Distributed assembly ("each GPU assembles the rows it owns").  Nothing global is built except what
// is cheap and index-only: dof tables, stencils, restriction, the raw per-row triplet counts (slab weights) and the
// brick-major permutation.  Each rank then assembles ITS rows (assemble_rows), renumbers their columns
// reference id -> brick-major id -> local [owned | halo by owner], and derives its send lists from its own rows:
// the matrix pattern is symmetric (every stress contributes d d^T), so "rank q reads my DOF i" <=> "my row i reads
// a DOF of q".
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_da_planes(int64_t n, const int32_t *__restrict__ vdof, const int32_t *__restrict__ perm,
                                                   const int32_t *__restrict__ raw_count, int axis, int extent, int shift, int nplanes,
                                                   uint16_t *__restrict__ plane, unsigned long long *__restrict__ weight)
{
    __shared__ unsigned long long h[kPlanLdsPlanes];
    const bool lds = nplanes <= kPlanLdsPlanes;
    if (lds) {
        for (int i = threadIdx.x; i < nplanes; i += 256) h[i] = 0ull;
        __syncthreads();
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int32_t dof = perm[i];
        const int pl = plane_of_dof(vdof, dof, axis, extent, shift);
        plane[i] = (uint16_t)pl;
        atomicAdd(lds ? &h[pl] : &weight[pl], (unsigned long long)raw_count[dof] + 2ull);
    }
    if (lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < nplanes; i += 256)
            if (h[i]) atomicAdd(&weight[i], h[i]);
    }
}

// owned rows: columns reference id -> brick-major id (in place); marks halo columns and, per local row, the ranks
// whose DOFs it reads (= the ranks that read this row's DOF)
__global__ __launch_bounds__(256) void k_da_mark(int64_t n_own, const int32_t *__restrict__ row_ptr, int32_t *__restrict__ col,
                                                 const int32_t *__restrict__ inv, const uint8_t *__restrict__ owner, int rank,
                                                 uint8_t *__restrict__ is_halo, uint32_t *__restrict__ needed_by, int *__restrict__ err = nullptr)
{
    // inv == null (slab-local assembly): ids stay the reference's; a column without an owner lies outside the rank's window (err)
    const int sub = threadIdx.x & 15;
    const int64_t group = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int64_t ngroups = ((int64_t)gridDim.x * 256) >> 4;
    for (int64_t l = group; l < n_own; l += ngroups) {
        uint32_t bits = 0u;
        for (int k = row_ptr[l] + sub; k < row_ptr[l + 1]; k += 16) {
            const int32_t cb = inv ? inv[col[k]] : col[k];
            col[k] = cb;
            const int q = owner[cb];
            if (q == 0xFF) {
                if (err) *err = 1;
            } else if (q != rank) {
                is_halo[cb] = 1; // same value from every writer
                bits |= 1u << q;
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) bits |= __shfl_xor(bits, o, 16);
        if (sub == 0) needed_by[l] = bits;
    }
}

// brick-major column ids -> local ids; tile_bnd[tile] = 1 when a row of the tile reads a halo column
__global__ __launch_bounds__(256) void k_da_localize(int64_t n_own, const int32_t *__restrict__ row_ptr, int32_t *__restrict__ col,
                                                     const int32_t *__restrict__ g2l, int tile_rows, int32_t *__restrict__ tile_bnd)
{
    const int sub = threadIdx.x & 15;
    const int64_t group = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int64_t ngroups = ((int64_t)gridDim.x * 256) >> 4;
    for (int64_t l = group; l < n_own; l += ngroups) {
        bool touches = false;
        for (int k = row_ptr[l] + sub; k < row_ptr[l + 1]; k += 16) {
            const int32_t lc = g2l[col[k]];
            col[k] = lc;
            touches |= lc >= n_own;
        }
        if (touches) tile_bnd[l / tile_rows] = 1;
    }
}

// reference DOF id of every halo column: global brick-major id g with a local id >= n_own -> perm[g]
__global__ __launch_bounds__(256) void k_da_halo_ref(int64_t n, const int32_t *__restrict__ g2l, const int32_t *__restrict__ perm, int64_t n_own,
                                                     int32_t *__restrict__ local_ref, const int32_t *__restrict__ dom = nullptr)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t g = dom ? dom[i] : (int32_t)i;
    const int32_t l = g2l[g];
    if (l >= n_own) local_ref[l] = perm ? perm[g] : g;
}

// the ranks this rank exchanges with: the union of the rows' needed_by bits (the pattern is symmetric: whom I read from reads from me)
__global__ __launch_bounds__(256) void k_da_peer_mask(int64_t n_own, const uint32_t *__restrict__ needed_by, uint32_t *__restrict__ mask)
{
    uint32_t m = 0u;
    for (int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x; l < n_own; l += (int64_t)gridDim.x * 256) m |= needed_by[l];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o, 64);
    if ((threadIdx.x & 63) == 0 && m) atomicOr(mask, m);
}

__global__ __launch_bounds__(256) void k_da_flag_send(int64_t n_own, const uint32_t *__restrict__ needed_by, int q, int32_t *__restrict__ f)
{
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l < n_own) f[l] = q < 0 ? (needed_by[l] == 0u) : ((needed_by[l] >> q) & 1u); // q < 0: rows that read no other rank
}

// local row order [interior | rows that read another rank's DOFs], both parts in ascending brick-major order (stable): the
// halo-touching SpMV tiles shrink from every tile that contains a surface brick (25-57 % of a slab's tiles) to the rows next to
// the cuts (a few per cent), everything else multiplies while the halo travels.  pos_int = exclusive scan of (needed_by == 0).
__global__ __launch_bounds__(256) void k_da_new_index(int64_t n_own, const uint32_t *__restrict__ needed_by, const int32_t *__restrict__ pos_int,
                                                      int64_t n_interior, const int32_t *__restrict__ row_ptr, int32_t *__restrict__ new_of,
                                                      int32_t *__restrict__ len_new)
{
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l > n_own) return;
    if (l == n_own) { len_new[n_own] = 0; return; }
    const int64_t nw = needed_by[l] == 0 ? (int64_t)pos_int[l] : n_interior + (l - pos_int[l]);
    new_of[l] = (int32_t)nw;
    len_new[nw] = row_ptr[l + 1] - row_ptr[l];
}

__global__ __launch_bounds__(256) void k_da_move_rows(int64_t n_own, const int32_t *__restrict__ new_of, const int32_t *__restrict__ row_ptr,
                                                      const int32_t *__restrict__ col, const double *__restrict__ val,
                                                      const int32_t *__restrict__ row_ptr_new, int32_t *__restrict__ col_new,
                                                      double *__restrict__ val_new)
{
    const int sub = threadIdx.x & 15;
    const int64_t group = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int64_t ngroups = ((int64_t)gridDim.x * 256) >> 4;
    for (int64_t l = group; l < n_own; l += ngroups) {
        const int src = row_ptr[l], dst = row_ptr_new[new_of[l]], len = row_ptr[l + 1] - src;
        for (int k = sub; k < len; k += 16) {
            col_new[dst + k] = col[src + k];
            val_new[dst + k] = val[src + k];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_da_move(int64_t n, const int32_t *__restrict__ new_of, const T *__restrict__ src, T *__restrict__ dst)
{
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l < n) dst[new_of[l]] = src[l];
}

__global__ __launch_bounds__(256) void k_da_g2l_own(int64_t n_own, const int32_t *__restrict__ own_global, int32_t *__restrict__ g2l)
{
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l < n_own) g2l[own_global[l]] = (int32_t)l;
}

// the brick-structured form for a rank's local rows: same rule as the single-GPU system (avs_reorder.hip)
static bool dist_brick_wanted(const avs_ctx *c, int64_t n_own)
{
    const int mode = c->opt.brick;
    if (c->brick_shift != 3) return false;
    return mode == 1 || (mode != 0 && n_own >= kBrickMinSystemRows);
}
static avs_status dist_build_brick(avs_ctx *c, PcgDist *d)
{
    d->brick.clear();
    d->brick.view(d->brick_view, d->vi);
    if (!dist_brick_wanted(c, d->n_own) || !d->local_ref.p || d->local_ref.n < (size_t)(d->n_own + d->n_halo)) return AVS_OK;
    BrickSource src;
    src.n_rows = d->n_own;
    src.n_cols = d->n_own + d->n_halo;
    src.nnz = d->nnz_local;
    src.row_ptr = d->row_ptr.p;
    src.col = d->col.p;
    src.vi = &d->vi;
    src.val = d->val.p;
    src.vdof = c->vdof.p;
    src.ref_id = d->local_ref.p;
    src.nx = c->desc.nx; src.ny = c->desc.ny; src.nz = c->desc.nz;
    src.levels = c->desc.levels;
    src.brick_shift = c->brick_shift;
    AVS_TRY(build_brick_form(d->brick, src, c->opt, c->stream));
    if (d->brick.ready) {
        const double fill = (double)d->n_own / (double)d->brick.ntiles;
        if (c->opt.brick != 1 && fill < kBrickMinFill) d->brick.release(); // (auto: the word stream serves part-filled tiles better)
        else {
            d->brick.view(d->brick_view, d->vi);
            d->brick_view.walk = fill >= kBrickEighthsFill ? 0 : 1;
            if (c->opt.brick_plan) {
                const int walk = d->brick_view.walk;
                AVS_TRY(d->brick.plan_walk(brick_partial_count(d->brick_view), walk, c->opt.brick_cost, c->stream));
                d->brick.view(d->brick_view, d->vi);
                d->brick_view.walk = walk;
            }
            return AVS_OK;
        }
    } else if (c->opt.brick != 1) {
        d->brick.release();
    }
    d->brick.view(d->brick_view, d->vi);
    return AVS_OK;
}

// Everything behind the choice of the owned rows, shared by the replicated-index assembly above and the slab-local one below: the rank's
// rows, halo numbering, send lists, local column ids, tile lists.  Ids live in an id space of `n_ids` entries (brick-major ids / the
// reference's ids); `dom` (null: 0 .. n_dom) lists, in brick-major order, the ids a halo column can have; owner / is_halo / g2l are
// indexed by id.  inv (reference id -> id, null: identity), perm (id -> reference id, null: identity).
static avs_status dist_assemble_tail(avs_ctx *c, PcgDist *d, DevBuf<int32_t> &ids, int64_t n_own, int64_t n_ids, const int32_t *dom, int64_t n_dom,
                                     const int32_t *inv, const int32_t *perm, DevBuf<uint8_t> &owner, DevBuf<uint8_t> &is_halo, DevBuf<int32_t> &g2l,
                                     DevBuf<int32_t> &flag, DevBuf<int32_t> &pos, DevBuf<int32_t> &scan_tmp)
{
    hipStream_t st = c->stream;
    const int rank = d->rank, world = d->world;
    const int64_t n = n_dom;
    DevBuf<int32_t> &halo_tmp = d->ws.halo_tmp;
    DevBuf<uint32_t> &needed_by = d->ws.needed_by;
    DevBuf<int> &mark_err = d->ws.mark_err;
    AVS_TRY(mark_err.reserve(1));
    AVS_HIP(hipMemsetAsync(mark_err.p, 0, sizeof(int), st));
    PhaseTrace tr(st, "tail", cur_opt().trace_phases != 0);
    // restriction (warm start, also the mass term of the right-hand side) and rows of this rank only;
    // columns still in the reference numbering
    AVS_TRY(build_initial_guess_rows(c, ids.p, n_own));
    tr.mark("initial guess rows");
    int64_t nnz_local = 0;
    AVS_TRY(assemble_rows(c, ids.p, n_own, d->row_ptr, d->col, d->val, d->rhs, &nnz_local, nullptr));
    tr.mark("assemble_rows");
    AVS_TRY(needed_by.reserve((size_t)n_own));
    if (n_own) hipLaunchKernelGGL(k_da_mark, dim3(8192), dim3(256), 0, st, n_own, d->row_ptr.p, d->col.p, inv, owner.p, rank, is_halo.p,
                                  needed_by.p, mark_err.p);
    uint32_t peer_mask = 0u; // (a slab talks to its neighbours: the per-rank scans below run for them only -- each is a scan + a host round trip)
    {
        DevBuf<uint32_t> &pm = d->ws.pm;
        AVS_TRY(pm.reserve(1));
        AVS_HIP(hipMemsetAsync(pm.p, 0, sizeof(uint32_t), st));
        if (n_own) hipLaunchKernelGGL(k_da_peer_mask, dim3(1024), dim3(256), 0, st, n_own, (const uint32_t *)needed_by.p, pm.p);
        int e = 0;
        AVS_HIP(hipMemcpyAsync(&e, mark_err.p, sizeof(int), hipMemcpyDeviceToHost, st));
        AVS_HIP(hipMemcpyAsync(&peer_mask, pm.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        AVS_HIP(hipStreamSynchronize(st));
        AVS_REQUIRE(e == 0, AVS_EINTERNAL, "slab-local assembly: a row reads a column outside the rank's window (index margin too small)");
    }
    tr.mark("mark columns");
    bool split = world > 1;
    split = split && cur_opt().dist_split_rows != 0;
    // A slab the brick-structured form will serve keeps the plain ascending (brick-major) row order: its tiles are whole bricks, and the
    // ones that read halo columns are moved to the end of the WALK order instead (build_brick_form), which keeps them full.  With the
    // [interior | halo-reading] split every brick next to a cut would fall into two part-filled tiles.
    if (dist_brick_wanted(c, n_own)) split = false;
    if (split && n_own) { // [interior | halo-reading] local row order (see k_da_new_index)
        DevBuf<int32_t> &new_of = d->alt_new_of, &len_new = d->alt_len, &rp2 = d->alt_row_ptr, &col2 = d->alt_col, &own2 = d->alt_own, &ids2 = d->alt_ids;
        DevBuf<double> &val2 = d->alt_val, &rhs2 = d->alt_rhs;
        DevBuf<uint32_t> &nb2 = d->alt_nb;
        AVS_TRY(new_of.reserve((size_t)n_own));
        AVS_TRY(len_new.reserve((size_t)n_own + 1));
        AVS_TRY(rp2.reserve((size_t)n_own + 1));
        AVS_TRY(col2.reserve((size_t)nnz_local));
        AVS_TRY(val2.reserve((size_t)nnz_local));
        AVS_TRY(own2.reserve((size_t)n_own));
        AVS_TRY(ids2.reserve((size_t)n_own));
        AVS_TRY(rhs2.reserve((size_t)n_own));
        AVS_TRY(nb2.reserve((size_t)n_own));
        hipLaunchKernelGGL(k_da_flag_send, dim3(grid256(n_own)), dim3(256), 0, st, n_own, (const uint32_t *)needed_by.p, -1, flag.p); // q = -1: "reads nobody"
        int64_t n_interior = 0;
        AVS_TRY(scan_flags(flag.p, pos.p, n_own, scan_tmp, &n_interior, st));
        hipLaunchKernelGGL(k_da_new_index, dim3(grid256(n_own + 1)), dim3(256), 0, st, n_own, (const uint32_t *)needed_by.p, (const int32_t *)pos.p,
                           n_interior, (const int32_t *)d->row_ptr.p, new_of.p, len_new.p);
        AVS_TRY(exclusive_scan_i32(len_new.p, rp2.p, n_own, scan_tmp.p, scan_tmp.n, st));
        hipLaunchKernelGGL(k_da_move_rows, dim3(8192), dim3(256), 0, st, n_own, (const int32_t *)new_of.p, (const int32_t *)d->row_ptr.p,
                           (const int32_t *)d->col.p, (const double *)d->val.p, (const int32_t *)rp2.p, col2.p, val2.p);
        const unsigned g1 = grid256(n_own);
        hipLaunchKernelGGL(k_da_move<int32_t>, dim3(g1), dim3(256), 0, st, n_own, (const int32_t *)new_of.p, (const int32_t *)d->own_global.p, own2.p);
        hipLaunchKernelGGL(k_da_move<int32_t>, dim3(g1), dim3(256), 0, st, n_own, (const int32_t *)new_of.p, (const int32_t *)ids.p, ids2.p);
        hipLaunchKernelGGL(k_da_move<double>, dim3(g1), dim3(256), 0, st, n_own, (const int32_t *)new_of.p, (const double *)d->rhs.p, rhs2.p);
        hipLaunchKernelGGL(k_da_move<uint32_t>, dim3(g1), dim3(256), 0, st, n_own, (const int32_t *)new_of.p, (const uint32_t *)needed_by.p, nb2.p);
        hipLaunchKernelGGL(k_da_g2l_own, dim3(g1), dim3(256), 0, st, n_own, (const int32_t *)own2.p, g2l.p);
        AVS_HIP(hipGetLastError());
        AVS_HIP(hipStreamSynchronize(st));
        auto swap_i = [](DevBuf<int32_t> &a, DevBuf<int32_t> &b) { std::swap(a.p, b.p); std::swap(a.n, b.n); };
        swap_i(d->row_ptr, rp2);
        swap_i(d->col, col2);
        swap_i(d->own_global, own2);
        swap_i(ids, ids2);
        std::swap(d->val.p, val2.p); std::swap(d->val.n, val2.n);
        std::swap(d->rhs.p, rhs2.p); std::swap(d->rhs.n, rhs2.n);
        std::swap(needed_by.p, nb2.p); std::swap(needed_by.n, nb2.n);
    }

    tr.mark("row order split");
    // halo numbering (grouped by owner, ascending) and send lists (ascending owned rows that read a DOF of q)
    std::vector<int64_t> recv_cnt((size_t)world, 0), send_cnt((size_t)world, 0);
    AVS_TRY(halo_tmp.reserve((size_t)(n > 0 ? n : 1)));
    int64_t n_halo = 0, n_send = 0;
    for (int q = 0; q < world; ++q) {
        if (q == rank || !((peer_mask >> q) & 1u)) continue;
        hipLaunchKernelGGL(k_plan_flag_halo, dim3(grid256(n)), dim3(256), 0, st, n, is_halo.p, owner.p, q, flag.p, dom);
        AVS_TRY(scan_flags(flag.p, pos.p, n, scan_tmp, &recv_cnt[(size_t)q], st));
        if (recv_cnt[(size_t)q])
            hipLaunchKernelGGL(k_plan_scatter, dim3(grid256(n)), dim3(256), 0, st, n, flag.p, pos.p, (const int32_t *)nullptr,
                               halo_tmp.p, g2l.p, (int32_t)(n_own + n_halo), dom);
        n_halo += recv_cnt[(size_t)q];
    }
    for (int q = 0; q < world; ++q) {
        if (q == rank || !n_own || !((peer_mask >> q) & 1u)) continue;
        hipLaunchKernelGGL(k_da_flag_send, dim3(grid256(n_own)), dim3(256), 0, st, n_own, needed_by.p, q, flag.p);
        AVS_TRY(scan_flags(flag.p, pos.p, n_own, scan_tmp, &send_cnt[(size_t)q], st));
        n_send += send_cnt[(size_t)q];
    }
    AVS_TRY(d->send_idx.reserve((size_t)n_send));
    AVS_TRY(d->sendbuf.reserve((size_t)n_send));
    {
        int64_t off = 0;
        for (int q = 0; q < world; ++q) {
            if (q == rank || !send_cnt[(size_t)q]) continue;
            hipLaunchKernelGGL(k_da_flag_send, dim3(grid256(n_own)), dim3(256), 0, st, n_own, needed_by.p, q, flag.p);
            AVS_TRY(exclusive_scan_i32(flag.p, pos.p, n_own, scan_tmp.p, scan_tmp.n, st));
            hipLaunchKernelGGL(k_plan_scatter, dim3(grid256(n_own)), dim3(256), 0, st, n_own, flag.p, pos.p, (const int32_t *)nullptr,
                               d->send_idx.p + off, (int32_t *)nullptr, 0); // local row index == local DOF index
            off += send_cnt[(size_t)q];
        }
    }
    d->peers.clear();
    d->send_counts.clear();
    d->recv_counts.clear();
    for (int q = 0; q < world; ++q)
        if (q != rank && (send_cnt[(size_t)q] || recv_cnt[(size_t)q])) {
            d->peers.push_back(q);
            d->send_counts.push_back((int32_t)send_cnt[(size_t)q]);
            d->recv_counts.push_back((int32_t)recv_cnt[(size_t)q]);
        }

    tr.mark("halo + send lists");
    // local column ids + tile lists
    const int T = spmv_tile_rows();
    const int64_t ntiles = (n_own + T - 1) / T;
    DevBuf<int32_t> &tile_bnd = d->ws.tile_bnd, &tile_int = d->ws.tile_int, &tile_pos = d->ws.tile_pos;
    AVS_TRY(tile_bnd.reserve((size_t)ntiles + 1));
    AVS_TRY(tile_int.reserve((size_t)ntiles + 1));
    AVS_TRY(tile_pos.reserve((size_t)ntiles + 1));
    AVS_HIP(hipMemsetAsync(tile_bnd.p, 0, ((size_t)ntiles + 1) * 4, st));
    if (n_own) hipLaunchKernelGGL(k_da_localize, dim3(8192), dim3(256), 0, st, n_own, d->row_ptr.p, d->col.p, g2l.p, T, tile_bnd.p);
    int64_t n_bnd = 0, n_int = 0;
    AVS_TRY(scan_flags(tile_bnd.p, tile_pos.p, ntiles, scan_tmp, &n_bnd, st));
    AVS_TRY(d->tiles_bnd.reserve((size_t)n_bnd));
    if (n_bnd)
        hipLaunchKernelGGL(k_plan_scatter, dim3(grid256(ntiles)), dim3(256), 0, st, ntiles, tile_bnd.p, tile_pos.p, (const int32_t *)nullptr,
                           d->tiles_bnd.p, (int32_t *)nullptr, 0);
    if (ntiles) hipLaunchKernelGGL(k_plan_not, dim3(grid256(ntiles)), dim3(256), 0, st, ntiles, tile_bnd.p, tile_int.p);
    AVS_TRY(scan_flags(tile_int.p, tile_pos.p, ntiles, scan_tmp, &n_int, st));
    AVS_TRY(d->tiles_int.reserve((size_t)n_int));
    if (n_int)
        hipLaunchKernelGGL(k_plan_scatter, dim3(grid256(ntiles)), dim3(256), 0, st, ntiles, tile_int.p, tile_pos.p, (const int32_t *)nullptr,
                           d->tiles_int.p, (int32_t *)nullptr, 0);

    // reference ids behind the local columns (owned: the assembled rows' DOFs; halo: through the brick-major permutation)
    AVS_TRY(d->local_ref.reserve((size_t)(n_own + n_halo)));
    if (n_own) AVS_HIP(hipMemcpyAsync(d->local_ref.p, ids.p, (size_t)n_own * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (n_halo) hipLaunchKernelGGL(k_da_halo_ref, dim3(grid256(n)), dim3(256), 0, st, n, (const int32_t *)g2l.p, perm, n_own, d->local_ref.p, dom);
    // warm start of the owned DOFs
    AVS_TRY(d->x0.reserve((size_t)n_own));
    AVS_TRY(d->x.reserve((size_t)n_own));
    if (n_own) hipLaunchKernelGGL(k_gather_i<double>, dim3(grid256(n_own)), dim3(256), 0, st, c->x0.p, ids.p, d->x0.p, n_own);
    AVS_HIP(hipGetLastError());
    AVS_HIP(hipStreamSynchronize(st)); // temporaries die here
    tr.mark("localize, tiles, x0");
    d->n_tiles_int = (int)n_int;
    d->n_tiles_bnd = (int)n_bnd;
    d->n_own = n_own;
    d->n_halo = n_halo;
    d->n_send = n_send;
    d->nnz_local = nnz_local;
    return AVS_OK;
}


// cuts[r] = first fine cell of rank r's slab (cuts[world] = extent) from the planes' owners (non-decreasing)
static void cuts_from_plane_owners(const std::vector<int> &po, int world, int shift, int extent, std::vector<int32_t> &cuts)
{
    cuts.assign((size_t)world + 1, extent);
    cuts[0] = 0;
    for (int r = 1; r < world; ++r) {
        int p = 0;
        while (p < (int)po.size() && po[(size_t)p] < r) ++p;
        const long long c = (long long)p << shift;
        cuts[(size_t)r] = (int32_t)(c < extent ? c : extent);
    }
}

// ---------------------------------------------------------------------------------------------
// Slab-local assembly (round 6; SURVEY 8(e): "each GPU assembles the rows it owns from its slab of the pyramids plus a halo").
// The context holds what a slab-local pre-pass lent it (avs_prepass_set_slab): lattices and dof tables valid inside the rank's window,
// GLOBAL reference ids, the ids of the window's DOFs.  Nothing here is sized or swept by the global DOF count except three
// lookup arrays indexed by reference id (owner, halo flag, local id: 6 B per DOF, filled by memset) -- the sweeps run over the window's
// DOFs (brick-major: the stable sort of their brick keys, the same relative order as the global permutation's), the rows are the
// reference's rows bit for bit, and the local order, halo order and send lists equal the replicated-index assembly's for the same cuts.
// ---------------------------------------------------------------------------------------------
struct CutTable {
    int world;
    int cuts[kMaxRanks + 1];
};
__global__ __launch_bounds__(256) void k_win_keys(const int32_t *__restrict__ vdof, const int32_t *__restrict__ wl, int64_t m, int nx, int ny, int nz,
                                                  int shift, int interleave, uint32_t *__restrict__ keys)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int4 rec = reinterpret_cast<const int4 *>(vdof)[wl[i]]; // (the key of avs_reorder.hip's k_brick_keys)
    const int level = rec.x & 0xff;
    int px = rec.y << level, py = rec.z << level, pz = rec.w << level;
    px = px < nx ? px : nx - 1;
    py = py < ny ? py : ny - 1;
    pz = pz < nz ? pz : nz - 1;
    const uint32_t bx = (uint32_t)(px >> shift), by = (uint32_t)(py >> shift), bz = (uint32_t)(pz >> shift);
    const uint32_t nbx = (uint32_t)((nx + (1 << shift) - 1) >> shift), nby = (uint32_t)((ny + (1 << shift) - 1) >> shift);
    uint32_t key = (bz * nby + by) * nbx + bx;
    if (interleave) {
        const uint32_t mk = (1u << shift) - 1u;
        key = (key << (3 * shift)) | ((((uint32_t)pz & mk) << (2 * shift)) | (((uint32_t)py & mk) << shift) | ((uint32_t)px & mk));
    }
    keys[i] = key;
}
// owner of every DOF of the window (by the position of its face along the cut axis), own flags
__global__ __launch_bounds__(256) void k_win_owner(int64_t m, const int32_t *__restrict__ vdof, const int32_t *__restrict__ dom, int axis, int extent,
                                                   CutTable T, int rank, uint8_t *__restrict__ owner_ref, int32_t *__restrict__ own_flag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int32_t id = dom[i];
    const int32_t *t = vdof + 4 * (int64_t)id;
    const int level = t[0] & 0xff;
    long long pos = (long long)t[1 + axis] << level;
    if (pos >= extent) pos = extent - 1;
    int r = 0;
    while (r + 1 < T.world && pos >= T.cuts[r + 1]) ++r;
    owner_ref[id] = (uint8_t)r;
    own_flag[i] = r == rank;
}
// per-plane weights of the rank's rows (non-zeros + 2, as the host planner), in units of 16 so that the sum over
// a plane of a 2048^2 cross-section fits the int32 the ranks exchange
__global__ __launch_bounds__(256) void k_win_plane_weights(int64_t n_own, const int32_t *__restrict__ ids, const int32_t *__restrict__ vdof,
                                                           const int32_t *__restrict__ row_ptr, int axis, int extent, int shift,
                                                           unsigned long long *__restrict__ weight)
{
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= n_own) return;
    atomicAdd(&weight[plane_of_dof(vdof, ids[l], axis, extent, shift)], (unsigned long long)(row_ptr[l + 1] - row_ptr[l]) + 2ull);
}
__global__ __launch_bounds__(256) void k_win_weights_pack(int n, const unsigned long long *__restrict__ w, int32_t *__restrict__ out)
{
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < n) out[i] = (int32_t)((w[i] + 15ull) >> 4);
}

avs_status dist_allreduce_i32(PcgDist *d, int32_t *dev, int64_t count, hipStream_t stream); // below

static avs_status dist_assemble_window(avs_ctx *c, PcgDist *d)
{
    hipStream_t st = c->stream;
    const SlabWindow &W = c->slab;
    const int rank = d->rank, world = d->world;
    AVS_REQUIRE(W.world == world && W.rank == rank, AVS_ESTATE, "the slab of the pre-pass (rank %d of %d) is not this context's rank (%d of %d)", W.rank,
                W.world, rank, world);
    AVS_REQUIRE(c->tables_ready && c->wlist[0].p, AVS_ESTATE, "slab-local context without dof tables / window lists");
    const int64_t n = c->n_vel, n_w = c->n_window[0];
    const int axis = W.axis;
    const int extent = axis == 0 ? c->desc.nx : (axis == 1 ? c->desc.ny : c->desc.nz);
    d->cuts.assign(W.cuts, W.cuts + world + 1);
    d->cut_axis = axis;
    PhaseTrace tr(st, "window", cur_opt().trace_phases != 0);

    // the window's velocity DOFs in brick-major order
    DevBuf<uint32_t> &keys_in = d->ws.keys_in, &keys_out = d->ws.keys_out;
    DevBuf<int32_t> &dom = d->ws.dom, &flag = d->ws.flag, &pos = d->ws.pos, &g2l = d->ws.g2l, &scan_tmp = d->ws.scan_tmp, &ids = d->ws.ids;
    DevBuf<uint8_t> &owner = d->ws.owner, &is_halo = d->ws.is_halo;
    DevBuf<char> &sort_tmp = d->ws.sort_tmp;
    AVS_TRY(keys_in.reserve((size_t)(n_w > 0 ? n_w : 1)));
    AVS_TRY(keys_out.reserve((size_t)(n_w > 0 ? n_w : 1)));
    AVS_TRY(dom.reserve((size_t)(n_w > 0 ? n_w : 1)));
    AVS_TRY(flag.reserve((size_t)n_w + 1));
    AVS_TRY(pos.reserve((size_t)n_w + 1));
    AVS_TRY(scan_tmp.reserve(scan_tmp_elems(n_w + 1)));
    AVS_TRY(g2l.reserve((size_t)(n > 0 ? n : 1)));
    AVS_TRY(owner.reserve((size_t)(n > 0 ? n : 1)));
    AVS_TRY(is_halo.reserve((size_t)(n > 0 ? n : 1)));
    AVS_HIP(hipMemsetAsync(g2l.p, 0xFF, (size_t)n * sizeof(int32_t), st));
    AVS_HIP(hipMemsetAsync(owner.p, 0xFF, (size_t)n, st));
    AVS_HIP(hipMemsetAsync(is_halo.p, 0, (size_t)n, st));
    tr.mark("alloc + memsets");
    int interleave = cur_opt().brick_interleave;
    {
        const int bs = c->brick_shift;
        const uint64_t nb = (uint64_t)((c->desc.nx >> bs) + 1) * ((c->desc.ny >> bs) + 1) * ((c->desc.nz >> bs) + 1);
        if ((nb << (3 * bs)) >= (1ull << 32)) interleave = 0;
    }
    if (n_w) {
        hipLaunchKernelGGL(k_win_keys, dim3(grid256(n_w)), dim3(256), 0, st, c->vdof.p, c->wlist[0].p, n_w, c->desc.nx, c->desc.ny, c->desc.nz,
                           c->brick_shift, interleave, keys_in.p);
        size_t tmp_bytes = 0;
        AVS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in.p, keys_out.p, c->wlist[0].p, dom.p, (size_t)n_w, 0, 32, st));
        AVS_TRY(sort_tmp.reserve(tmp_bytes > 0 ? tmp_bytes : 1));
        AVS_HIP(rocprim::radix_sort_pairs(sort_tmp.p, tmp_bytes, keys_in.p, keys_out.p, c->wlist[0].p, dom.p, (size_t)n_w, 0, 32, st)); // stable
        CutTable T{};
        T.world = world;
        for (int r = 0; r <= world; ++r) T.cuts[r] = W.cuts[r];
        hipLaunchKernelGGL(k_win_owner, dim3(grid256(n_w)), dim3(256), 0, st, n_w, c->vdof.p, dom.p, axis, extent, T, rank, owner.p, flag.p);
    }
    tr.mark("keys, sort, owners");
    int64_t n_own = 0;
    AVS_TRY(scan_flags(flag.p, pos.p, n_w, scan_tmp, &n_own, st));
    AVS_TRY(ids.reserve((size_t)(n_own > 0 ? n_own : 1)));
    AVS_TRY(d->own_global.reserve((size_t)n_own));
    if (n_own) {
        hipLaunchKernelGGL(k_plan_scatter, dim3(grid256(n_w)), dim3(256), 0, st, n_w, flag.p, pos.p, (const int32_t *)nullptr, ids.p, g2l.p, 0, dom.p);
        AVS_HIP(hipMemcpyAsync(d->own_global.p, ids.p, (size_t)n_own * sizeof(int32_t), hipMemcpyDeviceToDevice, st)); // (reference ids: avs_dist_get_solution scatters by them)
    }
    tr.mark("own list");
    AVS_TRY(dist_assemble_tail(c, d, ids, n_own, n, dom.p, n_w, nullptr, nullptr, owner, is_halo, g2l, flag, pos, scan_tmp));
    tr.mark("tail");

    // the cuts this frame's weights suggest for the next one: per-plane weights of the own rows, summed over the ranks
    d->next_cuts = d->cuts;
    if (!d->hosted) {
        int shift = c->desc.levels - 1;
        if (cur_opt().dist_plane_shift >= 0) shift = cur_opt().dist_plane_shift < shift ? cur_opt().dist_plane_shift : shift;
        else if (shift > 2) shift = 2;
        const int nplanes = (extent + (1 << shift) - 1) >> shift;
        DevBuf<unsigned long long> &weight = d->ws.weight;
        DevBuf<int32_t> &w32 = d->ws.w32;
        AVS_TRY(weight.reserve((size_t)nplanes));
        AVS_TRY(w32.reserve((size_t)nplanes));
        AVS_HIP(hipMemsetAsync(weight.p, 0, (size_t)nplanes * sizeof(unsigned long long), st));
        if (n_own)
            hipLaunchKernelGGL(k_win_plane_weights, dim3(grid256(n_own)), dim3(256), 0, st, n_own, (const int32_t *)d->local_ref.p, c->vdof.p,
                               (const int32_t *)d->row_ptr.p, axis, extent, shift, weight.p);
        hipLaunchKernelGGL(k_win_weights_pack, dim3(grid256(nplanes)), dim3(256), 0, st, nplanes, weight.p, w32.p);
        AVS_TRY(dist_allreduce_i32(d, w32.p, nplanes, st));
        std::vector<int32_t> hw((size_t)nplanes);
        AVS_HIP(hipMemcpyAsync(hw.data(), w32.p, hw.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        AVS_HIP(hipStreamSynchronize(st));
        std::vector<int64_t> w64(hw.begin(), hw.end());
        std::vector<int> po((size_t)nplanes);
        plane_owners_from_weights(w64.data(), nplanes, world, po.data());
        cuts_from_plane_owners(po, world, shift, extent, d->next_cuts);
    }
    return AVS_OK;
}

static avs_status dist_assemble_device(avs_ctx *c, PcgDist *d, int cut_axis, int extent)
{
    hipStream_t st = c->stream;
    const int64_t n = c->n_vel;
    const int rank = d->rank, world = d->world;
    AVS_REQUIRE(world <= 32, AVS_EINVAL, "at most 32 ranks");
    // Cut planes: the slabs are cut between planes of 2^shift fine cells.  The replicated planner (avs_dist_partition, avs_partition.cpp)
    // keeps 2^(levels-1) -- no top-level cell straddles a cut; here, where every rank assembles its own rows, nothing needs that (a DOF
    // belongs to the plane of its own position, whatever it reads is a halo column), and planes of at most 4 cells halve the granularity
    // of the balance at 4 levels: on the 8-way partition of the 512^3 beam a rank's share moved in steps of 12.5 % (868 k .. 994 k rows,
    // 30.5 .. 34.3 us per iteration in the loop-back measurement -- the slowest rank sets the pace).  AVS_DIST_PLANE_SHIFT overrides.
    int shift = c->desc.levels - 1;
    if (cur_opt().dist_plane_shift >= 0) shift = cur_opt().dist_plane_shift < shift ? cur_opt().dist_plane_shift : shift;
    else if (shift > 2) shift = 2;
    const int nplanes = (extent + (1 << shift) - 1) >> shift;
    AVS_REQUIRE(nplanes <= 65535, AVS_EINVAL, "too many cut planes");

    // global, index-only: raw triplet count per row (weights) and the brick-major permutation
    DevBuf<int32_t> &raw_count = d->ws.raw_count; // (work arrays kept across frames: see PcgDist::Scratch)
    AVS_TRY(count_raw_rows(c, raw_count));
    AVS_TRY(build_brick_permutation(c, c->brick_shift));

    DevBuf<uint16_t> &plane = d->ws.plane;
    DevBuf<unsigned long long> &weight = d->ws.weight;
    DevBuf<int32_t> &plane_owner = d->ws.plane_owner, &flag = d->ws.flag, &pos = d->ws.pos, &g2l = d->ws.g2l, &scan_tmp = d->ws.scan_tmp, &ids = d->ws.ids;
    DevBuf<uint8_t> &owner = d->ws.owner, &is_halo = d->ws.is_halo;
    AVS_TRY(plane.reserve((size_t)(n > 0 ? n : 1)));
    AVS_TRY(weight.reserve((size_t)nplanes));
    AVS_TRY(plane_owner.reserve((size_t)nplanes));
    AVS_TRY(flag.reserve((size_t)n + 1));
    AVS_TRY(pos.reserve((size_t)n + 1));
    AVS_TRY(g2l.reserve((size_t)(n > 0 ? n : 1)));
    AVS_TRY(scan_tmp.reserve(scan_tmp_elems(n + 1)));
    AVS_TRY(owner.reserve((size_t)(n > 0 ? n : 1)));
    AVS_TRY(is_halo.reserve((size_t)(n > 0 ? n : 1)));
    AVS_HIP(hipMemsetAsync(weight.p, 0, (size_t)nplanes * sizeof(unsigned long long), st));
    AVS_HIP(hipMemsetAsync(is_halo.p, 0, (size_t)n, st));
    AVS_HIP(hipMemsetAsync(g2l.p, 0xFF, (size_t)n * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_da_planes, dim3(2048), dim3(256), 0, st, n, c->vdof.p, c->perm.p, raw_count.p, cut_axis, extent, shift, nplanes,
                       plane.p, weight.p);
    std::vector<unsigned long long> h_w((size_t)nplanes);
    AVS_HIP(hipMemcpyAsync(h_w.data(), weight.p, h_w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> h_w64(h_w.begin(), h_w.end());
    std::vector<int> h_po((size_t)nplanes);
    plane_owners_from_weights(h_w64.data(), nplanes, world, h_po.data());
    cuts_from_plane_owners(h_po, world, shift, extent, d->cuts);
    d->next_cuts = d->cuts;
    d->cut_axis = cut_axis;
    AVS_HIP(hipMemcpyAsync(plane_owner.p, h_po.data(), h_po.size() * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_plan_owner, dim3(grid256(n)), dim3(256), 0, st, n, plane.p, plane_owner.p, rank, owner.p, flag.p);

    // owned rows (ascending brick-major id) and the DOFs behind them
    int64_t n_own = 0;
    AVS_TRY(scan_flags(flag.p, pos.p, n, scan_tmp, &n_own, st));
    AVS_TRY(d->own_global.reserve((size_t)n_own));
    AVS_TRY(ids.reserve((size_t)(n_own > 0 ? n_own : 1)));
    hipLaunchKernelGGL(k_plan_scatter, dim3(grid256(n)), dim3(256), 0, st, n, flag.p, pos.p, (const int32_t *)nullptr, d->own_global.p,
                       g2l.p, 0);
    if (n_own) hipLaunchKernelGGL(k_gather_i<int32_t>, dim3(grid256(n_own)), dim3(256), 0, st, c->perm.p, d->own_global.p, ids.p, n_own);

    return dist_assemble_tail(c, d, ids, n_own, n, nullptr, n, c->inv.p, c->perm.p, owner, is_halo, g2l, flag, pos, scan_tmp);
}

// storage form of the rank's local rows (avs_get_matrix_format when no global matrix exists)
bool dist_matrix_format(avs_ctx *c, avs_matrix_format *fmt)
{
    if (fmt) { fmt->brick_tiles = 0; fmt->brick_patterns = 0; fmt->brick_pattern_rows = 0; fmt->brick_bytes = 0; }
    PcgDist *d = c->dist;
    if (!d || !d->partitioned) return false;
    fmt->reordered = d->reordered ? 1 : 0;
    fmt->value_table_size = d->vi.table_size;
    fmt->column_bits = d->vi.col_bits;
    fmt->bytes_per_nonzero = d->vi.bytes_per_nonzero();
    fmt->tile_local_tables = d->vi.tile_tables ? 1 : 0;
    fmt->column_windows = d->vi.col_windows ? 1 : 0;
    if (d->brick.ready) {
        fmt->brick_tiles = d->brick.ntiles;
        fmt->brick_patterns = d->brick.patterns;
        fmt->brick_pattern_rows = d->brick.regular_rows;
        fmt->brick_bytes = d->brick.stored_bytes(d->n_own);
        fmt->brick_walk = d->brick_view.walk;
        fmt->brick_value_codes = d->brick.vc ? 1 : 0;
    }
    return true;
}

// the host planner (avs_partition.cpp) on a downloaded copy of the pattern: reference for the device planner
static avs_status plan_on_host(avs_ctx *c, PcgDist *d, int cut_axis, int extent, bool ro)
{
    hipStream_t st = c->stream;
    const int64_t n = c->n_vel, nnz = c->nnz;
    const int32_t *g_rp = ro ? c->p_row_ptr.p : c->row_ptr.p, *g_col = ro ? c->p_col.p : c->col.p;
    const double *g_val = ro ? c->p_val.p : c->val.p;
    std::vector<int32_t> h_rp((size_t)n + 1), h_col((size_t)nnz), h_tab((size_t)n * 4), h_owner((size_t)n);
    AVS_HIP(hipMemcpyAsync(h_rp.data(), g_rp, h_rp.size() * 4, hipMemcpyDeviceToHost, st));
    AVS_HIP(hipMemcpyAsync(h_col.data(), g_col, h_col.size() * 4, hipMemcpyDeviceToHost, st));
    AVS_HIP(hipMemcpyAsync(h_tab.data(), c->vdof.p, h_tab.size() * 4, hipMemcpyDeviceToHost, st));
    AVS_HIP(hipStreamSynchronize(st));
    if (ro) { // dof table in the new numbering
        std::vector<int32_t> h_perm((size_t)n), t2((size_t)n * 4);
        AVS_HIP(hipMemcpy(h_perm.data(), c->perm.p, h_perm.size() * 4, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) memcpy(&t2[(size_t)i * 4], &h_tab[(size_t)h_perm[(size_t)i] * 4], 16);
        h_tab.swap(t2);
    }
    AVS_TRY(avs_plan_owners(n, h_tab.data(), h_rp.data(), c->desc.levels, cut_axis, extent, d->world, h_owner.data()));
    avs_plan *plan = nullptr;
    AVS_TRY(avs_plan_create(n, h_rp.data(), h_col.data(), h_owner.data(), d->rank, d->world, &plan));
    avs_plan_sizes sz{};
    avs_plan_get_sizes(plan, &sz);
    std::vector<int32_t> own((size_t)sz.n_own), rpl((size_t)sz.n_own + 1), cl((size_t)sz.nnz_local), vs((size_t)sz.nnz_local),
        peers((size_t)sz.n_peers), sc((size_t)sz.n_peers), rc((size_t)sz.n_peers), sidx((size_t)sz.n_send);
    avs_plan_get_arrays(plan, own.data(), nullptr, rpl.data(), cl.data(), vs.data(), peers.data(), sc.data(), rc.data(), sidx.data());
    avs_plan_destroy(plan);
    d->n_own = sz.n_own;
    d->n_halo = sz.n_halo;
    d->n_send = sz.n_send;
    d->nnz_local = sz.nnz_local;
    d->peers = peers;
    d->send_counts = sc;
    d->recv_counts = rc;
    AVS_TRY(d->own_global.alloc((size_t)sz.n_own));
    AVS_TRY(d->send_idx.alloc((size_t)sz.n_send));
    AVS_TRY(d->sendbuf.alloc((size_t)sz.n_send));
    AVS_TRY(d->row_ptr.alloc((size_t)sz.n_own + 1));
    AVS_TRY(d->col.alloc((size_t)sz.nnz_local));
    AVS_TRY(d->val.alloc((size_t)sz.nnz_local));
    DevBuf<int32_t> d_vs;
    AVS_TRY(d_vs.alloc((size_t)sz.nnz_local));
    AVS_HIP(hipMemcpyAsync(d->own_global.p, own.data(), own.size() * 4, hipMemcpyHostToDevice, st));
    if (sz.n_send) AVS_HIP(hipMemcpyAsync(d->send_idx.p, sidx.data(), sidx.size() * 4, hipMemcpyHostToDevice, st));
    AVS_HIP(hipMemcpyAsync(d->row_ptr.p, rpl.data(), rpl.size() * 4, hipMemcpyHostToDevice, st));
    if (sz.nnz_local) {
        AVS_HIP(hipMemcpyAsync(d->col.p, cl.data(), cl.size() * 4, hipMemcpyHostToDevice, st));
        AVS_HIP(hipMemcpyAsync(d_vs.p, vs.data(), vs.size() * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_gather_i<double>, dim3(grid256(sz.nnz_local)), dim3(256), 0, st, g_val, d_vs.p, d->val.p, sz.nnz_local);
    }
    AVS_HIP(hipGetLastError());
    AVS_HIP(hipStreamSynchronize(st));
    // tiles of the local SpMV that read no halo column can run while the halo is in flight
    const int T = spmv_tile_rows();
    const int64_t ntiles = (sz.n_own + T - 1) / T;
    std::vector<int32_t> ti, tb;
    for (int64_t t = 0; t < ntiles; ++t) {
        const int64_t r0 = t * T, r1 = std::min<int64_t>(r0 + T, sz.n_own);
        bool touches = false;
        for (int32_t k = rpl[(size_t)r0]; k < rpl[(size_t)r1] && !touches; ++k) touches = cl[(size_t)k] >= sz.n_own;
        (touches ? tb : ti).push_back((int32_t)t);
    }
    d->n_tiles_int = (int)ti.size();
    d->n_tiles_bnd = (int)tb.size();
    AVS_TRY(d->tiles_int.alloc(ti.size()));
    AVS_TRY(d->tiles_bnd.alloc(tb.size()));
    if (!ti.empty()) AVS_HIP(hipMemcpy(d->tiles_int.p, ti.data(), ti.size() * 4, hipMemcpyHostToDevice));
    if (!tb.empty()) AVS_HIP(hipMemcpy(d->tiles_bnd.p, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
    return AVS_OK;
}

} // namespace avs

using namespace avs;

extern "C" {

avs_status avs_get_dof_table(avs_ctx *c, avs_index_kind kind, int32_t *table, avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && table, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->tables_ready, AVS_ESTATE, "dof tables not built: call avs_assemble first");
    AVS_HIP(hipSetDevice(c->desc.device));
    const LatBuf<int32_t> &t = kind == AVS_INDEX_VELOCITY ? c->vdof : (kind == AVS_INDEX_EDGE ? c->edof : c->cdof);
    const int64_t n = kind == AVS_INDEX_VELOCITY ? c->n_vel : (kind == AVS_INDEX_EDGE ? c->n_edge : c->n_center);
    AVS_HIP(copy_out(table, t.p, (size_t)n * 4 * sizeof(int32_t), where, c->stream));
    AVS_HIP(hipStreamSynchronize(c->stream));
    return AVS_OK;
}

avs_status avs_dist_get_unique_id(uint8_t id[AVS_UNIQUE_ID_BYTES])
{
    AVS_REQUIRE(id, AVS_EINVAL, "null argument");
    static_assert(sizeof(ncclUniqueId) <= AVS_UNIQUE_ID_BYTES, "unique id does not fit");
    ncclUniqueId u;
    AVS_NCCL(ncclGetUniqueId(&u));
    memset(id, 0, AVS_UNIQUE_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return AVS_OK;
}

static avs_status new_dist(avs_ctx *c, int rank, int world)
{
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(world >= 1 && world <= 32 && rank >= 0 && rank < world, AVS_EINVAL, "rank %d / world %d out of range", rank, world);
    dist_release(c);
    c->dist = new (std::nothrow) PcgDist();
    AVS_REQUIRE(c->dist, AVS_ENOMEM, "out of host memory");
    c->dist->rank = rank;
    c->dist->world = world;
    c->dist->device = c->desc.device;
    return AVS_OK;
}

avs_status avs_dist_init(avs_ctx *c, const uint8_t id[AVS_UNIQUE_ID_BYTES], int32_t rank, int32_t world)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && id, AVS_EINVAL, "null argument");
    AVS_HIP(hipSetDevice(c->desc.device));
    AVS_TRY(new_dist(c, rank, world));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    AVS_TRY(c->dist->blob_send.alloc(AVS_DIST_BLOB_BYTES)); // staging of the transport set-up collectives: allocated here, so that no
    AVS_TRY(c->dist->blob_recv.alloc((size_t)world * AVS_DIST_BLOB_BYTES)); // rank can fail an allocation BETWEEN two collectives
    AVS_TRY(c->dist->vote_word.alloc(1));
    AVS_NCCL(ncclCommInitRank(&c->dist->comm, world, u, rank));
    // second communicator for the point-to-point traffic (same ranks); without it the exchange shares
    // `comm` and stays on the solver stream (no overlap)
    if (ncclCommSplit(c->dist->comm, 0, rank, &c->dist->comm_p2p, nullptr) != ncclSuccess || !c->dist->comm_p2p) {
        c->dist->comm_p2p = c->dist->comm;
        c->dist->no_overlap = true;
    }
    return AVS_OK;
}

avs_status avs_local_group_create(int32_t world, avs_local_group **out)
{
    AVS_REQUIRE(out && world >= 1 && world <= 32, AVS_EINVAL, "bad argument");
    avs_local_group *g = new (std::nothrow) avs_local_group();
    AVS_REQUIRE(g, AVS_ENOMEM, "out of host memory");
    g->world = world;
    g->members.assign((size_t)world, nullptr);
    g->red.assign((size_t)world * 4, 0.);
    *out = g;
    return AVS_OK;
}

void avs_local_group_destroy(avs_local_group *g) { delete g; }

avs_status avs_dist_init_local(avs_ctx *c, avs_local_group *g, int32_t rank)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && g, AVS_EINVAL, "null argument");
    AVS_TRY(new_dist(c, rank, g->world));
    c->dist->group = g;
    std::lock_guard<std::mutex> lk(g->m);
    g->members[(size_t)rank] = c->dist;
    return AVS_OK;
}

avs_status avs_dist_init_hosted(avs_ctx *c, int32_t rank, int32_t world)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_TRY(new_dist(c, rank, world));
    c->dist->hosted = true;
    return AVS_OK;
}

avs_status avs_dist_export_blob(avs_ctx *c, uint8_t blob[AVS_DIST_BLOB_BYTES])
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && blob, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist && c->dist->direct_prepared, AVS_ESTATE, "no comm block: call avs_dist_assemble / avs_dist_partition first");
    memcpy(blob, c->dist->blob.data(), AVS_DIST_BLOB_BYTES);
    return AVS_OK;
}

// Measurement aid (tools/loopback_scaling.py): ONE rank of a world-size-W partition runs alone on one GPU with its peers
// looped back onto itself -- it pushes its boundary entries (into a scratch area of its own block), raises the flags its peers
// would raise, and all-gathers with itself.  Its halo values stay 0, so it solves a different (still SPD) system; what is
// representative is the TIME of an iteration of that rank's slab: real update + push + interior / halo-touching tiles +
// finalisation with W contributions, minus the xGMI latency.  Never used by a solve that is meant to be right.
static avs_status direct_connect_loopback(avs_ctx *c, PcgDist *d)
{
    AVS_REQUIRE(d->direct_prepared, AVS_ESTATE, "no comm block (assemble / partition first)");
    AVS_HIP(hipSetDevice(c->desc.device));
    // re-allocate the block with a scratch tail for the pushes
    const size_t bytes = kHeaderBytes + (size_t)((d->n_halo > 0 ? d->n_halo : 1) + (d->n_send > 0 ? d->n_send : 1)) * sizeof(double);
    if (d->comm_block) (void)hipFree(d->comm_block);
    d->comm_block = nullptr;
    void *p = nullptr;
    AVS_HIP(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
    AVS_HIP(hipMemset(p, 0, bytes));
    AVS_HIP(hipMemset((char *)p + offsetof(CommHeader, red), 0xFF, sizeof(CommHeader::red)));
    d->comm_block = p;
    d->comm_bytes = bytes;
    std::vector<uint8_t> blobs((size_t)d->world * AVS_DIST_BLOB_BYTES, 0);
    for (int q = 0; q < d->world; ++q) { // every "peer" is me
        DistBlob b;
        memcpy(&b, d->blob.data(), sizeof(b));
        b.rank = q;
        b.raw_ptr = (uint64_t)(uintptr_t)p;
        b.bytes = bytes;
        for (int k = 0; k < kMaxRanks; ++k) { b.recv_off_of[k] = -1; b.recv_cnt_of[k] = 0; }
        for (size_t i = 0; i < d->peers.size(); ++i)
            if (d->peers[i] == q) { // what I send to q lands in the scratch tail, at my send offset
                b.recv_off_of[d->rank] = (int32_t)(d->n_halo + d->send_offs[i]);
                b.recv_cnt_of[d->rank] = d->send_counts[i];
            }
        memcpy(blobs.data() + (size_t)q * AVS_DIST_BLOB_BYTES, &b, sizeof(b));
    }
    AVS_TRY(direct_connect(c, d, blobs.data(), false));
    // the flags my peers would raise are raised by my own push / finalisation
    DistDev h;
    AVS_HIP(hipMemcpy(&h, d->dd.p, sizeof(h), hipMemcpyDeviceToHost));
    CommHeader *mine = (CommHeader *)d->comm_block;
    for (int i = 0; i < h.npeers; ++i) {
        h.peer_hflag_dst[i] = &mine->hflag[h.peer_rank[i]];
        h.peer_hsum_dst[i] = &mine->hsum[h.peer_rank[i]];
    }
    h.paranoid = 0; // the halo area of a looped-back rank is never written: nothing to check
    h.inject_stale_round = -1;
    for (int q = 0; q < d->world; ++q) {
        h.all_red_dst[q] = &mine->red[0][q][0]; // as if rank q had written its sums into my block
    }
    // a peer that only sends to me (no entry from me to it) would never get its flag raised: give every receive a sender
    for (int i = 0; i < h.npeers; ++i)
        AVS_REQUIRE(h.recv_cnt[i] == 0 || h.send_off[i + 1] > h.send_off[i], AVS_EINTERNAL, "loopback: peer %d sends but does not receive", h.peer_rank[i]);
    AVS_HIP(hipMemcpy(d->dd.p, &h, sizeof(h), hipMemcpyHostToDevice));
    d->dd_host = h;
    return AVS_OK;
}

avs_status avs_dist_import_blobs(avs_ctx *c, const uint8_t *blobs)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist, AVS_ESTATE, "call avs_dist_init_hosted first");
    if (cur_opt().dist_loopback) return direct_connect_loopback(c, c->dist);
    AVS_REQUIRE(blobs, AVS_EINVAL, "null argument");
    AVS_TRY(direct_connect(c, c->dist, blobs));
    // hosted group: every rank calls this with the same blobs; the self-test's rounds rendezvous through the comm blocks.  There is
    // no other transport to fall back to, so a failure is an error.
    bool passed = true;
    AVS_TRY(run_selftest(c, c->dist, &passed));
    if (!passed) {
        c->dist->direct_ready = false;
        return AVS_ERCCL;
    }
    return AVS_OK;
}

avs_status avs_dist_partition(avs_ctx *c, int32_t cut_axis)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist, AVS_ESTATE, "call avs_dist_init / avs_dist_init_local first");
    AVS_REQUIRE(c->system_ready, AVS_ESTATE, "avs_assemble must succeed before avs_dist_partition");
    AVS_HIP(hipSetDevice(c->desc.device));
    AVS_REQUIRE(!c->slab.on, AVS_ESTATE, "the context holds a slab-local pre-pass (this rank's window only): only avs_dist_assemble works on it");
    PcgDist *d = c->dist;
    hipStream_t st = c->stream;
    const int64_t n = c->n_vel;
    if (cut_axis < 0) { // longest axis
        cut_axis = 0;
        if (c->desc.ny > c->desc.nx) cut_axis = 1;
        if (c->desc.nz > (cut_axis == 0 ? c->desc.nx : c->desc.ny)) cut_axis = 2;
    }
    AVS_REQUIRE(cut_axis <= 2, AVS_EINVAL, "cut_axis out of range");
    const int extent = cut_axis == 0 ? c->desc.nx : (cut_axis == 1 ? c->desc.ny : c->desc.nz);

    // the solve works on the brick-major numbering when it exists (avs_reorder.hip): partition THAT system,
    // so every rank's local rows keep the brick locality; results are mapped back in avs_dist_get_solution
    const bool ro = c->reordered;
    const double *g_rhs = ro ? c->p_rhs.p : c->rhs.p, *g_x0 = ro ? c->p_x0.p : c->x0.p;
    const bool host_plan = cur_opt().dist_host_plan != 0;
    d->vi.clear();
    d->brick.clear();   // (the brick-structured form is built for distributed assemblies only: avs_dist_assemble)
    d->brick.view(d->brick_view, d->vi);
    d->local_ref.release();
    if (host_plan) AVS_TRY(plan_on_host(c, d, cut_axis, extent, ro));
    else AVS_TRY(plan_on_device(c, d, cut_axis, extent, ro));
    avs_plan_sizes sz{};
    sz.n_own = d->n_own;
    sz.n_halo = d->n_halo;
    sz.nnz_local = d->nnz_local;
    sz.n_send = d->n_send;
    d->n_global = n;
    d->send_offs.assign(d->peers.size(), 0);
    d->recv_offs.assign(d->peers.size(), 0);
    {
        int32_t so = 0, ro2 = 0;
        for (size_t i = 0; i < d->peers.size(); ++i) {
            d->send_offs[i] = so;
            d->recv_offs[i] = ro2;
            so += d->send_counts[i];
            ro2 += d->recv_counts[i];
        }
    }
    if (ro && sz.nnz_local) // same lossless compression as the single-GPU solve, on the local rows
        AVS_TRY(build_matrix_index(d->row_ptr.p, d->col.p, d->val.p, sz.n_own, sz.nnz_local, (int64_t)sz.n_own + sz.n_halo, d->vi, st));
    AVS_TRY(d->rhs.alloc((size_t)sz.n_own));
    AVS_TRY(d->x0.alloc((size_t)sz.n_own));
    AVS_TRY(d->x.alloc((size_t)sz.n_own));
    if (sz.n_own) {
        const unsigned g = (unsigned)((sz.n_own + 255) / 256);
        hipLaunchKernelGGL(k_gather_i<double>, dim3(g), dim3(256), 0, st, g_rhs, d->own_global.p, d->rhs.p, sz.n_own);
        hipLaunchKernelGGL(k_gather_i<double>, dim3(g), dim3(256), 0, st, g_x0, d->own_global.p, d->x0.p, sz.n_own);
    }
    AVS_HIP(hipGetLastError());
    AVS_HIP(hipStreamSynchronize(st));
    if (!d->comm_stream) {
        AVS_HIP(hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking));
        AVS_HIP(hipEventCreateWithFlags(&d->ev_ready, hipEventDisableTiming));
        AVS_HIP(hipEventCreateWithFlags(&d->ev_halo, hipEventDisableTiming));
    }
    pcg_destroy(d->pcg);
    d->pcg = nullptr;
    AVS_TRY(pcg_create(&d->pcg, sz.n_own, sz.n_own + sz.n_halo, st));
    direct_release(d);
    d->direct_pending = true; // planning needs no peer; connecting the transport does, so it waits for avs_dist_solve
    if (d->hosted) AVS_TRY(direct_setup(c, d)); // hosted group: only allocates the block and fills the blob (no collective)
    d->partitioned = true;
    d->reordered = ro;
    d->solved = false;
    return AVS_OK;
}

avs_status avs_dist_assemble(avs_ctx *c, int32_t cut_axis, avs_assembly_info *info)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist, AVS_ESTATE, "call avs_dist_init / avs_dist_init_local first");
    AVS_HIP(hipSetDevice(c->desc.device));
    PcgDist *d = c->dist;
    hipStream_t st = c->stream;
    if (c->slab.on) {
        AVS_REQUIRE(cut_axis < 0 || cut_axis == c->slab.axis, AVS_EINVAL, "the context's lattices were cut along axis %d by the slab-local pre-pass", c->slab.axis);
        cut_axis = c->slab.axis;
    }
    if (cut_axis < 0) { // longest axis
        cut_axis = 0;
        if (c->desc.ny > c->desc.nx) cut_axis = 1;
        if (c->desc.nz > (cut_axis == 0 ? c->desc.nx : c->desc.ny)) cut_axis = 2;
    }
    AVS_REQUIRE(cut_axis <= 2, AVS_EINVAL, "cut_axis out of range");
    const int extent = cut_axis == 0 ? c->desc.nx : (cut_axis == 1 ? c->desc.ny : c->desc.nz);
    Timer t(st);
    avs_assembly_info ai{};
    t.start();
    AVS_TRY(build_stencils(c)); // dof tables + stencils: index-only; every rank builds all of them -- or, on a slab-local context, the ones around its slab
    ai.stencil_ms = t.stop();
    Scope scope("Build Octree Linear System"); // cpp:554: here only this rank's rows
    ai.guess_ms = 0.; // the restriction of the owned DOFs is part of system_ms here
    t.start();
    c->system_ready = false; // no global matrix in this mode
    c->reordered = false;
    d->vi.clear();
    if (c->slab.on) AVS_TRY(dist_assemble_window(c, d));
    else AVS_TRY(dist_assemble_device(c, d, cut_axis, extent));
    ai.system_ms = t.stop();
    t.start();
    d->n_global = c->n_vel;
    d->send_offs.assign(d->peers.size(), 0);
    d->recv_offs.assign(d->peers.size(), 0);
    {
        int32_t so = 0, ro2 = 0;
        for (size_t i = 0; i < d->peers.size(); ++i) {
            d->send_offs[i] = so;
            d->recv_offs[i] = ro2;
            so += d->send_counts[i];
            ro2 += d->recv_counts[i];
        }
    }
    if (d->nnz_local) // the rank's own dictionaries: its rows only hold a subset of the global values
        AVS_TRY(build_matrix_index(d->row_ptr.p, d->col.p, d->val.p, d->n_own, d->nnz_local, d->n_own + d->n_halo, d->vi, st));
    AVS_TRY(dist_build_brick(c, d)); // slabs of >= 2 M rows: the brick-structured form of the local rows
    if (!d->comm_stream) {
        AVS_HIP(hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking));
        AVS_HIP(hipEventCreateWithFlags(&d->ev_ready, hipEventDisableTiming));
        AVS_HIP(hipEventCreateWithFlags(&d->ev_halo, hipEventDisableTiming));
    }
    pcg_destroy(d->pcg);
    d->pcg = nullptr;
    AVS_TRY(pcg_create(&d->pcg, d->n_own, d->n_own + d->n_halo, st));
    direct_release(d);
    d->direct_pending = true;
    if (d->hosted) AVS_TRY(direct_setup(c, d));
    ai.csr_ms = t.stop();
    d->partitioned = true;
    d->reordered = !c->slab.on; // own_global holds brick-major ids: avs_dist_get_solution maps back through c->inv (slab-local: reference ids)
    d->solved = false;
    ai.n_velocity = c->n_vel;
    ai.n_edge = c->n_edge;
    ai.n_center = c->n_center;
    ai.nnz = d->nnz_local;
    ai.raw_triplets = 0;
    if (info) *info = ai;
    return AVS_OK;
}

static avs_status prepass_allreduce_cb(int32_t *dev, int64_t count, void *stream, void *user)
{
    return dist_allreduce_i32(static_cast<PcgDist *>(user), dev, count, reinterpret_cast<hipStream_t>(stream));
}

avs_status avs_dist_bind_prepass(avs_ctx *c, avs_prepass *pp, int32_t cut_axis, const int32_t *cuts)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && pp, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist, AVS_ESTATE, "call avs_dist_init / avs_dist_init_local first");
    PcgDist *d = c->dist;
    if (d->world <= 1 || !cuts) return avs_prepass_set_slab(pp, 0, nullptr, 1, 0, nullptr, nullptr);
    AVS_REQUIRE(d->comm || d->group, AVS_ESTATE, "a hosted group sums through the caller's own callback: avs_prepass_set_slab");
    if (cut_axis < 0) {
        cut_axis = 0;
        if (c->desc.ny > c->desc.nx) cut_axis = 1;
        if (c->desc.nz > (cut_axis == 0 ? c->desc.nx : c->desc.ny)) cut_axis = 2;
    }
    return avs_prepass_set_slab(pp, cut_axis, cuts, d->world, d->rank, prepass_allreduce_cb, d);
}

avs_status avs_dist_get_cuts(avs_ctx *c, int32_t which, int32_t *cut_axis, int32_t *cuts)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && cuts, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist && c->dist->partitioned && c->dist->cuts.size() == (size_t)c->dist->world + 1, AVS_ESTATE, "call avs_dist_assemble first");
    const std::vector<int32_t> &v = which ? c->dist->next_cuts : c->dist->cuts;
    for (size_t i = 0; i < v.size(); ++i) cuts[i] = v[i];
    if (cut_axis) *cut_axis = c->dist->cut_axis;
    return AVS_OK;
}

avs_status avs_dist_get_plan_sizes(avs_ctx *c, avs_plan_sizes *s)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && s, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist && c->dist->partitioned, AVS_ESTATE, "call avs_dist_partition first");
    s->n_own = c->dist->n_own;
    s->n_halo = c->dist->n_halo;
    s->nnz_local = c->dist->nnz_local;
    s->n_send = c->dist->n_send;
    s->n_peers = (int32_t)c->dist->peers.size();
    return AVS_OK;
}

avs_status avs_dist_get_plan_arrays(avs_ctx *c, int32_t *own_global, int32_t *row_ptr_local, int32_t *col_local, int32_t *send_idx,
                                    int32_t *peers, int32_t *send_counts, int32_t *recv_counts, int32_t *tiles_interior,
                                    int32_t *tiles_boundary)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist && c->dist->partitioned, AVS_ESTATE, "call avs_dist_partition first");
    PcgDist *d = c->dist;
    AVS_HIP(hipSetDevice(c->desc.device));
    auto down = [&](int32_t *dst, const int32_t *src, size_t count) -> avs_status {
        if (dst && count) AVS_HIP(hipMemcpy(dst, src, count * sizeof(int32_t), hipMemcpyDeviceToHost));
        return AVS_OK;
    };
    AVS_HIP(hipStreamSynchronize(c->stream));
    AVS_TRY(down(own_global, d->own_global.p, (size_t)d->n_own));
    AVS_TRY(down(row_ptr_local, d->row_ptr.p, (size_t)d->n_own + 1));
    AVS_TRY(down(col_local, d->col.p, (size_t)d->nnz_local));
    AVS_TRY(down(send_idx, d->send_idx.p, (size_t)d->n_send));
    AVS_TRY(down(tiles_interior, d->tiles_int.p, (size_t)d->n_tiles_int));
    AVS_TRY(down(tiles_boundary, d->tiles_bnd.p, (size_t)d->n_tiles_bnd));
    for (size_t i = 0; i < d->peers.size(); ++i) {
        if (peers) peers[i] = d->peers[i];
        if (send_counts) send_counts[i] = d->send_counts[i];
        if (recv_counts) recv_counts[i] = d->recv_counts[i];
    }
    return AVS_OK;
}

avs_status avs_dist_get_overlap_tiles(avs_ctx *c, int32_t *interior, int32_t *boundary)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && interior && boundary, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist && c->dist->partitioned, AVS_ESTATE, "call avs_dist_partition first");
    *interior = c->dist->n_tiles_int;
    *boundary = c->dist->n_tiles_bnd;
    return AVS_OK;
}

avs_status avs_dist_solve(avs_ctx *c, double tol, int32_t max_iters, avs_solve_info *info)
{
    avs::OptScope opt_scope_(c);
    avs::CancelScope cancel_scope_(c ? &c->cancel : nullptr);
    AVS_REQUIRE(c, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist && c->dist->partitioned, AVS_ESTATE, "call avs_dist_partition first");
    AVS_REQUIRE(tol >= 0. && max_iters >= 0, AVS_EINVAL, "tolerance / max_iterations out of range");
    AVS_HIP(hipSetDevice(c->desc.device));
    PcgDist *d = c->dist;
    AVS_REQUIRE(!d->hosted || d->direct_ready, AVS_ESTATE, "hosted group: exchange the blobs first (avs_dist_export_blob / avs_dist_import_blobs)");
    Scope scope("Solve Linear System"); // cpp:603
    if (d->direct_pending && !d->hosted) { // first solve on this plan: every rank is here, connect the transport
        AVS_TRY(direct_setup(c, d));
        d->direct_pending = false;
    }
    AVS_HIP(hipMemcpyAsync(d->x.p, d->x0.p, (size_t)d->n_own * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    CsrView A;
    A.n = d->n_own;
    A.nnz = d->nnz_local;
    A.row_ptr = d->row_ptr.p;
    A.col = d->col.p;
    A.val = d->val.p;
    d->vi.apply(A);
    A.no_precond = c->no_precond;
    A.brick = d->brick.ready ? &d->brick_view : nullptr;
    avs_solve_info local{};
    const avs_status rc = pcg_solve(d->pcg, A, d->rhs.p, d->x.p, tol, max_iters, c->stream, &local, d);
    if (rc != AVS_OK) {
        // a peer's flag timed out: epochs / tickets of the comm blocks are no longer in step -- every rank sees the same fault
        // (it waits for the same peer) and leaves the direct transport for this plan; the next solve uses the fallback
        if (rc == AVS_ERCCL && d->direct_ready) {
            (void)hipStreamSynchronize(c->stream);
            direct_release(d);
        }
        return rc;
    }
    local.n = d->n_global;
    if (info) *info = local;
    d->solved = true;
    return AVS_OK;
}

#ifdef AVS_PROBES
// measurement / test entry (include/avs_probe.h): the rank's local rows times a caller-supplied [owned | halo] vector, through the
// storage form and kernel the partitioned loops launch (halo in the tail of the vector: the RCCL transport's layout)
avs_status avs_dist_spmv_local_form(avs_ctx *c, const double *x_ext, double *y, int32_t fused_dot, double *dot_out)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && x_ext && y, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist && c->dist->partitioned, AVS_ESTATE, "call avs_dist_assemble / avs_dist_partition first");
    AVS_HIP(hipSetDevice(c->desc.device));
    PcgDist *d = c->dist;
    CsrView A;
    A.n = d->n_own;
    A.nnz = d->nnz_local;
    A.row_ptr = d->row_ptr.p;
    A.col = d->col.p;
    A.val = d->val.p;
    d->vi.apply(A);
    A.brick = d->brick.ready ? &d->brick_view : nullptr;
    if (A.n == 0) return AVS_OK;
    AVS_TRY(probe_spmv_form(A, x_ext, y, fused_dot != 0, dot_out, c->stream));
    AVS_HIP(hipStreamSynchronize(c->stream));
    return AVS_OK;
}
#endif

avs_status avs_dist_get_info(avs_ctx *c, avs_dist_info *info)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && info, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist, AVS_ESTATE, "call avs_dist_init / avs_dist_init_local first");
    PcgDist *d = c->dist;
    memset(info, 0, sizeof(*info));
    info->world_size = d->world;
    if (d->comm) {
        int cnt = 0;
        AVS_NCCL(ncclCommCount(d->comm, &cnt));
        info->rccl_ranks = cnt;
    }
    info->transport = d->transport;
    info->selftest_rounds = d->selftest_rounds;
    info->selftest_bad_entries = d->selftest_bad;
    info->paranoid = d->direct_ready ? d->dd_host.paranoid : 0;
    if (d->direct_ready) {
        info->graph_replay = c->opt.graph != 0;
        info->launches_per_iteration = 2; // update (+ push), SpMV over all tiles (+ all-gather + scalar step in its finalizer block)
        info->collectives_per_iteration = 0;
        return AVS_OK;
    }
    info->graph_replay = 0;
    const bool sr = dist_wants_single_reduction(d);
    info->launches_per_iteration = d->world == 1 ? 5 : (sr ? 5 : 8);
    info->collectives_per_iteration = d->world == 1 ? 0 : (sr ? 2 : 3);
    return AVS_OK;
}

avs_status avs_dist_get_solution(avs_ctx *c, double *x, int64_t n, avs_memspace where)
{
    avs::OptScope opt_scope_(c);
    AVS_REQUIRE(c && x, AVS_EINVAL, "null argument");
    AVS_REQUIRE(c->dist && c->dist->solved, AVS_ESTATE, "no solution: call avs_dist_solve first");
    PcgDist *d = c->dist;
    AVS_REQUIRE(n == d->n_global, AVS_EINVAL, "vector length mismatch");
    AVS_HIP(hipSetDevice(c->desc.device));
    hipStream_t st = c->stream;
    DevBuf<double> full;
    AVS_TRY(full.alloc((size_t)n));
    AVS_HIP(hipMemsetAsync(full.p, 0, (size_t)n * sizeof(double), st));
    if (d->n_own)
        hipLaunchKernelGGL(k_scatter_own, dim3((unsigned)((d->n_own + 255) / 256)), dim3(256), 0, st, d->x.p, d->own_global.p, full.p, d->n_own);
    if (d->world > 1 && !d->hosted) { // hosted group: the caller adds the per-rank vectors (owned entries are disjoint, the rest is 0)
        if (d->comm) AVS_NCCL(ncclAllReduce(full.p, full.p, (size_t)n, ncclDouble, ncclSum, d->comm, st));
        else {
            // in-process: sum the host copies in rank order
            avs_local_group *g = d->group;
            std::vector<double> mine((size_t)n);
            AVS_HIP(hipMemcpyAsync(mine.data(), full.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
            AVS_HIP(hipStreamSynchronize(st));
            static std::mutex gm;
            static std::vector<double> acc;
            g->barrier();
            {
                std::lock_guard<std::mutex> lk(gm);
                if (acc.size() != (size_t)n) acc.assign((size_t)n, 0.);
                if (d->rank == 0) std::fill(acc.begin(), acc.end(), 0.);
            }
            g->barrier();
            for (int r = 0; r < g->world; ++r) { // each owner writes disjoint entries; order is irrelevant
                if (r == d->rank) {
                    std::lock_guard<std::mutex> lk(gm);
                    for (int64_t i = 0; i < n; ++i) acc[(size_t)i] += mine[(size_t)i];
                }
                g->barrier();
            }
            AVS_HIP(hipMemcpyAsync(full.p, acc.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
            AVS_HIP(hipStreamSynchronize(st));
            g->barrier();
        }
    }
    // the gathered vector also becomes the context's solution (reference numbering), so that
    // avs_get_solution / avs_transfer_to_regular_grid work on every rank after a partitioned solve
    AVS_TRY(c->x.alloc((size_t)n));
    if (d->reordered) AVS_TRY(unpermute(c, full.p, c->x.p)); // back to the reference's DOF numbering
    else AVS_HIP(hipMemcpyAsync(c->x.p, full.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    narrow_solution_if_f32(c, c->x.p, n);
    c->solved = true;
    AVS_HIP(copy_out(x, c->x.p, (size_t)n * sizeof(double), where, st));
    AVS_HIP(hipStreamSynchronize(st));
    return AVS_OK;
}

} // extern "C"
