// avs_partition.cpp -- host-side domain decomposition planner (no device code).
//
// the octree is cut into spatial slabs along one axis, on multiples of
// 2^(levels-1) fine cells so that no coarse cell straddles a cut.  Every velocity DOF (face) is
// owned by the slab that contains its position; global DOF ids stay the reference's ids and each
// rank renumbers locally as [owned (ascending global id) | halo grouped by owner, ascending id].
// The send list towards a peer is the ascending list of owned DOFs that peer's rows reference, so
// sender and receiver agree on the order without exchanging index lists.
//
// Pure integer work on host arrays: it is exercised without a GPU by the gloo tests
// (tests/test_dist_gloo.py) and reused verbatim by avs_dist_partition on the GPU box.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "avs.h"

namespace avs {
void set_error(const char *fmt, ...);

// greedy prefix cuts over the planes' weights: rank r owns a contiguous run of planes.  Shared by the host planner
// below and the device planner (avs_dist.hip), so both cut the domain at the same places.
void plane_owners_from_weights(const int64_t *weight, int nplanes, int world_size, int *plane_owner)
{
    int64_t total = 0;
    for (int p = 0; p < nplanes; ++p) total += weight[p];
    int64_t acc = 0;
    int r = 0;
    for (int p = 0; p < nplanes; ++p) {
        // move to the next rank once this rank's share is reached, keeping at least one plane for
        // every remaining rank when there are enough planes
        const int64_t target = (total * (r + 1)) / world_size;
        const int planes_left = nplanes - p;
        const int ranks_left = world_size - r;
        if (r < world_size - 1 && ((acc >= target && acc > 0) || planes_left < ranks_left)) ++r;
        plane_owner[p] = r;
        acc += weight[p];
    }
}
} // namespace avs

struct avs_plan {
    int32_t rank = 0, world = 1;
    int64_t n_global = 0;
    std::vector<int32_t> own_global, halo_global;
    std::vector<int32_t> row_ptr_local, col_local, val_src;
    std::vector<int32_t> peers, send_counts, recv_counts, send_idx;
};

extern "C" {

avs_status avs_plan_owners(int64_t n, const int32_t *dof_table, const int32_t *row_ptr, int32_t levels,
                           int32_t cut_axis, int32_t extent_fine, int32_t world_size, int32_t *owner_out)
{
    if (n < 0 || !dof_table || !row_ptr || !owner_out || levels < 1 || levels > AVS_MAX_LEVELS || cut_axis < 0 ||
        cut_axis > 2 || extent_fine < 1 || world_size < 1) {
        avs::set_error("avs_plan_owners: bad argument");
        return AVS_EINVAL;
    }
    const int granule = 1 << (levels - 1); // cuts only between top-level cells
    const int nplanes = (extent_fine + granule - 1) / granule;
    // position of a face along the cut axis in fine cells; a face on the upper domain border
    // (pos == extent) belongs to the last plane
    auto plane_of = [&](int64_t d) {
        const int32_t *t = dof_table + 4 * d;
        const int level = t[0] & 0xff;
        int64_t pos = (int64_t)t[1 + cut_axis] << level;
        if (pos >= extent_fine) pos = extent_fine - 1;
        return (int)(pos / granule);
    };
    std::vector<int64_t> weight((size_t)nplanes, 0); // rows weighted by their nnz (SpMV cost)
    for (int64_t d = 0; d < n; ++d) weight[(size_t)plane_of(d)] += (int64_t)(row_ptr[d + 1] - row_ptr[d]) + 2;
    std::vector<int> plane_owner((size_t)nplanes, 0);
    avs::plane_owners_from_weights(weight.data(), nplanes, world_size, plane_owner.data());
    for (int64_t d = 0; d < n; ++d) owner_out[d] = plane_owner[(size_t)plane_of(d)];
    return AVS_OK;
}

avs_status avs_plan_create(int64_t n, const int32_t *row_ptr, const int32_t *col, const int32_t *owner, int32_t rank,
                           int32_t world_size, avs_plan **out)
{
    if (n < 0 || !row_ptr || !owner || !out || rank < 0 || rank >= world_size || world_size > 32 || (n > 0 && !col)) {
        avs::set_error("avs_plan_create: bad argument (world_size <= 32)");
        return AVS_EINVAL;
    }
    avs_plan *p = new (std::nothrow) avs_plan();
    if (!p) {
        avs::set_error("out of host memory");
        return AVS_ENOMEM;
    }
    p->rank = rank;
    p->world = world_size;
    p->n_global = n;
    try {
        // global -> local for owned DOFs, and "which ranks need my DOF" bit masks
        std::vector<int32_t> g2l((size_t)n, -1);
        for (int64_t d = 0; d < n; ++d)
            if (owner[d] == rank) {
                g2l[(size_t)d] = (int32_t)p->own_global.size();
                p->own_global.push_back((int32_t)d);
            }
        const int64_t n_own = (int64_t)p->own_global.size();
        std::vector<uint32_t> needed_by((size_t)n_own, 0u); // bit q: rank q references my DOF
        std::vector<uint8_t> is_halo((size_t)n, 0);
        int64_t nnz_local = 0;
        for (int64_t r = 0; r < n; ++r) {
            const int q = owner[r];
            if (q < 0 || q >= world_size) {
                delete p;
                avs::set_error("avs_plan_create: owner[%lld] = %d out of range", (long long)r, q);
                return AVS_EINVAL;
            }
            if (q == rank) {
                nnz_local += row_ptr[r + 1] - row_ptr[r];
                for (int32_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k)
                    if (owner[col[k]] != rank) is_halo[(size_t)col[k]] = 1;
            } else {
                for (int32_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
                    const int32_t c = col[k];
                    if (owner[c] == rank) needed_by[(size_t)g2l[(size_t)c]] |= (1u << q);
                }
            }
        }
        // halo: grouped by owner, ascending global id inside a group
        std::vector<int64_t> recv_cnt((size_t)world_size, 0), send_cnt((size_t)world_size, 0);
        for (int q = 0; q < world_size; ++q) {
            if (q == rank) continue;
            for (int64_t d = 0; d < n; ++d)
                if (is_halo[(size_t)d] && owner[d] == q) {
                    g2l[(size_t)d] = (int32_t)(n_own + (int64_t)p->halo_global.size());
                    p->halo_global.push_back((int32_t)d);
                    ++recv_cnt[(size_t)q];
                }
        }
        // send lists: ascending owned DOFs needed by q (own_global is ascending)
        for (int q = 0; q < world_size; ++q) {
            if (q == rank) continue;
            for (int64_t l = 0; l < n_own; ++l)
                if (needed_by[(size_t)l] & (1u << q)) {
                    p->send_idx.push_back((int32_t)l);
                    ++send_cnt[(size_t)q];
                }
        }
        for (int q = 0; q < world_size; ++q)
            if (q != rank && (send_cnt[(size_t)q] || recv_cnt[(size_t)q])) {
                p->peers.push_back(q);
                p->send_counts.push_back((int32_t)send_cnt[(size_t)q]);
                p->recv_counts.push_back((int32_t)recv_cnt[(size_t)q]);
            }
        // local CSR (rows in ascending global id, columns remapped, in-row order unchanged)
        p->row_ptr_local.resize((size_t)n_own + 1);
        p->col_local.resize((size_t)nnz_local);
        p->val_src.resize((size_t)nnz_local);
        int64_t w = 0;
        for (int64_t l = 0; l < n_own; ++l) {
            const int64_t r = p->own_global[(size_t)l];
            p->row_ptr_local[(size_t)l] = (int32_t)w;
            for (int32_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
                p->col_local[(size_t)w] = g2l[(size_t)col[k]];
                p->val_src[(size_t)w] = k;
                ++w;
            }
        }
        p->row_ptr_local[(size_t)n_own] = (int32_t)w;
    } catch (const std::bad_alloc &) {
        delete p;
        avs::set_error("out of host memory");
        return AVS_ENOMEM;
    }
    *out = p;
    return AVS_OK;
}

avs_status avs_plan_get_sizes(const avs_plan *p, avs_plan_sizes *s)
{
    if (!p || !s) {
        avs::set_error("null argument");
        return AVS_EINVAL;
    }
    s->n_own = (int64_t)p->own_global.size();
    s->n_halo = (int64_t)p->halo_global.size();
    s->nnz_local = (int64_t)p->col_local.size();
    s->n_send = (int64_t)p->send_idx.size();
    s->n_peers = (int32_t)p->peers.size();
    return AVS_OK;
}

avs_status avs_plan_get_arrays(const avs_plan *p, int32_t *own_global, int32_t *halo_global, int32_t *row_ptr_local,
                               int32_t *col_local, int32_t *val_src, int32_t *peers, int32_t *send_counts,
                               int32_t *recv_counts, int32_t *send_idx)
{
    if (!p) {
        avs::set_error("null argument");
        return AVS_EINVAL;
    }
    auto put = [](int32_t *dst, const std::vector<int32_t> &v) {
        if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(int32_t));
    };
    put(own_global, p->own_global);
    put(halo_global, p->halo_global);
    put(row_ptr_local, p->row_ptr_local);
    put(col_local, p->col_local);
    put(val_src, p->val_src);
    put(peers, p->peers);
    put(send_counts, p->send_counts);
    put(recv_counts, p->recv_counts);
    put(send_idx, p->send_idx);
    return AVS_OK;
}

void avs_plan_destroy(avs_plan *p) { delete p; }

} // extern "C"
