// hnsw_locality.hip -- locality codes for the visited set's bitmap blocks (visited_blocks_mem_add, hnsw_device.hip.h).
//
// Visited (lib/ohnsw.ml:256-268) is one array slot per node in the reference; on the device a wave has 9 KiB for it.  As a
// cache of node tags that is 4096 nodes, a quarter of what a walk at ef 512 visits.  As BITS it is 65 536 nodes -- if the
// nodes a walk visits are numbered close together.  Node ids are insertion order and know nothing of the geometry, so the
// index gets a second numbering that does: the locality code L, a bijection [0, n) -> [0, n) derived from the index's OWN
// upper layers (nothing but the graph and the vectors is needed, for any metric):
//   * every node v looks for its nearest node on layer 2 with the index's own search (its own vector as the target: the greedy
//     descent of lib/ohnsw.ml:865-867 down to layer 2, then Ohnsw.search_k on that layer with a small W); layer 2 holds n / M^1.5
//     nodes: a "cell" of about M^1.5 nodes per layer-2 node, 181 at M = 32.  The nodes of layer l do the same on layer l + 1;
//   * the nodes of the top layer are numbered in id order; the nodes of layer l by (number of the layer-(l+1) node their descent
//     reached, id); finally all nodes by (number of their layer-2 node, id).
// Nodes of one cell are consecutive, cells of one layer-3 neighbourhood are consecutive, and so on up.  On clustered data a
// walk's visited set then fills some hundred blocks of 256 consecutive codes a third full (tools/visited_policy_sim.py).
// The kernels never see L as a node number: ids, tie order, row addresses are untouched; L only keys the visited set, through
// lcode0[c][j] = L[nbr0[c][j]] read beside the adjacency row.  Results cannot depend on it.
#include "hnsw_internal.h"

#include <numeric>

using namespace hnsw_host;

namespace {
__global__ void __launch_bounds__(256)
lcode0_fill_kernel(const int32_t *nbr0, const int32_t *lcode, int64_t total, int32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int32_t v = nbr0[i];
    out[i] = v >= 0 ? lcode[v] : 0;
}
} // namespace

namespace hnsw_host {

static int build_codes(::hnsw_index *idx);

int build_locality_codes(::hnsw_index *idx) {
    if (idx->lcode_state == 1) return materialise_lcode0(idx);      // built before (or adopted from a file): the large table may have been dropped
    if (idx->lcode_state != 0) return HNSW_OK;
    const int rc = build_codes(idx);
    if (rc != HNSW_OK && idx->lcode_state == 0) idx->lcode_state = -1;   // the error is reported ONCE; later searches keep the tag cache
    return rc;
}

static int build_codes(::hnsw_index *idx) {
    const int64_t n = idx->iv.n;
    const int T = idx->iv.max_layer;
    if (n < 2 || T < 1 || idx->iv.entry_point < 0) { idx->lcode_state = -1; return HNSW_OK; }
    HIP_TRY(hipSetDevice(idx->device));
    HIP_TRY(hipDeviceSynchronize());
    const int lo = std::min(2, T);
    const int ef_b = std::max(1, std::min(64, env_int("HNSW_LCODE_EF", 32)));
    const int id_base = idx->iv.id_base;
    DevBuf entry, scratch, best, bdist, qmap;
    struct Guard { DevBuf &a, &b, &c, &d, &e; ~Guard() { a.release(); b.release(); c.release(); d.release(); e.release(); } } guard{entry, scratch, best, bdist, qmap};
    int rc;
    if ((rc = entry.ensure((size_t)n * 4)) || (rc = scratch.ensure((size_t)n * 16)) || (rc = best.ensure((size_t)n * 4)) ||
        (rc = bdist.ensure((size_t)n * 4)) || (rc = qmap.ensure((size_t)n * 4)))
        return rc;
    std::vector<uint8_t> lvl_h((size_t)n);
    HIP_TRY(hipMemcpy(lvl_h.data(), idx->dLvl, (size_t)n, hipMemcpyDeviceToHost));
    // ent[l - lo][v]: the nearest node of layer l to v that the index's own search finds: the greedy descent (Ohnsw.search_one,
    // lib/ohnsw.ml:865-867) down to layer l, then Ohnsw.search_k on layer l from there with W bounded by ef_b, its nearest result.
    // (The greedy descent alone is not enough: on clustered data it ends in the wrong cluster's node for half of the vectors, and
    // neighbours scatter over different local minima.)  Layer `lo` is searched for every node, the layers above only for the
    // nodes of the layer below (the only ones whose number depends on it).
    std::vector<std::vector<int32_t>> ent((size_t)(T - lo + 1));
    std::vector<int32_t> list;
    for (int l = lo; l <= T; ++l) {
        if ((rc = descent_entries(idx, (const float *)idx->dX, n, idx->iv.stride, l, (int32_t *)entry.p, (uint32_t *)scratch.p, nullptr))) return rc;
        HIP_TRY(hipMemcpy(best.p, entry.p, (size_t)n * 4, hipMemcpyDeviceToDevice));     // (targets not searched below keep the descent's node)
        const int32_t *d_qmap = nullptr;
        int64_t n_launch = n;
        if (l > lo) {
            list.clear();
            for (int64_t v = 0; v < n; ++v) if ((int)lvl_h[(size_t)v] >= l - 1) list.push_back((int32_t)v);
            n_launch = (int64_t)list.size();
            if (n_launch > 0) HIP_TRY(hipMemcpy(qmap.p, list.data(), list.size() * 4, hipMemcpyHostToDevice));
            d_qmap = (const int32_t *)qmap.p;
        }
        if (ef_b > 1 && (rc = layer_nearest_device(idx, l, (const float *)idx->dX, idx->iv.stride, n, d_qmap, n_launch, (const int32_t *)entry.p, ef_b,
                                                   (int32_t *)best.p, (float *)bdist.p))) return rc;
        ent[(size_t)(l - lo)].resize((size_t)n);
        HIP_TRY(hipMemcpy(ent[(size_t)(l - lo)].data(), best.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (ef_b > 1) {        // the layer search reports id_base-based ids; the descent 0-based ones (kept where nothing was searched)
            std::vector<int32_t> &e = ent[(size_t)(l - lo)];
            if (l > lo) { for (int32_t v : list) e[(size_t)v] -= id_base; }
            else for (int64_t v = 0; v < n; ++v) e[(size_t)v] -= id_base;
        }
    }
    const std::vector<uint8_t> &lvl = lvl_h;
    // numbers of the nodes of layer l (rank), from the top down; a node that is missing from the layer its descent "reached"
    // cannot occur (the descent moves along that layer's rows from the entry point, which is on every layer), but a bad
    // number would only cost locality, so it is clamped rather than trusted
    std::vector<int32_t> rank((size_t)n, 0), next((size_t)n, 0);
    std::vector<int64_t> keys;
    auto number_layer = [&](int l, const std::vector<int32_t> *parent) {
        keys.clear();
        for (int64_t v = 0; v < n; ++v)
            if ((int)lvl[(size_t)v] >= l) {
                const int64_t pr = parent ? (int64_t)rank[(size_t)std::max<int32_t>(0, std::min<int32_t>((int32_t)n - 1, (*parent)[(size_t)v]))] : 0;
                keys.push_back((pr << 32) | v);
            }
        std::sort(keys.begin(), keys.end());
        for (size_t i = 0; i < keys.size(); ++i) next[(size_t)(keys[i] & 0xFFFFFFFFll)] = (int32_t)i;
        rank.swap(next);
    };
    number_layer(T, nullptr);
    for (int l = T - 1; l >= lo; --l) number_layer(l, &ent[(size_t)(l + 1 - lo)]);
    number_layer(0, &ent[0]);                       // all nodes, by (number of their layer-`lo` node, id): the codes
    // the per-node table (n * 4 bytes) stays with the handle from here on; the per-slot table (n * max_degree0 * 4 bytes: the large
    // one) is made from it by materialise_lcode0 and can be dropped and made again in milliseconds (drop_lcode0)
    void *dL = nullptr;
    if (hipMalloc(&dL, (size_t)n * 4) != hipSuccess) {
        (void)hipGetLastError();
        idx->lcode_state = -1;                      // no room: the tag cache stays (not an error of the search)
        return HNSW_OK;
    }
    if (hipMemcpy(dL, rank.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(dL);
        idx->lcode_state = -1;                      // a failed build is not retried by every later search
        return fail(HNSW_ERR_HIP, "upload of the locality codes failed");
    }
    idx->dLcode = dL;
    idx->iv.lcode = (const int32_t *)dL;
    idx->info.device_bytes += (int64_t)n * 4;
    idx->lcode_state = 1;
    return materialise_lcode0(idx);
}

// lcode0[c][j] = L[nbr0[c][j]], the table the block filter reads beside the adjacency row: n * max_degree0 * 4 bytes
// (C5's shape: 2.56 GB), counted in hnsw_index_info.device_bytes while it exists
int materialise_lcode0(::hnsw_index *idx) {
    if (idx->lcode_state != 1 || idx->dLcode0) return HNSW_OK;
    HIP_TRY(hipSetDevice(idx->device));
    const size_t total = (size_t)idx->iv.n * (size_t)idx->iv.S0;
    void *dL0 = nullptr;
    if (hipMalloc(&dL0, std::max<size_t>(total, 1) * 4) != hipSuccess) {
        (void)hipGetLastError();
        idx->lcode_state = -1;                      // no room for the large table: the tag cache stays
        (void)hipFree(idx->dLcode);
        idx->info.device_bytes -= (int64_t)idx->iv.n * 4;
        idx->dLcode = nullptr; idx->iv.lcode = nullptr;
        return HNSW_OK;
    }
    hipLaunchKernelGGL(lcode0_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr,
                       (const int32_t *)idx->dNbr0, (const int32_t *)idx->dLcode, (int64_t)total, (int32_t *)dL0);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(dL0);
        idx->lcode_state = -1;
        return fail(HNSW_ERR_HIP, "filling the per-slot locality codes failed: %s", hipGetErrorString(e));
    }
    idx->dLcode0 = dL0;
    idx->iv.lcode0 = (const int32_t *)dL0;
    idx->info.device_bytes += (int64_t)total * 4;
    return HNSW_OK;
}

// frees the per-slot table again (no launch that reads it may be in flight: the caller has synchronised); the per-node codes stay
void drop_lcode0(::hnsw_index *idx) {
    if (!idx->dLcode0) return;
    (void)hipSetDevice(idx->device);
    (void)hipFree(idx->dLcode0);
    idx->dLcode0 = nullptr; idx->iv.lcode0 = nullptr;
    idx->info.device_bytes -= (int64_t)idx->iv.n * (int64_t)idx->iv.S0 * 4;
}

// the codes of a saved index (hnsw_index_load): adopted instead of built
int adopt_locality_codes(::hnsw_index *idx, const int32_t *codes) {
    if (idx->lcode_state != 0) return HNSW_OK;
    const int64_t n = idx->iv.n;
    HIP_TRY(hipSetDevice(idx->device));
    void *dL = nullptr;
    if (hipMalloc(&dL, (size_t)std::max<int64_t>(n, 1) * 4) != hipSuccess) { (void)hipGetLastError(); return HNSW_OK; }   // built later, if ever
    if (hipMemcpy(dL, codes, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(dL); return HNSW_OK; }
    idx->dLcode = dL;
    idx->iv.lcode = (const int32_t *)dL;
    idx->info.device_bytes += n * 4;
    idx->lcode_state = 1;
    return HNSW_OK;
}

} // namespace hnsw_host
