// hnsw_multi.hip -- one host process, several GPUs: the index replicated on every
// listed device, a query batch split into contiguous shards, one host thread per device.
// Each query is an independent read-only traversal (lib/ohnsw.ml:883-895 is a pure map over the
// batch's columns), so the concatenated shard results are the single-device results.
#include "hnsw_internal.h"

#include <thread>

using namespace hnsw_host;

struct hnsw_multi {
    std::vector<hnsw_index *> replicas;
};

namespace {
// shard g of G over nq queries: [g*nq/G, (g+1)*nq/G) -- the same bounds as sharding.shard_bounds
inline int64_t shard_lo(int64_t nq, int g, int G) { return (int64_t)((__int128)nq * g / G); }
} // namespace

extern "C" {

int32_t hnsw_multi_create(const hnsw_index_desc *desc, const int32_t *devices, int32_t n_devices, hnsw_multi **out) {
    if (!desc || !devices || !out) return fail(HNSW_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return fail(HNSW_ERR_BAD_ARG, "n_devices=%d must be in 1..64", n_devices);
    hnsw_multi *m = new hnsw_multi();
    for (int g = 0; g < n_devices; ++g) {
        hnsw_index *idx = nullptr;
        const int rc = hnsw_index_create(desc, devices[g], &idx);
        if (rc) { hnsw_multi_destroy(m); return rc; }   // the message of the failing create stays
        m->replicas.push_back(idx);
    }
    *out = m;
    return HNSW_OK;
}

int32_t hnsw_multi_destroy(hnsw_multi *m) {
    if (!m) return HNSW_OK;
    for (hnsw_index *idx : m->replicas) (void)hnsw_index_destroy(idx);
    delete m;
    return HNSW_OK;
}

int32_t hnsw_multi_num_replicas(const hnsw_multi *m, int32_t *n_devices) {
    if (!m || !n_devices) return fail(HNSW_ERR_BAD_ARG, "null argument");
    *n_devices = (int32_t)m->replicas.size();
    return HNSW_OK;
}

int32_t hnsw_multi_replica(hnsw_multi *m, int32_t g, hnsw_index **out) {
    if (!m || !out) return fail(HNSW_ERR_BAD_ARG, "null argument");
    if (g < 0 || g >= (int32_t)m->replicas.size()) return fail(HNSW_ERR_BAD_ARG, "replica %d out of range", g);
    *out = m->replicas[(size_t)g];
    return HNSW_OK;
}

int32_t hnsw_multi_search_batch(hnsw_multi *m, const float *queries, int64_t nq, int64_t q_stride,
                                const hnsw_search_params *params, int32_t *out_ids, float *out_dist,
                                uint32_t *out_ndist, uint32_t *out_nhops) {
    if (!m || m->replicas.empty()) return fail(HNSW_ERR_BAD_ARG, "null hnsw_multi");
    if (!params) return fail(HNSW_ERR_BAD_ARG, "null params");
    if (nq < 0) return fail(HNSW_ERR_BAD_ARG, "nq < 0");
    const int G = (int)m->replicas.size();
    if (G == 1 || nq == 0)
        return hnsw_search_batch(m->replicas[0], queries, nq, q_stride, params, out_ids, out_dist, out_ndist, out_nhops);
    if (!queries || !out_ids || !out_dist) return fail(HNSW_ERR_BAD_ARG, "bad buffers");
    const int k = params->k;
    std::vector<int> rcs((size_t)G, HNSW_OK);
    std::vector<std::string> msgs((size_t)G);
    std::vector<std::thread> workers;
    for (int g = 0; g < G; ++g) {
        const int64_t lo = shard_lo(nq, g, G), hi = shard_lo(nq, g + 1, G);
        if (hi <= lo) continue;
        workers.emplace_back([=, &rcs, &msgs]() {
            // hnsw_last_error is per thread: carry the message back to the caller's thread
            const int rc = hnsw_search_batch(m->replicas[(size_t)g], queries + lo * q_stride, hi - lo, q_stride, params,
                                             out_ids + lo * k, out_dist + lo * k,
                                             out_ndist ? out_ndist + lo : nullptr, out_nhops ? out_nhops + lo : nullptr);
            rcs[(size_t)g] = rc;
            if (rc) msgs[(size_t)g] = hnsw_last_error();
        });
    }
    for (std::thread &t : workers) t.join();
    for (int g = 0; g < G; ++g)
        if (rcs[(size_t)g]) return fail(rcs[(size_t)g], "replica %d: %s", g, msgs[(size_t)g].c_str());
    return HNSW_OK;
}

} // extern "C"
