// hnsw_multi.hip -- one host process, several GPUs: the index replicated on every listed
// device, a query batch split into contiguous shards, one shard per device, and ONE RCCL all-gather
// (ncclAllGather over xGMI) that leaves the full [nq][k] result resident on every device.
// Each query is an independent read-only traversal (lib/ohnsw.ml:883-895 is a pure map over the
// batch's columns), so the concatenated shard results are the single-device results.
//
// RCCL is bound at first use (dlopen of librccl.so.1 -- the soname PyTorch-ROCm's bundled copy also
// answers to, so a process ends up with one RCCL): a program that searches on one GPU never maps
// the 570 MB library.  Communicators come from ncclCommInitAll over the listed devices (one process,
// one communicator per device, collectives issued inside ncclGroupStart/End).
#include "hnsw_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

using namespace hnsw_host;

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

int load_rccl(RcclApi &api) {
    if (api.lib) return HNSW_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(HNSW_ERR_HIP, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
#define HNSW_RCCL_SYM(field, sym)                                                             \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, sym));                          \
    if (!api.field) return fail(HNSW_ERR_HIP, "RCCL symbol %s missing", sym)
    HNSW_RCCL_SYM(CommInitAll, "ncclCommInitAll");
    HNSW_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    HNSW_RCCL_SYM(CommAbort, "ncclCommAbort");
    HNSW_RCCL_SYM(AllGather, "ncclAllGather");
    HNSW_RCCL_SYM(Broadcast, "ncclBroadcast");
    HNSW_RCCL_SYM(GroupStart, "ncclGroupStart");
    HNSW_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    HNSW_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef HNSW_RCCL_SYM
    api.lib = h;
    return HNSW_OK;
}

// shard g of G over nq queries: [g*nq/G, (g+1)*nq/G) -- the same bounds as sharding.shard_bounds
inline int64_t shard_lo(int64_t nq, int g, int G) { return (int64_t)((__int128)nq * g / G); }

} // namespace

struct hnsw_multi {
    std::vector<hnsw_index *> replicas;
    std::vector<int> devices;
    bool distinct = true;                 // every device listed once: the RCCL path; otherwise (several replicas on
                                          // one device: tests) the gather is device-to-device copies
    RcclApi rccl;
    std::vector<ncclComm_t> comms;        // [G], created at the first device-resident search
    std::vector<hipStream_t> streams;     // [G]
    std::vector<hipEvent_t> done;         // [G] "this device's search is enqueued up to here" (the same-device copy arrangement)
    std::vector<DevBuf> dQ, dIds, dDist, dNd, dNh, dSt;   // per device: its query shard; the FULL [nq][k] result; per-shard counters
    int64_t last_nq = 0; int last_k = 0;
    std::vector<DevBuf> dFlag;            // per device: the launch's "any query flagged" word (see hnsw_search_batch)
    uint32_t *hFlags = nullptr;           // [G] pinned host words the flags are copied into
    // what the exchanges of this handle were made of (hnsw_multi_debug_counters): calls issued, summed over the devices
    int64_t n_allgather = 0, n_broadcast = 0, n_peer_copies = 0, n_repaired_shards = 0;
};

namespace {

#define RCCL_TRY(m, expr)                                                                     \
    do {                                                                                      \
        ncclResult_t r__ = (expr);                                                            \
        if (r__ != ncclSuccess) return fail(HNSW_ERR_HIP, "%s failed: %s", #expr, (m)->rccl.GetErrorString(r__)); \
    } while (0)

int ensure_streams(hnsw_multi *m) {
    const size_t G = m->replicas.size();
    if (m->streams.size() == G) return HNSW_OK;
    m->streams.assign(G, nullptr);
    m->dQ.resize(G); m->dIds.resize(G); m->dDist.resize(G); m->dNd.resize(G); m->dNh.resize(G); m->dSt.resize(G); m->dFlag.resize(G);
    if (!m->hFlags) HIP_TRY(hipHostMalloc((void **)&m->hFlags, G * sizeof(uint32_t), hipHostMallocPortable));
    for (size_t g = 0; g < G; ++g) {
        HIP_TRY(hipSetDevice(m->devices[g]));
        HIP_TRY(hipStreamCreateWithFlags(&m->streams[g], hipStreamNonBlocking));
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        m->done.push_back(e);
    }
    return HNSW_OK;
}

int ensure_comms(hnsw_multi *m) {
    if (!m->distinct || !m->comms.empty()) return HNSW_OK;
    int rc = load_rccl(m->rccl);
    if (rc) return rc;
    m->comms.assign(m->replicas.size(), nullptr);
    ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), (int)m->devices.size(), m->devices.data());
    if (r != ncclSuccess) {
        m->comms.clear();
        return fail(HNSW_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", (int)m->devices.size(), m->rccl.GetErrorString(r));
    }
    return HNSW_OK;
}

int sync_all(hnsw_multi *m);

// An exchange that was only PARTLY enqueued (one device's collective refused inside the group) must not be left standing:
// the devices whose collectives were accepted would wait on their streams for a peer that never joins -- harmless at
// communicator size 1, a hang of hnsw_multi_search_batch on a real node.  ncclCommAbort ends whatever those communicators
// have in flight (their kernels leave, the streams drain); the communicators are gone afterwards and the next search on
// the handle creates new ones (ensure_comms).  The call that hit the error returns it; its result tables are undefined.
void abort_comms(hnsw_multi *m) {
    for (size_t g = 0; g < m->comms.size(); ++g) {
        if (!m->comms[g]) continue;
        (void)hipSetDevice(m->devices[g]);
        (void)m->rccl.CommAbort(m->comms[g]);
    }
    m->comms.clear();
}

// Sharded search + gather: on return every device's dIds / dDist hold the full [nq][k] result.
int search_and_gather(hnsw_multi *m, const float *queries, int64_t nq, int64_t q_stride, const hnsw_search_params *params,
                      uint32_t *out_ndist, uint32_t *out_nhops) {
    const int G = (int)m->replicas.size();
    const int k = params->k;
    int rc;
    if ((rc = ensure_streams(m)) || (rc = ensure_comms(m))) return rc;
    const int d = m->replicas[0]->iv.d;
    const size_t full = (size_t)nq * k;
    // ---- per device: upload the shard, search it (results straight into the shard's slice of the full table).  Nothing
    //      below waits on the host until every device has its search AND the exchange enqueued: whether a shard needs
    //      the exactness fallback comes back as one word per device, read after the exchange ----
    for (int g = 0; g < G; ++g) {
        const int64_t lo = shard_lo(nq, g, G), hi = shard_lo(nq, g + 1, G);
        hnsw_index *idx = m->replicas[(size_t)g];
        HIP_TRY(hipSetDevice(m->devices[(size_t)g]));
        if ((rc = m->dIds[(size_t)g].ensure(std::max<size_t>(full, 1) * 4)) || (rc = m->dDist[(size_t)g].ensure(std::max<size_t>(full, 1) * 4)) ||
            (rc = m->dFlag[(size_t)g].ensure(16)))
            return rc;
        m->hFlags[g] = 0;
        if (hi <= lo) continue;
        const int64_t ns = hi - lo;
        const size_t qbytes = ((size_t)(ns - 1) * q_stride + d) * sizeof(float);
        if ((rc = m->dQ[(size_t)g].ensure(qbytes)) || (rc = m->dNd[(size_t)g].ensure((size_t)ns * 4)) ||
            (rc = m->dNh[(size_t)g].ensure((size_t)ns * 4)) || (rc = m->dSt[(size_t)g].ensure((size_t)ns * 4)))
            return rc;
        hipStream_t st = m->streams[(size_t)g];
        HIP_TRY(hipMemsetAsync(m->dFlag[(size_t)g].p, 0, 4, st));
        HIP_TRY(hipMemcpyAsync(m->dQ[(size_t)g].p, queries + lo * q_stride, qbytes, hipMemcpyHostToDevice, st));
        rc = search_batch_device_flag(idx, (const float *)m->dQ[(size_t)g].p, ns, q_stride, params,
                                      (int32_t *)m->dIds[(size_t)g].p + lo * k, (float *)m->dDist[(size_t)g].p + lo * k,
                                      (uint32_t *)m->dNd[(size_t)g].p, (uint32_t *)m->dNh[(size_t)g].p, (uint32_t *)m->dSt[(size_t)g].p,
                                      (uint32_t *)m->dFlag[(size_t)g].p, st);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(&m->hFlags[g], m->dFlag[(size_t)g].p, 4, hipMemcpyDeviceToHost, st));
    }
    // ---- the exchange: every device receives every shard (shards [r_lo, r_hi) only: the whole table, or, after the
    //      fallback, the repaired shards again) ----
    auto exchange = [&](int only_shard) -> int {
        if (!(G > 1 || m->distinct)) return HNSW_OK;
        if (m->distinct) {
            const bool equal = nq % G == 0 && only_shard < 0;
            RCCL_TRY(m, m->rccl.GroupStart());
            int err = HNSW_OK;
            // (tests) HNSW_MULTI_FAIL_ENQUEUE=g: device g's enqueue is answered as refused, after the devices before it were accepted
            const int fail_at = env_int("HNSW_MULTI_FAIL_ENQUEUE", -1);
            for (int g = 0; g < G && !err; ++g) {
                if (hipSetDevice(m->devices[(size_t)g]) != hipSuccess) { err = fail(HNSW_ERR_HIP, "hipSetDevice failed"); break; }
                if (g == fail_at) { err = fail(HNSW_ERR_HIP, "RCCL exchange failed: enqueue on device %d refused (HNSW_MULTI_FAIL_ENQUEUE)", g); break; }
                int32_t *ids = (int32_t *)m->dIds[(size_t)g].p;
                float *dd = (float *)m->dDist[(size_t)g].p;
                ncclResult_t r1 = ncclSuccess;
                if (equal) {       // in place: the send buffer is this rank's slice of the receive buffer
                    const size_t cnt = (size_t)(nq / G) * k;
                    r1 = m->rccl.AllGather(ids + (size_t)g * cnt, ids, cnt, ncclInt32, m->comms[(size_t)g], m->streams[(size_t)g]);
                    if (r1 == ncclSuccess) r1 = m->rccl.AllGather(dd + (size_t)g * cnt, dd, cnt, ncclFloat32, m->comms[(size_t)g], m->streams[(size_t)g]);
                    if (r1 == ncclSuccess) m->n_allgather += 2;
                } else {           // unequal shards: one in-place broadcast per shard (all-gather-v)
                    for (int r = 0; r < G && r1 == ncclSuccess; ++r) {
                        if (only_shard >= 0 && r != only_shard) continue;
                        const int64_t lo = shard_lo(nq, r, G), hi = shard_lo(nq, r + 1, G);
                        if (hi <= lo) continue;
                        r1 = m->rccl.Broadcast(ids + lo * k, ids + lo * k, (size_t)(hi - lo) * k, ncclInt32, r, m->comms[(size_t)g], m->streams[(size_t)g]);
                        if (r1 == ncclSuccess) r1 = m->rccl.Broadcast(dd + lo * k, dd + lo * k, (size_t)(hi - lo) * k, ncclFloat32, r, m->comms[(size_t)g], m->streams[(size_t)g]);
                        if (r1 == ncclSuccess) m->n_broadcast += 2;
                    }
                }
                if (r1 != ncclSuccess) err = fail(HNSW_ERR_HIP, "RCCL exchange failed: %s", m->rccl.GetErrorString(r1));
            }
            // the group is ALWAYS closed: an early return between GroupStart and GroupEnd would leave this thread
            // inside an open group and every later call on these communicators would misbehave
            ncclResult_t re = m->rccl.GroupEnd();
            if (err || re != ncclSuccess) {
                // part of the exchange may stand enqueued with no peer to meet: end it (see abort_comms) before anybody waits
                const std::string msg = err ? std::string(hnsw_last_error()) : std::string("ncclGroupEnd failed: ") + m->rccl.GetErrorString(re);
                abort_comms(m);
                return fail(HNSW_ERR_HIP, "%s; the communicators were aborted and are re-created by the next search", msg.c_str());
            }
        } else {
            // several replicas on one device (a test arrangement): RCCL refuses duplicate devices, and the
            // "exchange" between buffers of the same device is a device-to-device copy
            for (int g = 0; g < G; ++g) {
                HIP_TRY(hipSetDevice(m->devices[(size_t)g]));
                for (int r = 0; r < G; ++r) {
                    if (r == g || (only_shard >= 0 && r != only_shard)) continue;
                    const int64_t lo = shard_lo(nq, r, G), hi = shard_lo(nq, r + 1, G);
                    if (hi <= lo) continue;
                    // the copy runs on the RECEIVER's stream: it must not start before the sender's search is done
                    if (only_shard < 0) {
                        hipEvent_t ev = m->done[(size_t)r];
                        HIP_TRY(hipStreamWaitEvent(m->streams[(size_t)g], ev, 0));
                    }
                    HIP_TRY(hipMemcpyPeerAsync((int32_t *)m->dIds[(size_t)g].p + lo * k, m->devices[(size_t)g], (int32_t *)m->dIds[(size_t)r].p + lo * k,
                                               m->devices[(size_t)r], (size_t)(hi - lo) * k * 4, m->streams[(size_t)g]));
                    HIP_TRY(hipMemcpyPeerAsync((float *)m->dDist[(size_t)g].p + lo * k, m->devices[(size_t)g], (float *)m->dDist[(size_t)r].p + lo * k,
                                               m->devices[(size_t)r], (size_t)(hi - lo) * k * 4, m->streams[(size_t)g]));
                    m->n_peer_copies += 2;
                }
            }
        }
        return HNSW_OK;
    };
    if (!m->distinct) {   // the copy arrangement orders receiver streams behind sender streams with one event per device
        for (int g = 0; g < G; ++g) {
            HIP_TRY(hipSetDevice(m->devices[(size_t)g]));
            HIP_TRY(hipEventRecord(m->done[(size_t)g], m->streams[(size_t)g]));
        }
    }
    if ((rc = exchange(-1))) return rc;
    // ---- now the one host round trip: wait, read the flag words; a flagged shard (rare) is searched again with the
    //      global slab and its slice of the table is sent round once more ----
    if ((rc = sync_all(m))) return rc;
    for (int g = 0; g < G; ++g) {
        const int64_t lo = shard_lo(nq, g, G), hi = shard_lo(nq, g + 1, G);
        if (hi <= lo) continue;
        hnsw_index *idx = m->replicas[(size_t)g];
        HIP_TRY(hipSetDevice(m->devices[(size_t)g]));
        if (m->hFlags[g] & 1u) {
            rc = rerun_overflowed(idx, hi - lo, (const uint32_t *)m->dSt[(size_t)g].p,
                                  [&](const int32_t *qmap, int64_t c, uint32_t *slab, int32_t cap) {
                                      return search_rerun_device(idx, (const float *)m->dQ[(size_t)g].p, hi - lo, q_stride, params,
                                                                 (int32_t *)m->dIds[(size_t)g].p + lo * k, (float *)m->dDist[(size_t)g].p + lo * k,
                                                                 (uint32_t *)m->dNd[(size_t)g].p, (uint32_t *)m->dNh[(size_t)g].p,
                                                                 (uint32_t *)m->dSt[(size_t)g].p, qmap, c, slab, cap, nullptr);
                                  });
            if (rc) return rc;
            m->n_repaired_shards++;
            if ((rc = exchange(g))) return rc;
        }
        if (out_ndist) HIP_TRY(hipMemcpyAsync(out_ndist + lo, m->dNd[(size_t)g].p, (size_t)(hi - lo) * 4, hipMemcpyDeviceToHost, m->streams[(size_t)g]));
        if (out_nhops) HIP_TRY(hipMemcpyAsync(out_nhops + lo, m->dNh[(size_t)g].p, (size_t)(hi - lo) * 4, hipMemcpyDeviceToHost, m->streams[(size_t)g]));
    }
    m->last_nq = nq; m->last_k = k;
    return HNSW_OK;
}

int sync_all(hnsw_multi *m) {
    for (size_t g = 0; g < m->replicas.size(); ++g) {
        HIP_TRY(hipSetDevice(m->devices[g]));
        HIP_TRY(hipStreamSynchronize(m->streams[g]));
    }
    return HNSW_OK;
}

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
        m->devices.push_back(devices[g]);
        for (int h = 0; h < g; ++h) if (devices[h] == devices[g]) m->distinct = false;
    }
    *out = m;
    return HNSW_OK;
}

int32_t hnsw_multi_destroy(hnsw_multi *m) {
    if (!m) return HNSW_OK;
    for (size_t g = 0; g < m->streams.size(); ++g) {
        (void)hipSetDevice(m->devices[g]);
        if (m->streams[g]) { (void)hipStreamSynchronize(m->streams[g]); }
    }
    for (ncclComm_t c : m->comms) if (c) (void)m->rccl.CommDestroy(c);
    for (size_t g = 0; g < m->streams.size(); ++g) {
        (void)hipSetDevice(m->devices[g]);
        m->dQ[g].release(); m->dIds[g].release(); m->dDist[g].release(); m->dNd[g].release(); m->dNh[g].release(); m->dSt[g].release(); m->dFlag[g].release();
        if (m->streams[g]) (void)hipStreamDestroy(m->streams[g]);
        if (g < m->done.size()) (void)hipEventDestroy(m->done[g]);
    }
    if (m->hFlags) (void)hipHostFree(m->hFlags);
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

int32_t hnsw_multi_search_batch_device(hnsw_multi *m, const float *queries, int64_t nq, int64_t q_stride,
                                       const hnsw_search_params *params, int32_t **d_ids, float **d_dist) {
    if (!m || m->replicas.empty()) return fail(HNSW_ERR_BAD_ARG, "null hnsw_multi");
    if (!params) return fail(HNSW_ERR_BAD_ARG, "null params");
    if (nq < 1 || !queries) return fail(HNSW_ERR_BAD_ARG, "bad buffers (nq=%lld)", (long long)nq);
    if (q_stride < m->replicas[0]->iv.d) return fail(HNSW_ERR_BAD_ARG, "q_stride < d");
    {   // parameter errors are reported before any allocation sized by them
        int rcp = search_check(m->replicas[0], params);
        if (rcp) return rcp;
    }
    int rc = search_and_gather(m, queries, nq, q_stride, params, nullptr, nullptr);
    if (rc) { (void)sync_all(m); return rc; }
    if ((rc = sync_all(m))) return rc;
    for (size_t g = 0; g < m->replicas.size(); ++g) {
        if (d_ids) d_ids[g] = (int32_t *)m->dIds[g].p;
        if (d_dist) d_dist[g] = (float *)m->dDist[g].p;
    }
    return HNSW_OK;
}

int32_t hnsw_multi_debug_counters(const hnsw_multi *m, int64_t *out4) {
    if (!m || !out4) return fail(HNSW_ERR_BAD_ARG, "null argument");
    out4[0] = m->n_allgather; out4[1] = m->n_broadcast; out4[2] = m->n_peer_copies; out4[3] = m->n_repaired_shards;
    return HNSW_OK;
}

int32_t hnsw_multi_copy_result(hnsw_multi *m, int32_t g, int32_t *out_ids, float *out_dist) {
    if (!m || !out_ids || !out_dist) return fail(HNSW_ERR_BAD_ARG, "null argument");
    if (g < 0 || g >= (int32_t)m->replicas.size()) return fail(HNSW_ERR_BAD_ARG, "replica %d out of range", g);
    if (m->last_nq < 1 || m->streams.empty()) return fail(HNSW_ERR_BAD_ARG, "no device-resident result yet");
    HIP_TRY(hipSetDevice(m->devices[(size_t)g]));
    const size_t bytes = (size_t)m->last_nq * m->last_k * 4;
    HIP_TRY(hipMemcpy(out_ids, m->dIds[(size_t)g].p, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_dist, m->dDist[(size_t)g].p, bytes, hipMemcpyDeviceToHost));
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
    if (q_stride < m->replicas[0]->iv.d) return fail(HNSW_ERR_BAD_ARG, "q_stride < d");
    {   // parameter errors are reported before any device work (same text as the single-device call)
        int rcp = search_check(m->replicas[0], params);
        if (rcp) return rcp;
    }
    const int k = params->k;
    const size_t full = (size_t)nq * k * 4;
    int rc = search_and_gather(m, queries, nq, q_stride, params, out_ndist, out_nhops);
    if (rc) { (void)sync_all(m); return rc; }
    // the host gets the table from ONE device (it is complete everywhere after the exchange)
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(hipMemcpyAsync(out_ids, m->dIds[0].p, full, hipMemcpyDeviceToHost, m->streams[0]));
    HIP_TRY(hipMemcpyAsync(out_dist, m->dDist[0].p, full, hipMemcpyDeviceToHost, m->streams[0]));
    return sync_all(m);
}

} // extern "C"
