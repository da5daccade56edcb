// hnsw_internal.h -- shared host-side declarations of libhnsw_mi355x.so (not part of the ABI).
#pragma once
#include "../../include/hnsw_mi355x.h"
#include "hnsw_device.hip.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// 4-row batches in flight per wave for d <= 128 (NCH = 2): measured on C2: 4 (6 waves/SIMD) >= 3 (7) >= 2 (8): the batch is memory-bound, not occupancy-bound
#ifndef HNSW_RB_NCH2
#define HNSW_RB_NCH2 4
#endif

namespace hnsw_host {

int fail(int code, const char *fmt, ...);

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return ::hnsw_host::fail(e__ == hipErrorOutOfMemory ? HNSW_ERR_OOM : HNSW_ERR_HIP, \
                                     "%s failed: %s", #expr, hipGetErrorString(e__));        \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return HNSW_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        HIP_TRY(hipMalloc(&p, bytes));
        cap = bytes;
        return HNSW_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

inline int env_int(const char *name, int dflt) {
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}
inline int pick_nch(int nchunks) {
    const int per_lane = (nchunks + 15) / 16;
    for (int c : {1, 2, 4, 8, 16}) if (per_lane <= c) return c;
    return 0;
}
inline int pick_nslot(int ef) {
    for (int s : {1, 2, 4, 8, 16}) if (ef <= 64 * s) return s;
    return 0;
}
// ... of the knn kernel: rows of 65..256 dimensions (NCH 2 and 4: the shapes with hand-scheduled loops) also have W in three
// (ef 129..192) and six (ef 257..384) registers -- a W window that needs three registers pays for three (pop chain, flag masks,
// registers), not for four; the other row widths, the builder and the layer operators keep powers of two (pick_nslot)
inline int pick_nslot_knn(int ef, int nch) {
    static const int pow2 = env_int("HNSW_NSLOT_POW2", 0);      // (A/B: W in powers of two only, as rounds 1-5: profiles/r06_hop_phases.txt)
    if (!pow2 && (nch == 2 || nch == 4)) { for (int s : {1, 2, 3, 4, 6, 8, 16}) if (ef <= 64 * s) return s; return 0; }
    return pick_nslot(ef);
}
// index of a slot count in per-shape tables (hnsw_index::blk_choice)
inline int slot_class(int nslot) {
    switch (nslot) { case 1: return 0; case 2: return 1; case 3: return 2; case 4: return 3; case 6: return 4; case 8: return 5; default: return 6; }
}
constexpr int SLOT_CLASSES = 7;
inline int64_t padded_stride(int d) { return ((int64_t)d + 15) / 16 * 16; } // floats: rows are multiples of 64 B

// {upper_off, upper_lvl} of every node side by side (IndexView::upper_ref) from the two host tables
int upload_upper_ref(const int32_t *off, const uint8_t *lvl, int64_t n, void **dRef);

// copies [n][row_stride] host rows into a fresh zero-padded device table
int upload_vectors(const float *vectors, int64_t n, int d, int64_t row_stride, void **dX, size_t *bytes);

} // namespace hnsw_host

struct hnsw_index;
// one submitted batch (hnsw_search_submit / hnsw_search_wait): its own device buffers and stream
struct hnsw_request {
    hnsw_index *idx = nullptr;
    int64_t nq = 0, q_stride = 0;
    hnsw_search_params params{};
    hnsw_host::DevBuf q, ids, dist, nd, nh, st, flag;
    int stream = 0;
    uint32_t host_flag = 0;
};

struct hnsw_index {
    int device = -1;
    hnsw_dev::IndexView iv{};
    hnsw_index_info info{};
    void *dX8 = nullptr;                 // byte rows (hnsw_rows8.hip), nullptr when the data does not qualify
    void *dXm = nullptr, *dTail0 = nullptr; // split rows (hnsw_rows_split.hip), nullptr when the row shape does not qualify
    // locality codes (hnsw_locality.hip): per node and per layer-0 adjacency slot; built when the visited set first runs as
    // bitmap blocks.  lcode_state: 0 not built yet, 1 built, -1 cannot be built (no upper layer to derive an order from)
    void *dLcode = nullptr, *dLcode0 = nullptr;
    int lcode_state = 0;
    int blk_mode = -1;                   // option "visited_blocks": -1 automatic (measured per kernel shape on the index's own vectors), 0 never, 1 always
    int blk_choice[7][2] = {{-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}};   // [slot_class(NSLOT)][accept rule]: -1 undecided, 0 tag cache, else log2 of the block slots
    void *dX = nullptr, *dNbr0 = nullptr, *dNbrU = nullptr, *dOff = nullptr, *dLvl = nullptr, *dRef = nullptr;
    int64_t rowsU = 0;
    hnsw_host::DevBuf sQ, sIds, sDist, sNd, sNh, sSt, sFlag; // scratch for the host-buffer entry points (sFlag: the launch's "any query flagged" word)
    uint32_t *hFlag = nullptr, *hFlagDev = nullptr;          // the same word in pinned host memory (zero-copy calls) and its device address
    char *hSmall = nullptr, *hSmallDev = nullptr;            // page-locked block for small host-buffer calls (hnsw_search_batch: queries, ids, distances, counters)
    // option "device_fallback_slab_bytes": a slab of the caller's chosen size for the exactness fallback of
    // hnsw_search_batch_device, run on the caller's stream without a host round trip (dFbMap: the flagged queries' list)
    hnsw_host::DevBuf dFbSlab, dFbMap;
    int64_t fb_queries = 0;                                  // how many flagged queries one launch can repair (slab bytes / (4 n))
    hipStream_t hs[4] = {nullptr, nullptr, nullptr, nullptr}; // streams of the chunked host-buffer search and of requests (lazy)
    std::vector<hnsw_request *> free_requests;               // finished requests keep their buffers for the next submit
    std::vector<hnsw_request *> all_requests;                // every request ever created (released with the index)
    int live_requests = 0, next_stream = 0;
    int64_t resident_queries = 0;        // how many one-wave workgroups of the search kernel the chip holds (0 = not measured yet)
    int resident_per_cu = 0, cus = 0;    // ... per CU, and the CUs
    int64_t debug_last_nq = -1;          // HNSW_DEBUG_RESIDENT prints balanced_lds_pad's choice once per batch size
    int resident_nslot = 0; size_t resident_lds = 0;   // ... for this kernel variant / LDS size
    bool time_kernels = false;           // option "time_kernels": event triples around the launches of each device-entry call
    std::vector<hipEvent_t> tev;         // [3 * recorded calls]: before the pre-pass, before the search kernel, after it
    size_t tev_used = 0;
    std::vector<char> tev_ordered;       // per recorded call: did the ordering pre-pass run
    // scratch of the ordering pre-pass, one block per caller stream (kept between calls: work on one
    // stream is ordered, so the block is free again when the next call on that stream needs it)
    struct OrderScratch { hipStream_t st; void *p; size_t bytes; };
    std::vector<OrderScratch> order_scratch;
    int order_mode = -1;                 // option "order_queries": -1 automatic (batches larger than half of resident_queries), 0 never, 1 always
    int vt_bits_override = 0;
    int vt_grow_key = -1, vt_grow_bits = 0;   // knn_vt_bits' cached choice for (kernel variant, base size)
    int lds_pad = -1;                    // option "lds_pad": extra LDS bytes per search wave (-1 = balanced_lds_pad's choice)
    std::vector<std::pair<int, int>> prepared;   // (ef, accept rule) of every hnsw_index_prepare: what hnsw_index_save writes down
};

namespace hnsw_host {

// hnsw_rows8.hip: if every value of idx->dX is an integer in 0..255, build the byte copy (idx->dX8, iv.X8, iv.stride8)
int make_byte_rows(::hnsw_index *idx);
// hnsw_rows_split.hip: if a row ends 1..32 bytes past a 128-byte line (and there are no byte rows), build the split copy
// (idx->dXm / dTail0, iv.Xm / tail0 / stride_m / main_chunks / tail_chunks); call after make_byte_rows, graph in place
int make_split_rows(::hnsw_index *idx);

// hnsw_order.hip: the descent alone for nq device-resident queries: d_entry[q] = the node the greedy descent reaches on layer
// to_layer (lib/ohnsw.ml:865-867 stopped there); d_scratch: 4 * nq words
int descent_entries(::hnsw_index *idx, const float *d_queries, int64_t nq, int64_t q_stride, int32_t to_layer, int32_t *d_entry,
                    uint32_t *d_scratch, hipStream_t st);
// hnsw_capi.hip: the one-time costs of a process's first search (code objects, the handle's stream and flag word), paid at
// index construction: one query through the plain and the ordered launch
int warm_up(::hnsw_index *idx);
// hnsw_capi.hip: hnsw_index_prepare for construction paths (expected_ef): failures are swallowed, a search reports its own
void prepare_quietly(::hnsw_index *idx, int32_t ef, int32_t semantics);
// hnsw_capi.hip: a saved decision of the visited structure for the shape of (ef, rule): blocks (true) or the tag cache
void adopt_blk_choice(::hnsw_index *idx, int32_t ef, int32_t semantics, bool blocks);
// ... and the decisions a handle has made so far, as (ef representative of the shape, rule, blocks?) triples
void list_blk_choices(const ::hnsw_index *idx, std::vector<int32_t> &out3);
// hnsw_layer_ops.hip: Ohnsw.search_k on one layer for device-resident targets (one start node each, W bounded by ef): the
// nearest node found per target
int layer_nearest_device(::hnsw_index *idx, int32_t layer, const float *d_targets, int64_t t_stride, int64_t nq, const int32_t *d_qmap,
                         int64_t n_launch, const int32_t *d_starts, int32_t ef, int32_t *d_out_ids, float *d_out_dist);
// hnsw_locality.hip: builds idx->dLcode / dLcode0 (and iv.lcode / lcode0) once; lcode_state says how it went.  The per-slot
// table dLcode0 (n * max_degree0 * 4 bytes) exists only while a kernel shape uses the bitmap blocks or is being measured:
// drop_lcode0 frees it, materialise_lcode0 (also reached through build_locality_codes) makes it again from dLcode.
int build_locality_codes(::hnsw_index *idx);
int materialise_lcode0(::hnsw_index *idx);
void drop_lcode0(::hnsw_index *idx);
int adopt_locality_codes(::hnsw_index *idx, const int32_t *codes);

// Longest-first ordering of a large batch (hnsw_order.hip): runs the descent kernel and a radix sort
// on `st`; on success *block points to the handle's scratch for that stream (nothing to release)
// whose parts are returned in the other pointers.
// d_stage (optional): d_queries is the caller's registered HOST matrix seen from the device (zero-copy); the descent kernel
// then leaves a device-resident copy of every query there for the search kernel
int order_longest_first(::hnsw_index *idx, const float *d_queries, int64_t nq, int64_t q_stride, float *d_stage, hipStream_t st,
                        void **block, const int32_t **qmap, const int32_t **pre_entry, const uint32_t **pre_key,
                        const uint32_t **pre_nd, int32_t *pre_layer);

// hnsw_search_batch_device plus the optional device word that collects status bit 0 of the whole launch
// d_stage (optional, [nq][q_stride] device floats): d_queries points into registered host memory (see order_longest_first)
int search_batch_device_flag(::hnsw_index *idx, const float *d_queries, int64_t nq, int64_t q_stride,
                             const hnsw_search_params *params, int32_t *d_ids, float *d_dist,
                             uint32_t *d_ndist, uint32_t *d_nhops, uint32_t *d_status, uint32_t *d_any_flag, void *stream,
                             float *d_stage = nullptr);
// parameter / handle checks shared by every search entry point (HNSW_ERR_BAD_ARG, HNSW_ERR_EMPTY_INDEX, ...)
int search_check(const ::hnsw_index *idx, const hnsw_search_params *p);
// launch of the exactness fallback (see rerun_overflowed): `c` flagged queries, listed in qmap, searched again
// with a global slab for their tie lists
int search_rerun_device(::hnsw_index *idx, const float *d_queries, int64_t nq, int64_t q_stride, const hnsw_search_params *p,
                        int32_t *d_ids, float *d_dist, uint32_t *d_nd, uint32_t *d_nh, uint32_t *d_st,
                        const int32_t *qmap, int64_t c, uint32_t *slab, int32_t cap, hipStream_t st);

// log2 entries of the per-query LDS visited cache (never changes results)
inline int search_vt_bits(const hnsw_index *idx, int ef) {
    int b = idx->vt_bits_override ? idx->vt_bits_override : env_int("HNSW_VT_BITS", 0);
    if (b <= 0) {
        // Re-encounters of a node come soon after its first evaluation, so the cache need not
        // grow with ef: 2^11 tags (4 KiB, 32 waves/CU) cost 3.5 % re-evaluations on C2 and 2.6 %
        // at ef = 512 (measured), while 2^13 halves the resident waves.  One step more once the
        // W registers cap the occupancy anyway.
        b = ef <= 256 ? 11 : 12;
    }
    b = std::max(4, std::min(16, b));
    while (b < 16 && ((int64_t)0xFFFF << (b - 1)) < idx->iv.n) ++b;   // 16-bit tags must identify ids exactly, 0xFFFF = empty way
    return b;
}

// Exactness fallback of the host-buffer entry points: a query whose stack of tied, evicted, still
// expandable entries outgrew its 64 LDS slots (status bit 0) is searched again with a global slab
// that can hold every node.  launch(qmap, count, slab, slab_cap) starts the kernel for `count`
// flagged queries; it is synchronised here.
template <class Launch>
int rerun_overflowed(hnsw_index *idx, int64_t nq, const uint32_t *d_status, Launch &&launch, int64_t *n_rerun = nullptr) {
    std::vector<uint32_t> st((size_t)nq);
    HIP_TRY(hipMemcpy(st.data(), d_status, (size_t)nq * 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> flagged;
    for (int64_t i = 0; i < nq; ++i) if (st[(size_t)i] & 1u) flagged.push_back((int32_t)i);
    if (n_rerun) *n_rerun = (int64_t)flagged.size();
    if (flagged.empty()) return HNSW_OK;
    const int64_t n = idx->iv.n;
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(256, (512ll << 20) / (n * 4 + 1)));
    DevBuf dMap, dSlab;
    struct Guard { DevBuf &a, &b; ~Guard() { a.release(); b.release(); } } guard{dMap, dSlab};
    int rc;
    if ((rc = dMap.ensure((size_t)chunk * 4)) || (rc = dSlab.ensure((size_t)chunk * n * 4))) return rc;
    for (size_t f0 = 0; f0 < flagged.size(); f0 += (size_t)chunk) {
        const int64_t c = (int64_t)std::min<size_t>((size_t)chunk, flagged.size() - f0);
        HIP_TRY(hipMemcpy(dMap.p, flagged.data() + f0, (size_t)c * 4, hipMemcpyHostToDevice));
        if ((rc = launch((const int32_t *)dMap.p, c, (uint32_t *)dSlab.p, (int32_t)n))) return rc;
        HIP_TRY(hipDeviceSynchronize());
    }
    return HNSW_OK;
}

} // namespace hnsw_host
