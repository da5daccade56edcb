// hnsw_layer_ops.hip -- the layer-level functions of the search path as batched operators:
//   hnsw_search_layer_batch = Ohnsw.search_k (lib/ohnsw.ml:543-588) / Hnsw_algo.Search.search
//                             (lib/hnsw_algo.ml:350-391) on one layer from explicit start nodes
//   hnsw_search_one_batch   = Ohnsw.search_one (lib/ohnsw.ml:492-512) / Search.search_one
//                             (lib/hnsw_algo.ml:393-437)
// Same device code as the knn kernel (search_layer / greedy_descend of hnsw_device.hip.h): one
// 64-lane wave per target, W in registers, visited cache in LDS.
#include "hnsw_internal.h"

using hnsw_dev::IndexView;
using namespace hnsw_host;

namespace hnsw_dev {

struct LayerSearchArgs {
    const float *Q;
    int64_t q_stride;
    int64_t nq;
    const int32_t *starts;   // [nq][n_start] 0-based, -1 = absent
    int32_t n_start;
    int32_t layer;
    int32_t ef, k, fill, sem, vt_bits;
    int32_t *out_ids;
    float *out_dist;
    int32_t *out_cnt;
    uint32_t *out_ndist, *out_nhops, *out_status;
    const int32_t *qmap;
    uint32_t *ovf_g;
    int32_t ovf_gcap;
};

template <int NCH, int RB, int NSLOT, int METRIC>
__global__ void __launch_bounds__(64)
hnsw_layer_search_kernel(const IndexView iv, const LayerSearchArgs a) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    if ((int64_t)blockIdx.x >= a.nq) return;
    const int64_t q = a.qmap ? a.qmap[blockIdx.x] : (int64_t)blockIdx.x;
    WaveCtx cx = make_ctx(lds, a.vt_bits, lane);
    if (a.ovf_g) { cx.ovf.g = a.ovf_g + (int64_t)blockIdx.x * a.ovf_gcap; cx.ovf.gcap = a.ovf_gcap; }

    float4 qv[NCH];
    load_query<NCH>(qv, a.Q + q * a.q_stride, iv.d, cx.l16);
    visited_clear(cx);                                                   // lib/ohnsw.ml:553
    uint32_t n_dist = 0, n_hops = 0, status = 0;
    WList<NSLOT> w;
    wlist_init(w, a.ef, lane);                                           // :554
    __syncthreads();

    // start nodes -> visited + W (:555-557; Search.search recomputes their distances,
    // lib/hnsw_algo.ml:342-348), 64 at a time
    for (int base = 0; base < a.n_start; base += 64) {
        const int s = (base + lane < a.n_start) ? a.starts[q * a.n_start + base + lane] : -1;
        const bool valid = s >= 0;
        const uint64_t m = __ballot(valid);
        const int cnt = __popcll(m);
        if (cnt == 0) continue;
        const int pos = __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (valid) cx.cand_id[pos] = s;
        __syncthreads();
        eval_candidates<NCH, RB, METRIC>(iv, qv, cx.cand_id, cx.cand_key, cx.trash, cnt, cx.r, cx.l16);
        __syncthreads();
        n_dist += cnt;
        const uint32_t my_key = cx.cand_key[lane];
        const uint32_t my_id = (uint32_t)cx.cand_id[lane];
        uint32_t vword;
        (void)visited_mem(cx, lane < cnt ? my_id : 0u, vword);
        visited_add_masked(cx, my_id, vword, lane < cnt);
        for (int i = 0; i < cnt; ++i)
            wlist_insert(w, rdlane(my_key, i), rdlane(my_id, i), lane, cx.ovf, status);
        __syncthreads();
    }

    if (a.sem) search_layer<NCH, RB, NSLOT, METRIC, 1>(iv, qv, a.layer, w, a.ef, cx, n_dist, n_hops, status);
    else search_layer<NCH, RB, NSLOT, METRIC, 0>(iv, qv, a.layer, w, a.ef, cx, n_dist, n_hops, status);

    // W[0..k) ascending (result_minq, lib/ohnsw.ml:586-587)
    const int wbase = NSLOT * 64 - a.ef;
    int found = 0;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int idx = s * 64 + lane - wbase;
        const uint32_t hi = w.hi[s];
        const bool real = idx >= 0 && idx < a.k && hi < DUMMY_HI;
        found += __popcll(__ballot(real));
        if (idx >= 0 && idx < a.k) {
            int32_t oid = iv.id_base - 1;
            float od = a.fill == 0 ? __uint_as_float(0x7FC00000u) : __uint_as_float(0x7F800000u);
            if (real) { oid = (int32_t)key_id(w.lo[s]) + iv.id_base; od = key_to_dist<METRIC>(hi); }
            a.out_ids[q * a.k + idx] = oid;
            a.out_dist[q * a.k + idx] = od;
        }
    }
    if (lane == 0) {
        if (a.out_cnt) a.out_cnt[q] = found;
        if (a.out_ndist) a.out_ndist[q] = n_dist;
        if (a.out_nhops) a.out_nhops[q] = n_hops;
        if (a.out_status) a.out_status[q] = status;
    }
}

template <int NCH, int RB, int METRIC>
__global__ void __launch_bounds__(64)
hnsw_search_one_kernel(const IndexView iv, const float *Q, int64_t q_stride, int64_t nq, int32_t layer,
                       const int32_t *start, int32_t *out_node, float *out_dist) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    if (q >= nq) return;
    WaveCtx cx = make_ctx(lds, 4, lane);
    float4 qv[NCH];
    load_query<NCH>(qv, Q + q * q_stride, iv.d, cx.l16);
    int cur = start[q];
    if (lane == 0) cx.cand_id[0] = cur;
    __syncthreads();
    eval_candidates<NCH, RB, METRIC>(iv, qv, cx.cand_id, cx.cand_key, cx.trash, 1, cx.r, cx.l16);
    __syncthreads();
    uint32_t cur_key = cx.cand_key[0];
    uint32_t n_dist = 1;
    greedy_descend<NCH, RB, METRIC>(iv, qv, layer, layer, cur, cur_key, cx, n_dist);   // lib/ohnsw.ml:492-508
    if (lane == 0) {
        out_node[q] = cur + iv.id_base;
        out_dist[q] = key_to_dist<METRIC>(cur_key);
    }
}

} // namespace hnsw_dev

namespace {

using hnsw_dev::LayerSearchArgs;

template <int NCH, int RB, int NSLOT, int METRIC>
hipError_t launch_layer(const IndexView &iv, const LayerSearchArgs &a) {
    const size_t lds = hnsw_dev::wave_lds_words(a.vt_bits) * sizeof(uint32_t);
    hipLaunchKernelGGL((hnsw_dev::hnsw_layer_search_kernel<NCH, RB, NSLOT, METRIC>), dim3((unsigned)a.nq),
                       dim3(64), lds, nullptr, iv, a);
    return hipGetLastError();
}
template <int NCH, int RB, int METRIC>
hipError_t layer_slot(int nslot, const IndexView &iv, const LayerSearchArgs &a) {
    switch (nslot) {
    case 1: return launch_layer<NCH, RB, 1, METRIC>(iv, a);
    case 2: return launch_layer<NCH, RB, 2, METRIC>(iv, a);
    case 4: return launch_layer<NCH, RB, 4, METRIC>(iv, a);
    case 8: return launch_layer<NCH, RB, 8, METRIC>(iv, a);
    default: return launch_layer<NCH, RB, 16, METRIC>(iv, a);
    }
}
template <int METRIC>
hipError_t layer_nch(int nch, int nslot, const IndexView &iv, const LayerSearchArgs &a) {
    switch (nch) {
    case 1: return layer_slot<1, 8, METRIC>(nslot, iv, a);
    case 2: return layer_slot<2, HNSW_RB_NCH2, METRIC>(nslot, iv, a);
    case 4: return layer_slot<4, 2, METRIC>(nslot, iv, a);
    case 8: return layer_slot<8, 1, METRIC>(nslot, iv, a);
    default: return layer_slot<16, 1, METRIC>(nslot, iv, a);
    }
}
int launch_layer_args(hnsw_index *idx, const LayerSearchArgs &a) {
    const int nch = pick_nch(idx->iv.nchunks), nslot = pick_nslot(a.ef);
    hipError_t e = idx->info.metric == HNSW_METRIC_L2 ? layer_nch<0>(nch, nslot, idx->iv, a) : layer_nch<1>(nch, nslot, idx->iv, a);
    if (e != hipSuccess) return fail(HNSW_ERR_HIP, "layer search kernel launch failed: %s", hipGetErrorString(e));
    return HNSW_OK;
}

template <int METRIC>
hipError_t one_nch(int nch, const IndexView &iv, const float *Q, int64_t qs, int64_t nq, int32_t layer,
                   const int32_t *start, int32_t *out_node, float *out_dist) {
    const size_t lds = hnsw_dev::wave_lds_words(4) * sizeof(uint32_t);
    dim3 grid((unsigned)nq), block(64);
    switch (nch) {
    case 1: hipLaunchKernelGGL((hnsw_dev::hnsw_search_one_kernel<1, 8, METRIC>), grid, block, lds, nullptr, iv, Q, qs, nq, layer, start, out_node, out_dist); break;
    case 2: hipLaunchKernelGGL((hnsw_dev::hnsw_search_one_kernel<2, HNSW_RB_NCH2, METRIC>), grid, block, lds, nullptr, iv, Q, qs, nq, layer, start, out_node, out_dist); break;
    case 4: hipLaunchKernelGGL((hnsw_dev::hnsw_search_one_kernel<4, 2, METRIC>), grid, block, lds, nullptr, iv, Q, qs, nq, layer, start, out_node, out_dist); break;
    case 8: hipLaunchKernelGGL((hnsw_dev::hnsw_search_one_kernel<8, 1, METRIC>), grid, block, lds, nullptr, iv, Q, qs, nq, layer, start, out_node, out_dist); break;
    default: hipLaunchKernelGGL((hnsw_dev::hnsw_search_one_kernel<16, 1, METRIC>), grid, block, lds, nullptr, iv, Q, qs, nq, layer, start, out_node, out_dist); break;
    }
    return hipGetLastError();
}

// start ids: id_base-based int64 on the host -> 0-based int32, absent (-1) below id_base
int rebase_starts(const hnsw_index *idx, const int64_t *src, size_t count, bool allow_absent, std::vector<int32_t> &dst) {
    dst.resize(count);
    for (size_t i = 0; i < count; ++i) {
        const int64_t v = src[i] - idx->iv.id_base;
        if (v < 0) {
            if (!allow_absent) return fail(HNSW_ERR_BAD_ARG, "Vector.get: start node %lld out of range", (long long)src[i]); /* lib/ohnsw.ml:25 */
            dst[i] = -1;
            continue;
        }
        if (v >= idx->iv.n) return fail(HNSW_ERR_BAD_ARG, "Vector.get: start node %lld out of range", (long long)src[i]);
        dst[i] = (int32_t)v;
    }
    return HNSW_OK;
}

} // namespace

namespace hnsw_host {
// Ohnsw.search_k on `layer` for device-resident targets, one start node each, W bounded by ef, the NEAREST node found per target:
// d_out_ids[q] (id_base-based) / d_out_dist[q].  With d_qmap: n_launch waves, wave b searches target d_qmap[b] (the others'
// outputs are left alone).  Asynchronous on the null stream.  A target whose tie list overflows keeps its (still valid, possibly
// not nearest) result: the callers want a good neighbour, not the reference's answer.
int layer_nearest_device(::hnsw_index *idx, int32_t layer, const float *d_targets, int64_t t_stride, int64_t nq, const int32_t *d_qmap,
                         int64_t n_launch, const int32_t *d_starts, int32_t ef, int32_t *d_out_ids, float *d_out_dist) {
    if (n_launch <= 0) return HNSW_OK;
    LayerSearchArgs a{};
    a.Q = d_targets; a.q_stride = t_stride; a.nq = d_qmap ? n_launch : nq;
    a.starts = d_starts; a.n_start = 1; a.layer = layer;
    a.ef = ef; a.k = 1; a.fill = HNSW_FILL_OHNSW; a.sem = 0;
    a.vt_bits = search_vt_bits(idx, ef);
    a.out_ids = d_out_ids; a.out_dist = d_out_dist; a.qmap = d_qmap;
    return launch_layer_args(idx, a);
}
} // namespace hnsw_host

extern "C" {

int32_t hnsw_search_layer_batch(hnsw_index *idx, int32_t layer, const float *targets, int64_t nq,
                                int64_t t_stride, const int64_t *start_nodes, int32_t n_start,
                                const hnsw_search_params *p, int32_t *out_ids, float *out_dist,
                                int32_t *out_cnt, uint32_t *out_ndist, uint32_t *out_nhops) {
    if (!idx) return fail(HNSW_ERR_BAD_ARG, "null index");
    if (!p) return fail(HNSW_ERR_BAD_ARG, "null params");
    if (p->ef < 1 || p->k < 1 || p->k > p->ef) return fail(HNSW_ERR_BAD_ARG, "need 1 <= k <= ef (ef=%d k=%d)", p->ef, p->k);
    if (p->ef > 1024) return fail(HNSW_ERR_UNSUPPORTED, "ef=%d > 1024 not supported", p->ef);
    if (p->fill != HNSW_FILL_OHNSW && p->fill != HNSW_FILL_BA) return fail(HNSW_ERR_BAD_ARG, "bad fill %d", p->fill);
    if (p->semantics != HNSW_SEM_OHNSW && p->semantics != HNSW_SEM_FUNCTOR) return fail(HNSW_ERR_BAD_ARG, "bad semantics %d", p->semantics);
    if (layer < 0 || layer > idx->iv.max_layer) return fail(HNSW_ERR_BAD_ARG, "Hgraph.layer: layer %d out of range (max_layer %d)", layer, idx->iv.max_layer); /* lib/ohnsw.ml:326 */
    if (n_start < 1 || n_start > p->ef) return fail(HNSW_ERR_BAD_ARG, "need 1 <= n_start <= ef (n_start=%d ef=%d)", n_start, p->ef);
    if (nq == 0) return HNSW_OK;
    if (nq < 0 || nq > 0x7FFFFFFFLL || !targets || !start_nodes || !out_ids || !out_dist) return fail(HNSW_ERR_BAD_ARG, "bad buffers");
    if (t_stride < idx->iv.d) return fail(HNSW_ERR_BAD_ARG, "t_stride < d");
    if (idx->iv.n == 0) return fail(HNSW_ERR_EMPTY_INDEX, "search: empty hgraph");
    std::vector<int32_t> st;
    int rc = rebase_starts(idx, start_nodes, (size_t)nq * n_start, true, st);
    if (rc) return rc;

    HIP_TRY(hipSetDevice(idx->device));
    const int k = p->k;
    const size_t qbytes = ((size_t)(nq - 1) * t_stride + idx->iv.d) * sizeof(float);
    DevBuf dStart, dCnt;
    struct Guard { DevBuf &a, &b; ~Guard() { a.release(); b.release(); } } guard{dStart, dCnt};
    if ((rc = idx->sQ.ensure(qbytes)) || (rc = idx->sIds.ensure((size_t)nq * k * 4)) ||
        (rc = idx->sDist.ensure((size_t)nq * k * 4)) || (rc = idx->sNd.ensure((size_t)nq * 4)) ||
        (rc = idx->sNh.ensure((size_t)nq * 4)) || (rc = idx->sSt.ensure((size_t)nq * 4)) ||
        (rc = dStart.ensure(st.size() * 4)) || (rc = dCnt.ensure((size_t)nq * 4)))
        return rc;
    HIP_TRY(hipMemcpy(idx->sQ.p, targets, qbytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dStart.p, st.data(), st.size() * 4, hipMemcpyHostToDevice));

    LayerSearchArgs a{};
    a.Q = (const float *)idx->sQ.p; a.q_stride = t_stride; a.nq = nq;
    a.starts = (const int32_t *)dStart.p; a.n_start = n_start; a.layer = layer;
    a.ef = p->ef; a.k = k; a.fill = p->fill; a.sem = p->semantics;
    a.vt_bits = search_vt_bits(idx, p->ef);
    a.out_ids = (int32_t *)idx->sIds.p; a.out_dist = (float *)idx->sDist.p; a.out_cnt = (int32_t *)dCnt.p;
    a.out_ndist = (uint32_t *)idx->sNd.p; a.out_nhops = (uint32_t *)idx->sNh.p; a.out_status = (uint32_t *)idx->sSt.p;
    if ((rc = launch_layer_args(idx, a))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    rc = rerun_overflowed(idx, nq, (const uint32_t *)idx->sSt.p,
                          [&](const int32_t *qmap, int64_t c, uint32_t *slab, int32_t cap) {
                              LayerSearchArgs b = a;
                              b.nq = c; b.qmap = qmap; b.ovf_g = slab; b.ovf_gcap = cap;
                              return launch_layer_args(idx, b);
                          });
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out_ids, idx->sIds.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_dist, idx->sDist.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
    if (out_cnt) HIP_TRY(hipMemcpy(out_cnt, dCnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    if (out_ndist) HIP_TRY(hipMemcpy(out_ndist, idx->sNd.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    if (out_nhops) HIP_TRY(hipMemcpy(out_nhops, idx->sNh.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    return HNSW_OK;
}

int32_t hnsw_search_one_batch(hnsw_index *idx, int32_t layer, const float *targets, int64_t nq,
                              int64_t t_stride, const int64_t *start, int64_t *out_node, float *out_dist) {
    if (!idx) return fail(HNSW_ERR_BAD_ARG, "null index");
    if (layer < 0 || layer > idx->iv.max_layer) return fail(HNSW_ERR_BAD_ARG, "Hgraph.layer: layer %d out of range (max_layer %d)", layer, idx->iv.max_layer);
    if (nq == 0) return HNSW_OK;
    if (nq < 0 || nq > 0x7FFFFFFFLL || !targets || !start || !out_node) return fail(HNSW_ERR_BAD_ARG, "bad buffers");
    if (t_stride < idx->iv.d) return fail(HNSW_ERR_BAD_ARG, "t_stride < d");
    if (idx->iv.n == 0) return fail(HNSW_ERR_EMPTY_INDEX, "search_one: empty hgraph");
    std::vector<int32_t> st;
    int rc = rebase_starts(idx, start, (size_t)nq, false, st);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(idx->device));
    const size_t qbytes = ((size_t)(nq - 1) * t_stride + idx->iv.d) * sizeof(float);
    if ((rc = idx->sQ.ensure(qbytes)) || (rc = idx->sIds.ensure((size_t)nq * 4)) || (rc = idx->sDist.ensure((size_t)nq * 4)) ||
        (rc = idx->sNd.ensure((size_t)nq * 4)))
        return rc;
    HIP_TRY(hipMemcpy(idx->sQ.p, targets, qbytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(idx->sNd.p, st.data(), (size_t)nq * 4, hipMemcpyHostToDevice));
    const int nch = pick_nch(idx->iv.nchunks);
    hipError_t e = idx->info.metric == HNSW_METRIC_L2
                       ? one_nch<0>(nch, idx->iv, (const float *)idx->sQ.p, t_stride, nq, layer, (const int32_t *)idx->sNd.p, (int32_t *)idx->sIds.p, (float *)idx->sDist.p)
                       : one_nch<1>(nch, idx->iv, (const float *)idx->sQ.p, t_stride, nq, layer, (const int32_t *)idx->sNd.p, (int32_t *)idx->sIds.p, (float *)idx->sDist.p);
    if (e != hipSuccess) return fail(HNSW_ERR_HIP, "search_one kernel launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<int32_t> nodes((size_t)nq);
    HIP_TRY(hipMemcpy(nodes.data(), idx->sIds.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < nq; ++i) out_node[i] = nodes[(size_t)i];
    if (out_dist) HIP_TRY(hipMemcpy(out_dist, idx->sDist.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    return HNSW_OK;
}

} // extern "C"
