// hnsw_order.hip -- longest-first ordering of a large batch (see hnsw_descent_kernel): the descent
// kernel, then a radix sort of the queries by the distance of their layer-0 entry node, farthest first.
#include "hnsw_internal.h"

#include <hipcub/hipcub.hpp>

using hnsw_dev::IndexView;
using namespace hnsw_host;

namespace {
template <int METRIC>
hipError_t launch_descent(int nch, const IndexView &iv, const float *Q, int64_t qs, int64_t nq, int32_t *entry, uint32_t *key,
                          uint32_t *nd, uint32_t *sortkey, int32_t *index, hipStream_t st) {
    const size_t lds = hnsw_dev::wave_lds_words(4) * sizeof(uint32_t);
    dim3 grid((unsigned)nq), block(64);
    switch (nch) {
    case 1: hipLaunchKernelGGL((hnsw_dev::hnsw_descent_kernel<1, 8, METRIC>), grid, block, lds, st, iv, Q, qs, nq, entry, key, nd, sortkey, index); break;
    case 2: hipLaunchKernelGGL((hnsw_dev::hnsw_descent_kernel<2, HNSW_RB_NCH2, METRIC>), grid, block, lds, st, iv, Q, qs, nq, entry, key, nd, sortkey, index); break;
    case 4: hipLaunchKernelGGL((hnsw_dev::hnsw_descent_kernel<4, 2, METRIC>), grid, block, lds, st, iv, Q, qs, nq, entry, key, nd, sortkey, index); break;
    case 8: hipLaunchKernelGGL((hnsw_dev::hnsw_descent_kernel<8, 1, METRIC>), grid, block, lds, st, iv, Q, qs, nq, entry, key, nd, sortkey, index); break;
    default: hipLaunchKernelGGL((hnsw_dev::hnsw_descent_kernel<16, 1, METRIC>), grid, block, lds, st, iv, Q, qs, nq, entry, key, nd, sortkey, index); break;
    }
    return hipGetLastError();
}
} // namespace

namespace hnsw_host {

int order_longest_first(::hnsw_index *idx, const float *d_queries, int64_t nq, int64_t q_stride, hipStream_t st,
                        void **block, const int32_t **qmap, const int32_t **pre_entry, const uint32_t **pre_key,
                        const uint32_t **pre_nd) {
    *block = nullptr;
    const size_t n = (size_t)nq, slot = (n * 4 + 255) & ~(size_t)255;
    size_t temp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                           (const int32_t *)nullptr, (int32_t *)nullptr, (int)nq, 0, 32, st) != hipSuccess)
        return fail(HNSW_ERR_HIP, "radix sort sizing failed");
    temp_bytes = (temp_bytes + 255) & ~(size_t)255;
    const size_t need = 7 * slot + temp_bytes;
    // the handle keeps one scratch block per caller stream; it only ever grows
    hnsw_index::OrderScratch *sc = nullptr;
    for (auto &o : idx->order_scratch) if (o.st == st) sc = &o;
    if (!sc) { idx->order_scratch.push_back({st, nullptr, 0}); sc = &idx->order_scratch.back(); }
    if (sc->bytes < need) {
        if (sc->p) { HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(sc->p); sc->p = nullptr; sc->bytes = 0; }
        const size_t grow = need + need / 4;
        hipError_t me = hipMalloc(&sc->p, grow);
        if (me != hipSuccess) { (void)hipGetLastError(); sc->p = nullptr; return fail(me == hipErrorOutOfMemory ? HNSW_ERR_OOM : HNSW_ERR_HIP, "hipMalloc(%zu) for the ordering pre-pass failed", grow); }
        sc->bytes = grow;
    }
    char *base = (char *)sc->p;
    hipError_t e;
    int32_t *entry = (int32_t *)(base + 0 * slot);
    uint32_t *key = (uint32_t *)(base + 1 * slot), *nd = (uint32_t *)(base + 2 * slot);
    uint32_t *sortkey = (uint32_t *)(base + 3 * slot), *sorted = (uint32_t *)(base + 4 * slot);
    int32_t *index = (int32_t *)(base + 5 * slot), *order = (int32_t *)(base + 6 * slot);
    void *temp = base + 7 * slot;
    const int nch = pick_nch(idx->iv.nchunks);
    e = idx->info.metric == HNSW_METRIC_L2 ? launch_descent<0>(nch, idx->iv, d_queries, q_stride, nq, entry, key, nd, sortkey, index, st)
                                           : launch_descent<1>(nch, idx->iv, d_queries, q_stride, nq, entry, key, nd, sortkey, index, st);
    if (e == hipSuccess)
        e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const uint32_t *)sortkey, sorted, (const int32_t *)index, order, (int)nq, 0, 32, st);
    if (e != hipSuccess) return fail(HNSW_ERR_HIP, "ordering pre-pass failed: %s", hipGetErrorString(e));
    *block = base; *qmap = order; *pre_entry = entry; *pre_key = key; *pre_nd = nd;
    return HNSW_OK;
}

} // namespace hnsw_host
