// hnsw_order.hip -- longest-first ordering of a large batch (see hnsw_descent_kernel): the descent
// kernel, then a radix sort of the queries by the distance of their layer-0 entry node, farthest first.
#include "hnsw_internal.h"

#include <hipcub/hipcub.hpp>

using hnsw_dev::IndexView;
using namespace hnsw_host;

namespace {
// The launch order only has to be roughly right, so for batches of moderate size the sort is one
// workgroup doing a counting sort into ORDER_BUCKETS buckets spread linearly over the range of the
// keys present (keys are order-preserving bit patterns of the distances): min/max, histogram in
// LDS, exclusive scan, scatter.  Which of two queries of one bucket comes first is left to the
// atomics: per-query results do not depend on the launch order.  ~6 us for 10 k queries against
// 28 us for the five launches of a full radix sort.
constexpr int ORDER_BUCKETS = 2048;
constexpr int ORDER_THREADS = 1024;
// n <= PER * ORDER_THREADS: every key is read from memory once, all loads in flight together, and kept
// in registers (larger batches use the device-wide radix sort).
template <int PER>
__global__ void __launch_bounds__(ORDER_THREADS)
order_bucket_kernel(const uint32_t *sortkey, int32_t n, int32_t *order) {
    __shared__ uint32_t cnt[ORDER_BUCKETS];
    __shared__ uint32_t red_min[ORDER_THREADS / 64], red_max[ORDER_THREADS / 64];
    __shared__ uint32_t part[ORDER_THREADS];
    const int t = threadIdx.x;
    uint32_t keys[PER > 0 ? PER : 1];
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    if (PER > 0) {
#pragma unroll
        for (int j = 0; j < PER; ++j) { const int i = t + j * ORDER_THREADS; keys[j] = sortkey[i < n ? i : 0]; }
#pragma unroll
        for (int j = 0; j < PER; ++j) if (t + j * ORDER_THREADS < n) { lo = keys[j] < lo ? keys[j] : lo; hi = keys[j] > hi ? keys[j] : hi; }
    } else {
        for (int i = t; i < n; i += ORDER_THREADS) { const uint32_t k = sortkey[i]; lo = k < lo ? k : lo; hi = k > hi ? k : hi; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t a = __shfl_xor(lo, o), b = __shfl_xor(hi, o);
        lo = a < lo ? a : lo; hi = b > hi ? b : hi;
    }
    if ((t & 63) == 0) { red_min[t >> 6] = lo; red_max[t >> 6] = hi; }
    for (int i = t; i < ORDER_BUCKETS; i += ORDER_THREADS) cnt[i] = 0;
    __syncthreads();
    lo = red_min[0]; hi = red_max[0];
#pragma unroll
    for (int w = 1; w < ORDER_THREADS / 64; ++w) { lo = red_min[w] < lo ? red_min[w] : lo; hi = red_max[w] > hi ? red_max[w] : hi; }
    // monotone map of [lo, hi] onto the buckets (float: no 64-bit division; clamped for rounding)
    const float scale = (float)ORDER_BUCKETS / ((float)(hi - lo) + 1.0f);
    auto bucket = [&](uint32_t k) { const uint32_t b = (uint32_t)((float)(k - lo) * scale); return b < (uint32_t)ORDER_BUCKETS ? b : (uint32_t)ORDER_BUCKETS - 1u; };
    if (PER > 0) {
#pragma unroll
        for (int j = 0; j < PER; ++j) if (t + j * ORDER_THREADS < n) atomicAdd(&cnt[bucket(keys[j])], 1u);
    } else {
        for (int i = t; i < n; i += ORDER_THREADS) atomicAdd(&cnt[bucket(sortkey[i])], 1u);
    }
    __syncthreads();
    // exclusive scan of the 2048 counters: two per thread, a scan of the pair sums inside each wave (no barriers), then the
    // 16 wave totals (one barrier where the step-by-step block scan had twenty)
    const uint32_t c0 = cnt[2 * t], c1 = cnt[2 * t + 1];
    uint32_t incl = c0 + c1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if ((t & 63) >= o) incl += v;
    }
    if ((t & 63) == 63) part[t >> 6] = incl;
    __syncthreads();
    uint32_t wave_base = 0;
#pragma unroll
    for (int w = 0; w < ORDER_THREADS / 64; ++w) wave_base += w < (t >> 6) ? part[w] : 0u;
    const uint32_t base = wave_base + incl - (c0 + c1);
    cnt[2 * t] = base; cnt[2 * t + 1] = base + c0;
    __syncthreads();
    if (PER > 0) {
#pragma unroll
        for (int j = 0; j < PER; ++j) { const int i = t + j * ORDER_THREADS; if (i < n) order[atomicAdd(&cnt[bucket(keys[j])], 1u)] = i; }
    } else {
        for (int i = t; i < n; i += ORDER_THREADS) order[atomicAdd(&cnt[bucket(sortkey[i])], 1u)] = i;
    }
}

template <int METRIC, int ROWS>
hipError_t launch_descent_rows(int nch, const IndexView &iv, const float *Q, int64_t qs, int64_t nq, int32_t to_layer, int32_t *entry, uint32_t *key,
                               uint32_t *nd, uint32_t *sortkey, int32_t *index, float *stage, hipStream_t st) {
    const size_t lds = hnsw_dev::wave_lds_words(4) * sizeof(uint32_t);
    dim3 grid((unsigned)nq), block(64);
    constexpr bool B = ROWS == 2;            // byte rows: a quarter of the registers per row in flight
    switch (nch) {
    case 1: hipLaunchKernelGGL((hnsw_dev::hnsw_descent_kernel<1, 8, METRIC, ROWS>), grid, block, lds, st, iv, Q, qs, nq, to_layer, entry, key, nd, sortkey, index, stage); break;
    case 2: hipLaunchKernelGGL((hnsw_dev::hnsw_descent_kernel<2, (B ? 8 : HNSW_RB_NCH2), METRIC, ROWS>), grid, block, lds, st, iv, Q, qs, nq, to_layer, entry, key, nd, sortkey, index, stage); break;
    case 4: hipLaunchKernelGGL((hnsw_dev::hnsw_descent_kernel<4, (B ? 4 : 2), METRIC, ROWS>), grid, block, lds, st, iv, Q, qs, nq, to_layer, entry, key, nd, sortkey, index, stage); break;
    case 8: hipLaunchKernelGGL((hnsw_dev::hnsw_descent_kernel<8, (B ? 2 : 1), METRIC, ROWS>), grid, block, lds, st, iv, Q, qs, nq, to_layer, entry, key, nd, sortkey, index, stage); break;
    default: hipLaunchKernelGGL((hnsw_dev::hnsw_descent_kernel<16, 1, METRIC, ROWS>), grid, block, lds, st, iv, Q, qs, nq, to_layer, entry, key, nd, sortkey, index, stage); break;
    }
    return hipGetLastError();
}
template <int METRIC>
hipError_t launch_descent(int nch, const IndexView &iv, const float *Q, int64_t qs, int64_t nq, int32_t to_layer, int32_t *entry, uint32_t *key,
                          uint32_t *nd, uint32_t *sortkey, int32_t *index, float *stage, hipStream_t st) {
    return iv.X8 ? launch_descent_rows<METRIC, 2>(nch, iv, Q, qs, nq, to_layer, entry, key, nd, sortkey, index, stage, st)
                 : launch_descent_rows<METRIC, -1>(nch, iv, Q, qs, nq, to_layer, entry, key, nd, sortkey, index, stage, st);
}
} // namespace

namespace hnsw_host {

int descent_entries(::hnsw_index *idx, const float *d_queries, int64_t nq, int64_t q_stride, int32_t to_layer, int32_t *d_entry,
                    uint32_t *d_scratch, hipStream_t st) {
    if (nq <= 0) return HNSW_OK;
    const int nch = pick_nch(idx->iv.nchunks);
    uint32_t *key = d_scratch, *nd = d_scratch + nq, *sortkey = d_scratch + 2 * nq;
    int32_t *index = (int32_t *)(d_scratch + 3 * nq);
    const hipError_t e = idx->info.metric == HNSW_METRIC_L2
        ? launch_descent<0>(nch, idx->iv, d_queries, q_stride, nq, to_layer, d_entry, key, nd, sortkey, index, nullptr, st)
        : launch_descent<1>(nch, idx->iv, d_queries, q_stride, nq, to_layer, d_entry, key, nd, sortkey, index, nullptr, st);
    if (e != hipSuccess) return fail(HNSW_ERR_HIP, "descent launch failed: %s", hipGetErrorString(e));
    return HNSW_OK;
}

int order_longest_first(::hnsw_index *idx, const float *d_queries, int64_t nq, int64_t q_stride, float *d_stage, hipStream_t st,
                        void **block, const int32_t **qmap, const int32_t **pre_entry, const uint32_t **pre_key,
                        const uint32_t **pre_nd, int32_t *pre_layer) {
    // The pre-pass may stop above layer 1 and leave the rest of the descent to the search kernel: on C2
    // stopping at layer 2 costs 58 instead of 78 us and orders almost as well (0.827 against 0.835 ms
    // per step), stopping at layer 3 orders badly (0.885 ms).  Default: the whole descent, so that
    // the two kernels' shares of the work are the descent and the layer-0 walk.
    const int32_t to_layer = std::max(1, std::min(idx->iv.max_layer, env_int("HNSW_ORDER_STOP_LAYER", 1)));
    *pre_layer = to_layer;
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
    e = idx->info.metric == HNSW_METRIC_L2 ? launch_descent<0>(nch, idx->iv, d_queries, q_stride, nq, to_layer, entry, key, nd, sortkey, index, d_stage, st)
                                           : launch_descent<1>(nch, idx->iv, d_queries, q_stride, nq, to_layer, entry, key, nd, sortkey, index, d_stage, st);
    if (e == hipSuccess) {
        if (nq <= 16 * ORDER_THREADS && !env_int("HNSW_ORDER_FULL_SORT", 0)) {
            static_assert(ORDER_BUCKETS == 2 * ORDER_THREADS, "two counters per thread in the scan");
            hipLaunchKernelGGL(order_bucket_kernel<16>, dim3(1), dim3(ORDER_THREADS), 0, st, (const uint32_t *)sortkey, (int32_t)nq, order);
            e = hipGetLastError();
        } else {
            e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const uint32_t *)sortkey, sorted, (const int32_t *)index, order, (int)nq, 0, 32, st);
        }
    }
    if (e != hipSuccess) return fail(HNSW_ERR_HIP, "ordering pre-pass failed: %s", hipGetErrorString(e));
    *block = base; *qmap = order; *pre_entry = entry; *pre_key = key; *pre_nd = nd;
    return HNSW_OK;
}

} // namespace hnsw_host
