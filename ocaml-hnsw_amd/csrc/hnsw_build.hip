// hnsw_build.hip -- host driver of the batched GPU graph builder + graph export.
//
// Restates Ohnsw.build_batch_bigarray / Ohnsw.insert (lib/ohnsw.ml:766-857) batch-wise on the
// device (kernels: hnsw_build_device.hip.h).  The level law is the reference's
// (round_nearest(-ln(U) * 1/ln M), lib/ohnsw.ml:781,844); the RNG is this library's own seeded
// splitmix64 (OCaml's Random cannot and need not be reproduced: search parity is defined GIVEN a
// graph).  A node that raises max_layer ends its batch and becomes the entry point (:832-836).
#include "hnsw_build_device.hip.h"
#include "hnsw_internal.h"

#include <hipcub/hipcub.hpp>

#include <cmath>

using namespace hnsw_host;
using hnsw_dev::BatchView;
using hnsw_dev::BuildView;
using hnsw_dev::IndexView;
using hnsw_dev::MergeArgs;
using hnsw_dev::SelectArgs;
using hnsw_dev::SelectOpArgs;

namespace {

inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline double rng_unit(uint64_t *s) { return ((double)(splitmix64(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

template <int NCH, int RB, int METRIC>
hipError_t launch_k1(int nslot, const BuildView &bv, const BatchView &bt, hipStream_t st) {
    const size_t lds = hnsw_dev::wave_lds_words(bv.vt_bits) * sizeof(uint32_t);
    dim3 grid((unsigned)bt.B), block(64);
    switch (nslot) {
    case 1: hipLaunchKernelGGL((hnsw_dev::build_search_kernel<NCH, RB, 1, METRIC>), grid, block, lds, st, bv, bt); break;
    case 2: hipLaunchKernelGGL((hnsw_dev::build_search_kernel<NCH, RB, 2, METRIC>), grid, block, lds, st, bv, bt); break;
    case 4: hipLaunchKernelGGL((hnsw_dev::build_search_kernel<NCH, RB, 4, METRIC>), grid, block, lds, st, bv, bt); break;
    default: hipLaunchKernelGGL((hnsw_dev::build_search_kernel<NCH, RB, 8, METRIC>), grid, block, lds, st, bv, bt); break;
    }
    return hipGetLastError();
}
template <int METRIC>
hipError_t dispatch_k1(int nch, int nslot, const BuildView &bv, const BatchView &bt, hipStream_t st) {
    switch (nch) {
    case 1: return launch_k1<1, 8, METRIC>(nslot, bv, bt, st);
    case 2: return launch_k1<2, HNSW_RB_NCH2, METRIC>(nslot, bv, bt, st);
    case 4: return launch_k1<4, 2, METRIC>(nslot, bv, bt, st);
    case 8: return launch_k1<8, 1, METRIC>(nslot, bv, bt, st);
    default: return launch_k1<16, 1, METRIC>(nslot, bv, bt, st);
    }
}
template <int METRIC>
hipError_t dispatch_k2(int nch, const BuildView &bv, const BatchView &bt, const SelectArgs &sa, hipStream_t st) {
    const size_t lds = (2 * (size_t)bv.cand_stride + 64) * sizeof(uint32_t);
    dim3 grid((unsigned)(sa.rec_end - sa.rec_begin)), block(64);
    switch (nch) {
    case 1: hipLaunchKernelGGL((hnsw_dev::build_select_kernel<1, METRIC>), grid, block, lds, st, bv, bt, sa); break;
    case 2: hipLaunchKernelGGL((hnsw_dev::build_select_kernel<2, METRIC>), grid, block, lds, st, bv, bt, sa); break;
    case 4: hipLaunchKernelGGL((hnsw_dev::build_select_kernel<4, METRIC>), grid, block, lds, st, bv, bt, sa); break;
    case 8: hipLaunchKernelGGL((hnsw_dev::build_select_kernel<8, METRIC>), grid, block, lds, st, bv, bt, sa); break;
    default: hipLaunchKernelGGL((hnsw_dev::build_select_kernel<16, METRIC>), grid, block, lds, st, bv, bt, sa); break;
    }
    return hipGetLastError();
}
template <int METRIC>
hipError_t dispatch_k4(int nch, const BuildView &bv, const MergeArgs &ma, hipStream_t st) {
    dim3 grid((unsigned)ma.n_edges), block(64);
    switch (nch) {
    case 1: hipLaunchKernelGGL((hnsw_dev::build_merge_kernel<1, 8, METRIC>), grid, block, 0, st, bv, ma); break;
    case 2: hipLaunchKernelGGL((hnsw_dev::build_merge_kernel<2, 4, METRIC>), grid, block, 0, st, bv, ma); break;
    case 4: hipLaunchKernelGGL((hnsw_dev::build_merge_kernel<4, 2, METRIC>), grid, block, 0, st, bv, ma); break;
    case 8: hipLaunchKernelGGL((hnsw_dev::build_merge_kernel<8, 1, METRIC>), grid, block, 0, st, bv, ma); break;
    default: hipLaunchKernelGGL((hnsw_dev::build_merge_kernel<16, 1, METRIC>), grid, block, 0, st, bv, ma); break;
    }
    return hipGetLastError();
}

template <int METRIC>
hipError_t dispatch_link(int nch, const BuildView &bv, const hnsw_dev::LinkArgs &la, hipStream_t st) {
    dim3 grid(1), block(64);
    switch (nch) {
    case 1: hipLaunchKernelGGL((hnsw_dev::build_link_sequential_kernel<1, 8, METRIC>), grid, block, 0, st, bv, la); break;
    case 2: hipLaunchKernelGGL((hnsw_dev::build_link_sequential_kernel<2, 4, METRIC>), grid, block, 0, st, bv, la); break;
    case 4: hipLaunchKernelGGL((hnsw_dev::build_link_sequential_kernel<4, 2, METRIC>), grid, block, 0, st, bv, la); break;
    case 8: hipLaunchKernelGGL((hnsw_dev::build_link_sequential_kernel<8, 1, METRIC>), grid, block, 0, st, bv, la); break;
    default: hipLaunchKernelGGL((hnsw_dev::build_link_sequential_kernel<16, 1, METRIC>), grid, block, 0, st, bv, la); break;
    }
    return hipGetLastError();
}

#define HIP_TRY_B(expr)                                                                      \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            rc = fail(e__ == hipErrorOutOfMemory ? HNSW_ERR_OOM : HNSW_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
            goto done;                                                                       \
        }                                                                                    \
    } while (0)

} // namespace

// One attempt with a removal buffer of rem_scale * (2 * max_batch * 2M) entries; *rem_overflow tells the
// caller that some step produced more removals than that (the graph is then discarded, never patched).
static int32_t build_attempt(const float *vectors, int64_t n, int32_t d, int64_t row_stride,
                             const hnsw_build_params *p, int32_t device, hnsw_index **out,
                             int64_t rem_scale, bool *rem_overflow) {
    *rem_overflow = false;
    if (!out || !p) return fail(HNSW_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    if (n < 1 || n > 0x7FFFFFF0LL) return fail(HNSW_ERR_BAD_ARG, "n=%lld out of range", (long long)n);
    if (!vectors || d < 1 || row_stride < d) return fail(HNSW_ERR_BAD_ARG, "bad vectors/d/row_stride");
    if (p->metric != HNSW_METRIC_L2 && p->metric != HNSW_METRIC_IP) return fail(HNSW_ERR_BAD_ARG, "bad metric");
    const int M = p->num_connections, efc = p->num_nodes_search_construction;
    if (M < 2 || 2 * M > 64) return fail(HNSW_ERR_UNSUPPORTED, "num_connections=%d must be in 2..32", M);
    if (efc < 1 || efc > 512) return fail(HNSW_ERR_UNSUPPORTED, "num_nodes_search_construction=%d must be in 1..512", efc);
    const int nchunks = (d + 3) / 4;
    const int nch = pick_nch(nchunks);
    if (!nch) return fail(HNSW_ERR_UNSUPPORTED, "d=%d > 1024 not supported", d);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(HNSW_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)"); }
    if (device < 0 || device >= ndev) return fail(HNSW_ERR_BAD_ARG, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));

    // ---- levels (lib/ohnsw.ml:781): the first node draws nothing (:774-778) ----
    const double level_mult = 1.0 / std::log((double)M);                                  // :844
    std::vector<uint8_t> lvl((size_t)n, 0);
    {
        uint64_t rng = p->seed;
        for (int64_t i = 1; i < n; ++i) {
            const double u = rng_unit(&rng);
            int l = (int)std::floor(-std::log(u) * level_mult + 0.5);
            lvl[(size_t)i] = (uint8_t)std::min(l, 15);
        }
    }
    int lcap = 1;
    for (int64_t i = 0; i < n; ++i) lcap = std::max(lcap, (int)lvl[(size_t)i] + 1);
    std::vector<int32_t> off((size_t)n, -1);
    int64_t rowsU = 0;
    for (int64_t i = 0; i < n; ++i) if (lvl[(size_t)i]) { off[(size_t)i] = (int32_t)rowsU; rowsU += lvl[(size_t)i]; }

    hnsw_index *idx = new hnsw_index();
    idx->device = device;
    int rc = HNSW_OK;
    const int S0 = 2 * M, SU = M;
    const int64_t stride = padded_stride(d);
    size_t xbytes = 0;
    const int bdiv = p->batch_div > 0 ? p->batch_div : 16;
    const int bmax = p->max_batch > 0 ? p->max_batch : 8192;
    const int cand_stride = (efc + 63) / 64 * 64;
    const int nslot = pick_nslot(efc);
    const int64_t maxrec = (int64_t)bmax * lcap;
    const int64_t max_edges = (int64_t)bmax * S0;
    const int64_t rem_cap = max_edges * 2 * rem_scale;
    uint32_t rem_over = 0;
    hipStream_t st = nullptr;
    void *dNodes = nullptr, *dRecOf = nullptr, *dRecNode = nullptr, *dCandId = nullptr, *dCandKey = nullptr,
         *dCandCnt = nullptr, *dEdges = nullptr, *dEdgesSorted = nullptr, *dRem = nullptr, *dRemCnt = nullptr, *dTemp = nullptr;
    size_t temp_bytes = 0;
    BuildView bv{};
    int32_t *hp_nodes[2] = {nullptr, nullptr}, *hp_rec_of[2] = {nullptr, nullptr}, *hp_rec_node[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int64_t batch_no = 0;
    std::vector<int32_t> rec_begin((size_t)lcap + 1, 0);
    int cur_max = 0, entry = 0;

    if ((rc = upload_vectors(vectors, n, d, row_stride, &idx->dX, &xbytes))) goto done;
    HIP_TRY_B(hipMalloc(&idx->dNbr0, (size_t)n * S0 * 4));
    HIP_TRY_B(hipMemset(idx->dNbr0, 0xFF, (size_t)n * S0 * 4));
    HIP_TRY_B(hipMalloc(&idx->dNbrU, (size_t)std::max<int64_t>(rowsU, 1) * SU * 4));
    HIP_TRY_B(hipMemset(idx->dNbrU, 0xFF, (size_t)std::max<int64_t>(rowsU, 1) * SU * 4));
    HIP_TRY_B(hipMalloc(&idx->dOff, (size_t)n * 4));
    HIP_TRY_B(hipMemcpy(idx->dOff, off.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY_B(hipMalloc(&idx->dLvl, (size_t)n));
    HIP_TRY_B(hipMemcpy(idx->dLvl, lvl.data(), (size_t)n, hipMemcpyHostToDevice));
    if ((rc = upload_upper_ref(off.data(), lvl.data(), n, &idx->dRef))) goto done;
    idx->rowsU = rowsU; idx->iv.rowsU = rowsU;
    {
        IndexView &iv = idx->iv;
        iv.X = (const float *)idx->dX; iv.stride = stride; iv.n = n; iv.d = d; iv.nchunks = nchunks;
        iv.nbr0 = (const int32_t *)idx->dNbr0; iv.S0 = S0; iv.SU = SU;
        iv.nbrU = (const int32_t *)idx->dNbrU; iv.upper_off = (const int32_t *)idx->dOff;
        iv.upper_lvl = (const uint8_t *)idx->dLvl; iv.upper_ref = (const int2 *)idx->dRef;
        iv.max_layer = 0; iv.entry_point = 0; iv.id_base = p->id_base;
    }
    bv.iv = idx->iv; bv.nbr0_w = (int32_t *)idx->dNbr0; bv.nbrU_w = (int32_t *)idx->dNbrU;
    bv.efc = efc; bv.cand_stride = cand_stride;
    bv.vt_bits = efc <= 256 ? 11 : 12;
    while (bv.vt_bits < 16 && ((int64_t)0xFFFF << (bv.vt_bits - 1)) < n) ++bv.vt_bits;   // tags identify ids exactly, 0xFFFF = empty way

    HIP_TRY_B(hipStreamCreate(&st));
    HIP_TRY_B(hipMalloc(&dNodes, (size_t)bmax * 4));
    HIP_TRY_B(hipMalloc(&dRecOf, (size_t)bmax * lcap * 4));
    HIP_TRY_B(hipMalloc(&dRecNode, (size_t)maxrec * 4));
    HIP_TRY_B(hipMalloc(&dCandId, (size_t)maxrec * cand_stride * 4));
    HIP_TRY_B(hipMalloc(&dCandKey, (size_t)maxrec * cand_stride * 4));
    HIP_TRY_B(hipMalloc(&dCandCnt, (size_t)maxrec * 4));
    HIP_TRY_B(hipMalloc(&dEdges, (size_t)max_edges * 8));
    HIP_TRY_B(hipMalloc(&dEdgesSorted, (size_t)max_edges * 8));
    HIP_TRY_B(hipMalloc(&dRem, (size_t)rem_cap * 8));
    HIP_TRY_B(hipMalloc(&dRemCnt, 16));
    HIP_TRY_B(hipMemset(dRemCnt, 0, 16));
    HIP_TRY_B(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, (uint64_t *)dEdges, (uint64_t *)dEdgesSorted, (int)max_edges, 0, 64, st));
    HIP_TRY_B(hipMalloc(&dTemp, std::max<size_t>(temp_bytes, 16)));

    for (int k = 0; k < 2; ++k) {   // pinned, double buffered: batch b+2 waits for batch b's uploads
        HIP_TRY_B(hipHostMalloc((void **)&hp_nodes[k], (size_t)bmax * 4, hipHostMallocDefault));
        HIP_TRY_B(hipHostMalloc((void **)&hp_rec_of[k], (size_t)bmax * lcap * 4, hipHostMallocDefault));
        HIP_TRY_B(hipHostMalloc((void **)&hp_rec_node[k], (size_t)maxrec * 4, hipHostMallocDefault));
        HIP_TRY_B(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    }

    // ---- batches, in node order (fold_cols, lib/ohnsw.ml:848); node 0 is the first entry point ----
    for (int64_t pos = 1; pos < n;) {
        int64_t bsz = std::max<int64_t>(1, std::min<int64_t>(bmax, pos / bdiv));
        int64_t end = std::min(n, pos + bsz);
        for (int64_t j = pos; j < end; ++j)
            if ((int)lvl[(size_t)j] > cur_max) { end = j + 1; break; }       // :832-836
        const int B = (int)(end - pos);
        const int kb = (int)(batch_no & 1);
        if (batch_no >= 2) HIP_TRY_B(hipEventSynchronize(ev[kb]));
        int32_t *h_nodes = hp_nodes[kb], *h_rec_of = hp_rec_of[kb], *h_rec_node = hp_rec_node[kb];
        // records: (layer, batch slot) for layer <= min(level, cur_max), ordered by layer then slot
        int nrec = 0;
        std::fill(h_rec_of, h_rec_of + (size_t)B * lcap, -1);
        for (int l = 0; l <= cur_max; ++l) {
            rec_begin[(size_t)l] = nrec;
            for (int i = 0; i < B; ++i) {
                h_nodes[i] = (int32_t)(pos + i);
                if ((int)lvl[(size_t)(pos + i)] >= l) { h_rec_of[(size_t)i * lcap + l] = nrec; h_rec_node[nrec] = (int32_t)(pos + i); nrec++; }
            }
        }
        rec_begin[(size_t)cur_max + 1] = nrec;
        HIP_TRY_B(hipMemcpyAsync(dNodes, h_nodes, (size_t)B * 4, hipMemcpyHostToDevice, st));
        HIP_TRY_B(hipMemcpyAsync(dRecOf, h_rec_of, (size_t)B * lcap * 4, hipMemcpyHostToDevice, st));
        HIP_TRY_B(hipMemcpyAsync(dRecNode, h_rec_node, (size_t)nrec * 4, hipMemcpyHostToDevice, st));
        HIP_TRY_B(hipEventRecord(ev[kb], st));

        BatchView bt{};
        bt.nodes = (const int32_t *)dNodes; bt.rec_of = (const int32_t *)dRecOf; bt.B = B; bt.lcap = lcap;
        bt.cur_max_layer = cur_max; bt.entry = entry;
        bt.cand_id = (int32_t *)dCandId; bt.cand_key = (uint32_t *)dCandKey; bt.cand_cnt = (int32_t *)dCandCnt;
        bv.iv.max_layer = cur_max; bv.iv.entry_point = entry;
        HIP_TRY_B(p->metric == HNSW_METRIC_L2 ? dispatch_k1<0>(nch, nslot, bv, bt, st) : dispatch_k1<1>(nch, nslot, bv, bt, st));

        for (int l = cur_max; l >= 0; --l) {
            const int rb = rec_begin[(size_t)l], re = rec_begin[(size_t)l + 1];
            if (re == rb) continue;
            const int R = l == 0 ? S0 : SU;                                    // :818
            SelectArgs sa{};
            sa.rec_node = (const int32_t *)dRecNode; sa.rec_begin = rb; sa.rec_end = re; sa.layer = l; sa.R = R;
            sa.edges = (uint64_t *)dEdges;
            HIP_TRY_B(p->metric == HNSW_METRIC_L2 ? dispatch_k2<0>(nch, bv, bt, sa, st) : dispatch_k2<1>(nch, bv, bt, sa, st));
            if (B == 1) {
                // a batch of one node: the link step of Ohnsw.insert exactly, neighbour by neighbour (:820-829)
                hnsw_dev::LinkArgs la{};
                la.q = (int32_t)pos; la.layer = l; la.R = R;
                HIP_TRY_B(p->metric == HNSW_METRIC_L2 ? dispatch_link<0>(nch, bv, la, st) : dispatch_link<1>(nch, bv, la, st));
                continue;
            }
            const int n_edges = (re - rb) * R;
            size_t tb = temp_bytes;
            HIP_TRY_B(hipcub::DeviceRadixSort::SortKeys(dTemp, tb, (uint64_t *)dEdges, (uint64_t *)dEdgesSorted, n_edges, 0, 64, st));
            HIP_TRY_B(hipMemsetAsync(dRemCnt, 0, 4, st));
            MergeArgs ma{};
            ma.edges = (const uint64_t *)dEdgesSorted; ma.n_edges = n_edges; ma.layer = l; ma.R = R;
            ma.removals = (uint64_t *)dRem; ma.rem_cnt = (uint32_t *)dRemCnt; ma.rem_cap = (uint32_t)rem_cap;
            HIP_TRY_B(p->metric == HNSW_METRIC_L2 ? dispatch_k4<0>(nch, bv, ma, st) : dispatch_k4<1>(nch, bv, ma, st));
            // one thread per possible removal: the count is only known on the device (clamped to the buffer there)
            const unsigned rem_threads = (unsigned)std::min<int64_t>(std::max<int64_t>((int64_t)n_edges * 2, 4096), rem_cap);
            hipLaunchKernelGGL(hnsw_dev::build_unlink_kernel, dim3((rem_threads + 255) / 256), dim3(256), 0, st, bv,
                               (const uint64_t *)dRem, (const uint32_t *)dRemCnt, (uint32_t)rem_cap, l);
            HIP_TRY_B(hipGetLastError());
        }
        if ((int)lvl[(size_t)(end - 1)] > cur_max) { cur_max = lvl[(size_t)(end - 1)]; entry = (int)(end - 1); } // :832-836
        pos = end;
        batch_no++;
    }
    hipLaunchKernelGGL(hnsw_dev::build_compact_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (int32_t *)idx->dNbr0, n, S0);
    if (rowsU > 0)
        hipLaunchKernelGGL(hnsw_dev::build_compact_kernel, dim3((unsigned)((rowsU + 3) / 4)), dim3(256), 0, st, (int32_t *)idx->dNbrU, rowsU, SU);
    HIP_TRY_B(hipGetLastError());
    HIP_TRY_B(hipStreamSynchronize(st));
    HIP_TRY_B(hipMemcpy(&rem_over, (const uint32_t *)dRemCnt + 1, 4, hipMemcpyDeviceToHost));
    if (rem_over) { *rem_overflow = true; rc = fail(HNSW_ERR_DEGREE_OVERFLOW, "a build step produced %u symmetric removals, more than the %lld the buffer holds", rem_over, (long long)rem_cap); goto done; }

    idx->iv.max_layer = cur_max; idx->iv.entry_point = entry;
    {
        hnsw_index_info &inf = idx->info;
        inf.n = n; inf.d = d; inf.metric = p->metric; inf.id_base = p->id_base; inf.max_degree0 = S0;
        inf.max_degree = SU; inf.max_layer = cur_max; inf.entry_point = (int64_t)entry + p->id_base;
        inf.device_bytes = (int64_t)(xbytes + (size_t)n * S0 * 4 + (size_t)std::max<int64_t>(rowsU, 1) * SU * 4 + (size_t)n * 13);
        inf.row_stride_bytes = stride * 4; inf.device = device;
    }
done:
    for (void *q : {dNodes, dRecOf, dRecNode, dCandId, dCandKey, dCandCnt, dEdges, dEdgesSorted, dRem, dRemCnt, dTemp}) if (q) (void)hipFree(q);
    for (int k = 0; k < 2; ++k) {
        if (hp_nodes[k]) (void)hipHostFree(hp_nodes[k]);
        if (hp_rec_of[k]) (void)hipHostFree(hp_rec_of[k]);
        if (hp_rec_node[k]) (void)hipHostFree(hp_rec_node[k]);
        if (ev[k]) (void)hipEventDestroy(ev[k]);
    }
    if (st) (void)hipStreamDestroy(st);
    if (!rc) rc = make_byte_rows(idx);      // the finished index serves searches from the byte copy where the data allows
    if (!rc) rc = make_split_rows(idx);     // ... or from split rows where a row ends just past a 128-byte line
    if (rc) { hnsw_index_destroy(idx); return rc; }
    (void)warm_up(idx);                     // ... and its first search call does not pay for the process's code loading (an optimisation:
    prepare_quietly(idx, p->expected_ef, p->expected_semantics);      //  failures are left to the first search) nor for its shape's one-time decisions
    *out = idx;
    return HNSW_OK;
}

extern "C" {

int32_t hnsw_build(const float *vectors, int64_t n, int32_t d, int64_t row_stride,
                   const hnsw_build_params *p, int32_t device, hnsw_index **out) {
    // Removals (links dropped when a neighbour's row is re-selected) are buffered per step; a step that
    // produces more than the buffer holds invalidates the attempt -- a dropped removal would leave an
    // asymmetric link, which Graph.Test.invariant (lib/ohnsw.ml:217-225) forbids -- and the build is repeated
    // with four times the room (deterministic: same graph as if the buffer had been large from the start).
    int32_t rc = HNSW_OK;
    for (int64_t scale = 1; scale <= 64; scale *= 4) {
        bool overflow = false;
        rc = build_attempt(vectors, n, d, row_stride, p, device, out, scale, &overflow);
        if (!overflow) return rc;
    }
    return rc;   // HNSW_ERR_DEGREE_OVERFLOW with the message of the last attempt
}

// ---- select_neighbours operator ----------------------------------------------------------------------
int32_t hnsw_select_neighbours_batch(hnsw_index *idx, const float *targets, int64_t nb, int64_t t_stride,
                                     const int32_t *cand, const int32_t *cand_cnt, int32_t cand_stride,
                                     int32_t num_neighbours, int32_t keep_all_if_few, const int32_t *cand_degree,
                                     int32_t *out, int32_t *out_cnt) {
    if (!idx) return fail(HNSW_ERR_BAD_ARG, "null index");
    if (nb == 0) return HNSW_OK;
    if (nb < 0 || !targets || !cand || !cand_cnt || !out || !out_cnt) return fail(HNSW_ERR_BAD_ARG, "bad buffers");
    if (t_stride < idx->iv.d) return fail(HNSW_ERR_BAD_ARG, "t_stride < d");
    if (num_neighbours < 1 || num_neighbours > 64) return fail(HNSW_ERR_UNSUPPORTED, "num_neighbours must be in 1..64");
    if (cand_stride < 1 || cand_stride > 1024) return fail(HNSW_ERR_UNSUPPORTED, "cand_stride must be in 1..1024");
    const int base = idx->iv.id_base;
    std::vector<int32_t> c0((size_t)nb * cand_stride, 0);
    for (int64_t b = 0; b < nb; ++b) {
        if (cand_cnt[b] < 0 || cand_cnt[b] > cand_stride) return fail(HNSW_ERR_BAD_ARG, "cand_cnt out of range");
        if (keep_all_if_few && cand_cnt[b] <= num_neighbours && cand_cnt[b] > 64) return fail(HNSW_ERR_UNSUPPORTED, "keep-all needs <= 64 candidates");
        if (cand_degree) {
            int forced = 0;
            for (int j = 0; j < cand_cnt[b]; ++j) forced += cand_degree[b * cand_stride + j] <= 1;
            // forced >= M: the reference's loop only checks the bound after an addition, so it can
            // return more than num_neighbours (hnsw_algo.ml:591-592, 600-606); refuse instead of truncating
            if (forced >= num_neighbours && cand_cnt[b] > num_neighbours)
                return fail(HNSW_ERR_DEGREE_OVERFLOW, "do_not_isolate forces %d >= num_neighbours=%d candidates", forced, num_neighbours);
        }
        for (int j = 0; j < cand_cnt[b]; ++j) {
            const int64_t v = (int64_t)cand[b * cand_stride + j] - base;
            if (v < 0 || v >= idx->iv.n) return fail(HNSW_ERR_BAD_ARG, "Vector.get: candidate id out of range");
            c0[(size_t)(b * cand_stride + j)] = (int32_t)v;
        }
    }
    HIP_TRY(hipSetDevice(idx->device));
    DevBuf dT, dC, dN, dO, dOc, dDg;
    int rc;
    const size_t tbytes = ((size_t)(nb - 1) * t_stride + idx->iv.d) * 4;
    if ((rc = dT.ensure(tbytes)) || (rc = dC.ensure(c0.size() * 4)) || (rc = dN.ensure((size_t)nb * 4)) ||
        (rc = dO.ensure((size_t)nb * num_neighbours * 4)) || (rc = dOc.ensure((size_t)nb * 4))) {
        dT.release(); dC.release(); dN.release(); dO.release(); dOc.release();
        return rc;
    }
    auto cleanup = [&]() { dT.release(); dC.release(); dN.release(); dO.release(); dOc.release(); dDg.release(); };
    if (cand_degree) {
        if ((rc = dDg.ensure((size_t)nb * cand_stride * 4))) { cleanup(); return rc; }
        if (hipMemcpy(dDg.p, cand_degree, (size_t)nb * cand_stride * 4, hipMemcpyHostToDevice) != hipSuccess) { cleanup(); return fail(HNSW_ERR_HIP, "upload failed"); }
    }
    if (hipMemcpy(dT.p, targets, tbytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dC.p, c0.data(), c0.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dN.p, cand_cnt, (size_t)nb * 4, hipMemcpyHostToDevice) != hipSuccess) { cleanup(); return fail(HNSW_ERR_HIP, "upload failed"); }
    SelectOpArgs sa{};
    sa.targets = (const float *)dT.p; sa.t_stride = t_stride; sa.cand = (const int32_t *)dC.p; sa.cand_cnt = (const int32_t *)dN.p;
    sa.cand_stride = cand_stride; sa.nb = (int32_t)nb; sa.R = num_neighbours; sa.keep_all_if_few = keep_all_if_few;
    sa.out = (int32_t *)dO.p; sa.out_cnt = (int32_t *)dOc.p; sa.cand_deg = cand_degree ? (const int32_t *)dDg.p : nullptr;
    const size_t lds = ((size_t)4 * cand_stride + 128) * 4;
    const int nch = pick_nch(idx->iv.nchunks);
    dim3 grid((unsigned)nb), block(64);
#define LAUNCH_SEL(N, R_, M_) hipLaunchKernelGGL((hnsw_dev::select_neighbours_kernel<N, R_, M_>), grid, block, lds, 0, idx->iv, sa)
    if (idx->info.metric == HNSW_METRIC_L2) {
        switch (nch) { case 1: LAUNCH_SEL(1, 8, 0); break; case 2: LAUNCH_SEL(2, 4, 0); break; case 4: LAUNCH_SEL(4, 2, 0); break; case 8: LAUNCH_SEL(8, 1, 0); break; default: LAUNCH_SEL(16, 1, 0); }
    } else {
        switch (nch) { case 1: LAUNCH_SEL(1, 8, 1); break; case 2: LAUNCH_SEL(2, 4, 1); break; case 4: LAUNCH_SEL(4, 2, 1); break; case 8: LAUNCH_SEL(8, 1, 1); break; default: LAUNCH_SEL(16, 1, 1); }
    }
#undef LAUNCH_SEL
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dO.p, (size_t)nb * num_neighbours * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_cnt, dOc.p, (size_t)nb * 4, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(HNSW_ERR_HIP, "select_neighbours failed: %s", hipGetErrorString(e));
    for (int64_t i = 0; i < nb * num_neighbours; ++i) if (out[i] >= 0) out[i] += base;
    return HNSW_OK;
}

// ---- export (the inverse of hnsw_index_create's flatten) ------------------------------------------
int32_t hnsw_index_export_layer0(const hnsw_index *idx, int32_t *deg0, int32_t *nbr0) {
    if (!idx || !deg0 || !nbr0) return fail(HNSW_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(idx->device));
    const int64_t n = idx->iv.n; const int S0 = idx->iv.S0; const int base = idx->iv.id_base;
    HIP_TRY(hipMemcpy(nbr0, idx->dNbr0, (size_t)n * S0 * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) {
        int w = 0;
        int32_t *row = nbr0 + i * S0;
        for (int j = 0; j < S0; ++j) if (row[j] >= 0) row[w++] = row[j] + base;
        deg0[i] = w;
        for (int j = w; j < S0; ++j) row[j] = -1;
    }
    return HNSW_OK;
}

int32_t hnsw_index_export_upper_count(const hnsw_index *idx, int32_t layer, int64_t *n_nodes) {
    if (!idx || !n_nodes) return fail(HNSW_ERR_BAD_ARG, "null argument");
    if (layer < 1 || layer > idx->iv.max_layer) return fail(HNSW_ERR_BAD_ARG, "layer %d out of range", layer);
    HIP_TRY(hipSetDevice(idx->device));
    std::vector<uint8_t> lvl((size_t)idx->iv.n);
    HIP_TRY(hipMemcpy(lvl.data(), idx->dLvl, (size_t)idx->iv.n, hipMemcpyDeviceToHost));
    int64_t c = 0;
    for (int64_t i = 0; i < idx->iv.n; ++i) c += lvl[(size_t)i] >= layer;
    *n_nodes = c;
    return HNSW_OK;
}

int32_t hnsw_index_export_upper(const hnsw_index *idx, int32_t layer, int64_t *nodes, int32_t *deg, int32_t *nbr) {
    if (!idx || !nodes || !deg || !nbr) return fail(HNSW_ERR_BAD_ARG, "null argument");
    if (layer < 1 || layer > idx->iv.max_layer) return fail(HNSW_ERR_BAD_ARG, "layer %d out of range", layer);
    HIP_TRY(hipSetDevice(idx->device));
    const int64_t n = idx->iv.n; const int SU = idx->iv.SU; const int base = idx->iv.id_base;
    std::vector<uint8_t> lvl((size_t)n);
    std::vector<int32_t> off((size_t)n), rows((size_t)std::max<int64_t>(idx->rowsU, 1) * SU);
    HIP_TRY(hipMemcpy(lvl.data(), idx->dLvl, (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(off.data(), idx->dOff, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rows.data(), idx->dNbrU, rows.size() * 4, hipMemcpyDeviceToHost));
    int64_t s = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (lvl[(size_t)i] < layer) continue;
        const int32_t *row = &rows[((size_t)off[(size_t)i] + (layer - 1)) * SU];
        int w = 0;
        for (int j = 0; j < SU; ++j) if (row[j] >= 0) nbr[s * SU + w++] = row[j] + base;
        for (int j = w; j < SU; ++j) nbr[s * SU + j] = -1;
        nodes[s] = i + base; deg[s] = w;
        s++;
    }
    return HNSW_OK;
}


// ---- per-layer degree statistics: Hgraph.Stats (lib/hnsw.ml:353-375) -----------------------------
// min_max_connectivity (:361-368) folds the layer's connections map from (1000000, -1, 0, 0., []): min / max / mean of
// the neighbour-list lengths and the list of nodes without a neighbour.  One thread per node over the device-resident
// tables; the keys of layer l >= 1 are the nodes whose level reaches l (upper_ref), of layer 0 every node.
namespace {
struct StatsAcc { unsigned long long cnt, sum, iso; int mi, ma; };

__global__ void __launch_bounds__(256)
layer_stats_kernel(const int32_t *nbr0, int32_t S0, const int32_t *nbrU, int32_t SU, const int2 *upper_ref, int64_t n,
                   int32_t layer, StatsAcc *acc, int64_t *iso_ids, unsigned long long iso_cap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t *row; int w;
    if (layer == 0) { row = nbr0 + i * S0; w = S0; }
    else {
        const int2 ref = upper_ref[i];
        if (ref.y < layer) return;                                   // not a key of this layer
        row = nbrU + ((int64_t)ref.x + (layer - 1)) * SU; w = SU;
    }
    int dg = 0;
    for (int j = 0; j < w; ++j) dg += row[j] >= 0;                   // Neighbours.length
    atomicAdd(&acc->cnt, 1ull); atomicAdd(&acc->sum, (unsigned long long)dg);
    atomicMin(&acc->mi, dg); atomicMax(&acc->ma, dg);
    if (dg == 0) {
        const unsigned long long at = atomicAdd(&acc->iso, 1ull);
        if (at < iso_cap) iso_ids[at] = i;
    }
}

// runs the kernel once; with iso != nullptr also collects the isolated nodes (0-based, unordered)
int layer_stats_device(const hnsw_index *idx, int32_t layer, StatsAcc *out, std::vector<int64_t> *iso) {
    if (layer < 0 || layer > idx->iv.max_layer) return fail(HNSW_ERR_BAD_ARG, "layer %d out of range", layer);
    HIP_TRY(hipSetDevice(idx->device));
    const int64_t n = idx->iv.n;
    StatsAcc h{0, 0, 0, 1000000, -1};
    if (n == 0) { *out = h; if (iso) iso->clear(); return HNSW_OK; }
    DevBuf dAcc, dIso;
    struct Guard { DevBuf &a, &b; ~Guard() { a.release(); b.release(); } } guard{dAcc, dIso};
    int rc;
    if ((rc = dAcc.ensure(sizeof(StatsAcc)))) return rc;
    unsigned long long cap = 0;
    for (int pass = 0; pass < 2; ++pass) {                          // second pass only when there are isolated nodes to list
        HIP_TRY(hipMemcpy(dAcc.p, &h, sizeof h, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(layer_stats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, idx->iv.nbr0, idx->iv.S0,
                           idx->iv.nbrU, idx->iv.SU, idx->iv.upper_ref, n, layer, (StatsAcc *)dAcc.p, (int64_t *)dIso.p, cap);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(out, dAcc.p, sizeof *out, hipMemcpyDeviceToHost));
        if (!iso || out->iso == 0 || pass == 1) break;
        cap = out->iso;
        if ((rc = dIso.ensure((size_t)cap * 8))) return rc;
    }
    if (iso) {
        iso->resize((size_t)out->iso);
        if (out->iso) HIP_TRY(hipMemcpy(iso->data(), dIso.p, (size_t)out->iso * 8, hipMemcpyDeviceToHost));
    }
    return HNSW_OK;
}
} // namespace

int32_t hnsw_index_layer_stats(const hnsw_index *idx, int32_t layer, hnsw_layer_stats *out) {
    if (!idx || !out) return fail(HNSW_ERR_BAD_ARG, "null argument");
    StatsAcc a;
    int rc = layer_stats_device(idx, layer, &a, nullptr);
    if (rc) return rc;
    out->num_nodes = (int64_t)a.cnt; out->min_degree = a.mi; out->max_degree = a.ma;       // empty layer: 1000000 / -1 / nan, as the fold's
    out->mean_degree = (double)a.sum / (double)a.cnt;                                      // initial value leaves them (0. /. 0.)
    out->num_isolated = (int64_t)a.iso;
    return HNSW_OK;
}

int32_t hnsw_index_layer_isolated(const hnsw_index *idx, int32_t layer, int64_t *ids, int64_t cap, int64_t *count) {
    if (!idx || !count || cap < 0 || (cap > 0 && !ids)) return fail(HNSW_ERR_BAD_ARG, "null argument");
    StatsAcc a;
    std::vector<int64_t> iso;
    int rc = layer_stats_device(idx, layer, &a, &iso);
    if (rc) return rc;
    std::sort(iso.begin(), iso.end(), [](int64_t x, int64_t y) { return x > y; });   // key :: isolated during an ascending fold: descending ids
    *count = (int64_t)iso.size();
    for (int64_t i = 0; i < std::min<int64_t>(cap, (int64_t)iso.size()); ++i) ids[i] = iso[(size_t)i] + idx->iv.id_base;
    return HNSW_OK;
}

// ---- flattened-index file: header + vectors + layer 0 + upper layers (little endian) ---------------
namespace {
struct FileHeader {
    char magic[8];        // "HNSWMI35"
    uint32_t version;     // 1: vectors + graph; 2: + the trailer "PREP" (decisions, locality codes)
    int32_t d, metric, id_base, max_degree0, max_degree, max_layer, reserved;
    int64_t n, entry_point;
};
bool wr(FILE *f, const void *p, size_t bytes) { return bytes == 0 || fwrite(p, 1, bytes, f) == bytes; }
bool rd(FILE *f, void *p, size_t bytes) { return bytes == 0 || fread(p, 1, bytes, f) == bytes; }
} // namespace

int32_t hnsw_index_save(const hnsw_index *idx, const char *path) {
    if (!idx || !path) return fail(HNSW_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(idx->device));
    const int64_t n = idx->iv.n; const int d = idx->iv.d; const int S0 = idx->iv.S0, SU = idx->iv.SU;
    FILE *f = fopen(path, "wb");
    if (!f) return fail(HNSW_ERR_BAD_ARG, "cannot open %s for writing", path);
    FileHeader h{};
    memcpy(h.magic, "HNSWMI35", 8); h.version = 2; h.d = d; h.metric = idx->info.metric; h.id_base = idx->iv.id_base;
    h.max_degree0 = S0; h.max_degree = idx->info.max_degree; h.max_layer = idx->iv.max_layer; h.n = n;
    h.entry_point = idx->info.entry_point;
    bool ok = wr(f, &h, sizeof h);
    {   // vectors, unpadded, in chunks
        const int64_t stride = idx->iv.stride;
        const int64_t chunk = std::max<int64_t>(1, (64ll << 20) / (stride * 4));
        std::vector<float> stage((size_t)chunk * stride), packed((size_t)chunk * d);
        for (int64_t r0 = 0; ok && r0 < n; r0 += chunk) {
            const int64_t nr = std::min(chunk, n - r0);
            if (hipMemcpy(stage.data(), (const float *)idx->dX + r0 * stride, (size_t)nr * stride * 4, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
            for (int64_t i = 0; i < nr; ++i) memcpy(&packed[(size_t)i * d], &stage[(size_t)i * stride], (size_t)d * 4);
            ok = wr(f, packed.data(), (size_t)nr * d * 4);
        }
    }
    std::vector<int32_t> deg0((size_t)std::max<int64_t>(n, 1)), nbr0((size_t)std::max<int64_t>(n, 1) * S0);
    if (ok && n > 0) ok = hnsw_index_export_layer0(idx, deg0.data(), nbr0.data()) == HNSW_OK;
    ok = ok && wr(f, deg0.data(), (size_t)n * 4) && wr(f, nbr0.data(), (size_t)n * S0 * 4);
    for (int l = 1; ok && l <= idx->iv.max_layer; ++l) {
        int64_t c = 0;
        ok = hnsw_index_export_upper_count(idx, l, &c) == HNSW_OK;
        std::vector<int64_t> nodes((size_t)std::max<int64_t>(c, 1));
        std::vector<int32_t> deg((size_t)std::max<int64_t>(c, 1)), nbr((size_t)std::max<int64_t>(c, 1) * SU);
        ok = ok && hnsw_index_export_upper(idx, l, nodes.data(), deg.data(), nbr.data()) == HNSW_OK;
        ok = ok && wr(f, &c, 8) && wr(f, nodes.data(), (size_t)c * 8) && wr(f, deg.data(), (size_t)c * 4) && wr(f, nbr.data(), (size_t)c * SU * 4);
    }
    {   // format 2: what the handle has learnt -- the locality codes (if built) and the visited-structure decisions per kernel shape
        std::vector<int32_t> dec;
        list_blk_choices(idx, dec);
        for (const auto &pr : idx->prepared) {                       // prepared shapes without a decision of their own (W in one / two registers): kept
            bool have = false;
            for (size_t i = 0; i + 2 < dec.size(); i += 3) have = have || (dec[i] == pr.first && dec[i + 1] == (pr.second ? 1 : 0));
            if (!have) { dec.push_back(pr.first); dec.push_back(pr.second); dec.push_back(-1); }
        }
        const uint32_t n_dec = (uint32_t)(dec.size() / 3), has_codes = idx->lcode_state == 1 && idx->dLcode ? 1u : 0u;
        ok = ok && wr(f, "PREP", 4) && wr(f, &n_dec, 4) && wr(f, dec.data(), dec.size() * 4) && wr(f, &has_codes, 4);
        if (ok && has_codes) {
            std::vector<int32_t> codes((size_t)n);
            ok = hipMemcpy(codes.data(), idx->dLcode, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess && wr(f, codes.data(), (size_t)n * 4);
        }
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail(HNSW_ERR_HIP, "writing %s failed", path);
    return HNSW_OK;
}

int32_t hnsw_index_load(const char *path, int32_t device, hnsw_index **out) {
    if (!path || !out) return fail(HNSW_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(HNSW_ERR_BAD_ARG, "cannot open %s", path);
    FileHeader h{};
    if (!rd(f, &h, sizeof h) || memcmp(h.magic, "HNSWMI35", 8) != 0 || (h.version != 1 && h.version != 2)) { fclose(f); return fail(HNSW_ERR_BAD_ARG, "%s is not a flattened hnsw index (format 1 or 2)", path); }
    if (h.n < 0 || h.n > 0x7FFFFFF0LL || h.d < 1 || h.max_degree0 < 1 || h.max_degree0 > 64 || h.max_layer < 0 || h.max_layer > 255) { fclose(f); return fail(HNSW_ERR_BAD_ARG, "%s: corrupt header", path); }
    const int64_t n = h.n;
    const int SU = h.max_layer > 0 ? h.max_degree : 1;
    if (h.max_layer > 0 && (SU < 1 || SU > 64)) { fclose(f); return fail(HNSW_ERR_BAD_ARG, "%s: corrupt header", path); }
    std::vector<float> X((size_t)std::max<int64_t>(n, 1) * h.d);
    std::vector<int32_t> deg0((size_t)std::max<int64_t>(n, 1)), nbr0((size_t)std::max<int64_t>(n, 1) * h.max_degree0);
    bool ok = rd(f, X.data(), (size_t)n * h.d * 4) && rd(f, deg0.data(), (size_t)n * 4) && rd(f, nbr0.data(), (size_t)n * h.max_degree0 * 4);
    std::vector<std::vector<int64_t>> nodes((size_t)h.max_layer);
    std::vector<std::vector<int32_t>> deg((size_t)h.max_layer), nbr((size_t)h.max_layer);
    std::vector<hnsw_layer_desc> layers((size_t)std::max(h.max_layer, 1));
    for (int l = 0; ok && l < h.max_layer; ++l) {
        int64_t c = 0;
        ok = rd(f, &c, 8) && c >= 0 && c <= n;
        if (!ok) break;
        nodes[(size_t)l].resize((size_t)std::max<int64_t>(c, 1)); deg[(size_t)l].resize((size_t)std::max<int64_t>(c, 1)); nbr[(size_t)l].resize((size_t)std::max<int64_t>(c, 1) * SU);
        ok = rd(f, nodes[(size_t)l].data(), (size_t)c * 8) && rd(f, deg[(size_t)l].data(), (size_t)c * 4) && rd(f, nbr[(size_t)l].data(), (size_t)c * SU * 4);
        layers[(size_t)l] = hnsw_layer_desc{c, nodes[(size_t)l].data(), deg[(size_t)l].data(), nbr[(size_t)l].data()};
    }
    // format 2: decisions and codes (an unreadable trailer is ignored: everything in it can be made again)
    std::vector<int32_t> dec, codes;
    if (ok && h.version >= 2) {
        char tag[4];
        uint32_t n_dec = 0, has_codes = 0;
        bool tok = rd(f, tag, 4) && memcmp(tag, "PREP", 4) == 0 && rd(f, &n_dec, 4) && n_dec <= 64;
        if (tok) { dec.resize((size_t)n_dec * 3); tok = rd(f, dec.data(), dec.size() * 4) && rd(f, &has_codes, 4); }
        if (tok && has_codes == 1 && n > 0) { codes.resize((size_t)n); tok = rd(f, codes.data(), (size_t)n * 4); }
        if (tok && !codes.empty()) {          // a permutation of 0 .. n-1, or not used
            std::vector<uint8_t> seen((size_t)n, 0);
            for (int64_t i = 0; tok && i < n; ++i) {
                const int32_t c = codes[(size_t)i];
                tok = c >= 0 && c < n && !seen[(size_t)c];
                if (tok) seen[(size_t)c] = 1;
            }
        }
        if (!tok) { dec.clear(); codes.clear(); }
    }
    fclose(f);
    if (!ok) return fail(HNSW_ERR_BAD_ARG, "%s: truncated or corrupt", path);
    hnsw_index_desc d{};
    d.vectors = X.data(); d.n = n; d.d = h.d; d.row_stride = h.d; d.metric = h.metric; d.id_base = h.id_base;
    d.max_degree0 = h.max_degree0; d.max_degree = h.max_degree; d.max_layer = h.max_layer; d.entry_point = h.entry_point;
    d.deg0 = deg0.data(); d.nbr0 = nbr0.data(); d.upper = layers.data();
    const int rc = hnsw_index_create(&d, device, out);   // re-validates every id and degree
    if (rc) return rc;
    hnsw_index *idx = *out;
    if (!codes.empty()) (void)adopt_locality_codes(idx, codes.data());
    for (size_t i = 0; i + 2 < dec.size(); i += 3)
        if (dec[i + 2] >= 0) adopt_blk_choice(idx, dec[i], dec[i + 1], dec[i + 2] > 0);
    // every saved shape is prepared again (residency, code objects); its decision is already in place, nothing is measured
    for (size_t i = 0; i + 2 < dec.size(); i += 3) prepare_quietly(idx, dec[i], dec[i + 1]);
    return HNSW_OK;
}

} // extern "C"
