// hnsw_capi.hip -- host side of libhnsw_mi355x.so: the C ABI of include/hnsw_mi355x.h.
//
// Flattens the reference's graph containers (Ohnsw.Hgraph.t lib/ohnsw.ml:307-312 /
// Hnsw.Ba.Hgraph.t lib/hnsw.ml:342-348, as handed over by the OCaml shim) into HBM-resident
// tables and launches the gfx950 kernels of hnsw_device.hip.h.  No CPU fallback: without a
// usable device every compute entry point fails with HNSW_ERR_NO_DEVICE.
#include "hnsw_internal.h"

#include <mutex>

using hnsw_dev::IndexView;
using hnsw_dev::SearchArgs;
using namespace hnsw_host;

namespace hnsw_host {

thread_local std::string g_last_error;

// Host ranges obtained through hnsw_host_alloc / hnsw_host_register, with the device address the runtime gave them.  The
// host-buffer entry points access a caller's matrix directly from the device (zero-copy) only when it lies inside one of
// THESE ranges: what the runtime's pointer queries say about memory somebody else pinned (or pinned once and freed) is not
// trusted with a kernel's loads and stores.
//
// A range also remembers who may still be READING it: an entry point that returns before the device is done with the
// caller's matrix (hnsw_search_batch_h2d: the kernels read it in place; hnsw_search_submit: the upload is a DMA out of it)
// leaves an event behind its last reader on that stream (range_reader_enqueued).  hnsw_host_unregister / hnsw_host_free wait
// for those events before the pages are unpinned or freed: unregistering early is slow, never a GPU page fault (round 4's
// second suite abort was exactly that: DESIGN.md section 7).
struct InFlight { hipStream_t st; hipEvent_t ev; };
enum { RANGE_ALLOCATED = 0, RANGE_REGISTERED = 1, RANGE_FOREIGN = 2 };   // hipHostMalloc / hipHostRegister by this library / pinned by somebody else
struct HostRange { const char *p; size_t bytes; char *dev; int kind; std::vector<InFlight> readers; };
std::mutex g_ranges_mu;
std::vector<HostRange> g_ranges;

void *registered_device_address(const void *p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_ranges_mu);
    for (const HostRange &r : g_ranges)
        if (r.dev && (const char *)p >= r.p && (const char *)p + bytes <= r.p + r.bytes) return r.dev + ((const char *)p - r.p);
    return nullptr;
}

// `st` (of the current device) has just been given work that reads [p, p + bytes) after the entry point returns
void range_reader_enqueued(const void *p, size_t bytes, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_ranges_mu);
    for (HostRange &r : g_ranges) {
        if (!((const char *)p >= r.p && (const char *)p + bytes <= r.p + r.bytes)) continue;
        // Whatever goes wrong with the event, the guarantee stands: without an event behind the reader the stream is drained
        // HERE, before the entry point returns (slow, never a page unpinned under a running kernel).
        for (InFlight &f : r.readers)
            if (f.st == st) {      // work on one stream is ordered: the later record covers the earlier reader
                if (hipEventRecord(f.ev, st) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(st); }
                return;
            }
        InFlight f{st, nullptr};
        if (hipEventCreateWithFlags(&f.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(st); return; }
        if (hipEventRecord(f.ev, st) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(f.ev); (void)hipStreamSynchronize(st); return; }
        r.readers.push_back(f);
        return;
    }
}

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int upload_vectors(const float *vectors, int64_t n, int d, int64_t row_stride, void **dX, size_t *bytes) {
    const int64_t stride = padded_stride(d);
    const size_t xbytes = (size_t)std::max<int64_t>(n, 1) * stride * sizeof(float);
    *bytes = xbytes;
    if (hipMalloc(dX, xbytes) != hipSuccess) { (void)hipGetLastError(); return fail(HNSW_ERR_OOM, "hipMalloc(%zu) for vectors failed", xbytes); }
    if (n == 0) return HNSW_OK;
    if (stride == row_stride) {
        if (hipMemcpy(*dX, vectors, xbytes, hipMemcpyHostToDevice) != hipSuccess) return fail(HNSW_ERR_HIP, "vector upload failed");
        return HNSW_OK;
    }
    const int64_t chunk_rows = std::max<int64_t>(1, (64ll << 20) / (stride * 4));
    std::vector<float> stage((size_t)chunk_rows * stride);
    for (int64_t r0 = 0; r0 < n; r0 += chunk_rows) {
        const int64_t nr = std::min(chunk_rows, n - r0);
        std::fill(stage.begin(), stage.begin() + (size_t)nr * stride, 0.0f);
        for (int64_t i = 0; i < nr; ++i)
            memcpy(&stage[(size_t)i * stride], vectors + (r0 + i) * row_stride, sizeof(float) * (size_t)d);
        if (hipMemcpy((float *)*dX + r0 * stride, stage.data(), (size_t)nr * stride * 4, hipMemcpyHostToDevice) != hipSuccess)
            return fail(HNSW_ERR_HIP, "vector upload failed");
    }
    return HNSW_OK;
}

int upload_upper_ref(const int32_t *off, const uint8_t *lvl, int64_t n, void **dRef) {
    std::vector<int2> ref((size_t)std::max<int64_t>(n, 1));
    for (int64_t i = 0; i < n; ++i) ref[(size_t)i] = make_int2(off[(size_t)i], (int)lvl[(size_t)i]);
    if (hipMalloc(dRef, ref.size() * sizeof(int2)) != hipSuccess) { (void)hipGetLastError(); return fail(HNSW_ERR_OOM, "hipMalloc for the upper-row table failed"); }
    if (hipMemcpy(*dRef, ref.data(), ref.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) return fail(HNSW_ERR_HIP, "upper-row table upload failed");
    return HNSW_OK;
}

} // namespace hnsw_host

// ---- kernel dispatch ---------------------------------------------------------------------------
// the knn kernel's variants live in hnsw_search_variants.hip, one object per (metric, accept rule, row shape)
#define HNSW_DECL_VARIANT(m, s, f)                                                                          \
    namespace hnsw_host {                                                                                    \
    hipError_t search_launch_##m##_##s##_##f(int nch, int nslot, const IndexView &iv, const SearchArgs &a, hipStream_t st); \
    int search_occupancy_##m##_##s##_##f(int nch, int nslot, size_t lds, int blk);                                    \
    }
HNSW_DECL_VARIANT(0, 0, 0) HNSW_DECL_VARIANT(0, 0, 1) HNSW_DECL_VARIANT(0, 0, 2) HNSW_DECL_VARIANT(0, 0, 3)
HNSW_DECL_VARIANT(0, 1, 0) HNSW_DECL_VARIANT(0, 1, 1) HNSW_DECL_VARIANT(0, 1, 2) HNSW_DECL_VARIANT(0, 1, 3)
HNSW_DECL_VARIANT(1, 0, 0) HNSW_DECL_VARIANT(1, 0, 1) HNSW_DECL_VARIANT(1, 0, 2) HNSW_DECL_VARIANT(1, 0, 3)
HNSW_DECL_VARIANT(1, 1, 0) HNSW_DECL_VARIANT(1, 1, 1) HNSW_DECL_VARIANT(1, 1, 2) HNSW_DECL_VARIANT(1, 1, 3)
#undef HNSW_DECL_VARIANT

// probe queries of the visited-structure measurement (knn_blk_bits): query j = the midpoint between node j * step's vector and
// its first layer-0 neighbour's (the node's own vector if it has none); rows of `stride` floats, padding zero as in the table
__global__ void __launch_bounds__(64)
probe_queries_kernel(const IndexView iv, int64_t step, float *out) {
    const int64_t v = (int64_t)blockIdx.x * step;
    const int32_t nb = iv.nbr0[v * iv.S0];
    const int64_t u = nb >= 0 ? (int64_t)nb : v;
    for (int64_t i = threadIdx.x; i < iv.stride; i += 64)
        out[(int64_t)blockIdx.x * iv.stride + i] = 0.5f * (iv.X[v * iv.stride + i] + iv.X[u * iv.stride + i]);
}

namespace {

typedef hipError_t (*search_launch_fn)(int, int, const IndexView &, const SearchArgs &, hipStream_t);
typedef int (*search_occupancy_fn)(int, int, size_t, int);
const search_launch_fn k_launch[2][2][4] = {
    {{search_launch_0_0_0, search_launch_0_0_1, search_launch_0_0_2, search_launch_0_0_3}, {search_launch_0_1_0, search_launch_0_1_1, search_launch_0_1_2, search_launch_0_1_3}},
    {{search_launch_1_0_0, search_launch_1_0_1, search_launch_1_0_2, search_launch_1_0_3}, {search_launch_1_1_0, search_launch_1_1_1, search_launch_1_1_2, search_launch_1_1_3}}};
const search_occupancy_fn k_occupancy[2][2][4] = {
    {{search_occupancy_0_0_0, search_occupancy_0_0_1, search_occupancy_0_0_2, search_occupancy_0_0_3}, {search_occupancy_0_1_0, search_occupancy_0_1_1, search_occupancy_0_1_2, search_occupancy_0_1_3}},
    {{search_occupancy_1_0_0, search_occupancy_1_0_1, search_occupancy_1_0_2, search_occupancy_1_0_3}, {search_occupancy_1_1_0, search_occupancy_1_1_1, search_occupancy_1_1_2, search_occupancy_1_1_3}}};

// the knn kernel's row format: 2 = byte rows (hnsw_rows8.hip), 3 = split fp32 rows (hnsw_rows_split.hip), else plain fp32
// rows, 1 = every chunk of the lane grid inside the row
inline int variant_full(const hnsw_index *idx) {
    if (idx->iv.X8) return 2;
    if (idx->iv.Xm) return 3;
    return idx->iv.nchunks == 16 * pick_nch(idx->iv.nchunks) ? 1 : 0;
}

template <int METRIC>
hipError_t dispatch_dist(int nch, const IndexView &iv, const float *Q, int64_t qs, int64_t nq,
                         const int32_t *ids, int32_t m, float *out, hipStream_t st) {
    // blockIdx.y strides over a query's ids: enough blocks to fill the chip twice over (8 192 waves: what it holds at four waves
    // per SIMD, twice), no more -- every block starts by loading its query, and a block that then evaluates thirty-two batches
    // amortises that better than one that evaluates four (bench_dist's shape, profiles/r06_dist_ab.txt: 6.07 / 5.77 TB/s at 8 192
    // waves on two boxes, 5.85 / 5.76 at 65 536 -- rounds 1-5's grid: within the noise of one box; HNSW_DIST_WAVES: tuning)
    const int64_t want_waves = env_int("HNSW_DIST_WAVES", 8192);
    const int64_t per_query = std::max<int64_t>(1, (want_waves + std::max<int64_t>(nq, 1) - 1) / std::max<int64_t>(nq, 1));
    const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(256, (m + 15) / 16), per_query));
    dim3 grid((unsigned)nq, gy), block(64);
    switch (nch) {
    case 1: hipLaunchKernelGGL((hnsw_dev::hnsw_distance_kernel<1, METRIC>), grid, block, 0, st, iv, Q, qs, nq, ids, m, out); break;
    case 2: hipLaunchKernelGGL((hnsw_dev::hnsw_distance_kernel<2, METRIC>), grid, block, 0, st, iv, Q, qs, nq, ids, m, out); break;
    case 4: hipLaunchKernelGGL((hnsw_dev::hnsw_distance_kernel<4, METRIC>), grid, block, 0, st, iv, Q, qs, nq, ids, m, out); break;
    case 8: hipLaunchKernelGGL((hnsw_dev::hnsw_distance_kernel<8, METRIC>), grid, block, 0, st, iv, Q, qs, nq, ids, m, out); break;
    default: hipLaunchKernelGGL((hnsw_dev::hnsw_distance_kernel<16, METRIC>), grid, block, 0, st, iv, Q, qs, nq, ids, m, out); break;
    }
    return hipGetLastError();
}

// log2 entries of the knn kernel's visited cache: search_vt_bits' choice, then grown for free.  A kernel variant whose
// REGISTERS cap it at few waves per CU (C3's split-row kernel: 106 VGPRs, 16 waves) leaves LDS unused; a larger cache then
// costs no residency and saves re-evaluations, which on clustered data are many (1.18 M x 100 unit vectors around 256
// directions, M 32, ef 256: 8239 evaluations per query with 2^11 tags against 5015 in the oracle, 6592 with 2^12: 5.22 ->
// 4.36 ms per 10 k batch; 2^13 would halve the residency: 5.17 ms).  So: the largest cache (up to 2^14 tags) that keeps the
// waves per CU the variant reaches with the base size.  Never changes results; "vt_bits" / HNSW_VT_BITS still override.
int knn_vt_bits(hnsw_index *idx, int ef, int semf) {
    const int base = search_vt_bits(idx, ef);
    if (idx->vt_bits_override || env_int("HNSW_VT_BITS", 0) > 0 || !env_int("HNSW_VT_GROW", 1)) return base;
    const int nch = pick_nch(idx->iv.nchunks), nslot = pick_nslot_knn(ef, nch);
    const int vkey = ((nslot * 2 + semf) * 4 + variant_full(idx)) * 32 + base;
    if (idx->vt_grow_key == vkey) return idx->vt_grow_bits;
    const search_occupancy_fn occ = k_occupancy[idx->info.metric == HNSW_METRIC_L2 ? 0 : 1][semf][variant_full(idx)];
    const int occ0 = occ(nch, nslot, hnsw_dev::wave_lds_words(base) * sizeof(uint32_t), 0);
    int b = base;
    while (occ0 > 0 && b < 14 && occ(nch, nslot, hnsw_dev::wave_lds_words(b + 1) * sizeof(uint32_t), 0) >= occ0) ++b;
    idx->vt_grow_key = vkey; idx->vt_grow_bits = b;
    return b;
}

int launch_search_args(hnsw_index *idx, SearchArgs &a, hipStream_t st);

// Visited as bitmap blocks over the locality codes (visited_blocks_mem_add; hnsw_locality.hip) for the kernels with W in four
// or more registers (ef > 128: the walks that visit several times what the tag cache holds): log2 of the block slots, or 0 =
// the tag cache.  The slots are as many as fit the LDS of the waves per CU the variant reaches with the tag cache (C5's and
// C3's kernels, 16 waves per CU: 2^8 blocks = 65 536 codes).
// Which of the two forgets less depends on the DATA -- on clustered vectors the blocks end C5's shape's 40 % repeated
// evaluations (1-4 % left); on structureless vectors no numbering has locality, a block holds one visited node and the tag
// cache wins -- so unless the caller decides (option "visited_blocks" 0 / 1, HNSW_VISITED_BLOCKS), the handle MEASURES, once
// per kernel shape: 256 of the index's own vectors are searched both ways and the evaluations counted; the blocks are taken
// when they save at least 5 %.  The measurement (and building the codes: one descent per node per upper layer, a sort) runs
// inside the first search call that needs the answer and synchronises the device; indices of fewer than 200 000 nodes are
// not measured (a walk cannot visit much more than the tag cache holds).  Never changes results.
// The per-slot code table (n * max_degree0 * 4 bytes: 2.56 GB for 10 M nodes at M 32) is only worth its memory while some kernel
// shape of the handle runs the bitmap blocks.  After a measurement that chose the tag cache -- structureless data -- it is freed
// again unless another shape took the blocks; the per-node codes (n * 4 bytes) stay, so a later measurement for another shape
// re-makes the table with one fill kernel instead of the whole build.  Only called with the device idle (the measurement
// synchronises; nothing in flight reads the table when no shape has chosen it).
void release_unused_lcode0(hnsw_index *idx) {
    if (!idx->dLcode0) return;
    const int mode = idx->blk_mode >= 0 ? idx->blk_mode : env_int("HNSW_VISITED_BLOCKS", -1);
    if (mode == 1) return;                   // the caller asked for blocks wherever they can run: keep the table
    for (auto &c : idx->blk_choice) if (c[0] > 0 || c[1] > 0) return;
    drop_lcode0(idx);
}

// the largest block directory (log2 slots) that keeps the waves per CU of this shape's kernel, 0 = the shape cannot run the blocks
int blk_capacity_bits(hnsw_index *idx, int ef, int semf) {
    const int nch = pick_nch(idx->iv.nchunks), nslot = pick_nslot_knn(ef, nch), vt = knn_vt_bits(idx, ef, semf);
    if (nslot < 3) return 0;
    const search_occupancy_fn occ = k_occupancy[idx->info.metric == HNSW_METRIC_L2 ? 0 : 1][semf][variant_full(idx)];
    // (the waves per CU of the tag-cache kernel, or of the block kernel at its smallest directory if its registers allow fewer)
    const int occ0 = std::min(occ(nch, nslot, hnsw_dev::wave_lds_words(vt) * sizeof(uint32_t), 0),
                              occ(nch, nslot, hnsw_dev::search_lds_words(vt, 6) * sizeof(uint32_t), 1));
    int bits = 0;
    for (int b = 6; b <= 10; ++b)
        if (hnsw_dev::wave_lds_words_blocks(b) * sizeof(uint32_t) <= 65536 && occ0 > 0 &&
            occ(nch, nslot, hnsw_dev::search_lds_words(vt, b) * sizeof(uint32_t), 1) >= occ0) bits = b;
    if (bits == 0) return 0;
    {   // HNSW_BLK_BITS (tests): a smaller directory than fits, down to one set of eight slots -- evictions and contested slots on small graphs
        const int forced = env_int("HNSW_BLK_BITS", 0);
        if (forced >= 3) bits = std::min(bits, forced);
    }
    return bits;
}

// does option "visited_blocks" at -1 consider this shape at all?
bool blk_auto_eligible(const hnsw_index *idx, int nslot) {
    if (idx->iv.n < env_int("HNSW_VISITED_BLOCKS_MIN_N", 200000)) return false;
    // Left to itself the handle only considers the FLOAT32 shapes whose hand-scheduled loop has the block filter (rows of 65..256
    // dimensions -- full, ragged or split --, W in three to eight registers: C3's and C5's kernels): those kernels are bound by
    // row requests, so fewer evaluations are less time.  The byte-row loops have the filter too (an explicit "visited_blocks" 1
    // runs it, and the C++ loop of every other shape), but a byte-row kernel is bound by the LATENCY of a hop, and the filter's
    // four dependent LDS round trips and ~90 vector instructions cost a hop more than the evaluations it saves: the harder
    // SIFT-like set at ef 192, 8 % fewer evaluations, 0.98 -> 1.10 ms per 10 k batch (profiles/r05_ab_bytes_blocks.txt).
    return variant_full(idx) != 2 && idx->iv.nchunks > 16 && idx->iv.nchunks <= 64 && nslot <= 8;
}

int knn_blk_bits(hnsw_index *idx, int ef, int semf) {
    const int nslot = pick_nslot_knn(ef, pick_nch(idx->iv.nchunks));
    const int ls = slot_class(nslot);
    const int mode = idx->blk_mode >= 0 ? idx->blk_mode : env_int("HNSW_VISITED_BLOCKS", -1);
    if (mode == 0 || nslot < 3 || idx->lcode_state < 0) return 0;
    if (mode < 0 && !blk_auto_eligible(idx, nslot)) return 0;
    int &choice = idx->blk_choice[ls][semf];
    if (choice >= 0) return choice;
    const int bits = blk_capacity_bits(idx, ef, semf), vt = knn_vt_bits(idx, ef, semf);
    if (bits == 0) return choice = 0;
    if (build_locality_codes(idx) != HNSW_OK || idx->lcode_state != 1 || !idx->dLcode0) return choice = 0;
    if (mode == 1) return choice = bits;
    // measure: 256 probe queries -- the midpoint between every (n / 256)-th vector of the index and its first layer-0 neighbour:
    // where queries of the data's own distribution fall, without being stored vectors themselves (a stored vector finds itself
    // at distance 0 and its walk is shorter than a real query's) -- searched both ways, k = 1, evaluations counted
    const int64_t nq = std::min<int64_t>(256, idx->iv.n), step = idx->iv.n / nq;
    DevBuf out, probes;
    if (out.ensure((size_t)nq * 16) != HNSW_OK || probes.ensure((size_t)nq * idx->iv.stride * 4) != HNSW_OK) { out.release(); probes.release(); return choice = 0; }
    hipLaunchKernelGGL(probe_queries_kernel, dim3((unsigned)nq), dim3(64), 0, nullptr, idx->iv, step, (float *)probes.p);
    uint64_t sum[2] = {0, 0};
    bool ok = hipGetLastError() == hipSuccess;
    for (int pass = 0; pass < 2 && ok; ++pass) {
        SearchArgs a{};
        a.Q = (const float *)probes.p; a.q_stride = idx->iv.stride; a.nq = nq; a.ef = ef; a.k = 1; a.fill = HNSW_FILL_OHNSW; a.sem = semf;
        a.vt_bits = vt; a.blk_bits = pass ? bits : 0; a.prio_tail = 0x7FFFFFFF;
        a.out_ids = (int32_t *)out.p; a.out_dist = (float *)out.p + nq; a.out_ndist = (uint32_t *)out.p + 2 * nq; a.out_nhops = (uint32_t *)out.p + 3 * nq;
        std::vector<uint32_t> nd((size_t)nq);
        ok = launch_search_args(idx, a, nullptr) == HNSW_OK && hipDeviceSynchronize() == hipSuccess &&
             hipMemcpy(nd.data(), a.out_ndist, (size_t)nq * 4, hipMemcpyDeviceToHost) == hipSuccess;
        for (uint32_t v : nd) sum[pass] += v;
    }
    out.release(); probes.release();
    if (!ok) { (void)hipGetLastError(); choice = 0; release_unused_lcode0(idx); return 0; }
    choice = (double)sum[1] <= 0.95 * (double)sum[0] ? bits : 0;
    release_unused_lcode0(idx);     // the tags won and no other shape uses the blocks: the per-slot table (GBs at 10 M nodes) goes again
    if (env_int("HNSW_DEBUG_VISITED", 0))
        fprintf(stderr, "hnsw: visited set for ef %d (W in %d registers, rule %d): tag cache %.0f evaluations per probe query, 2^%d blocks %.0f -> %s\n",
                ef, nslot, semf, (double)sum[0] / (double)nq, bits, (double)sum[1] / (double)nq, choice ? "blocks" : "tags");
    return choice;
}
// LDS bytes of one search wave of this index at this ef (no padding)
size_t knn_lds_bytes(hnsw_index *idx, int ef, int semf) {
    return hnsw_dev::search_lds_words(knn_vt_bits(idx, ef, semf), knn_blk_bits(idx, ef, semf)) * sizeof(uint32_t);
}

// how many one-wave workgroups of the search kernel for this ef are resident on the device at once (no LDS padding)
int64_t resident_queries(hnsw_index *idx, int ef, int semf) {
    // cached in the handle; the answer depends on the kernel variant's registers and LDS
    const int nch = pick_nch(idx->iv.nchunks), nslot = pick_nslot_knn(ef, nch);
    const size_t lds = knn_lds_bytes(idx, ef, semf);
    const int vkey = (nslot * 2 + semf) * 4 + variant_full(idx);
    if (idx->resident_queries && idx->resident_nslot == vkey && idx->resident_lds == lds) return idx->resident_queries;
    const int per_cu = k_occupancy[idx->info.metric == HNSW_METRIC_L2 ? 0 : 1][semf][variant_full(idx)](nch, nslot, lds, knn_blk_bits(idx, ef, semf) > 0);
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, idx->device) != hipSuccess) { (void)hipGetLastError(); cus = 0; }
    const int64_t v = (per_cu > 0 && cus > 0) ? (int64_t)per_cu * cus : (int64_t)1 << 40;   // unknown: never reorder
    idx->resident_queries = v; idx->resident_nslot = vkey; idx->resident_lds = lds;
    idx->resident_per_cu = (per_cu > 0 && cus > 0) ? per_cu : 0; idx->cus = cus;
    return v;
}

// LDS bytes to request beyond what a search wave uses, for a launch of nq > resident queries (longest first).
// With one workgroup per query the chip holds `resident` of them and the launch is ceil(nq / resident) passes of
// which the last is part empty: on C2 (10 000 queries, 7168 resident) 4336 waves run one query and stop while 2832
// run two, and the second half of the launch has under half the chip's memory requests in flight.  Holding
// nq / passes queries at a time instead gives every slot the same share of the walk and the launch ends together
// (C2: 0.78 -> 0.74 ms per call).  The only per-launch handle on residency is the LDS a workgroup asks for:
// gfx950 hands LDS out in 1280-byte granules, 128 per CU, so k granules per wave hold floor(128 / k) waves.
// Chosen: the fewest waves per CU that still cover nq / passes.
int balanced_lds_pad(hnsw_index *idx, int64_t nq, int ef, int semf) {
    const int forced = idx->lds_pad >= 0 ? idx->lds_pad : env_int("HNSW_LDS_PAD", -1);
    if (forced >= 0) {     // an explicit request is clamped to what a workgroup may ask for beside its own scratch
        const int64_t base_f = (int64_t)knn_lds_bytes(idx, ef, semf);
        return (int)std::max<int64_t>(0, std::min<int64_t>(std::min(forced, 32768), 65536 - base_f));
    }
    const int64_t resident = resident_queries(idx, ef, semf);
    if (nq <= resident || idx->resident_per_cu <= 0) return 0;
    // byte rows: a quarter of the bytes per evaluation, the launch is bound by the latency of a hop, not by the
    // memory system, and holds as many queries as the registers allow (C2: 0.60 ms per call at 8192 held, 0.65 at 5376)
    if (idx->iv.X8) return 0;
    constexpr int64_t GRANULE = 1280, GRANULES_PER_CU = 128;
    const int64_t base = (int64_t)knn_lds_bytes(idx, ef, semf);
    const int64_t passes = (nq + resident - 1) / resident;
    const int64_t want_per_cu = (nq + passes * idx->cus - 1) / (passes * idx->cus);
    const int64_t k0 = (base + GRANULE - 1) / GRANULE;
    int64_t best = k0;
    for (int64_t k = k0; k * GRANULE <= 65536; ++k) {         // a workgroup may ask for 64 KiB at most
        const int64_t w = std::min<int64_t>(idx->resident_per_cu, GRANULES_PER_CU / k);
        if (w < want_per_cu) break;
        best = k;
    }
    int64_t pad = best > k0 ? best * GRANULE - base : 0;
    // The granule arithmetic above is gfx950's (1280-byte granules, 128 per CU: profiles/r02_residency_sweep.txt).  The
    // runtime has the last word: the padded request must still hold the wanted waves per CU by ITS count, and fit the
    // largest dynamic LDS a workgroup may ask for; otherwise the padding is stepped back.
    if (pad > 0) {
        int max_lds = 0;
        if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, idx->device) != hipSuccess) { (void)hipGetLastError(); max_lds = 65536; }
        const int nch_ = pick_nch(idx->iv.nchunks), nslot_ = pick_nslot_knn(ef, nch_);
        const search_occupancy_fn occ = k_occupancy[idx->info.metric == HNSW_METRIC_L2 ? 0 : 1][semf][variant_full(idx)];
        while (pad > 0 && (base + pad > max_lds || occ(nch_, nslot_, (size_t)(base + pad), knn_blk_bits(idx, ef, semf) > 0) < want_per_cu)) pad = pad > GRANULE ? pad - GRANULE : 0;
    }
    if (idx->debug_last_nq != nq && env_int("HNSW_DEBUG_RESIDENT", 0) && ((idx->debug_last_nq = nq), true))
        fprintf(stderr, "hnsw: nq %lld, %d waves/CU x %d CUs resident, %lld passes -> want %lld waves/CU: LDS %lld + %lld B\n",
                (long long)nq, idx->resident_per_cu, idx->cus, (long long)passes, (long long)want_per_cu, (long long)base, (long long)pad);
    return (int)pad;
}

// Issue priorities of an ordered launch (SearchArgs::prio_head / prio_tail; the kernel's comment says why): the first
// eighth of the blocks that start at once (the walks predicted longest) and, when the launch is one full pass and a
// partial second one, the blocks that have to wait for a slot.  HNSW_PRIO="head,tail" overrides the bounds (tuning).
void launch_priorities(hnsw_index *idx, int64_t nq, int ef, int semf, SearchArgs &a) {
    a.prio_head = 0; a.prio_tail = 0x7FFFFFFF;
    if (const char *e = getenv("HNSW_PRIO")) {
        int h = 0, t = 0x7FFFFFFF;
        if (sscanf(e, "%d,%d", &h, &t) >= 1) { a.prio_head = h; a.prio_tail = t; }
        return;
    }
    const int64_t resident = resident_queries(idx, ef, semf);
    if (idx->resident_per_cu <= 0) return;
    a.prio_head = (int32_t)(std::min(nq, resident) / 8);
    if (nq > resident && nq < 2 * resident) a.prio_tail = (int32_t)resident;
}

int check_params(const hnsw_index *idx, const hnsw_search_params *p) {
    if (!idx) return fail(HNSW_ERR_BAD_ARG, "null index");
    if (!p) return fail(HNSW_ERR_BAD_ARG, "null params");
    if (p->ef < 1 || p->k < 1) return fail(HNSW_ERR_BAD_ARG, "ef and k must be >= 1 (ef=%d k=%d)", p->ef, p->k);
    if (p->k > p->ef)
        return fail(HNSW_ERR_BAD_ARG, "k=%d > ef=%d: the reference silently misbehaves here "
                    "(lib/hnsw_algo.ml:257-258); rejected", p->k, p->ef);
    if (p->ef > 1024) return fail(HNSW_ERR_UNSUPPORTED, "ef=%d > 1024 not supported", p->ef);
    if (p->fill != HNSW_FILL_OHNSW && p->fill != HNSW_FILL_BA) return fail(HNSW_ERR_BAD_ARG, "bad fill %d", p->fill);
    if (p->semantics != HNSW_SEM_OHNSW && p->semantics != HNSW_SEM_FUNCTOR && p->semantics != HNSW_SEM_FUNCTOR_NEAREST_K)
        return fail(HNSW_ERR_BAD_ARG, "bad semantics %d", p->semantics);
    if (idx->iv.entry_point < 0) return fail(HNSW_ERR_EMPTY_INDEX, "knn: empty hgraph");
    return HNSW_OK;
}

} // namespace

// The flagged queries of a launch (status bit 0), listed on the device: map[0 .. cap) = their indices, -1 beyond the count
// (the re-run kernel's blocks with a -1 entry leave at once); map[cap] = how many there were.  One workgroup.
__global__ void __launch_bounds__(1024)
flagged_list_kernel(const uint32_t *status, int64_t nq, int32_t *map, int32_t cap) {
    __shared__ int32_t count;
    if (threadIdx.x == 0) count = 0;
    for (int i = threadIdx.x; i < cap; i += blockDim.x) map[i] = -1;
    __syncthreads();
    for (int64_t q = threadIdx.x; q < nq; q += blockDim.x)
        if (status[q] & 1u) { const int at = atomicAdd(&count, 1); if (at < cap) map[at] = (int32_t)q; }
    __syncthreads();
    if (threadIdx.x == 0) map[cap] = count;
}

namespace hnsw_host {
// the handle's stream for the host-buffer call and its page-locked "any query flagged" word
int ensure_host_call_state(hnsw_index *idx) {
    if (!idx->hs[0]) HIP_TRY(hipStreamCreateWithFlags(&idx->hs[0], hipStreamNonBlocking));
    if (!idx->hFlag) {
        HIP_TRY(hipHostMalloc((void **)&idx->hFlag, 64, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&idx->hFlagDev, idx->hFlag, 0));
    }
    return HNSW_OK;
}

// The one-time costs of a process's FIRST search belong to index construction, which the reference's benchmark times on its own
// (benchmark/benchmark.ml:66-80 before :89-96): the code objects of the search kernels and of the ordering pre-pass (the
// runtime loads a translation unit's code when its first kernel is launched: 1.4 ms and 5.2 ms, tools/cold_probe.py), the
// handle's stream and flag word.  One query (node 0's vector, ef 1) through the plain and through the ordered launch; results
// are discarded.  HNSW_WARM_UP=0 leaves them to the first call (7 ms instead of 0.6 ms for a 10 k batch).
static int warm_up_steps(hnsw_index *idx);
int warm_up(hnsw_index *idx) {
    if (!env_int("HNSW_WARM_UP", 1) || idx->iv.n < 1 || idx->iv.entry_point < 0) return HNSW_OK;
    // Nothing here is needed for a valid index: a failure (a stream, 64 bytes, a dummy launch) clears the HIP error and leaves the
    // one-time costs to the first search call, which reports its own errors.  Callers ignore the return value.
    const int rc = warm_up_steps(idx);
    if (rc) (void)hipGetLastError();
    return rc;
}
static int warm_up_steps(hnsw_index *idx) {
    int rc = ensure_host_call_state(idx);
    if (rc) return rc;
    DevBuf out;
    if ((rc = out.ensure(64))) return rc;
    hnsw_search_params p{};
    p.ef = 1; p.k = 1; p.fill = HNSW_FILL_OHNSW; p.semantics = HNSW_SEM_OHNSW;
    const int mode = idx->order_mode;
    for (int ordered = 0; ordered < 2 && !rc; ++ordered) {
        idx->order_mode = ordered;
        rc = search_batch_device_flag(idx, (const float *)idx->dX, 1, idx->iv.stride, &p, (int32_t *)out.p, (float *)out.p + 1,
                                      nullptr, nullptr, (uint32_t *)out.p + 2, nullptr, idx->hs[0]);
    }
    idx->order_mode = mode;
    const hipError_t e = hipStreamSynchronize(idx->hs[0]);
    out.release();
    if (!rc && e != hipSuccess) rc = fail(HNSW_ERR_HIP, "warm-up search failed: %s", hipGetErrorString(e));
    return rc;
}
} // namespace hnsw_host

// ---- ABI ---------------------------------------------------------------------------------------
extern "C" {

int32_t hnsw_abi_version(void) { return HNSW_ABI_VERSION; }
const char *hnsw_last_error(void) { return hnsw_host::g_last_error.c_str(); }

int32_t hnsw_device_count(int32_t *count) {
    if (!count) return fail(HNSW_ERR_BAD_ARG, "null count");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); return fail(HNSW_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e)); }
    *count = c;
    return HNSW_OK;
}

int32_t hnsw_index_create(const hnsw_index_desc *d, int32_t device, hnsw_index **out) {
    if (!d || !out) return fail(HNSW_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    if (d->n < 0 || d->n > 0x7FFFFFF0LL) return fail(HNSW_ERR_BAD_ARG, "n=%lld out of range", (long long)d->n);
    if (d->d < 1) return fail(HNSW_ERR_BAD_ARG, "d=%d", d->d);
    if (d->row_stride < d->d) return fail(HNSW_ERR_BAD_ARG, "row_stride < d");
    if (d->metric != HNSW_METRIC_L2 && d->metric != HNSW_METRIC_IP) return fail(HNSW_ERR_BAD_ARG, "bad metric %d", d->metric);
    if (d->max_degree0 < 1 || d->max_degree0 > 64) return fail(HNSW_ERR_UNSUPPORTED, "max_degree0=%d must be in 1..64", d->max_degree0);
    if (d->max_layer < 0 || d->max_layer > 255) return fail(HNSW_ERR_BAD_ARG, "max_layer=%d", d->max_layer);
    if (d->max_layer > 0 && (d->max_degree < 1 || d->max_degree > 64)) return fail(HNSW_ERR_UNSUPPORTED, "max_degree=%d must be in 1..64", d->max_degree);
    if (d->n > 0 && (!d->vectors || !d->deg0 || !d->nbr0)) return fail(HNSW_ERR_BAD_ARG, "null vectors/deg0/nbr0");
    if (d->max_layer > 0 && !d->upper) return fail(HNSW_ERR_BAD_ARG, "null upper");
    const int nchunks = (d->d + 3) / 4;
    if (pick_nch(nchunks) == 0) return fail(HNSW_ERR_UNSUPPORTED, "d=%d > 1024 not supported", d->d);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(HNSW_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)"); }
    if (device < 0 || device >= ndev) return fail(HNSW_ERR_BAD_ARG, "device %d out of range (have %d)", device, ndev);

    const int64_t n = d->n;
    const int32_t base = d->id_base;
    const int S0 = d->max_degree0, SU = d->max_layer > 0 ? d->max_degree : 1;
    int64_t ep = d->entry_point - base;
    if (n == 0 || ep < 0) ep = -1;
    if (ep >= n) return fail(HNSW_ERR_BAD_ARG, "Hgraph.set_entry_point: invalid node"); /* lib/ohnsw.ml:343 */

    // ---- layer 0: validate, rebase to 0, pad with -1 ----
    std::vector<int32_t> nbr0((size_t)n * S0, -1);
    for (int64_t i = 0; i < n; ++i) {
        const int dg = d->deg0[i];
        if (dg < 0) return fail(HNSW_ERR_BAD_ARG, "deg0[%lld] < 0", (long long)i);
        if (dg > S0) return fail(HNSW_ERR_DEGREE_OVERFLOW, "node %lld has %d layer-0 neighbours > max_degree0=%d", (long long)i + base, dg, S0);
        for (int j = 0; j < dg; ++j) {
            const int64_t v = (int64_t)d->nbr0[i * S0 + j] - base;
            if (v < 0 || v >= n) return fail(HNSW_ERR_BAD_ARG, "Vector.get: neighbour id %lld of node %lld out of range", (long long)v + base, (long long)i + base); /* lib/ohnsw.ml:25 */
            nbr0[(size_t)i * S0 + j] = (int32_t)v;
        }
    }
    // ---- upper layers: per node, rows for layers 1..lvl(node), contiguous ----
    std::vector<uint8_t> lvl((size_t)std::max<int64_t>(n, 1), 0);
    for (int l = 1; l <= d->max_layer; ++l) {
        const hnsw_layer_desc &L = d->upper[l - 1];
        if (L.n_nodes < 0 || (L.n_nodes > 0 && (!L.nodes || !L.deg || !L.nbr))) return fail(HNSW_ERR_BAD_ARG, "layer %d: null arrays", l);
        for (int64_t s = 0; s < L.n_nodes; ++s) {
            const int64_t v = L.nodes[s] - base;
            if (v < 0 || v >= n) return fail(HNSW_ERR_BAD_ARG, "layer %d: node id out of range", l);
            lvl[(size_t)v] = (uint8_t)std::max<int>(lvl[(size_t)v], l);
        }
    }
    std::vector<int32_t> off((size_t)std::max<int64_t>(n, 1), -1);
    int64_t rowsU = 0;
    for (int64_t i = 0; i < n; ++i) if (lvl[(size_t)i]) { off[(size_t)i] = (int32_t)rowsU; rowsU += lvl[(size_t)i]; }
    if (rowsU > 0x7FFFFFF0LL) return fail(HNSW_ERR_UNSUPPORTED, "too many upper rows");
    std::vector<int32_t> nbrU((size_t)std::max<int64_t>(rowsU, 1) * SU, -1);
    for (int l = 1; l <= d->max_layer; ++l) {
        const hnsw_layer_desc &L = d->upper[l - 1];
        for (int64_t s = 0; s < L.n_nodes; ++s) {
            const int64_t v = L.nodes[s] - base;
            const int dg = L.deg[s];
            if (dg < 0) return fail(HNSW_ERR_BAD_ARG, "layer %d: negative degree", l);
            if (dg > SU) return fail(HNSW_ERR_DEGREE_OVERFLOW, "node %lld has %d neighbours on layer %d > max_degree=%d", (long long)v + base, dg, l, SU);
            int32_t *row = &nbrU[((size_t)off[(size_t)v] + (l - 1)) * SU];
            for (int j = 0; j < dg; ++j) {
                const int64_t u = (int64_t)L.nbr[s * SU + j] - base;
                if (u < 0 || u >= n) return fail(HNSW_ERR_BAD_ARG, "layer %d: neighbour id out of range", l);
                row[j] = (int32_t)u;
            }
        }
    }

    HIP_TRY(hipSetDevice(device));
    hnsw_index *idx = new hnsw_index();
    idx->device = device;
    auto bail = [&](int code) { hnsw_index_destroy(idx); return code; };

    // ---- vectors: rows zero-padded to a multiple of 64 B so every float4 chunk is in bounds ----
    const int64_t stride = padded_stride(d->d);
    size_t xbytes = 0;
    { int rcv = upload_vectors(d->vectors, n, d->d, d->row_stride, &idx->dX, &xbytes); if (rcv) return bail(rcv); }
    auto upload = [&](void **dst, const void *src, size_t bytes) -> bool {
        if (hipMalloc(dst, std::max<size_t>(bytes, 16)) != hipSuccess) return false;
        return bytes == 0 || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!upload(&idx->dNbr0, nbr0.data(), nbr0.size() * 4) || !upload(&idx->dNbrU, nbrU.data(), nbrU.size() * 4) ||
        !upload(&idx->dOff, off.data(), off.size() * 4) || !upload(&idx->dLvl, lvl.data(), lvl.size())) {
        (void)hipGetLastError();
        return bail(fail(HNSW_ERR_OOM, "graph upload failed"));
    }

    { int rcr = upload_upper_ref(off.data(), lvl.data(), n, &idx->dRef); if (rcr) return bail(rcr); }
    IndexView &iv = idx->iv;
    iv.upper_ref = (const int2 *)idx->dRef;
    iv.X = (const float *)idx->dX; iv.stride = stride; iv.n = n; iv.d = d->d; iv.nchunks = nchunks;
    iv.nbr0 = (const int32_t *)idx->dNbr0; iv.S0 = S0; iv.SU = SU;
    iv.nbrU = (const int32_t *)idx->dNbrU; iv.upper_off = (const int32_t *)idx->dOff;
    iv.upper_lvl = (const uint8_t *)idx->dLvl;
    iv.max_layer = d->max_layer; iv.entry_point = (int32_t)ep; iv.id_base = base;
    idx->rowsU = rowsU; iv.rowsU = rowsU;

    hnsw_index_info &inf = idx->info;
    inf.n = n; inf.d = d->d; inf.metric = d->metric; inf.id_base = base; inf.max_degree0 = S0;
    inf.max_degree = d->max_degree; inf.max_layer = d->max_layer; inf.entry_point = ep < 0 ? base - 1 : ep + base;
    inf.device_bytes = (int64_t)(xbytes + nbr0.size() * 4 + nbrU.size() * 4 + off.size() * 12 + lvl.size());
    inf.row_stride_bytes = stride * 4; inf.device = device;
    { int rc8 = make_byte_rows(idx); if (rc8) return bail(rc8); }
    { int rcs = make_split_rows(idx); if (rcs) return bail(rcs); }
    (void)warm_up(idx);      // an optimisation: whatever fails in it is left to the first search call
    prepare_quietly(idx, d->expected_ef, d->expected_semantics);
    *out = idx;
    return HNSW_OK;
}

int32_t hnsw_index_destroy(hnsw_index *idx) {
    if (!idx) return HNSW_OK;
    if (idx->device >= 0) (void)hipSetDevice(idx->device);
    for (void *p : {idx->dX, idx->dX8, idx->dXm, idx->dTail0, idx->dLcode, idx->dLcode0, idx->dNbr0, idx->dNbrU, idx->dOff, idx->dLvl, idx->dRef}) if (p) (void)hipFree(p);
    idx->dFbSlab.release(); idx->dFbMap.release();
    idx->sQ.release(); idx->sIds.release(); idx->sDist.release(); idx->sNd.release(); idx->sNh.release(); idx->sSt.release(); idx->sFlag.release();
    (void)hipDeviceSynchronize();                      // requests never waited for
    for (hnsw_request *r : idx->all_requests) {
        r->q.release(); r->ids.release(); r->dist.release(); r->nd.release(); r->nh.release(); r->st.release(); r->flag.release();
        delete r;
    }
    for (hipStream_t st : idx->hs) if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t e : idx->tev) (void)hipEventDestroy(e);
    for (auto &o : idx->order_scratch) if (o.p) (void)hipFree(o.p);
    if (idx->hFlag) (void)hipHostFree(idx->hFlag);
    if (idx->hSmall) (void)hipHostFree(idx->hSmall);
    delete idx;
    return HNSW_OK;
}

int32_t hnsw_index_get_info(const hnsw_index *idx, hnsw_index_info *info) {
    if (!idx || !info) return fail(HNSW_ERR_BAD_ARG, "null argument");
    *info = idx->info;
    info->row_format = idx->iv.X8 ? HNSW_ROWS_BYTES : idx->iv.Xm ? HNSW_ROWS_SPLIT : HNSW_ROWS_F32;
    return HNSW_OK;
}

int32_t hnsw_index_locality_codes(hnsw_index *idx, int32_t *out) {
    if (!idx || !out) return fail(HNSW_ERR_BAD_ARG, "null argument");
    int rc = build_locality_codes(idx);
    if (rc) return rc;
    if (idx->lcode_state != 1) return fail(HNSW_ERR_UNSUPPORTED, "no locality codes: the index has no upper layer (or no room for the tables)");
    HIP_TRY(hipMemcpy(out, idx->dLcode, (size_t)idx->iv.n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipDeviceSynchronize());
    release_unused_lcode0(idx);              // introspection needs the per-node codes only
    return HNSW_OK;
}

int32_t hnsw_index_visited_blocks(hnsw_index *idx, const hnsw_search_params *params, int32_t *log2_slots) {
    if (!log2_slots) return fail(HNSW_ERR_BAD_ARG, "null argument");
    int rc = check_params(idx, params);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(idx->device));
    *log2_slots = knn_blk_bits(idx, params->ef, params->semantics ? 1 : 0);
    return HNSW_OK;
}

int32_t hnsw_index_prepare(hnsw_index *idx, const hnsw_search_params *params) {
    int rc = check_params(idx, params);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(idx->device));
    const int semf = params->semantics ? 1 : 0;
    (void)knn_blk_bits(idx, params->ef, semf);            // the visited structure of this kernel shape (codes, measurement)
    (void)resident_queries(idx, params->ef, semf);        // the shape's residency, cached in the handle
    // the code object of the shape's translation unit: one query (node 0's vector) through the plain and the ordered launch
    if ((rc = ensure_host_call_state(idx))) return rc;
    DevBuf out;
    if ((rc = out.ensure(64))) return rc;
    hnsw_search_params p = *params;
    p.k = 1;
    const int mode = idx->order_mode;
    for (int ordered = 0; ordered < 2 && !rc; ++ordered) {
        idx->order_mode = ordered;
        rc = search_batch_device_flag(idx, (const float *)idx->dX, 1, idx->iv.stride, &p, (int32_t *)out.p, (float *)out.p + 1,
                                      nullptr, nullptr, (uint32_t *)out.p + 2, nullptr, idx->hs[0]);
    }
    idx->order_mode = mode;
    const hipError_t e = hipStreamSynchronize(idx->hs[0]);
    out.release();
    if (rc) return rc;
    if (e != hipSuccess) return fail(HNSW_ERR_HIP, "hnsw_index_prepare: the trial search failed: %s", hipGetErrorString(e));
    const std::pair<int, int> key(params->ef, params->semantics);
    if (std::find(idx->prepared.begin(), idx->prepared.end(), key) == idx->prepared.end()) idx->prepared.push_back(key);
    return HNSW_OK;
}

extern "C++" {
namespace hnsw_host {
void prepare_quietly(hnsw_index *idx, int32_t ef, int32_t semantics) {
    if (ef <= 0 || idx->iv.n < 1 || idx->iv.entry_point < 0) return;
    hnsw_search_params p{};
    p.ef = ef; p.k = 1; p.fill = HNSW_FILL_OHNSW; p.semantics = semantics;
    const std::string keep = g_last_error;
    if (hnsw_index_prepare(idx, &p) != HNSW_OK) { (void)hipGetLastError(); g_last_error = keep; }     // an optimisation: the search reports its own errors
}
void adopt_blk_choice(hnsw_index *idx, int32_t ef, int32_t semantics, bool blocks) {
    if (ef < 1 || ef > 1024) return;
    const int semf = semantics ? 1 : 0, nslot = pick_nslot_knn(ef, pick_nch(idx->iv.nchunks));
    if (nslot < 3 || idx->blk_choice[slot_class(nslot)][semf] >= 0) return;
    int bits = 0;
    if (blocks && idx->lcode_state == 1 && (bits = blk_capacity_bits(idx, ef, semf)) > 0 && materialise_lcode0(idx) == HNSW_OK && idx->dLcode0) {
        idx->blk_choice[slot_class(nslot)][semf] = bits;
        return;
    }
    if (!blocks) idx->blk_choice[slot_class(nslot)][semf] = 0;       // (blocks that cannot be honoured here stay undecided: measured on demand)
}
void list_blk_choices(const hnsw_index *idx, std::vector<int32_t> &out3) {
    static const int rep_ef[SLOT_CLASSES] = {64, 128, 192, 256, 384, 512, 1024};     // an ef of every slot class
    const bool wide = pick_nch(idx->iv.nchunks) == 2 || pick_nch(idx->iv.nchunks) == 4;
    for (int c = 0; c < SLOT_CLASSES; ++c)
        for (int s = 0; s < 2; ++s) {
            if (idx->blk_choice[c][s] < 0) continue;
            if (!wide && (c == 2 || c == 4)) continue;              // three / six registers: rows of 65..256 dimensions only
            out3.push_back(rep_ef[c]); out3.push_back(s); out3.push_back(idx->blk_choice[c][s] > 0 ? 1 : 0);
        }
}
} // namespace hnsw_host
}

int32_t hnsw_index_row_bytes(const hnsw_index *idx, int64_t *row_bytes) {
    if (!idx || !row_bytes) return fail(HNSW_ERR_BAD_ARG, "null argument");
    *row_bytes = idx->iv.X8 ? (int64_t)idx->iv.d : (int64_t)idx->iv.d * 4;
    return HNSW_OK;
}

int32_t hnsw_index_set_option(hnsw_index *idx, const char *name, int64_t value) {
    if (!idx || !name) return fail(HNSW_ERR_BAD_ARG, "null argument");
    // (an option that changes the kernel variant or its LDS makes the cached per-shape choices stale: the visited structure's
    // directory size follows the variant's occupancy, the residency follows the LDS)
    auto forget_shape_choices = [&]() {
        for (auto &c : idx->blk_choice) c[0] = c[1] = -1;
        idx->resident_queries = 0; idx->vt_grow_key = -1;
    };
    if (!strcmp(name, "vt_bits")) { idx->vt_bits_override = (int)value; forget_shape_choices(); return HNSW_OK; }
    if (!strcmp(name, "lds_pad")) { idx->lds_pad = value < 0 ? -1 : (int)std::min<int64_t>(value, 32768); return HNSW_OK; }
    if (!strcmp(name, "byte_rows")) {     // 0: search the fp32 rows even where a byte copy exists; otherwise: use it where it exists
        idx->iv.X8 = value != 0 ? (const uint8_t *)idx->dX8 : nullptr;
        forget_shape_choices();
        return HNSW_OK;
    }
    if (!strcmp(name, "split_rows")) {    // 0: search the plain fp32 rows even where a split copy exists; -1: ... and free the copy; otherwise: use it where it exists
        if (value < 0 && idx->dXm) {
            HIP_TRY(hipSetDevice(idx->device));
            HIP_TRY(hipDeviceSynchronize());             // launches that still read the copy
            idx->info.device_bytes -= idx->iv.n * idx->iv.stride_m + idx->iv.n * idx->iv.S0 * 16 * idx->iv.tail_chunks;
            (void)hipFree(idx->dXm); (void)hipFree(idx->dTail0);
            idx->dXm = nullptr; idx->dTail0 = nullptr; idx->iv.tail0 = nullptr;
        }
        idx->iv.Xm = value > 0 ? (const float *)idx->dXm : nullptr;
        forget_shape_choices();
        return HNSW_OK;
    }
    if (!strcmp(name, "time_kernels")) { idx->time_kernels = value != 0; return HNSW_OK; }
    if (!strcmp(name, "visited_blocks")) {   // -1: measured per kernel shape (default); 0: the tag cache; 1: bitmap blocks wherever the codes can be built
        idx->blk_mode = value < 0 ? -1 : (value ? 1 : 0);
        forget_shape_choices();               // (the LDS per wave, hence the residency, may differ)
        return HNSW_OK;
    }
    if (!strcmp(name, "device_fallback_slab_bytes")) {
        // room for the tie lists of value / (4 n) flagged queries per call (a flagged query may need a slot per node)
        HIP_TRY(hipSetDevice(idx->device));
        HIP_TRY(hipDeviceSynchronize());
        idx->dFbSlab.release(); idx->dFbMap.release(); idx->fb_queries = 0;
        if (value <= 0) return HNSW_OK;
        const int64_t per_query = std::max<int64_t>(idx->iv.n, 1) * 4;
        const int64_t q = std::min<int64_t>(value / per_query, 65536);
        if (q < 1) return fail(HNSW_ERR_BAD_ARG, "device_fallback_slab_bytes=%lld holds no query: one needs 4 n = %lld bytes", (long long)value, (long long)per_query);
        int rc;
        if ((rc = idx->dFbSlab.ensure((size_t)(q * per_query))) || (rc = idx->dFbMap.ensure((size_t)(q + 1) * 4))) { idx->dFbSlab.release(); idx->dFbMap.release(); return rc; }
        idx->fb_queries = q;
        return HNSW_OK;
    }
    if (!strcmp(name, "order_queries")) { idx->order_mode = value < 0 ? -1 : (value ? 1 : 0); return HNSW_OK; }
    return fail(HNSW_ERR_BAD_ARG, "unknown option %s", name);
}

extern "C++" {
namespace {
int launch_search_args(hnsw_index *idx, SearchArgs &a, hipStream_t st);
}
namespace hnsw_host {
int search_check(const hnsw_index *idx, const hnsw_search_params *p) { return check_params(idx, p); }
// the exactness fallback's launch: the `c` flagged queries listed in qmap are searched again with a global
// slab for their tie lists; results overwrite their rows of d_ids / d_dist (launched on the null stream)
int search_rerun_device(hnsw_index *idx, const float *d_queries, int64_t nq, int64_t q_stride, const hnsw_search_params *p,
                        int32_t *d_ids, float *d_dist, uint32_t *d_nd, uint32_t *d_nh, uint32_t *d_st,
                        const int32_t *qmap, int64_t c, uint32_t *slab, int32_t cap, hipStream_t st) {
    SearchArgs a{};
    a.Q = d_queries; a.q_stride = q_stride; a.nq = c; a.ef = p->ef; a.k = p->k;
    a.fill = p->fill; a.sem = p->semantics;
    a.vt_bits = knn_vt_bits(idx, p->ef, p->semantics ? 1 : 0);
    a.blk_bits = knn_blk_bits(idx, p->ef, p->semantics ? 1 : 0);
    a.out_ids = d_ids; a.out_dist = d_dist; a.out_ndist = d_nd; a.out_nhops = d_nh; a.out_status = d_st;
    a.qmap = qmap; a.q_limit = nq; a.ovf_g = slab; a.ovf_gcap = cap; a.prio_tail = 0x7FFFFFFF;
    return launch_search_args(idx, a, st);
}
} // namespace hnsw_host
} // extern "C++"
namespace {
int launch_search_args(hnsw_index *idx, SearchArgs &a, hipStream_t st) {
    const int nch = pick_nch(idx->iv.nchunks), nslot = pick_nslot_knn(a.ef, nch);
    hipError_t e = k_launch[idx->info.metric == HNSW_METRIC_L2 ? 0 : 1][a.sem ? 1 : 0][variant_full(idx)](nch, nslot, idx->iv, a, st);
    if (e != hipSuccess) return fail(HNSW_ERR_HIP, "search kernel launch failed: %s", hipGetErrorString(e));
    return HNSW_OK;
}
} // namespace

extern "C++" {
namespace hnsw_host {
int search_batch_device_flag(hnsw_index *idx, const float *d_queries, int64_t nq, int64_t q_stride,
                             const hnsw_search_params *params, int32_t *d_ids, float *d_dist,
                             uint32_t *d_ndist, uint32_t *d_nhops, uint32_t *d_status, uint32_t *d_any_flag, void *stream,
                             float *d_stage);
}
}
int32_t hnsw_search_batch_device(hnsw_index *idx, const float *d_queries, int64_t nq, int64_t q_stride,
                                 const hnsw_search_params *params, int32_t *d_ids, float *d_dist,
                                 uint32_t *d_ndist, uint32_t *d_nhops, uint32_t *d_status, void *stream) {
    return search_batch_device_flag(idx, d_queries, nq, q_stride, params, d_ids, d_dist, d_ndist, d_nhops, d_status, nullptr, stream);
}
extern "C++" int hnsw_host::search_batch_device_flag(hnsw_index *idx, const float *d_queries, int64_t nq, int64_t q_stride,
                                 const hnsw_search_params *params, int32_t *d_ids, float *d_dist,
                                 uint32_t *d_ndist, uint32_t *d_nhops, uint32_t *d_status, uint32_t *d_any_flag, void *stream,
                                 float *d_stage) {
    int rc = check_params(idx, params);
    if (rc) return rc;
    if (nq < 0 || nq > 0x7FFFFFFFLL) return fail(HNSW_ERR_BAD_ARG, "nq out of range");
    if (nq == 0) return HNSW_OK;
    if (!d_queries || !d_ids || !d_dist) return fail(HNSW_ERR_BAD_ARG, "null buffer");
    if (q_stride < idx->iv.d) return fail(HNSW_ERR_BAD_ARG, "q_stride < d");
    HIP_TRY(hipSetDevice(idx->device));
    SearchArgs a{};
    a.Q = d_queries; a.q_stride = q_stride; a.nq = nq; a.ef = params->ef; a.k = params->k; a.fill = params->fill; a.sem = params->semantics;
    a.vt_bits = knn_vt_bits(idx, params->ef, params->semantics ? 1 : 0);
    a.blk_bits = knn_blk_bits(idx, params->ef, params->semantics ? 1 : 0);
    a.out_ids = d_ids; a.out_dist = d_dist; a.out_ndist = d_ndist; a.out_nhops = d_nhops; a.out_status = d_status;
    a.any_flag = d_any_flag;
    a.prio_tail = 0x7FFFFFFF;
    // A batch larger than the chip holds at once is searched longest walk first (hnsw_order.hip):
    // per-query results are unchanged, the launch's drain phase is made of short walks.
    void *block = nullptr;
    hipEvent_t *ev = nullptr;
    if (idx->time_kernels && idx->tev_used + 3 <= 3 * 4096) {
        while (idx->tev.size() < idx->tev_used + 3) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            idx->tev.push_back(e);
        }
        ev = &idx->tev[idx->tev_used];      // claimed (tev_used advanced) only once all three are recorded
        HIP_TRY(hipEventRecord(ev[0], (hipStream_t)stream));
    }
    const int mode = idx->order_mode >= 0 ? idx->order_mode : env_int("HNSW_ORDER_QUERIES", -1);
    // Ordered when more than half of what the chip holds: a batch that fits is faster too with its long walks
    // dispatched first and spread over the CUs (C2, 7168 queries: 0.59 -> 0.45 ms byte rows, 0.68 -> 0.65 ms fp32);
    // below that the pre-pass costs more than it returns.
    if (mode != 0 && (mode == 1 || 2 * nq > resident_queries(idx, params->ef, params->semantics ? 1 : 0))) {
        rc = order_longest_first(idx, d_queries, nq, q_stride, d_stage, (hipStream_t)stream, &block, &a.qmap, &a.pre_entry, &a.pre_key, &a.pre_nd, &a.pre_layer);
        if (rc) return rc;
        if (d_stage) a.Q = d_stage;        // the descent kernel left a device-resident copy of the (host-resident) queries
        a.q_limit = nq;
        a.lds_pad = balanced_lds_pad(idx, nq, params->ef, params->semantics ? 1 : 0);
        launch_priorities(idx, nq, params->ef, params->semantics ? 1 : 0, a);
    }
    if (ev) HIP_TRY(hipEventRecord(ev[1], (hipStream_t)stream));
    rc = launch_search_args(idx, a, (hipStream_t)stream);
    if (!rc && idx->fb_queries > 0 && d_status && !d_any_flag) {
        // opt-in exact mode of the device-pointer entry point (option "device_fallback_slab_bytes"): the queries the launch
        // flagged are listed on the device and searched again with the slab, on the caller's stream, no host round trip.
        // The re-run rewrites their results, counters and status words (bit 0 clear: a slab slot per node cannot overflow);
        // with more flagged queries than the slab holds, the ones left over keep their flag.  Two small launches per call.
        const int32_t cap = (int32_t)idx->fb_queries;
        hipLaunchKernelGGL(flagged_list_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const uint32_t *)d_status, nq, (int32_t *)idx->dFbMap.p, cap);
        if (hipGetLastError() != hipSuccess) return fail(HNSW_ERR_HIP, "flagged-query listing failed");
        // (at most min(cap, nq) queries can be listed: no more blocks than that; the blocks past the count leave at once)
        rc = search_rerun_device(idx, a.Q, nq, q_stride, params, d_ids, d_dist, d_ndist, d_nhops, d_status, (const int32_t *)idx->dFbMap.p,
                                 std::min<int64_t>(cap, nq), (uint32_t *)idx->dFbSlab.p, (int32_t)std::min<int64_t>(idx->iv.n, 0x7FFFFFFF), (hipStream_t)stream);
    }
    if (ev && !rc) {
        HIP_TRY(hipEventRecord(ev[2], (hipStream_t)stream));
        idx->tev_used += 3;                 // an early return above leaves the triple unclaimed: nothing half-recorded is ever read
        idx->tev_ordered.resize(idx->tev_used / 3);
        idx->tev_ordered[idx->tev_used / 3 - 1] = block != nullptr;
    }
    return rc;
}

int32_t hnsw_index_kernel_times(hnsw_index *idx, double *search_ms, double *prepass_ms, int32_t *calls) {
    if (!idx) return fail(HNSW_ERR_BAD_ARG, "null index");
    HIP_TRY(hipSetDevice(idx->device));
    double s = 0, p = 0;
    const size_t n = idx->tev_used / 3;
    for (size_t i = 0; i < n; ++i) {
        float a = 0, b = 0;
        HIP_TRY(hipEventSynchronize(idx->tev[3 * i + 2]));
        HIP_TRY(hipEventElapsedTime(&a, idx->tev[3 * i], idx->tev[3 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&b, idx->tev[3 * i + 1], idx->tev[3 * i + 2]));
        if (idx->tev_ordered[i]) p += a;
        s += b;
    }
    idx->tev_used = 0;
    if (search_ms) *search_ms = n ? s / (double)n : 0.0;
    if (prepass_ms) *prepass_ms = n ? p / (double)n : 0.0;
    if (calls) *calls = (int32_t)n;
    return HNSW_OK;
}

int32_t hnsw_search_batch(hnsw_index *idx, const float *queries, int64_t nq, int64_t q_stride,
                          const hnsw_search_params *params, int32_t *out_ids, float *out_dist,
                          uint32_t *out_ndist, uint32_t *out_nhops) {
    int rc = check_params(idx, params);
    if (rc) return rc;
    if (nq == 0) return HNSW_OK;
    if (nq < 0 || !queries || !out_ids || !out_dist) return fail(HNSW_ERR_BAD_ARG, "bad buffers");
    if (q_stride < idx->iv.d) return fail(HNSW_ERR_BAD_ARG, "q_stride < d");
    HIP_TRY(hipSetDevice(idx->device));
    const int k = params->k;
    const size_t qbytes = ((size_t)(nq - 1) * q_stride + idx->iv.d) * sizeof(float);
    if ((rc = idx->sQ.ensure(qbytes)) || (rc = idx->sIds.ensure((size_t)nq * k * 4)) ||
        (rc = idx->sDist.ensure((size_t)nq * k * 4)) || (rc = idx->sNd.ensure((size_t)nq * 4)) ||
        (rc = idx->sNh.ensure((size_t)nq * 4)) || (rc = idx->sSt.ensure((size_t)nq * 4)) || (rc = idx->sFlag.ensure(16)))
        return rc;
    // Upload, search (ordered longest walk first when the batch is larger than the chip holds) and download on one
    // of the handle's streams, ONE stream synchronisation at the end.
    //
    // Page-locked matrices of the caller (hnsw_host_alloc / hnsw_host_register) are not copied at all: the device reads the queries straight
    // from the caller's matrix -- each query once, by the wave that searches it (the descent pre-pass keeps a device copy
    // for the search kernel), so the 5 MB of a 10 k x 128 batch cross PCIe UNDER the descent instead of before it -- and
    // the kernel writes each query's results straight into the caller's result matrices as the query finishes, so there is
    // no download step behind the launch either.  Pageable matrices are staged through hipMemcpyAsync as before.  Whether
    // any query needs the exactness fallback comes back as one word (pinned host memory, the kernel stores it), not as a
    // scan of nq status words.  (Splitting the batch into chunks on two streams to overlap copies and search measured
    // 1.08 against 1.06 ms in round 1.)  HNSW_ZERO_COPY=0 switches the direct access off.
    if ((rc = ensure_host_call_state(idx))) return rc;
    hipStream_t st = idx->hs[0];
    // device address of a range the caller registered with hnsw_host_register, or nullptr
    auto mapped = [&](const void *p, size_t bytes) -> void * {
        static const int enabled = env_int("HNSW_ZERO_COPY", 1);
        if (!enabled || !p || bytes == 0) return nullptr;
        return registered_device_address(p, bytes);
    };
    const float *zq = (const float *)mapped(queries, qbytes);
    int32_t *zi = (int32_t *)mapped(out_ids, (size_t)nq * k * 4);
    float *zd = (float *)mapped(out_dist, (size_t)nq * k * 4);
    uint32_t *znd = out_ndist ? (uint32_t *)mapped(out_ndist, (size_t)nq * 4) : nullptr;
    uint32_t *znh = out_nhops ? (uint32_t *)mapped(out_nhops, (size_t)nq * 4) : nullptr;
    if (!zi || !zd) zi = nullptr, zd = nullptr;                      // results: both matrices or neither
    // A SMALL batch from ordinary memory (a single query: Ohnsw.knn, test/test.ml:122) goes through a page-locked block of the
    // handle's own instead of three staged copies: the queries are copied into it by the host, the device reads them and writes
    // the results there, the host copies them out -- what is left of the call is one launch and one synchronisation.
    constexpr size_t SMALL = 32768;
    const size_t rbytes = (size_t)nq * k * 4;
    const bool small = !zq && !zi && !znd && !znh && qbytes <= SMALL && rbytes <= SMALL && (size_t)nq * 4 <= SMALL && env_int("HNSW_SMALL_CALLS", 1);
    if (small) {
        if (!idx->hSmall) {
            HIP_TRY(hipHostMalloc((void **)&idx->hSmall, 5 * SMALL, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void **)&idx->hSmallDev, idx->hSmall, 0));
        }
        memcpy(idx->hSmall, queries, qbytes);
        zq = (const float *)idx->hSmallDev;
        zi = (int32_t *)(idx->hSmallDev + SMALL); zd = (float *)(idx->hSmallDev + 2 * SMALL);
        if (out_ndist) znd = (uint32_t *)(idx->hSmallDev + 3 * SMALL);
        if (out_nhops) znh = (uint32_t *)(idx->hSmallDev + 4 * SMALL);
    }
    auto small_out = [&]() {
        if (!small) return;
        memcpy(out_ids, idx->hSmall + SMALL, rbytes); memcpy(out_dist, idx->hSmall + 2 * SMALL, rbytes);
        if (out_ndist) memcpy(out_ndist, idx->hSmall + 3 * SMALL, (size_t)nq * 4);
        if (out_nhops) memcpy(out_nhops, idx->hSmall + 4 * SMALL, (size_t)nq * 4);
    };
    const float *dQ = zq ? zq : (const float *)idx->sQ.p;            // where the queries can be read from the device
    int32_t *dI = zi ? zi : (int32_t *)idx->sIds.p;
    float *dD = zi ? zd : (float *)idx->sDist.p;
    uint32_t *dNd = znd ? znd : (uint32_t *)idx->sNd.p, *dNh = znh ? znh : (uint32_t *)idx->sNh.p;
    *(volatile uint32_t *)idx->hFlag = 0;
    if (!zq) HIP_TRY(hipMemcpyAsync(idx->sQ.p, queries, qbytes, hipMemcpyHostToDevice, st));
    rc = search_batch_device_flag(idx, dQ, nq, q_stride, params, dI, dD, dNd, dNh, (uint32_t *)idx->sSt.p, idx->hFlagDev, st,
                                  zq ? (float *)idx->sQ.p : nullptr);
    if (rc) { (void)hipStreamSynchronize(st); return rc; }
    auto copy_out = [&](hipStream_t s_) -> int {
        if (!zi) {
            HIP_TRY(hipMemcpyAsync(out_ids, idx->sIds.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s_));
            HIP_TRY(hipMemcpyAsync(out_dist, idx->sDist.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s_));
        }
        if (out_ndist && !znd) HIP_TRY(hipMemcpyAsync(out_ndist, idx->sNd.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s_));
        if (out_nhops && !znh) HIP_TRY(hipMemcpyAsync(out_nhops, idx->sNh.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s_));
        return HNSW_OK;
    };
    rc = copy_out(st);
    {   // no return while a copy into the caller's arrays may still be queued
        const hipError_t es = hipStreamSynchronize(st);
        if (rc) return rc;
        if (es != hipSuccess) return fail(HNSW_ERR_HIP, "search failed: %s", hipGetErrorString(es));
    }
    if (!(*(volatile uint32_t *)idx->hFlag & 1u)) { small_out(); return HNSW_OK; }      // (the flag: written by the kernel, pinned host memory)
    // Exactness fallback for queries whose tie-overflow stack outgrew its LDS slots (rare: the rows
    // of the whole batch are then copied out again)
    int64_t n_rerun = 0;
    rc = rerun_overflowed(idx, nq, (const uint32_t *)idx->sSt.p,
                          [&](const int32_t *qmap, int64_t c, uint32_t *slab, int32_t cap) {
                              return search_rerun_device(idx, dQ, nq, q_stride, params, dI, dD, dNd, dNh, (uint32_t *)idx->sSt.p,
                                                         qmap, c, slab, cap, nullptr);
                          }, &n_rerun);
    if (rc) return rc;
    if (n_rerun > 0) {
        rc = copy_out(nullptr);
        const hipError_t es = hipDeviceSynchronize();
        if (rc) return rc;
        if (es != hipSuccess) return fail(HNSW_ERR_HIP, "result download failed: %s", hipGetErrorString(es));
    }
    small_out();
    return HNSW_OK;
}

int32_t hnsw_search_batch_h2d(hnsw_index *idx, const float *queries, int64_t nq, int64_t q_stride,
                              const hnsw_search_params *params, int32_t *d_ids, float *d_dist,
                              uint32_t *d_ndist, uint32_t *d_nhops, uint32_t *d_status, void *stream) {
    int rc = check_params(idx, params);
    if (rc) return rc;
    if (nq == 0) return HNSW_OK;
    if (nq < 0 || nq > 0x7FFFFFFFLL || !queries || !d_ids || !d_dist) return fail(HNSW_ERR_BAD_ARG, "bad buffers");
    if (q_stride < idx->iv.d) return fail(HNSW_ERR_BAD_ARG, "q_stride < d");
    HIP_TRY(hipSetDevice(idx->device));
    const size_t qbytes = ((size_t)(nq - 1) * q_stride + idx->iv.d) * sizeof(float);
    if ((rc = idx->sQ.ensure(qbytes))) return rc;
    static const int zero_copy = env_int("HNSW_ZERO_COPY", 1);
    const float *zq = zero_copy ? (const float *)registered_device_address(queries, qbytes) : nullptr;
    if (!zq) HIP_TRY(hipMemcpyAsync(idx->sQ.p, queries, qbytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    // registered matrix: read by the device directly, the pre-pass (when there is one) leaves the device copy in sQ
    rc = search_batch_device_flag(idx, zq ? zq : (const float *)idx->sQ.p, nq, q_stride, params, d_ids, d_dist, d_ndist, d_nhops, d_status,
                                  nullptr, stream, zq ? (float *)idx->sQ.p : nullptr);
    // the call returns while the device still reads the caller's matrix (in place, or as the source of the DMA above): the
    // range remembers it, so that hnsw_host_unregister / hnsw_host_free wait instead of pulling the pages from under a kernel
    range_reader_enqueued(queries, qbytes, (hipStream_t)stream);
    return rc;
}

int32_t hnsw_search_submit(hnsw_index *idx, const float *queries, int64_t nq, int64_t q_stride,
                           const hnsw_search_params *params, hnsw_request **out) {
    if (!out) return fail(HNSW_ERR_BAD_ARG, "null out");
    *out = nullptr;
    int rc = check_params(idx, params);
    if (rc) return rc;
    if (nq < 1 || nq > 0x7FFFFFFFLL || !queries) return fail(HNSW_ERR_BAD_ARG, "bad buffers (nq=%lld)", (long long)nq);
    if (q_stride < idx->iv.d) return fail(HNSW_ERR_BAD_ARG, "q_stride < d");
    HIP_TRY(hipSetDevice(idx->device));
    hnsw_request *r;
    if (!idx->free_requests.empty()) { r = idx->free_requests.back(); idx->free_requests.pop_back(); }
    else { r = new hnsw_request(); idx->all_requests.push_back(r); }
    auto give_back = [&](int code) { idx->free_requests.push_back(r); return code; };
    r->idx = idx; r->nq = nq; r->q_stride = q_stride; r->params = *params;
    r->stream = idx->next_stream; idx->next_stream = (idx->next_stream + 1) & 3;
    if (!idx->hs[r->stream] && hipStreamCreateWithFlags(&idx->hs[r->stream], hipStreamNonBlocking) != hipSuccess)
        return give_back(fail(HNSW_ERR_HIP, "hipStreamCreate failed"));
    const int k = params->k;
    const size_t qbytes = ((size_t)(nq - 1) * q_stride + idx->iv.d) * sizeof(float);
    if ((rc = r->q.ensure(qbytes)) || (rc = r->ids.ensure((size_t)nq * k * 4)) || (rc = r->dist.ensure((size_t)nq * k * 4)) ||
        (rc = r->nd.ensure((size_t)nq * 4)) || (rc = r->nh.ensure((size_t)nq * 4)) || (rc = r->st.ensure((size_t)nq * 4)) ||
        (rc = r->flag.ensure(16)))
        return give_back(rc);
    hipStream_t st = idx->hs[r->stream];
    if (hipMemcpyAsync(r->q.p, queries, qbytes, hipMemcpyHostToDevice, st) != hipSuccess)
        return give_back(fail(HNSW_ERR_HIP, "query upload failed"));
    range_reader_enqueued(queries, qbytes, st);    // page-locked source: the DMA above outlives this call (see hnsw_host_unregister)
    if (hipMemsetAsync(r->flag.p, 0, 4, st) != hipSuccess) return give_back(fail(HNSW_ERR_HIP, "hipMemsetAsync failed"));
    rc = search_batch_device_flag(idx, (const float *)r->q.p, nq, q_stride, params, (int32_t *)r->ids.p, (float *)r->dist.p,
                                  (uint32_t *)r->nd.p, (uint32_t *)r->nh.p, (uint32_t *)r->st.p, (uint32_t *)r->flag.p, st);
    if (rc) return give_back(rc);
    // the results follow the search on the request's stream: hnsw_search_wait only has to wait for them
    r->host_flag = 0;
    idx->live_requests++;
    *out = r;
    return HNSW_OK;
}

int32_t hnsw_search_wait(hnsw_request *r, int32_t *out_ids, float *out_dist, uint32_t *out_ndist, uint32_t *out_nhops) {
    if (!r || !r->idx) return fail(HNSW_ERR_BAD_ARG, "null request");
    hnsw_index *idx = r->idx;
    auto done = [&](int code) { r->idx = nullptr; idx->live_requests--; idx->free_requests.push_back(r); return code; };
    if (!out_ids || !out_dist) return done(fail(HNSW_ERR_BAD_ARG, "null result buffers"));
    if (hipSetDevice(idx->device) != hipSuccess) return done(fail(HNSW_ERR_HIP, "hipSetDevice failed"));
    hipStream_t st = idx->hs[r->stream];
    const int k = r->params.k;
    const int64_t nq = r->nq;
    // results and the "any query flagged" word in one go; the exactness fallback (as in hnsw_search_batch) only if set
    uint32_t flag = 0;
    {
        hipError_t e0 = hipMemcpyAsync(out_ids, r->ids.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st);
        if (e0 == hipSuccess) e0 = hipMemcpyAsync(out_dist, r->dist.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st);
        if (e0 == hipSuccess && out_ndist) e0 = hipMemcpyAsync(out_ndist, r->nd.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st);
        if (e0 == hipSuccess && out_nhops) e0 = hipMemcpyAsync(out_nhops, r->nh.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st);
        if (e0 == hipSuccess) e0 = hipMemcpyAsync(&flag, r->flag.p, 4, hipMemcpyDeviceToHost, st);
        // synchronise whatever happened: copies already queued target the caller's arrays and `flag` (a stack word),
        // and the request goes back to the pool only once its stream is idle
        const hipError_t es = hipStreamSynchronize(st);
        if (e0 == hipSuccess) e0 = es;
        if (e0 != hipSuccess) return done(fail(HNSW_ERR_HIP, "search failed: %s", hipGetErrorString(e0)));
    }
    if (!(flag & 1u)) return done(HNSW_OK);
    int rc = rerun_overflowed(idx, nq, (const uint32_t *)r->st.p,
                              [&](const int32_t *qmap, int64_t c, uint32_t *slab, int32_t cap) {
                                  return search_rerun_device(idx, (const float *)r->q.p, nq, r->q_stride, &r->params, (int32_t *)r->ids.p,
                                                             (float *)r->dist.p, (uint32_t *)r->nd.p, (uint32_t *)r->nh.p, (uint32_t *)r->st.p,
                                                             qmap, c, slab, cap, st);
                              });
    if (rc) return done(rc);
    hipError_t e = hipMemcpyAsync(out_ids, r->ids.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(out_dist, r->dist.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && out_ndist) e = hipMemcpyAsync(out_ndist, r->nd.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && out_nhops) e = hipMemcpyAsync(out_nhops, r->nh.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st);
    { const hipError_t es = hipStreamSynchronize(st); if (e == hipSuccess) e = es; }
    if (e != hipSuccess) return done(fail(HNSW_ERR_HIP, "result download failed: %s", hipGetErrorString(e)));
    return done(HNSW_OK);
}

namespace {
void remember_range(const void *p, size_t bytes, void *dev, int kind) {
    std::lock_guard<std::mutex> lk(g_ranges_mu);
    g_ranges.push_back({(const char *)p, bytes, (char *)dev, kind, {}});
}
// takes the range that STARTS at p off the list (no launch can pick it for direct access any more) and waits for the
// readers that earlier asynchronous calls left on it; false: no such range
bool retire_range(const void *p, int *kind) {
    std::vector<InFlight> readers;
    {
        std::lock_guard<std::mutex> lk(g_ranges_mu);
        size_t i = 0;
        while (i < g_ranges.size() && g_ranges[i].p != (const char *)p) ++i;
        if (i == g_ranges.size()) return false;
        *kind = g_ranges[i].kind;
        readers.swap(g_ranges[i].readers);
        g_ranges.erase(g_ranges.begin() + (long)i);
    }
    for (InFlight &f : readers) {
        if (hipEventSynchronize(f.ev) != hipSuccess) (void)hipGetLastError();
        (void)hipEventDestroy(f.ev);
    }
    return true;
}
} // namespace

int32_t hnsw_host_register(void *p, int64_t bytes) {
    if (!p || bytes <= 0) return fail(HNSW_ERR_BAD_ARG, "hnsw_host_register: null buffer or bytes <= 0");
    const char *b = (const char *)p, *e_ = b + bytes;
    {   // this library's own list decides about overlaps (the runtime answers "success" for an array that merely STARTS
        // inside an existing registration and leaves the rest pageable: measured on ROCm 7.2)
        std::lock_guard<std::mutex> lk(g_ranges_mu);
        for (const HostRange &r : g_ranges) {
            if (b >= r.p && e_ <= r.p + r.bytes) return HNSW_OK;          // registered already: twice is not an error
            if (b < r.p + r.bytes && r.p < e_)
                return fail(HNSW_ERR_BAD_ARG, "hnsw_host_register: part of the %lld-byte array is registered already, the rest is not "
                            "(unregister the shorter range first)", (long long)bytes);
        }
    }
    auto locked = [](const void *q) {
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    hipError_t e = hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        // Page-locked by somebody else -- registered (hipErrorHostMemoryAlreadyRegistered) or allocated page-locked (hipHostMalloc,
        // a torch pin_memory tensor: the runtime then answers "invalid argument") --: theirs to unpin.  Copies out of it run at
        // pinned speed anyway; the library neither maps it for direct access nor ever unregisters it.
        if (e == hipErrorHostMemoryAlreadyRegistered || (locked(b) && locked(e_ - 1))) {
            remember_range(p, (size_t)bytes, nullptr, RANGE_FOREIGN);
            return HNSW_OK;
        }
        return fail(HNSW_ERR_HIP, "hipHostRegister(%lld bytes) failed: %s", (long long)bytes, hipGetErrorString(e));
    }
    // "success" is believed only if every page answers as page-locked host memory (at most 4096 probes: the step grows with
    // the array) -- a foreign registration that covers only the array's beginning, or its two ends, leaves pageable pages
    // that a kernel's loads would fault on
    const size_t step = std::max<size_t>(4096, (((size_t)bytes / 4096) + 4095) / 4096 * 4096);
    bool whole = locked(e_ - 1);
    for (size_t o = 0; whole && o < (size_t)bytes; o += step) whole = locked(b + o);
    if (!whole) {
        if (hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
        return fail(HNSW_ERR_BAD_ARG, "hnsw_host_register: part of the %lld-byte array is registered already (by someone else), the rest is not",
                    (long long)bytes);
    }
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) { (void)hipGetLastError(); dev = nullptr; }   // no mapping: copies only
    remember_range(p, (size_t)bytes, dev, RANGE_REGISTERED);
    return HNSW_OK;
}

int32_t hnsw_host_unregister(void *p) {
    if (!p) return fail(HNSW_ERR_BAD_ARG, "hnsw_host_unregister: null buffer");
    int kind = 0;
    {   // a block of hnsw_host_alloc is freed with hnsw_host_free, not unregistered
        std::lock_guard<std::mutex> lk(g_ranges_mu);
        for (const HostRange &r : g_ranges)
            if (r.p == (const char *)p && r.kind == RANGE_ALLOCATED) return fail(HNSW_ERR_BAD_ARG, "hnsw_host_unregister: this block came from hnsw_host_alloc (hnsw_host_free releases it)");
    }
    if (!retire_range(p, &kind))       // (waits for the asynchronous calls that still read the range)
        return fail(HNSW_ERR_BAD_ARG, "hnsw_host_unregister: no range registered through hnsw_host_register starts at %p", p);
    if (kind == RANGE_FOREIGN) return HNSW_OK;                 // not this library's registration to undo
    hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(HNSW_ERR_HIP, "hipHostUnregister failed: %s", hipGetErrorString(e)); }
    return HNSW_OK;
}

int32_t hnsw_host_alloc(void **out, int64_t bytes) {
    if (!out || bytes <= 0) return fail(HNSW_ERR_BAD_ARG, "hnsw_host_alloc: null out or bytes <= 0");
    *out = nullptr;
    void *p = nullptr, *dev = nullptr;
    hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(e == hipErrorOutOfMemory ? HNSW_ERR_OOM : HNSW_ERR_HIP, "hipHostMalloc(%lld bytes) failed: %s", (long long)bytes, hipGetErrorString(e)); }
    if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) { (void)hipGetLastError(); dev = nullptr; }
    remember_range(p, (size_t)bytes, dev, RANGE_ALLOCATED);
    *out = p;
    return HNSW_OK;
}

int32_t hnsw_host_free(void *p) {
    if (!p) return HNSW_OK;
    int kind = 0;
    {
        std::lock_guard<std::mutex> lk(g_ranges_mu);
        bool mine = false;
        for (const HostRange &r : g_ranges) mine = mine || (r.p == (const char *)p && r.kind == RANGE_ALLOCATED);
        if (!mine) return fail(HNSW_ERR_BAD_ARG, "hnsw_host_free: %p is not a block of hnsw_host_alloc", p);
    }
    (void)retire_range(p, &kind);      // (waits for the asynchronous calls that still read the block)
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(HNSW_ERR_HIP, "hipHostFree failed: %s", hipGetErrorString(e)); }
    return HNSW_OK;
}

int32_t hnsw_knn(hnsw_index *idx, const float *query, const hnsw_search_params *params,
                 int32_t *out_ids, float *out_dist, int32_t *out_count) {
    if (!idx || !query) return fail(HNSW_ERR_BAD_ARG, "null argument");
    int rc = hnsw_search_batch(idx, query, 1, idx->iv.d, params, out_ids, out_dist, nullptr, nullptr);
    if (rc) return rc;
    if (out_count) {
        int c = 0;
        while (c < params->k && out_ids[c] >= idx->iv.id_base) ++c;
        *out_count = c;
    }
    return HNSW_OK;
}

int32_t hnsw_distance_batch_device(hnsw_index *idx, const float *d_queries, int64_t nq, int64_t q_stride,
                                   const int32_t *d_ids, int32_t m, float *d_out, void *stream) {
    if (!idx) return fail(HNSW_ERR_BAD_ARG, "null index");
    if (nq == 0 || m == 0) return HNSW_OK;
    if (nq < 0 || m < 0 || !d_queries || !d_ids || !d_out) return fail(HNSW_ERR_BAD_ARG, "bad buffers");
    if (q_stride < idx->iv.d) return fail(HNSW_ERR_BAD_ARG, "q_stride < d");
    HIP_TRY(hipSetDevice(idx->device));
    const int nch = pick_nch(idx->iv.nchunks);
    hipError_t e = idx->info.metric == HNSW_METRIC_L2
                       ? dispatch_dist<0>(nch, idx->iv, d_queries, q_stride, nq, d_ids, m, d_out, (hipStream_t)stream)
                       : dispatch_dist<1>(nch, idx->iv, d_queries, q_stride, nq, d_ids, m, d_out, (hipStream_t)stream);
    if (e != hipSuccess) return fail(HNSW_ERR_HIP, "distance kernel launch failed: %s", hipGetErrorString(e));
    return HNSW_OK;
}

int32_t hnsw_distance_batch(hnsw_index *idx, const float *queries, int64_t nq, int64_t q_stride,
                            const int32_t *ids, int32_t m, float *out) {
    if (!idx) return fail(HNSW_ERR_BAD_ARG, "null index");
    if (nq == 0 || m == 0) return HNSW_OK;
    if (nq < 0 || m < 0 || !queries || !ids || !out) return fail(HNSW_ERR_BAD_ARG, "bad buffers");
    for (int64_t i = 0; i < nq * m; ++i) {
        const int64_t v = (int64_t)ids[i] - idx->iv.id_base;
        if (v < 0 || v >= idx->iv.n) return fail(HNSW_ERR_BAD_ARG, "Vector.get: id out of range");
    }
    HIP_TRY(hipSetDevice(idx->device));
    int rc;
    const size_t qbytes = ((size_t)(nq - 1) * q_stride + idx->iv.d) * sizeof(float);
    if ((rc = idx->sQ.ensure(qbytes)) || (rc = idx->sIds.ensure((size_t)nq * m * 4)) || (rc = idx->sDist.ensure((size_t)nq * m * 4))) return rc;
    HIP_TRY(hipMemcpy(idx->sQ.p, queries, qbytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(idx->sIds.p, ids, (size_t)nq * m * 4, hipMemcpyHostToDevice));
    rc = hnsw_distance_batch_device(idx, (const float *)idx->sQ.p, nq, q_stride, (const int32_t *)idx->sIds.p, m, (float *)idx->sDist.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, idx->sDist.p, (size_t)nq * m * 4, hipMemcpyDeviceToHost));
    return HNSW_OK;
}

} // extern "C"
