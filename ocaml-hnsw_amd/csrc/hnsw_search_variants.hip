// hnsw_search_variants.hip -- instantiations of hnsw_search_kernel (hnsw_device.hip.h) for ONE
// (metric, accept rule, row format) triple; build.py compiles this file once per triple
// (-DHNSW_V_METRIC= -DHNSW_V_SEMF= -DHNSW_V_FULL=), in parallel.  Each object exports a launcher and an
// occupancy query over the (NCH, NSLOT) grid; hnsw_capi.hip picks the object by the triple.
// The triple is a compile-time parameter of the kernel because the register allocation of a kernel is
// that of its worst path: with both accept rules and both row shapes in one kernel the d = 128 Ohnsw
// variant needed 88 VGPRs (5 waves/SIMD); on its own it needs 64 (8 waves/SIMD).
#include "hnsw_internal.h"

#ifndef HNSW_V_METRIC
#error "compile with -DHNSW_V_METRIC=0|1 -DHNSW_V_SEMF=0|1 -DHNSW_V_FULL=0|1|2|3 (rows: ragged fp32, full fp32, bytes, split fp32)"
#endif

using hnsw_dev::IndexView;
using hnsw_dev::SearchArgs;

namespace {

constexpr int M_ = HNSW_V_METRIC, S_ = HNSW_V_SEMF;
constexpr int F_ = HNSW_V_FULL;
// 4-row batches in flight per wave: a byte row is a quarter of the registers
#ifndef HNSW_RB_BYTES_NCH2
#define HNSW_RB_BYTES_NCH2 4
#endif
constexpr int RB1 = 8, RB2 = F_ == 2 ? HNSW_RB_BYTES_NCH2 : HNSW_RB_NCH2, RB4 = F_ == 2 ? 4 : 2, RB8 = F_ == 2 ? 2 : 1, RB16 = 1;

template <int NCH, int RB, int NSLOT>
hipError_t launch_one(const IndexView &iv, const SearchArgs &a, hipStream_t st) {
    const size_t lds = hnsw_dev::search_lds_words(a.vt_bits, a.blk_bits) * sizeof(uint32_t) + (size_t)a.lds_pad;
    // Visited as bitmap blocks (a.blk_bits > 0; W in three or more registers only) is a kernel of its own: the tag-cache kernels
    // keep their registers
    if constexpr (NSLOT >= 3) {
        if (a.blk_bits > 0 && iv.lcode0 && iv.lcode) {
            hipLaunchKernelGGL((hnsw_dev::hnsw_search_kernel<NCH, RB, NSLOT, M_, S_, F_, 1>), dim3((unsigned)a.nq), dim3(64), lds, st, iv, a);
            return hipGetLastError();
        }
    }
    SearchArgs b = a;
    b.blk_bits = 0;
    hipLaunchKernelGGL((hnsw_dev::hnsw_search_kernel<NCH, RB, NSLOT, M_, S_, F_>), dim3((unsigned)a.nq),
                       dim3(64), lds, st, iv, b);
    return hipGetLastError();
}
template <int NCH, int RB, int NSLOT>
int occupancy_one(size_t lds, int blk) {
    int nb = 0;
    hipError_t e;
    if constexpr (NSLOT >= 3) {
        e = blk ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, hnsw_dev::hnsw_search_kernel<NCH, RB, NSLOT, M_, S_, F_, 1>, 64, lds)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, hnsw_dev::hnsw_search_kernel<NCH, RB, NSLOT, M_, S_, F_>, 64, lds);
    } else {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, hnsw_dev::hnsw_search_kernel<NCH, RB, NSLOT, M_, S_, F_>, 64, lds);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); nb = 0; }
    return nb;
}

template <int NCH, int RB>
hipError_t launch_slot(int nslot, const IndexView &iv, const SearchArgs &a, hipStream_t st) {
    switch (nslot) {
    case 1: return launch_one<NCH, RB, 1>(iv, a, st);
    case 2: return launch_one<NCH, RB, 2>(iv, a, st);
    case 4: return launch_one<NCH, RB, 4>(iv, a, st);
    case 8: return launch_one<NCH, RB, 8>(iv, a, st);
    // three and six registers: rows of 65..256 dimensions only (pick_nslot_knn never asks for them elsewhere)
    case 3: if constexpr (NCH == 2 || NCH == 4) return launch_one<NCH, RB, 3>(iv, a, st); else return hipErrorInvalidValue;
    case 6: if constexpr (NCH == 2 || NCH == 4) return launch_one<NCH, RB, 6>(iv, a, st); else return hipErrorInvalidValue;
    default: return launch_one<NCH, RB, 16>(iv, a, st);
    }
}
template <int NCH, int RB>
int occupancy_slot(int nslot, size_t lds, int blk) {
    switch (nslot) {
    case 1: return occupancy_one<NCH, RB, 1>(lds, blk);
    case 2: return occupancy_one<NCH, RB, 2>(lds, blk);
    case 4: return occupancy_one<NCH, RB, 4>(lds, blk);
    case 8: return occupancy_one<NCH, RB, 8>(lds, blk);
    case 3: if constexpr (NCH == 2 || NCH == 4) return occupancy_one<NCH, RB, 3>(lds, blk); else return 0;
    case 6: if constexpr (NCH == 2 || NCH == 4) return occupancy_one<NCH, RB, 6>(lds, blk); else return 0;
    default: return occupancy_one<NCH, RB, 16>(lds, blk);
    }
}

} // namespace

#define HNSW_V_CAT2(a, b, c, d) a##b##_##c##_##d
#define HNSW_V_CAT(a, b, c, d) HNSW_V_CAT2(a, b, c, d)

namespace hnsw_host {

hipError_t HNSW_V_CAT(search_launch_, HNSW_V_METRIC, HNSW_V_SEMF, HNSW_V_FULL)(int nch, int nslot, const IndexView &iv, const SearchArgs &a, hipStream_t st) {
    switch (nch) {
    case 1: return launch_slot<1, RB1>(nslot, iv, a, st);
    case 2: return launch_slot<2, RB2>(nslot, iv, a, st);
    case 4: return launch_slot<4, RB4>(nslot, iv, a, st);
    case 8: return launch_slot<8, RB8>(nslot, iv, a, st);
    default: return launch_slot<16, RB16>(nslot, iv, a, st);
    }
}
int HNSW_V_CAT(search_occupancy_, HNSW_V_METRIC, HNSW_V_SEMF, HNSW_V_FULL)(int nch, int nslot, size_t lds, int blk) {
    switch (nch) {
    case 1: return occupancy_slot<1, RB1>(nslot, lds, blk);
    case 2: return occupancy_slot<2, RB2>(nslot, lds, blk);
    case 4: return occupancy_slot<4, RB4>(nslot, lds, blk);
    case 8: return occupancy_slot<8, RB8>(nslot, lds, blk);
    default: return occupancy_slot<16, RB16>(nslot, lds, blk);
    }
}

} // namespace hnsw_host
