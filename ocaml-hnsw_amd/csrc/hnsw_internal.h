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
inline int64_t padded_stride(int d) { return ((int64_t)d + 15) / 16 * 16; } // floats: rows are multiples of 64 B

// copies [n][row_stride] host rows into a fresh zero-padded device table
int upload_vectors(const float *vectors, int64_t n, int d, int64_t row_stride, void **dX, size_t *bytes);

} // namespace hnsw_host

struct hnsw_index {
    int device = -1;
    hnsw_dev::IndexView iv{};
    hnsw_index_info info{};
    void *dX = nullptr, *dNbr0 = nullptr, *dNbrU = nullptr, *dOff = nullptr, *dLvl = nullptr;
    int64_t rowsU = 0;
    hnsw_host::DevBuf sQ, sIds, sDist, sNd, sNh, sSt; // scratch for the host-buffer entry points
    int vt_bits_override = 0;
};
