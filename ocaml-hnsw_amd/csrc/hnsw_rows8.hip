// hnsw_rows8.hip -- byte rows: a lossless copy of the vectors for data whose every value is an integer in 0..255.
// SIFT descriptors are such data (the hdf5 files the reference's benchmark reads, benchmark/dataset.ml, hold them as
// float32).  A row of d such floats is d bytes of information in 4d bytes of memory, and the search is bound by the
// bytes it gathers: the knn kernel's ROWS = 2 variants (hnsw_device.hip.h: hop_round) read the byte row, convert each
// byte back to the float it came from and run the unchanged arithmetic on it -- same distances bit for bit, a quarter
// of the traffic per evaluation.  The fp32 rows stay: the builder, the layer operators and hnsw_distance_batch use
// them, and option "byte_rows" = 0 sends the knn kernel back to them.
#include "hnsw_internal.h"

using namespace hnsw_host;

namespace {

// flag[0] is cleared when a value is not an integer in 0..255.  -0.0 counts as 0: in (x - q)^2 and in x * q summed from
// +0 under round-to-nearest the sign of a zero x never reaches the result (a sum that starts at +0 cannot become -0).
__global__ void __launch_bounds__(256)
rows_are_bytes_kernel(const float *X, int64_t stride, int64_t n, int32_t d, int32_t *flag) {
    const int64_t total = n * (int64_t)d;
    bool ok = true;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / d;
        const float v = X[row * stride + (e - row * d)];
        ok = ok && v >= 0.0f && v <= 255.0f && v == truncf(v);
    }
    if (!ok) flag[0] = 0;
}

// one thread per dword of the byte rows: bytes 4c..4c+3 of a row are dims 4c..4c+3 (0 beyond d)
__global__ void __launch_bounds__(256)
pack_byte_rows_kernel(const float *X, int64_t stride, int64_t n, int32_t d, uint32_t *X8, int32_t words_per_row) {
    const int64_t total = n * (int64_t)words_per_row;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = w / words_per_row;
        const int c = (int)(w - row * words_per_row);
        uint32_t u = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 4 * c + j;
            if (e < d) u |= (uint32_t)X[row * stride + e] << (8 * j);
        }
        X8[w] = u;
    }
}

} // namespace

namespace hnsw_host {

int make_byte_rows(::hnsw_index *idx) {
    if (!idx || !idx->dX || idx->iv.n <= 0) return HNSW_OK;
    if (!env_int("HNSW_BYTE_ROWS", 1)) return HNSW_OK;
    HIP_TRY(hipSetDevice(idx->device));
    const int64_t n = idx->iv.n;
    const int32_t d = idx->iv.d;
    const int32_t row_bytes = 64 * pick_nch(idx->iv.nchunks);     // the lane grid of the kernel: 16 lanes x NCH dwords
    int32_t *dflag = nullptr;
    HIP_TRY(hipMalloc((void **)&dflag, 16));
    const int32_t one = 1;
    int32_t ok = 0;
    hipError_t e = hipMemcpy(dflag, &one, 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const int blocks = (int)std::min<int64_t>(65536, (n * (int64_t)d + 255) / 256);
        hipLaunchKernelGGL(rows_are_bytes_kernel, dim3((unsigned)std::max(1, blocks)), dim3(256), 0, 0,
                           (const float *)idx->dX, idx->iv.stride, n, d, dflag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&ok, dflag, 4, hipMemcpyDeviceToHost);
    (void)hipFree(dflag);
    if (e != hipSuccess) return fail(HNSW_ERR_HIP, "byte-row check failed: %s", hipGetErrorString(e));
    if (env_int("HNSW_DEBUG_ROWS", 0)) fprintf(stderr, "hnsw: byte-row check n=%lld d=%d stride=%lld -> %d\n", (long long)n, d, (long long)idx->iv.stride, ok);
    if (!ok) return HNSW_OK;
    const size_t bytes = (size_t)n * (size_t)row_bytes;
    if (hipMalloc(&idx->dX8, bytes) != hipSuccess) {       // no room for the copy: not an error, the fp32 rows serve
        (void)hipGetLastError();
        idx->dX8 = nullptr;
        return HNSW_OK;
    }
    const int32_t words = row_bytes / 4;
    const int blocks = (int)std::min<int64_t>(65536, (n * (int64_t)words + 255) / 256);
    hipLaunchKernelGGL(pack_byte_rows_kernel, dim3((unsigned)std::max(1, blocks)), dim3(256), 0, 0,
                       (const float *)idx->dX, idx->iv.stride, n, d, (uint32_t *)idx->dX8, words);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(idx->dX8); idx->dX8 = nullptr; return fail(HNSW_ERR_HIP, "byte-row packing failed: %s", hipGetErrorString(e)); }
    idx->iv.X8 = (const uint8_t *)idx->dX8;
    idx->iv.stride8 = row_bytes;
    idx->info.device_bytes += (int64_t)bytes;
    return HNSW_OK;
}

} // namespace hnsw_host
