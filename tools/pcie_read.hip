// pcie_read.hip -- what the host protocol's pre-pass is bound by: the rate at which the device reads a page-locked host
// matrix in place (the 5.12 MB of a 10 k x 128 float32 query batch, and a 64 MB one), by load width and by how the reads
// are spread over waves, beside the DMA engines' rate for the same bytes (hipMemcpyAsync host -> device).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pcie_read tools/pcie_read.hip && /tmp/pcie_read
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every wave streams consecutive 64 x W bytes blocks, grid-strided; the sum keeps the loads alive
template <typename T>
__global__ void __launch_bounds__(64) stream_read(const T *src, size_t n_items, float *out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n_items; i += (size_t)gridDim.x * 64) {
        const T v = __builtin_nontemporal_load(src + i);
        const float *f = reinterpret_cast<const float *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc += f[k];
    }
    if (acc == 12345.678f) out[0] = acc;
}

// the pre-pass's pattern: one wave per 512-byte row (a query), 8 bytes per lane, and the copy into device memory beside it
__global__ void __launch_bounds__(64) row_read(const float2 *src, size_t rows, float2 *dst) {
    const size_t q = blockIdx.x;
    if (q >= rows) return;
    dst[q * 64 + threadIdx.x] = src[q * 64 + threadIdx.x];
}

// when does each row arrive?  (the wave's clock when its 512 bytes are in registers, by block index)
__global__ void __launch_bounds__(64) row_read_timed(const float2 *src, size_t rows, float2 *dst, long long *when) {
    const size_t q = blockIdx.x;
    if (q >= rows) return;
    const float2 v = src[q * 64 + threadIdx.x];
    dst[q * 64 + threadIdx.x] = v;
    if (threadIdx.x == 0) when[q] = wall_clock64() + (v.x == 12345.678f ? 1 : 0);
}

template <typename F>
static float median_ms(F &&launch, hipStream_t st, int reps = 15) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    std::vector<float> ts;
    for (int r = 0; r < reps + 3; ++r) {
        (void)hipEventRecord(a, st);
        launch();
        (void)hipEventRecord(b, st);
        (void)hipEventSynchronize(b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, a, b);
        if (r >= 3) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ts[ts.size() / 2];
}

int main() {
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *d_out;
    CHECK(hipMalloc(&d_out, 4));
    for (size_t bytes : {(size_t)5120000, (size_t)64 << 20}) {
        void *h = nullptr, *hd = nullptr, *d = nullptr;
        CHECK(hipHostMalloc(&h, bytes, hipHostMallocPortable | hipHostMallocMapped));
        CHECK(hipHostGetDevicePointer(&hd, h, 0));
        CHECK(hipMalloc(&d, bytes));
        for (size_t i = 0; i < bytes / 4; ++i) ((float *)h)[i] = (float)(i & 255);
        printf("# %zu bytes of page-locked host memory\n", bytes);
        const float tc = median_ms([&] { (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st); }, st);
        printf("hipMemcpyAsync host -> device                      %8.4f ms  %6.1f GB/s\n", tc, bytes / tc * 1e-6);
        for (int grid : {256, 1024, 4096, 16384}) {
            const float t4 = median_ms([&] { stream_read<float><<<grid, 64, 0, st>>>((const float *)hd, bytes / 4, d_out); }, st);
            const float t8 = median_ms([&] { stream_read<v2f><<<grid, 64, 0, st>>>((const v2f *)hd, bytes / 8, d_out); }, st);
            const float t16 = median_ms([&] { stream_read<v4f><<<grid, 64, 0, st>>>((const v4f *)hd, bytes / 16, d_out); }, st);
            printf("kernel reads in place, %5d waves: 4 B / lane %8.4f ms %6.1f GB/s | 8 B %8.4f ms %6.1f GB/s | 16 B %8.4f ms %6.1f GB/s\n",
                   grid, t4, bytes / t4 * 1e-6, t8, bytes / t8 * 1e-6, t16, bytes / t16 * 1e-6);
        }
        const size_t rows = bytes / 512;
        const float tr = median_ms([&] { row_read<<<(unsigned)rows, 64, 0, st>>>((const float2 *)hd, rows, (float2 *)d); }, st);
        printf("one wave per 512-byte row, copied to device memory  %8.4f ms  %6.1f GB/s\n", tr, bytes / tr * 1e-6);
        {   // arrival order: quartiles of the block index against the time of arrival
            long long *d_when;
            CHECK(hipMalloc(&d_when, rows * 8));
            int khz = 0;
            CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
            for (int rep = 0; rep < 3; ++rep) {
                row_read_timed<<<(unsigned)rows, 64, 0, st>>>((const float2 *)hd, rows, (float2 *)d, d_when);
                CHECK(hipStreamSynchronize(st));
            }
            std::vector<long long> when(rows);
            CHECK(hipMemcpy(when.data(), d_when, rows * 8, hipMemcpyDeviceToHost));
            const long long t0 = *std::min_element(when.begin(), when.end());
            printf("arrival of the rows by block index (us after the first; last row of each eighth of the grid, and the latest row in it):");
            for (int e = 0; e < 8; ++e) {
                const size_t a = rows * e / 8, b = rows * (e + 1) / 8;
                const long long mx = *std::max_element(when.begin() + a, when.begin() + b);
                printf("  [%d] %.1f / %.1f", e, (when[b - 1] - t0) * 1e3 / khz, (mx - t0) * 1e3 / khz);
            }
            printf("\n");
            CHECK(hipFree(d_when));
        }
        CHECK(hipFree(d));
        CHECK(hipHostFree(h));
    }
    return 0;
}
