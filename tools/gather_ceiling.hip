// gather_ceiling.hip -- what the chip delivers for the search kernel's access shape with nothing else in the way: every
// wave requests random 128-byte rows, four rows per wave-instruction (16 lanes x 4 bytes, twice: bytes 0..63 and 64..127
// of the row, exactly the byte-row kernel's two global_load_dword per batch), INFLIGHT batches in flight before the first
// is consumed, 8 waves per SIMD on every CU.  Rows are chosen by a per-group hash, so consecutive requests of a wave are
// independent (the search's are not: its next row ids come out of the previous rows' distances).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_ceiling tools/gather_ceiling.hip && /tmp/gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int INFLIGHT>
__global__ void __launch_bounds__(64) gather(const uint8_t *table, uint32_t row_mask, int iters, uint32_t *out) {
    const int lane = threadIdx.x, r = lane >> 4, l16 = lane & 15;
    uint32_t state = mix(blockIdx.x * 4u + r + 12345u);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t a[INFLIGHT], b[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            state = mix(state + 0x9e3779b9U);
            const uint8_t *row = table + (uint64_t)(state & row_mask) * 128u + 4u * l16;
            a[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(row));
            b[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(row + 64));
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) acc += a[j] ^ b[j];
    }
    if (acc == 0x12345678u) out[0] = acc;     // keeps the loads alive
}

template <int INFLIGHT>
__global__ void __launch_bounds__(64) gather_plain(const uint8_t *table, uint32_t row_mask, int iters, uint32_t *out) {
    const int lane = threadIdx.x, r = lane >> 4, l16 = lane & 15;
    uint32_t state = mix(blockIdx.x * 4u + r + 12345u);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t a[INFLIGHT], b[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            state = mix(state + 0x9e3779b9U);
            const uint8_t *row = table + (uint64_t)(state & row_mask) * 128u + 4u * l16;
            a[j] = *reinterpret_cast<const uint32_t *>(row);
            b[j] = *reinterpret_cast<const uint32_t *>(row + 64);
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) acc += a[j] ^ b[j];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename K>
static int run(const char *name, K kernel, const uint8_t *d_table, uint32_t rows, int inflight, uint32_t *d_out) {
    const int waves = 8192, iters = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), 0, 0, d_table, rows - 1, iters, d_out);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)waves * iters * inflight * 4 * 128;
    printf("  %-28s %2d batches (x 4 rows) in flight per wave: %.3f ms, %.2f TB/s of rows\n", name, inflight, best, bytes / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    uint32_t *d_out;
    CHECK(hipMalloc(&d_out, 4));
    for (uint32_t mb : {128u, 1024u}) {
        const uint32_t rows = (uint32_t)(((uint64_t)mb << 20) >> 7);     // a power of two: the row index is masked
        if (rows == 0 || (rows & (rows - 1)) != 0) { fprintf(stderr, "bad table size\n"); return 1; }
        uint8_t *d_table;
        CHECK(hipMalloc(&d_table, (size_t)rows * 128));
        CHECK(hipMemset(d_table, 1, (size_t)rows * 128));
        printf("random 128-byte rows of a %u MB table, 8192 waves (8 per SIMD):\n", mb); fflush(stdout);
        if (run("plain loads", gather_plain<1>, d_table, rows, 1, d_out)) return 1;
        if (run("plain loads", gather_plain<2>, d_table, rows, 2, d_out)) return 1;
        if (run("plain loads", gather_plain<4>, d_table, rows, 4, d_out)) return 1;
        if (run("plain loads", gather_plain<8>, d_table, rows, 8, d_out)) return 1;
        if (run("nontemporal loads", gather<4>, d_table, rows, 4, d_out)) return 1;
        CHECK(hipFree(d_table));
    }
    return 0;
}
