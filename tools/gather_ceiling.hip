// gather_ceiling.hip -- what the chip delivers for the search kernels' access shapes with nothing else in the way.
//
// Every wave requests random whole rows the way hop_round does: a row is read by a 16-lane group, four rows per
// wave-instruction, INFLIGHT batches (x 4 rows) requested before the first is consumed.  Rows are chosen by a per-group
// hash, so consecutive requests of a wave are independent (the search's are not: its next row ids come out of the previous
// rows' distances) -- this is the memory system's ceiling for the shape, not a model of the search.
//
//   byte rows   (ROWS = 2):  128-byte rows, two global_load_dword per batch (bytes 0..63 and 64..127 of the row)
//   float rows  (ROWS = 0/1): row_bytes in {384, 400, 512}, stride in {384, 448, 512}: global_load_dwordx4 per 16-lane
//                             chunk l16, l16 + 16 (lanes past the row end re-read chunk 0, as the ragged kernel does)
//
// Round 4: the shapes C5 and C3 actually read -- 384-byte rows (d = 96), 400-byte rows in a 448-byte
// stride (d = 100), 384 + 16-byte split rows, 512-byte rows -- from 1 / 4 / 6 GB tables, at 4 / 5 / 8 waves per SIMD
// (residency held by an LDS request, as balanced_lds_pad does) and 1..8 batches in flight.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_ceiling tools/gather_ceiling.hip && /tmp/gather_ceiling [quick]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// 128-byte rows, the byte-row kernel's two dword loads per batch
template <int INFLIGHT>
__global__ void __launch_bounds__(64) gather_bytes(const uint8_t *table, uint32_t rows, int iters, uint32_t *out) {
    extern __shared__ uint32_t pad_lds[];
    const int lane = threadIdx.x, r = lane >> 4, l16 = lane & 15;
    uint32_t state = mix(blockIdx.x * 4u + r + 12345u);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t a[INFLIGHT], b[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            state = mix(state + 0x9e3779b9U);
            const uint32_t id = (uint32_t)(((uint64_t)state * rows) >> 32);
            const uint8_t *row = table + (uint64_t)id * 128u + 4u * l16;
            a[j] = *reinterpret_cast<const uint32_t *>(row);
            b[j] = *reinterpret_cast<const uint32_t *>(row + 64);
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) acc += a[j] ^ b[j];
    }
    if (acc == 0x12345678u) out[0] = acc;     // keeps the loads alive
}

// float rows of `chunks` float4 (NCH = 2: chunk l16 and chunk 16 + l16; a lane whose second chunk is past the row end
// re-reads chunk 0).  TAIL = 1: the last chunk (chunk index `chunks - 1`) is NOT read from the row -- the "C3 without its
// fourth line" layout, where dims 96..99 travel with the adjacency row.
template <int INFLIGHT, int TAIL>
__global__ void __launch_bounds__(64) gather_f32(const uint8_t *table, uint32_t rows, uint32_t stride, int chunks, int iters, uint32_t *out) {
    extern __shared__ uint32_t pad_lds[];
    const int lane = threadIdx.x, r = lane >> 4, l16 = lane & 15;
    uint32_t state = mix(blockIdx.x * 4u + r + 12345u);
    const int nread = chunks - TAIL;
    const uint32_t c1 = (16 + l16) < nread ? (uint32_t)(16 + l16) * 16u : 0u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float4 a[INFLIGHT], b[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            state = mix(state + 0x9e3779b9U);
            const uint32_t id = (uint32_t)(((uint64_t)state * rows) >> 32);
            const uint8_t *row = table + (uint64_t)id * stride;
            a[j] = *reinterpret_cast<const float4 *>(row + 16u * l16);
            b[j] = *reinterpret_cast<const float4 *>(row + c1);
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) acc += a[j].x + b[j].y + a[j].z + b[j].w;
    }
    if (acc == 1234.5678f) out[0] = 1;
}

struct Shape { const char *name; int kind; uint32_t stride; int chunks; int tail; uint32_t useful; };

template <typename K, typename... A>
static int time_kernel(K kernel, int waves, size_t lds, float *best_ms, A... args) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), lds, 0, args...);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        CHECK(hipGetLastError());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    *best_ms = best;
    return 0;
}

static int run(const Shape &s, const uint8_t *d_table, uint64_t table_bytes, int wps, int inflight, uint32_t *d_out) {
    const uint32_t rows = (uint32_t)(table_bytes / s.stride);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int waves = wps * 4 * cus;                     // exactly what is resident at once
    // hold the residency with an LDS request: gfx950 hands LDS out in 1280-byte granules, 128 per CU
    const size_t lds = wps >= 8 ? 0 : (size_t)(128 / (4 * wps)) * 1280;
    const int iters = (int)std::max<double>(4.0, 24e9 / ((double)waves * inflight * 4 * s.useful));   // ~24 GB of rows per launch
    float ms = 0;
    int rc = 0;
#define LAUNCH_B(I) rc = time_kernel(gather_bytes<I>, waves, lds, &ms, d_table, rows, iters, d_out)
#define LAUNCH_F(I) rc = s.tail ? time_kernel(gather_f32<I, 1>, waves, lds, &ms, d_table, rows, s.stride, s.chunks, iters, d_out) \
                                : time_kernel(gather_f32<I, 0>, waves, lds, &ms, d_table, rows, s.stride, s.chunks, iters, d_out)
    if (s.kind == 2) { switch (inflight) { case 1: LAUNCH_B(1); break; case 2: LAUNCH_B(2); break; case 4: LAUNCH_B(4); break; default: LAUNCH_B(8); } }
    else { switch (inflight) { case 1: LAUNCH_F(1); break; case 2: LAUNCH_F(2); break; case 4: LAUNCH_F(4); break; default: LAUNCH_F(8); } }
    if (rc) return rc;
    const double nrows = (double)waves * iters * inflight * 4;
    // 128-byte lines a row costs on average: rows start at multiples of the stride
    double lines = 0;
    {
        const uint32_t rb = (uint32_t)(s.chunks - s.tail) * 16u;
        for (uint32_t o = 0; o < 128; ++o) { const uint32_t off = (uint32_t)(((uint64_t)o * s.stride) % 128); lines += (off + rb - 1) / 128 + 1; }
        lines /= 128;
    }
    printf("  %-34s %4.1f GB  %d waves/SIMD  %d x4 rows in flight: %8.3f ms  %5.2f TB/s useful  %5.2f TB/s of lines  %5.1f G lines/s\n",
           s.name, table_bytes / 1e9, wps, inflight, ms, nrows * s.useful / (ms * 1e-3) / 1e12,
           nrows * lines * 128 / (ms * 1e-3) / 1e12, nrows * lines / (ms * 1e-3) / 1e9);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    uint32_t *d_out;
    CHECK(hipMalloc(&d_out, 4));
    const Shape shapes[] = {
        {"128-B byte rows (C2 byte rows)", 2, 128, 8, 0, 128},
        {"512-B rows (C2 float32 rows)", 0, 512, 32, 0, 512},
        {"384-B rows (C5, d = 96)", 0, 384, 24, 0, 384},
        {"400-B rows, stride 448 (C3, d = 100)", 0, 448, 25, 0, 400},
        {"384 of 400 B, stride 384 (C3 split)", 0, 384, 25, 1, 384},
    };
    const uint64_t GB = 1ull << 30;
    std::vector<uint64_t> sizes = quick ? std::vector<uint64_t>{GB} : std::vector<uint64_t>{GB / 8, GB, 4 * GB, 6 * GB};
    for (uint64_t bytes : sizes) {
        uint8_t *d_table;
        CHECK(hipMalloc(&d_table, bytes + 4096));
        CHECK(hipMemset(d_table, 1, bytes + 4096));
        CHECK(hipDeviceSynchronize());
        printf("table of %.3f GB, random whole rows, independent requests:\n", bytes / 1e9); fflush(stdout);
        for (const Shape &s : shapes) {
            if (s.kind == 2 && bytes > GB) continue;             // byte rows: the C2 table is 128 MB
            for (int wps : {4, 5, 8}) {
                if (s.kind == 2 && wps != 8) continue;
                for (int inflight : {1, 2, 4, 8}) {
                    if (quick && inflight != 4) continue;
                    if (run(s, d_table, bytes, wps, inflight, d_out)) return 1;
                }
            }
        }
        CHECK(hipFree(d_table));
    }
    return 0;
}
