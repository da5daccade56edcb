// mem_latency.hip -- dependent-load latency of the chip the search kernel's hop is made of: one wave chases a random
// cycle through buffers of growing size (L2 -> Infinity Cache -> HBM), each load's address taken from the previous
// load's data, 128-byte stride (one line per step, like a byte row or an adjacency row of the C2 index).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mem_latency tools/mem_latency.hip && /tmp/mem_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void chase(const uint32_t *buf, uint32_t start, int steps, uint32_t *out, long long *cycles) {
    uint32_t i = start;
    const long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) i = buf[(size_t)i * 32];      // 32 dwords = 128 B per node
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = i; cycles[0] = t1 - t0; }
}

int main() {
    const int steps = 20000;
    uint32_t *d_out; long long *d_cyc;
    CHECK(hipMalloc(&d_out, 4)); CHECK(hipMalloc(&d_cyc, 8));
    int rate_khz = 0;
    CHECK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    printf("# wall clock %d kHz; one wave, %d dependent 4-byte loads, nodes 128 B apart, random cycle\n", rate_khz, steps);
    for (size_t mb : {1, 2, 8, 32, 64, 128, 256, 512, 1024, 4096}) {
        const size_t nodes = mb * (1u << 20) / 128;
        std::vector<uint32_t> perm(nodes);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937_64 rng(7);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint32_t> host(nodes * 32, 0u);
        for (size_t k = 0; k < nodes; ++k) host[(size_t)perm[k] * 32] = perm[(k + 1) % nodes];   // one cycle through all nodes
        uint32_t *d_buf;
        CHECK(hipMalloc(&d_buf, host.size() * 4));
        CHECK(hipMemcpy(d_buf, host.data(), host.size() * 4, hipMemcpyHostToDevice));
        double best = 1e30;
        double best_ev = 1e30;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d_buf, perm[(size_t)rep * 977 % nodes], steps, d_out, d_cyc);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            long long c = 0;
            float ms = 0;
            CHECK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, (double)c / steps / (rate_khz * 1e-6));      // ns per load
            best_ev = std::min(best_ev, (double)ms * 1e6 / steps);
        }
        printf("%6zu MB: %7.1f ns per dependent load (device clock), %7.1f ns (HIP events around the launch)\n", mb, best, best_ev);
        CHECK(hipFree(d_buf));
    }
    return 0;
}
