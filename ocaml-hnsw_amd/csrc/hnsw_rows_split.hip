// hnsw_rows_split.hip -- split rows: a second copy of the vectors for rows that end a little past a 128-byte line.
//
// The search is bound by the 128-byte line requests it makes (profiles/r04_gather_ceiling.txt: the chip serves about
// 50-60 G random lines per second whatever the row shape).  A float32 row of d = 100 is 400 bytes = three lines + 16
// bytes, so every evaluation costs FOUR requests and the fourth fetches 16 useful bytes (C3: 1.29x the algorithmic
// traffic).  When 4d mod 128 is in 1..32 the index keeps
//   * Xm    [n][128 * k] : dims 0 .. 32k-1 of every row, 128-byte aligned: exactly k lines per evaluation, and
//   * tail0 [n][S0][4T]  : the remaining T = 1 or 2 float4 chunks of node nbr0[c][j], stored at (c, j) -- beside the
//                          neighbour in the layer-0 adjacency of the node that is being expanded.  A hop evaluates the
//                          unvisited neighbours of ONE node c, so all their tails lie in S0 * 16T contiguous bytes
//                          (d = 100, M = 32: 1 KiB = 8 lines per hop instead of one extra line per evaluation).
// The knn kernel's ROWS = 3 variants (hnsw_device.hip.h: hop_round) read these on layer 0; lane grid, operands and order
// of the arithmetic are those of the plain float32 row, so distances are bit-identical.  The float32 rows stay: the
// descent, the builder, the layer operators and hnsw_distance_batch use them, and option "split_rows" = 0 sends the knn
// kernel back to them.  Semantics: lib/ohnsw.ml:570-573 (iterate the adjacency of c, distance to each unvisited neighbour).
#include "hnsw_internal.h"

using namespace hnsw_host;

namespace {

// one thread per float4 of the main rows
__global__ void __launch_bounds__(256)
pack_main_rows_kernel(const float4 *X, int64_t stride4, int64_t n, int32_t main_chunks, float4 *Xm) {
    const int64_t total = n * (int64_t)main_chunks;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / main_chunks;
        const int c = (int)(e - row * main_chunks);
        Xm[e] = X[row * stride4 + c];
    }
}

// one thread per (node, adjacency slot, tail chunk): the tail chunk of that neighbour (zeros for a hole); rows of X are
// zero padded to a multiple of 16 floats, so dims past d read as 0
__global__ void __launch_bounds__(256)
pack_tails_kernel(const float4 *X, int64_t stride4, const int32_t *nbr0, int64_t slots, int32_t main_chunks,
                  int32_t tail_chunks, float4 *tail0) {
    const int64_t total = slots * tail_chunks;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slot = e / tail_chunks;
        const int t = (int)(e - slot * tail_chunks);
        const int32_t v = nbr0[slot];
        tail0[e] = v >= 0 ? X[(int64_t)v * stride4 + main_chunks + t] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

} // namespace

namespace hnsw_host {

int make_split_rows(::hnsw_index *idx) {
    if (!idx || !idx->dX || !idx->dNbr0 || idx->iv.n <= 0) return HNSW_OK;
    if (idx->dX8) return HNSW_OK;                        // byte rows are a single line per vector already
    if (!env_int("HNSW_SPLIT_ROWS", 1)) return HNSW_OK;
    const int nchunks = idx->iv.nchunks;
    const int T = nchunks % 8;                           // float4 chunks past the last whole 128-byte line
    if ((T != 1 && T != 2) || nchunks < 9) return HNSW_OK;
    HIP_TRY(hipSetDevice(idx->device));
    const int64_t n = idx->iv.n;
    const int main_chunks = nchunks - T, S0 = idx->iv.S0;
    const size_t main_bytes = (size_t)n * main_chunks * 16, tail_bytes = (size_t)n * S0 * T * 16;
    void *dXm = nullptr, *dTail = nullptr;
    if (hipMalloc(&dXm, main_bytes) != hipSuccess || hipMalloc(&dTail, tail_bytes) != hipSuccess) {
        (void)hipGetLastError();                         // no room for the copy: not an error, the plain rows serve
        if (dXm) (void)hipFree(dXm);
        return HNSW_OK;
    }
    const int64_t stride4 = idx->iv.stride / 4;
    int blocks = (int)std::min<int64_t>(65536, (n * (int64_t)main_chunks + 255) / 256);
    hipLaunchKernelGGL(pack_main_rows_kernel, dim3((unsigned)std::max(1, blocks)), dim3(256), 0, 0,
                       (const float4 *)idx->dX, stride4, n, main_chunks, (float4 *)dXm);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) {
        blocks = (int)std::min<int64_t>(65536, (n * (int64_t)S0 * T + 255) / 256);
        hipLaunchKernelGGL(pack_tails_kernel, dim3((unsigned)std::max(1, blocks)), dim3(256), 0, 0,
                           (const float4 *)idx->dX, stride4, (const int32_t *)idx->dNbr0, n * (int64_t)S0, main_chunks, T, (float4 *)dTail);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(dXm); (void)hipFree(dTail); return fail(HNSW_ERR_HIP, "split-row packing failed: %s", hipGetErrorString(e)); }
    idx->dXm = dXm; idx->dTail0 = dTail;
    idx->iv.Xm = (const float *)dXm; idx->iv.tail0 = (const float *)dTail;
    idx->iv.stride_m = main_chunks * 16; idx->iv.main_chunks = main_chunks; idx->iv.tail_chunks = T;
    idx->info.device_bytes += (int64_t)(main_bytes + tail_bytes);
    return HNSW_OK;
}

} // namespace hnsw_host
