// issue_latency.hip -- what one wave alone pays for the instruction patterns a layer-0 hop is made of (gfx950): each
// pattern is a dependent chain repeated REP times inside one asm block, timed with s_memtime (shader cycles), one
// wave on an idle chip, (second column) 4 waves per SIMD on one CU running the same chain, and (third) 8 waves per SIMD on
// every CU (512 workgroups of 1024 threads: two per CU) -- the second and third tell how many instructions of a kind a SIMD
// issues per cycle when it has waves to choose from.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_latency tools/issue_latency.hip && /tmp/issue_latency
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
constexpr int REP = 64, LOOPS = 64;

#define PATTERN_KERNEL(NAME, BODY, CLOB...)                                                                        \
    __global__ void NAME(uint32_t *out, long long *cycles) {                                                       \
        uint32_t v = threadIdx.x, w = threadIdx.x * 3u, s = 5, s2 = 1;                                             \
        uint64_t m = 0, vv = threadIdx.x;                                                                                            \
        __shared__ uint32_t lds[256];                                                                              \
        lds[threadIdx.x & 255] = threadIdx.x;                                                                      \
        uint32_t la = (threadIdx.x & 63) * 4;                                                                      \
        __syncthreads();                                                                                           \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                         \
        for (int l = 0; l < LOOPS; ++l)                                                                            \
            asm volatile(R64(BODY) : "+&v"(v), "+&v"(w), "+&s"(s), "+&s"(s2), "+&s"(m), "+&v"(vv) : "v"(la) : "vcc", "scc", "memory", ##CLOB); \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                         \
        if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = v + w + s + s2 + (uint32_t)m + (uint32_t)vv;                                 \
    }

// operands: %0 v, %1 w (vector), %2 s, %3 s2 (scalar), %4 m (scalar pair), %5 vv (vector pair), %6 la (lds address)
PATTERN_KERNEL(k_valu_dep, "v_add_u32_e32 %0, 1, %0\n\t")
PATTERN_KERNEL(k_valu_indep, "v_add_u32_e32 %0, 1, %0\n\tv_add_u32_e32 %1, 1, %1\n\t")
PATTERN_KERNEL(k_salu_dep, "s_add_u32 %2, %2, 1\n\t")
PATTERN_KERNEL(k_salu_indep, "s_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t")
PATTERN_KERNEL(k_valu_salu_mix, "v_add_u32_e32 %0, 1, %0\n\ts_add_u32 %2, %2, 1\n\t")
PATTERN_KERNEL(k_readlane_mov, "v_readlane_b32 %2, %0, 5\n\tv_mov_b32_e32 %0, %2\n\t")
PATTERN_KERNEL(k_readlane_salu_mov, "v_readlane_b32 %2, %0, 5\n\ts_add_u32 %2, %2, 1\n\tv_mov_b32_e32 %0, %2\n\t")
PATTERN_KERNEL(k_cmp_bcnt_mov, "v_cmp_gt_u32_e32 vcc, 7, %0\n\ts_bcnt1_i32_b64 %2, vcc\n\tv_mov_b32_e32 %0, %2\n\t")
PATTERN_KERNEL(k_cmp_sgpr_bcnt_mov, "v_cmp_gt_u32_e64 %4, 7, %0\n\ts_bcnt1_i32_b64 %2, %4\n\tv_mov_b32_e32 %0, %2\n\t")
PATTERN_KERNEL(k_ff1_readlane, "s_ff1_i32_b32 %3, %2\n\tv_readlane_b32 %2, %0, %3\n\t")
PATTERN_KERNEL(k_m0_writelane, "s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %3, m0\n\t", "m0")
PATTERN_KERNEL(k_exec_dpp, "s_lshl_b64 exec, -1, %2\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_mov_b64 exec, -1\n\t")
PATTERN_KERNEL(k_dpp_nop, "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
PATTERN_KERNEL(k_dpp_fill, "v_add_u32_e32 %1, 1, %1\n\tv_add_u32_e32 %1, 1, %1\n\tv_add_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
PATTERN_KERNEL(k_branch_taken, "s_branch 1f\n\ts_nop 0\n1:\n\t")
PATTERN_KERNEL(k_cbranch_not_taken, "s_cmp_eq_u32 %2, -1\n\ts_cbranch_scc1 1f\n1:\n\t")
PATTERN_KERNEL(k_cbranch_taken, "s_cmp_lg_u32 %2, -1\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:\n\t")
PATTERN_KERNEL(k_lds_roundtrip, "ds_write_b32 %6, %0\n\tds_read_b32 %0, %6\n\ts_waitcnt lgkmcnt(0)\n\t")
PATTERN_KERNEL(k_lds_read, "ds_read_b32 %0, %6\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32_e32 %0, 0, %0\n\t")
PATTERN_KERNEL(k_snop0, "s_nop 0\n\t")
PATTERN_KERNEL(k_snop2, "s_nop 2\n\t")
PATTERN_KERNEL(k_cmp_vcc_cndmask, "v_cmp_gt_u32_e32 vcc, 7, %0\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\t")
PATTERN_KERNEL(k_dot4, "v_dot4_u32_u8 %0, %1, %1, %0\n\t")
PATTERN_KERNEL(k_mad64, "v_mad_u64_u32 %5, vcc, %0, %1, %5\n\t")
PATTERN_KERNEL(k_max_cmp_branch, "s_max_u32 %3, %3, %2\n\ts_cmp_eq_u32 %3, -3\n\ts_cbranch_scc1 1f\n1:\n\t")

struct Pattern { const char *name; void (*fn)(uint32_t *, long long *); int n_instr; };

int main() {
    uint32_t *d_out; long long *d_cyc;
    CHECK(hipMalloc(&d_out, 4 * 1024 * 512)); CHECK(hipMalloc(&d_cyc, 8));
    const Pattern pats[] = {
        {"v_add dependent chain", k_valu_dep, 1}, {"2 independent v_add", k_valu_indep, 2},
        {"s_add dependent chain", k_salu_dep, 1}, {"2 independent s_add", k_salu_indep, 2},
        {"v_add ; s_add (independent)", k_valu_salu_mix, 2},
        {"v_readlane -> v_mov (VALU->SGPR->VALU)", k_readlane_mov, 2},
        {"v_readlane -> s_add -> v_mov", k_readlane_salu_mov, 3},
        {"v_cmp vcc -> s_bcnt1 -> v_mov", k_cmp_bcnt_mov, 3},
        {"v_cmp sgpr -> s_bcnt1 -> v_mov", k_cmp_sgpr_bcnt_mov, 3},
        {"s_ff1 -> v_readlane (lane from SGPR) -> s_ff1", k_ff1_readlane, 2},
        {"s_mov m0 -> v_writelane", k_m0_writelane, 2},
        {"s_lshl exec -> v_mov_dpp -> s_mov exec", k_exec_dpp, 3},
        {"s_nop 1 ; dependent DPP add", k_dpp_nop, 2},
        {"2 v_add ; dependent DPP add", k_dpp_fill, 3},
        {"s_branch taken (over one s_nop)", k_branch_taken, 1},
        {"s_cmp ; s_cbranch not taken", k_cbranch_not_taken, 2},
        {"s_cmp ; s_cbranch taken (over one s_nop)", k_cbranch_taken, 2},
        {"ds_write ; ds_read ; wait", k_lds_roundtrip, 3},
        {"ds_read ; wait ; use", k_lds_read, 3},
        {"s_nop 0", k_snop0, 1}, {"s_nop 2", k_snop2, 1},
        {"v_cmp vcc ; s_nop 1 ; v_cndmask", k_cmp_vcc_cndmask, 3},
        {"v_dot4_u32_u8 accumulate chain", k_dot4, 1},
        {"v_mad_u64_u32 chain", k_mad64, 1},
        {"s_max ; s_cmp ; s_cbranch not taken", k_max_cmp_branch, 3},
    };
    printf("# cycles per repetition of the pattern (s_memtime): one wave alone | 16 waves on the CU (4 per SIMD), each its own chain | 32 waves on every CU (8 per SIMD)\n");
    for (const Pattern &p : pats) {
        double c[3];
        float wall_ms = 1e30f;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int mode = 0; mode < 3; ++mode) {
            long long best = 1ll << 60;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(p.fn, dim3(mode == 2 ? 512 : 1), dim3(mode ? 1024 : 64), 0, 0, d_out, d_cyc);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipDeviceSynchronize());
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (mode == 2 && ms < wall_ms) wall_ms = ms;
                long long cyc = 0;
                CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
                if (cyc < best) best = cyc;
            }
            c[mode] = (double)best / (REP * LOOPS);
        }
        // third column's launch as a whole: 8192 waves x REP x LOOPS repetitions over 1024 SIMDs in wall_ms
        printf("%-48s %2d instr: %7.1f | %7.1f | %7.1f   (whole launch: %.1f us = %.2f repetitions per SIMD per us)\n", p.name, p.n_instr, c[0], c[1], c[2],
               wall_ms * 1e3, 8.0 * REP * LOOPS / (wall_ms * 1e3));
    }
    return 0;
}
