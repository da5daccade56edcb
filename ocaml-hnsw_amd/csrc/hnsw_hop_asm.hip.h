// hnsw_hop_asm.hip.h -- the layer-0 loop of Ohnsw.search_k (lib/ohnsw.ml:543-588) and the descent above it
// (Ohnsw.search_one, :492-508), written instruction by instruction for gfx950.
//
// Shapes: rows of 65..128 dimensions (NCH = 2) -- and, for byte rows and for full / ragged float32 rows, of 129..256 (NCH = 4:
// the rounds with twice the loads and arithmetic per batch) --, ef <= 64 / 65..128 / 129..256 / 257..512 (W in one / two / four / eight key
// registers per lane -- and three / six for ef 129..192 / 257..384; hnsw_hop_loop.inc has one body for one register, one for two
// and one for three and more, included once per shape by the generated table hnsw_hop_instances.inc; what depends on the slot
// count above two is generated too: hnsw_hop_slots.inc):
//   * byte rows and a byte-valued query (exact integer arithmetic, see hop_round), L2 -- the headline shape; its descent too --
//     and inner product;
//   * float32 rows, L2 and inner product: full, ragged and split rows ("The same loops over FLOAT32 rows" below);
//   * each of them for the Ohnsw accept rule and for the functor rule (Hnsw_algo.Search: the loop then leaves a hop when an
//     entry would enter the tie set, see HNSW_LOOP_SEM in hnsw_hop_loop.inc and search_layer).
// Everything else takes search_layer's C++ loop; the blocks compute exactly what that loop computes (same pops, same
// evaluations, same insertions, same counters), so the two are interchangeable and tests/ compare both against the oracle.
//
// Why by hand (numbers: tools/issue_latency.hip -> profiles/r03_issue_latency.txt): a CU issues ONE scalar instruction per
// cycle for its four SIMDs and a SIMD one vector wave-instruction per ~2.4 cycles, so in a loaded launch scalar
// instructions are the dearest, and hipcc's hop was ~125 vector + ~115 scalar instructions, a third of them glue
// (loop-carried register copies, boolean flags materialised in mask registers and re-tested, skip branches around single
// instructions).  A wave that runs alone -- the longest walk and the late starters are what a 10 k launch ends with --
// issues one instruction per 4 cycles, waits 16 more whenever a scalar instruction reads what a vector one has just
// written, and pays ~20 for every taken branch.  Here a hop is one fall-through chain: pop (s_ff1 + v_readlane +
// v_writelane for the flag), adjacency row (or the row fetched speculatively during the previous hop), visited filter,
// compaction through LDS, one round of 4 / 8 / 16 rows with every load issued before the first is consumed, integer dot
// products, ONE transposing reduction for the round's sums, accept ballot, and the insertion loop; what is seldom taken
// sits behind the loop.
#pragma once

namespace hnsw_dev {

#ifndef HNSW_ASM_LOOP
#define HNSW_ASM_LOOP 1
#endif
#ifndef HNSW_ASM_LOOP_F32       /* the float32-row instantiations (0: those shapes keep search_layer's C++ loop) */
#define HNSW_ASM_LOOP_F32 1
#endif
#ifndef HNSW_ASM_LOOP_SPLIT     /* ... and the split-row ones */
#define HNSW_ASM_LOOP_SPLIT 1
#endif
#ifndef HNSW_ASM_LOOP_8SLOTS    /* ... and W in eight registers (ef 257..512) */
#define HNSW_ASM_LOOP_8SLOTS 1
#endif
#ifndef HNSW_ASM_LOOP_BYTES4    /* ... and byte rows of 129..256 dimensions */
#define HNSW_ASM_LOOP_BYTES4 1
#endif
#ifndef HNSW_ASM_LOOP_F32N4      /* ... and float32 rows of 129..256 dimensions */
#define HNSW_ASM_LOOP_F32N4 1
#endif
#ifndef HNSW_ASM_LOOP_SEM1      /* the instantiations for the functor accept rule (0: that rule keeps the C++ loop) */
#define HNSW_ASM_LOOP_SEM1 1
#endif
// measurement builds (-DHNSW_ASM_PHASE=k, tools/asm_phases.sh): shader-clock cycles spent between probe point k and k + 1 of
// every hop, summed into the n_dist counter.  Points: 0 hop start, 1 adjacency row in registers, 2 fresh list written,
// 3 round evaluated and accept mask known, 4 insertions done.  s[90:93] are used by name (declared clobbered).
#ifdef HNSW_ASM_PHASE
#define HNSW_PROBE_BEGIN "s_memtime s[90:91]\n\ts_waitcnt lgkmcnt(0)\n\t"
#define HNSW_PROBE_END "s_memtime s[92:93]\n\ts_waitcnt lgkmcnt(0)\n\ts_sub_u32 s92, s92, s90\n\ts_add_u32 %[nd], %[nd], s92\n\t"
#define HNSW_PROBE(K) HNSW_PROBE_SEL(K, HNSW_ASM_PHASE)
#define HNSW_PROBE_SEL(K, P) HNSW_PROBE_SEL2(K, P)
#define HNSW_PROBE_SEL2(K, P) HNSW_PROBE_##K##_##P
#define HNSW_PROBE_0_0 HNSW_PROBE_BEGIN
#define HNSW_PROBE_1_0 HNSW_PROBE_END
#define HNSW_PROBE_2_0
#define HNSW_PROBE_3_0
#define HNSW_PROBE_4_0
#define HNSW_PROBE_0_1
#define HNSW_PROBE_1_1 HNSW_PROBE_BEGIN
#define HNSW_PROBE_2_1 HNSW_PROBE_END
#define HNSW_PROBE_3_1
#define HNSW_PROBE_4_1
#define HNSW_PROBE_0_2
#define HNSW_PROBE_1_2
#define HNSW_PROBE_2_2 HNSW_PROBE_BEGIN
#define HNSW_PROBE_3_2 HNSW_PROBE_END
#define HNSW_PROBE_4_2
#define HNSW_PROBE_0_3
#define HNSW_PROBE_1_3
#define HNSW_PROBE_2_3
#define HNSW_PROBE_3_3 HNSW_PROBE_BEGIN
#define HNSW_PROBE_4_3 HNSW_PROBE_END
#define HNSW_PROBE_CLOBBER , "s90", "s91", "s92", "s93"
#else
#define HNSW_PROBE(K)
#define HNSW_PROBE_CLOBBER
#endif
#ifdef HNSW_ASM_STATS      /* measurement build: status bits 8.. count the hops whose adjacency row had been fetched speculatively */
#define HNSW_ASM_COUNT_HIT "s_add_u32 %[st], %[st], 0x100\n\t"
#else
#define HNSW_ASM_COUNT_HIT
#endif
// Where a hop loop starts relative to a 64-byte instruction line: k dwords past one.  A lone wave's query time moves by
// up to 4 % with k, with a period of 8 dwords (which branch targets end a 32-byte fetch window): two-slot loop on C2 0.1369 ms
// at k = 2 or 10, 0.141 at 4 / 12, 0.1425-0.1435 at 0 / 6 / 8 / 14; the loaded launches follow by ~1 %.  Pinned per loop so
// that code added in front of a loop does not move it.  (tools/mkvariant.sh -DHNSW_ASM_ALIGN_PAD[1|4]=k)
#ifndef HNSW_ASM_ALIGN_PAD
#define HNSW_ASM_ALIGN_PAD 2
#endif
#ifndef HNSW_ASM_ALIGN_PAD4
#define HNSW_ASM_ALIGN_PAD4 6        // four-slot loop, ef 192: 0.2116 ms at 6, 0.214 at 0 / 4, 0.221 at 2
#endif
#ifndef HNSW_ASM_ALIGN_PAD1
#define HNSW_ASM_ALIGN_PAD1 0        // one-slot loop, ef 64: within 1.5 % over k
#endif
#define HNSW_STR2(x) #x
#define HNSW_STR(x) HNSW_STR2(x)
#define HNSW_ASM_ALIGN_K(K) ".p2align 6\n\t.rept " HNSW_STR(K) "\n\ts_nop 0\n\t.endr\n"
#define HNSW_ASM_ALIGN HNSW_ASM_ALIGN_K(HNSW_ASM_ALIGN_PAD)
#ifndef HNSW_ASM_PREFETCH
#define HNSW_ASM_PREFETCH 1
#endif

// One hand-scheduled layer-0 loop per shape: an explicit specialisation of this template (hnsw_hop_loop.inc, instantiated by the
// generated table hnsw_hop_instances.inc) with available = true and
//     static bool run(iv, w, cx, rs, qv, n_dist, n_hops, status [, maxhops])
// which runs the layer-0 search to completion (false) or -- functor rule, SEM 1 -- until an entry would enter the tie set (true:
// the hop is finished by search_layer from `rs`).  On entry W holds the start node (unexpanded) and the visited set knows it.
// qv: the query as float4 chunks (float32 rows; byte rows read cx.qb); maxhops: debugging builds of the two-slot loop only.
template <int NCH, int NSLOT, int METRIC, int ROWS, int SEM, int BLK> struct HopLoop { static constexpr bool available = false; };

__device__ __forceinline__ uint32_t lds_offset(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

#define HNSW_DPP_ALL " row_mask:0xf bank_mask:0xf"
#define HNSW_DPP_BC " row_mask:0xf bank_mask:0xf bound_ctrl:1"

// ids of batch B of a round -> LDS read.  Group r reads the round's candidate NB r + B, clamped to the last candidate of
// the list (a group past the end re-reads a row that is in flight anyway; its key is masked out below).
//   sx = address of the round's first candidate; lastad = address of the list's last; r4 = 4 r per lane; SH = log2 NB
#define HNSW_ID_READ0(ID, SH)                                                            \
    "v_lshl_add_u32 %[t0], %[r4], " #SH ", %[sx]\n\t"                                    \
    "v_min_u32 %[t0], %[lastad], %[t0]\n\t"                                              \
    "ds_read_b32 " ID ", %[t0]\n\t"
#define HNSW_ID_READN(ID)   /* the next batch: one candidate further, clamped again */    \
    "v_add_u32_e32 %[t0], 4, %[t0]\n\t"                                                  \
    "v_min_u32 %[t0], %[lastad], %[t0]\n\t"                                              \
    "ds_read_b32 " ID ", %[t0]\n\t"
// row address (one 64-bit multiply-add) and the row's two dwords per lane
#define HNSW_ROW_LOAD(ID, AD, DA, DB)                                                    \
    "v_mad_u64_u32 " AD ", vcc, " ID ", %[st8], %[xl]\n\t"                               \
    "global_load_dword " DA ", " AD ", off\n\t"                                          \
    "global_load_dword " DB ", " AD ", off offset:64\n\t"
// x.q -> TA and x.x -> DA for this lane's 8 dimensions; HNSW_COMBINE: |x - q|^2 - q.q = x.x - 2 x.q -> DA.
// gfx950 hazard (not interlocked, and nothing inserts wait states into inline assembly): the result of a v_dot4 may be
// read or overwritten by a DIFFERENT vector instruction only 3 wait states later (a following v_dot4 that takes it
// as its accumulator is fine), so the combines come after other batches' dot products or after an s_nop.  These and the
// other hand-counted states of this file (VALU-written SGPR / VCC -> VALU 2, -> lane select 4; VALU-written VGPR -> DPP 2)
// are checked on the disassembled code object by tools/check_asm_hazards.py (tests/test_asm_hazards.py, CPU side).
#define HNSW_DOTS(DA, DB, TA)                                                            \
    "v_dot4_u32_u8 " TA ", " DA ", %[qb0], 0\n\t"                                        \
    "v_dot4_u32_u8 " DA ", " DA ", " DA ", %[q2v]\n\t"                                     \
    "v_dot4_u32_u8 " TA ", " DB ", %[qb1], " TA "\n\t"                                   \
    "v_dot4_u32_u8 " DA ", " DB ", " DB ", " DA "\n\t"
#define HNSW_COMBINE(DA, TA) "v_mad_i32_i24 " DA ", " TA ", -2, " DA "\n\t"
// the same for the inner product (hop_round: METRIC 1 on byte rows): x.q alone -> DA; the combine's slot stays a wait state, so
// that the rounds' hand-counted distances are those of the L2 text
#define HNSW_DOTS_IP(DA, DB, TA)                                                         \
    "v_dot4_u32_u8 " TA ", " DA ", %[qb0], 0\n\t"                                        \
    "v_dot4_u32_u8 " DA ", " DB ", %[qb1], " TA "\n\t"
#define HNSW_COMBINE_IP(DA, TA) "s_nop 0\n\t"
// the group's integer sum SRC -> key (dist_to_key of (float)sum): L2 the float's bits; inner product 1 - sum, sign-flipped
#define HNSW_B8_KEY_L2(SRC) "v_cvt_f32_i32_e32 %[ckey], " SRC "\n\t"
#define HNSW_B8_KEY_IP(SRC)                                                              \
    "v_cvt_f32_i32_e32 %[t1], " SRC "\n\t"                                               \
    "v_sub_f32_e32 %[t1], 1.0, %[t1]\n\t"                                                \
    "v_ashrrev_i32_e32 %[t0], 31, %[t1]\n\t"                                             \
    "v_or_b32_e32 %[t0], 0x80000000, %[t0]\n\t"                                          \
    "v_xor_b32_e32 %[ckey], %[t1], %[t0]\n\t"

// The entry that falls off W is at the new maximum's distance (label LBL; BACK: the way on; LASTLO: the top slot's low halves).
// A dummy: nothing happens.  Ohnsw rule (HNSW_EVICT_TIE_PUSH): a real, unexpanded one stays in the candidate queue
// (lib/ohnsw.ml:568 is false for it): pushed on the tie list in LDS.  The list holds entries at distance od; it is alive
// while od == max(W).d (a list left over from a larger maximum is restarted here); full (64 entries): the query is flagged
// and searched again by the caller.  Functor rule (HNSW_EVICT_TIE_BAIL): the list is a set with other entries and another
// pop order (search_layer): the loop is left with the candidate still in `fresh`, see hnsw_hop_loop.inc.
#define HNSW_EVICT_TIE_PUSH(LBL, BACK, LASTLO, FULL)                                                                        \
    LBL ":\n\t"                                                                                                             \
    "s_cmp_eq_u32 %[wmax], -2\n\t"                                                                                          \
    "s_cbranch_scc1 " BACK "\n\t"                                        /* W still holds dummies: nothing real falls off */ \
    "v_readlane_b32 %[t], " LASTLO ", 63\n\t"                                                                               \
    "s_bitcmp1_b32 %[t], 31\n\t"                                                                                            \
    "s_cbranch_scc1 " BACK "\n\t"                                        /* expanded: gone for good */                     \
    "s_cmp_eq_u32 %[od], %[nw]\n\t"                                                                                         \
    "s_cselect_b32 %[oc], %[oc], 0\n\t"                                                                                     \
    "s_mov_b32 %[od], %[nw]\n\t"                                                                                            \
    "s_cmp_ge_u32 %[oc], 64\n\t"                                                                                            \
    "s_cbranch_scc1 " FULL "f\n\t"                                                                                          \
    "s_sub_u32 %[t], %[t], 1\n\t"                                                                                           \
    "s_lshl2_add_u32 %[tmp], %[oc], %[cand]\n\t"                                                                            \
    "v_mov_b32_e32 %[t0], %[tmp]\n\t"                                                                                       \
    "v_mov_b32_e32 %[t1], %[t]\n\t"                                                                                         \
    "s_mov_b64 exec, 1\n\t"                                                                                                 \
    "ds_write_b32 %[t0], %[t1] offset:768\n\t"                                                                              \
    "s_mov_b64 exec, -1\n\t"                                                                                                \
    "s_add_u32 %[oc], %[oc], 1\n\t"                                                                                         \
    "s_branch " BACK "\n"                                                                                                   \
    FULL ":\n\t"                                                                                                            \
    "s_or_b32 %[st], %[st], 1\n\t"                                        /* list full: flagged, the host searches again */ \
    "s_branch " BACK "\n"
#define HNSW_EVICT_TIE_BAIL(LBL, BACK, LASTLO, FULL)                                                                        \
    LBL ":\n\t"                                                                                                             \
    "s_cmp_eq_u32 %[wmax], -2\n\t"                                                                                          \
    "s_cbranch_scc1 " BACK "\n\t"                                        /* W still holds dummies: nothing real falls off */ \
    "s_branch 98f\n"

// The insertion loop of one round (labels 10 loop entry, 110 next candidate, 19 done): insert_island2's steps, ordered for a
// wave that runs alone and counted for a chip that is full.  Alone: a scalar instruction that reads what a vector
// instruction has just written (v_readlane, v_cmp -> SGPR) waits ~16 cycles beyond its issue slot and a taken branch costs
// ~20 (profiles/r03_issue_latency.txt), so the candidate's key halves, max(W)'s predecessor and the rank compares are all
// issued before the first scalar use, and the common way through (rank in the upper slot, then the next candidate or none)
// falls through.  Full: the CU's four SIMDs share one scalar unit, so scalar instructions are the dearest: W is sorted, so
// any key of the upper slot below the candidate puts the rank there (one count gives the position: its SCC picks the slot);
// the shift is a v_cndmask_b32_dpp under a mask made by one s_bfm (no EXEC writes); the tie list is not cleared when max(W)
// drops -- it carries the distance it was filled at (od) and is alive only while that is max(W).d.
// The lower-slot shift and the rare cases are in HNSW_INSERT_RARE, behind the hop loop.
#define HNSW_INSERT_LOOP                                                                                                    \
    "10:\n\t"                                                                                                               \
    "s_cmp_eq_u64 %[fresh], 0\n\t"                                                                                          \
    "s_cbranch_scc1 19f\n"                                                                                                  \
    "110:\n\t"                                                                                                              \
    "s_ff1_i32_b64 %[i], %[fresh]\n\t"                                                                                      \
    "v_readlane_b32 %[kd], %[ckey], %[i]\n\t"                                                                               \
    "v_readlane_b32 %[klo], %[cid], %[i]\n\t"                                                                               \
    "v_readlane_b32 %[nw], %[h1], 62\n\t"                                                                                   \
    "v_cmp_eq_u32_e64 %[um0], %[kd], %[h0]\n\t"                                                                             \
    "v_cmp_eq_u32_e64 %[um1], %[kd], %[h1]\n\t"                                                                             \
    "v_cmp_gt_u32_e64 %[g0], %[kd], %[h0]\n\t"                                                                              \
    "v_cmp_gt_u32_e32 vcc, %[kd], %[h1]\n\t"                                                                                \
    "s_cmp_ge_u32 %[kd], %[wmax]\n\t"                                                                                       \
    "s_cbranch_scc1 " HNSW_SEM_REJECT "\n\t"                                              /* no longer below max(W): rejected, :574 */      \
    "s_or_b64 %[um0], %[um0], %[um1]\n\t"                                                                                   \
    "s_cbranch_scc1 14f\n\t"                                              /* members of W at this very distance */          \
    "s_max_u32 %[nw], %[nw], %[kd]\n\t"                                   /* the new max(W).d (this key if it ranks last) */ \
    "s_cmp_eq_u32 %[nw], %[wmax]\n\t"                                                                                       \
    "s_cbranch_scc1 15f\n"                                                /* the entry falling off ties with it */          \
    "12:\n\t"                                                                                                               \
    "s_bcnt1_i32_b64 m0, vcc\n\t"                                         /* upper-slot keys below: any -> the rank is there */ \
    "s_cbranch_scc0 13f\n"                                                                                                  \
    "120:\n\t"                                                            /* upper slot from lane p (bits 5:0) on */        \
    "s_bfm_b64 vcc, m0, 0\n\t"                                            /* lanes below m0 keep their keys */               \
    "v_cndmask_b32_dpp %[h1], %[h1], %[h1], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_cndmask_b32_dpp %[l1], %[l1], %[l1], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_writelane_b32 %[h1], %[kd], m0\n\t"                                                                                  \
    "v_writelane_b32 %[l1], %[klo], m0\n\t"                                                                                 \
    "s_mov_b32 %[wmax], %[nw]\n"                                                                                            \
    "18:\n\t"                                                                                                               \
    "s_bitset0_b64 %[fresh], %[i]\n\t"                                                                                      \
    "s_cmp_lg_u64 %[fresh], 0\n\t"                                                                                          \
    "s_cbranch_scc1 110b\n"                                               /* the next accepted candidate */                 \
    "19:\n\t"

#define HNSW_INSERT_RARE                                                                                                    \
    "13:\n\t"                                                             /* no upper-slot key below: the rank is the lower slot's count */ \
    "s_bcnt1_i32_b64 m0, %[g0]\n\t"                                                                                         \
    "s_bitcmp1_b32 m0, 6\n\t"                                                                                               \
    "s_cbranch_scc1 120b\n"                                               /* all 64 below: lane 0 of the upper slot (p & 63 = 0) */ \
    "130:\n\t"                                                            /* lower slot from lane p on; the upper slot moves whole */ \
    "v_readlane_b32 %[sx], %[h0], 63\n\t"                                                                                   \
    "v_readlane_b32 %[tmp], %[l0], 63\n\t"                                                                                  \
    "v_mov_b32_dpp %[h1], %[h1] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_mov_b32_dpp %[l1], %[l1] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "s_bfm_b64 vcc, m0, 0\n\t"                                                                                              \
    "v_cndmask_b32_dpp %[h0], %[h0], %[h0], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_cndmask_b32_dpp %[l0], %[l0], %[l0], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_writelane_b32 %[h1], %[sx], 0\n\t"                                                                                   \
    "v_writelane_b32 %[l1], %[tmp], 0\n\t"                                                                                  \
    "v_writelane_b32 %[h0], %[kd], m0\n\t"                                                                                  \
    "v_writelane_b32 %[l0], %[klo], m0\n\t"                                                                                 \
    "s_mov_b32 %[wmax], %[nw]\n\t"                                                                                          \
    "s_bitset0_b64 %[fresh], %[i]\n\t"                                                                                      \
    "s_cmp_lg_u64 %[fresh], 0\n\t"                                                                                          \
    "s_cbranch_scc1 110b\n\t"                                                                                               \
    "s_branch 19b\n"                                                                                                        \
    /* rare: distance tie inside W.  The full rank: keys at a smaller distance + members at this distance with a smaller id; a */ \
    /* member with this id (flag either way) means the node is already in W (a re-evaluation the visited cache forgot): ignored */ \
    "14:\n\t"                                                                                                               \
    "v_cmp_eq_u32_e64 %[um0], %[kd], %[h0]\n\t"                                                                             \
    "v_cmp_eq_u32_e64 %[um1], %[kd], %[h1]\n\t"                                                                             \
    "s_bcnt1_i32_b64 %[p], %[g0]\n\t"                                                                                       \
    "s_bcnt1_i32_b64 %[t], vcc\n\t"                                                                                         \
    "s_add_u32 %[p], %[p], %[t]\n\t"                                                                                        \
    "v_and_b32_e32 %[t0], 0x7fffffff, %[l0]\n\t"                          /* id + 1 without the flag */                     \
    "v_and_b32_e32 %[t1], 0x7fffffff, %[l1]\n\t"                                                                            \
    "v_cmp_gt_u32_e32 vcc, %[klo], %[t0]\n\t"                                                                               \
    "s_and_b64 vcc, vcc, %[um0]\n\t"                                                                                        \
    "s_bcnt1_i32_b64 %[t], vcc\n\t"                                                                                         \
    "s_add_u32 %[p], %[p], %[t]\n\t"                                                                                        \
    "v_cmp_gt_u32_e32 vcc, %[klo], %[t1]\n\t"                                                                               \
    "s_and_b64 vcc, vcc, %[um1]\n\t"                                                                                        \
    "s_bcnt1_i32_b64 %[t], vcc\n\t"                                                                                         \
    "s_add_u32 %[p], %[p], %[t]\n\t"                                                                                        \
    "v_cmp_eq_u32_e32 vcc, %[klo], %[t0]\n\t"                                                                               \
    "s_and_b64 %[um0], vcc, %[um0]\n\t"                                                                                     \
    "v_cmp_eq_u32_e32 vcc, %[klo], %[t1]\n\t"                                                                               \
    "s_and_b64 %[um1], vcc, %[um1]\n\t"                                                                                     \
    "s_or_b64 %[um0], %[um0], %[um1]\n\t"                                                                                   \
    "s_cbranch_scc1 18b\n\t"                                              /* already in W */                                \
    "s_max_u32 %[nw], %[nw], %[kd]\n\t"                                                                                     \
    "s_cmp_eq_u32 %[nw], %[wmax]\n\t"                                                                                       \
    "s_cbranch_scc1 151f\n"                                                                                                 \
    "141:\n\t"                                                                                                              \
    "s_mov_b32 m0, %[p]\n\t"                                                                                                \
    "s_cmp_lt_u32 %[p], 64\n\t"                                                                                             \
    "s_cbranch_scc1 130b\n\t"                                                                                               \
    "s_branch 120b\n"                                                                                                       \
    /* rare: the entry that falls off is at the new maximum's distance.  A dummy: nothing happens.  A real, unexpanded one   */ \
    /* stays in the candidate queue (lib/ohnsw.ml:568 is false for it): pushed on the tie list.  The list holds entries at */ \
    /* distance od; it is alive while od == max(W).d (a list left over from a larger maximum is restarted here)            */ \
    HNSW_EVICT_TIE("15", "12b", "%[l1]", "16")                                                                                                        \
    HNSW_EVICT_TIE("151", "141b", "%[l1]", "161")

// accept ballot of a round: candidate index of this lane within the round = byte SHAPE of the per-lane constant co (0xff on
// a lane that holds no candidate's sum), valid below cnt, and below the current max(W) (lib/ohnsw.ml:574)
// (cnt = candidates not yet evaluated when the round starts).  The first half does not depend on the sums: it fills the
// two wait states a DPP read of a just-written register needs anyway.
#define HNSW_ACCEPT_EARLY(SHAPE)                                                         \
    "v_add_u32_e32 %[cid], 1, %[cid]\n\t"           /* ids -> low key halves: id + 1, unexpanded */          \
    "v_cmp_gt_u32_sdwa vcc, %[cnt], %[co] src0_sel:DWORD src1_sel:BYTE_" #SHAPE "\n\t"
#define HNSW_ACCEPT_LATE                                                                 \
    "v_cmp_" HNSW_SEM_ACCEPT "_u32_e64 %[fresh], %[wmax], %[ckey]\n\t"  /* Ohnsw: below max(W); functor rule: not above */ \
    "s_and_b64 %[fresh], %[fresh], vcc\n\t"

// label 4: the hop's node is in klo: count it, then its adjacency row (Graph.adjacent, :570).  The row was requested during
// the previous hop if the guess of the next node was right (nine hops in ten): that path falls through; a miss leaves the
// line (44, HNSW_HOP_ADJACENCY_MISS) and comes back at 6 -- a taken branch costs a lone wave five issue slots.
#define HNSW_HOP_ADJACENCY \
        "4:\n\t"                            \
        "s_add_u32 %[nh], %[nh], 1\n\t"     \
        HNSW_SPLIT_HOP                      \
        "s_cmp_lg_u32 %[kd], %[pref]\n\t"   \
        "s_cbranch_scc1 44f\n\t"            \
        HNSW_ASM_COUNT_HIT                  \
        "s_waitcnt vmcnt(0)\n\t"            \
        "v_mov_b32_e32 %[nb], %[pnb]\n"
#define HNSW_HOP_ADJACENCY_MISS \
        "44:\n\t"                                                                                                             \
        "s_mul_i32 %[tmp], %[kd], %[rowb]\n\t"   /* (id + 1) rows: the base is one row early; the caller checked the table's size */ \
        "v_lshl_add_u32 %[t0], %[lane], 2, %[tmp]\n\t"                                                                        \
        "s_waitcnt vmcnt(0)\n\t"  /* a wrong guess still in flight is drained first (its target is pnb) */                    \
        "v_mov_b32_e32 %[nb], -1\n\t"                                                                                         \
        "s_mov_b64 exec, %[rowm]\n\t"                                                                                         \
        "global_load_dword %[nb], %[t0], %[nbrm]\n\t"                                                                          \
        "s_mov_b64 exec, -1\n\t"                                                                                              \
        "s_waitcnt vmcnt(0)\n\t"                                                                                              \
        "s_branch 6b\n"

// label 6: visited filter (Visited.mem, :571), first half: the set's word is requested, the tag computed
#define HNSW_HOP_FILTER_ISSUE \
        "6:\n\t"                                                               \
  /* visited filter (Visited.mem, :571): 2-way set of 16-bit tags per word */  \
        "v_and_b32_e32 %[va], %[setm], %[nb]\n\t"                              \
        "v_lshl_add_u32 %[va], %[va], 2, %[vtb]\n\t"                           \
        "ds_read_b32 %[vw], %[va]\n\t"                                         \
        "v_lshrrev_b32_e32 %[tag], %[setb], %[nb]\n\t"                         \
        "v_cmp_lt_i32_e32 vcc, -1, %[nb]\n\t"                                  

// The cache's ways (visited_three_ways, hnsw_device.hip.h): the one- and two-slot loops always run two 16-bit ways; the four- and
// eight-slot loops (ef > 128) take three 10-bit ways when the index is small enough -- a uniform branch on the shift (16 | 10) to
// the three-field compare behind the loop (label 96, HNSW_VT_RUNTIME_RARE), which comes back at 97
#define HNSW_VT_TWO_WAYS_TEST
#define HNSW_VT_TWO_WAYS_JOIN "\t"
#define HNSW_VT_TWO_WAYS_SHIFT "16"
#define HNSW_VT_TWO_WAYS_RARE
#define HNSW_VT_RUNTIME_TEST                                                             \
        "s_cmp_lg_u32 %[tsh], 16\n\t"                                                    \
        "s_cbranch_scc1 96f\n\t"
#define HNSW_VT_RUNTIME_JOIN "97:\n\t"
#define HNSW_VT_RUNTIME_SHIFT "%[tsh]"
#define HNSW_VT_RUNTIME_RARE                                                             \
        "96:\n\t"                                                                        \
        "v_bfe_u32 %[t0], %[vw], 0, 10\n\t"                                              \
        "v_cmp_ne_u32_e64 %[um0], %[t0], %[tag]\n\t"                                     \
        "v_bfe_u32 %[t0], %[vw], 10, 10\n\t"                                             \
        "v_cmp_ne_u32_e64 %[um1], %[t0], %[tag]\n\t"                                     \
        "s_and_b64 %[fresh], %[um0], %[um1]\n\t"                                         \
        "v_bfe_u32 %[t0], %[vw], 20, 10\n\t"                                             \
        "v_cmp_ne_u32_e64 %[um0], %[t0], %[tag]\n\t"                                     \
        "s_and_b64 %[fresh], %[fresh], %[um0]\n\t"                                       \
        "s_branch 97b\n"
// visited filter, second half: 2-way compare -> fresh; nothing fresh: next hop (1b); Visited.add (:572) and the list of
// fresh neighbours in row order; the round variables
#define HNSW_HOP_FILTER_COMPACT \
        "s_waitcnt lgkmcnt(0)\n\t"                                                     \
        HNSW_VT_WAYS_TEST                                                              \
        "v_cmp_ne_u32_sdwa %[um0], %[vw], %[tag] src0_sel:WORD_0 src1_sel:DWORD\n\t"   \
        "v_cmp_ne_u32_sdwa %[um1], %[vw], %[tag] src0_sel:WORD_1 src1_sel:DWORD\n\t"   \
        "s_and_b64 %[fresh], %[um0], %[um1]\n"                                         \
        HNSW_VT_WAYS_JOIN                                                              \
        "s_and_b64 %[fresh], %[fresh], vcc\n\t"                                        \
        "s_cbranch_scc0 1b\n\t"  /* nothing fresh: next hop */                         \
  /* Visited.add (:572) and the list of fresh neighbours, in row order */              \
        "s_bcnt1_i32_b64 %[cnt], %[fresh]\n\t"                                         \
        "s_mov_b64 exec, %[fresh]\n\t"                                                 \
        "v_mbcnt_lo_u32_b32 %[t0], exec_lo, 0\n\t"  /* exec is the fresh mask here */  \
        "v_mbcnt_hi_u32_b32 %[t0], exec_hi, %[t0]\n\t"                                 \
        "v_lshl_or_b32 %[vw], %[vw], " HNSW_VT_SHIFT ", %[tag]\n\t"                     \
        "ds_write_b32 %[va], %[vw]\n\t"                                                \
        "v_lshl_add_u32 %[t0], %[t0], 2, %[cand]\n\t"                                  \
        "ds_write_b32 %[t0], %[nb]\n\t"                                                \
        HNSW_SPLIT_COMPACT                                                             \
        "s_mov_b64 exec, -1\n\t"                                                       \
        "s_add_u32 %[nd], %[nd], %[cnt]\n\t"                                           \
        "s_lshl2_add_u32 %[lastad], %[cnt], %[candm4]\n\t"                             \
        "s_mov_b32 %[sx], %[cand]\n"                                                   

// ---- Visited as bitmap blocks over the locality codes (visited_blocks_mem_add, hnsw_device.hip.h), for the loops with W in four
// or eight registers instantiated with HNSW_LOOP_BLK 1.  The hop reads lcode0's row beside the adjacency row (nc / pnc beside nb /
// pnb: same lane, same offset, base lcm), and the filter is the five steps of the C++ routine, same order, same arithmetic (a
// build without the loops must count the same evaluations): the set's eight directory words (block << 8 | stamp), the way that
// carries the lane's block or else the least recently touched one (largest (age << 3) | way, age = (now - stamp) mod 256), 1 the
// bit of the lanes that hit, 2 every valid lane's (block << 8) | now into its way, 3 read back: the lane owns the slot iff the word
// carries its block, 4 owners that missed clear the slot's bitmap, 5 owners whose bit was clear set it.  Scratch: the first
// fifteen row registers (no rows are in flight between the rounds of two hops); um0..um3 are free between the peek and the
// insertion.  vcc holds the valid lanes from label 6 on (every compare in between writes a scalar pair).
// The scratch registers by number: HNSW_BLK_R(I) = HNSW_BLK_REG_I, which hnsw_hop_loop.inc defines per row format -- the float32
// loops' row registers are named (v40 ..: consecutive, so the directory is read and a bitmap cleared 128 bits at a time), the byte
// loops' are operands (d0 .. d7, id0 .. id3, ta, tb, ckey: one dword per LDS instruction).
#define HNSW_BLK_R(I) HNSW_BLK_REG_##I
#define HNSW_BLK_F32_REG(I) "v[" HNSW_STR(HNSW_F32_BASE) "+" #I "]"
#define HNSW_BLK_F32_REG4(I) "v[" HNSW_STR(HNSW_F32_BASE) "+" #I ":" HNSW_STR(HNSW_F32_BASE) "+" #I "+3]"
#define HNSW_BLK_DIR_READ_F32                                                   \
        "ds_read_b128 " HNSW_BLK_F32_REG4(0) ", %[va]\n\t"                      \
        "ds_read_b128 " HNSW_BLK_F32_REG4(4) ", %[va] offset:16\n\t"
#define HNSW_BLK_DIR_READ_B8                                                    \
        "ds_read_b32 %[d0], %[va]\n\t"                                          \
        "ds_read_b32 %[d1], %[va] offset:4\n\t"                                 \
        "ds_read_b32 %[d2], %[va] offset:8\n\t"                                 \
        "ds_read_b32 %[d3], %[va] offset:12\n\t"                                \
        "ds_read_b32 %[d4], %[va] offset:16\n\t"                                \
        "ds_read_b32 %[d5], %[va] offset:20\n\t"                                \
        "ds_read_b32 %[d6], %[va] offset:24\n\t"                                \
        "ds_read_b32 %[d7], %[va] offset:28\n\t"
// (the zeros: register 0 .. 3 of the scratch set, resp. register 0 alone)
#define HNSW_BLK_ZERO_F32                                                       \
        "v_mov_b32_e32 " HNSW_BLK_F32_REG(0) ", 0\n\t"                          \
        "v_mov_b32_e32 " HNSW_BLK_F32_REG(1) ", 0\n\t"                          \
        "v_mov_b32_e32 " HNSW_BLK_F32_REG(2) ", 0\n\t"                          \
        "v_mov_b32_e32 " HNSW_BLK_F32_REG(3) ", 0\n\t"
#define HNSW_BLK_ZERO_B8 "v_mov_b32_e32 %[d0], 0\n\t"
#define HNSW_BLK_CLEAR_F32(AD)                                                  \
        "ds_write_b128 " AD ", " HNSW_BLK_F32_REG4(0) "\n\t"                    \
        "ds_write_b128 " AD ", " HNSW_BLK_F32_REG4(0) " offset:16\n\t"
#define HNSW_BLK_CLEAR_B8(AD)                                                   \
        "ds_write_b32 " AD ", %[d0]\n\t"                                        \
        "ds_write_b32 " AD ", %[d0] offset:4\n\t"                               \
        "ds_write_b32 " AD ", %[d0] offset:8\n\t"                               \
        "ds_write_b32 " AD ", %[d0] offset:12\n\t"                              \
        "ds_write_b32 " AD ", %[d0] offset:16\n\t"                              \
        "ds_write_b32 " AD ", %[d0] offset:20\n\t"                              \
        "ds_write_b32 " AD ", %[d0] offset:24\n\t"                              \
        "ds_write_b32 " AD ", %[d0] offset:28\n\t"
#define HNSW_HOP_ADJACENCY_BLK \
        "4:\n\t"                            \
        "s_add_u32 %[nh], %[nh], 1\n\t"     \
        HNSW_SPLIT_HOP                      \
        "s_cmp_lg_u32 %[kd], %[pref]\n\t"   \
        "s_cbranch_scc1 44f\n\t"            \
        HNSW_ASM_COUNT_HIT                  \
        "s_waitcnt vmcnt(0)\n\t"            \
        "v_mov_b32_e32 %[nb], %[pnb]\n\t"   \
        "v_mov_b32_e32 %[nc], %[pnc]\n"
#define HNSW_HOP_ADJACENCY_MISS_BLK \
        "44:\n\t"                                                                                                             \
        "s_mul_i32 %[tmp], %[kd], %[rowb]\n\t"                                                                                \
        "v_lshl_add_u32 %[t0], %[lane], 2, %[tmp]\n\t"                                                                        \
        "s_waitcnt vmcnt(0)\n\t"                                                                                              \
        "v_mov_b32_e32 %[nb], -1\n\t"                                                                                         \
        "v_mov_b32_e32 %[nc], 0\n\t"                                                                                          \
        "s_mov_b64 exec, %[rowm]\n\t"                                                                                         \
        "global_load_dword %[nb], %[t0], %[nbrm]\n\t"                                                                          \
        "global_load_dword %[nc], %[t0], %[lcm]\n\t"                                                                           \
        "s_mov_b64 exec, -1\n\t"                                                                                              \
        "s_waitcnt vmcnt(0)\n\t"                                                                                              \
        "s_branch 6b\n"
#define HNSW_HOP_PREFETCH_LOAD_BLK \
        "8:\n\t"                                                  \
        "s_mul_i32 %[tmp], %[pref], %[rowb]\n\t"                  \
        "v_lshl_add_u32 %[t1], %[lane], 2, %[tmp]\n\t"            \
        "v_mov_b32_e32 %[pnb], -1\n\t"                            \
        "v_mov_b32_e32 %[pnc], 0\n\t"                             \
        "s_mov_b64 exec, %[rowm]\n\t"                             \
        "global_load_dword %[pnb], %[t1], %[nbrm]\n\t"             \
        "global_load_dword %[pnc], %[t1], %[lcm]\n\t"              \
        "s_mov_b64 exec, -1\n"                                    \
        "9:\n\t"
// label 6: the lane's block, its set's directory words requested, the valid lanes
#define HNSW_HOP_FILTER_ISSUE_BLK \
        "6:\n\t"                                                               \
        "v_lshrrev_b32_e32 %[tag], 8, %[nc]\n\t"          /* block number */   \
        "v_and_b32_e32 %[va], %[bsm], %[tag]\n\t"         /* set */            \
        "v_lshl_add_u32 %[va], %[va], 5, %[vtb]\n\t"      /* its eight words */ \
        HNSW_BLK_DIR_READ                                                      \
        "v_cmp_lt_i32_e32 vcc, -1, %[nb]\n\t"
// one way: hit -> t0 = way; (age << 3) | way -> running maximum t1   (the compare's mask is read three instructions later)
#define HNSW_BLK_WAY(I) \
        "v_lshrrev_b32_e32 " HNSW_BLK_R(8) ", 8, " HNSW_BLK_R(I) "\n\t"                        \
        "v_cmp_eq_u32_e64 %[um0], " HNSW_BLK_R(8) ", %[tag]\n\t"                               \
        "v_sub_u32_e32 " HNSW_BLK_R(8) ", %[tmp], " HNSW_BLK_R(I) "\n\t"                        \
        "v_and_b32_e32 " HNSW_BLK_R(8) ", 0xff, " HNSW_BLK_R(8) "\n\t"                          \
        "v_cndmask_b32_e64 %[t0], %[t0], " #I ", %[um0]\n\t"                                   \
        "v_lshl_or_b32 " HNSW_BLK_R(8) ", " HNSW_BLK_R(8) ", 3, " #I "\n\t"                     \
        "v_max_u32_e32 %[t1], %[t1], " HNSW_BLK_R(8) "\n\t"
#define HNSW_HOP_FILTER_COMPACT_BLK \
        "s_bfe_u32 %[tmp], %[nh], 0x80001\n\t"            /* now = (hops >> 1) & 255 */                         \
        "v_mov_b32_e32 %[t0], -1\n\t"                                                                           \
        "v_mov_b32_e32 %[t1], 0\n\t"                                                                            \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
        HNSW_BLK_WAY(0) HNSW_BLK_WAY(1) HNSW_BLK_WAY(2) HNSW_BLK_WAY(3)                                         \
        HNSW_BLK_WAY(4) HNSW_BLK_WAY(5) HNSW_BLK_WAY(6) HNSW_BLK_WAY(7)                                         \
        "v_cmp_lt_i32_e64 %[um1], -1, %[t0]\n\t"                                 /* the lanes that hit */       \
        "v_and_b32_e32 %[t1], 7, %[t1]\n\t"                                      /* the victim's way */          \
        "v_subrev_u32_e32 " HNSW_BLK_R(8) ", %[vtb], %[va]\n\t"                  /* set * 32 */                  \
        "v_cndmask_b32_e64 %[t0], %[t1], %[t0], %[um1]\n\t"                      /* the lane's way */            \
        "v_lshlrev_b32_e32 " HNSW_BLK_R(8) ", 3, " HNSW_BLK_R(8) "\n\t"          /* set * 8 slots * 32 bytes */  \
        "v_lshl_add_u32 %[vw], %[t0], 2, %[va]\n\t"                              /* its directory word */        \
        "v_lshl_add_u32 " HNSW_BLK_R(8) ", %[t0], 5, " HNSW_BLK_R(8) "\n\t"                                      \
        "v_add_u32_e32 " HNSW_BLK_R(13) ", %[bmb], " HNSW_BLK_R(8) "\n\t"        /* the slot's bitmap */         \
        "v_bfe_u32 " HNSW_BLK_R(8) ", %[nc], 5, 3\n\t"                                                          \
        "v_lshl_add_u32 " HNSW_BLK_R(12) ", " HNSW_BLK_R(8) ", 2, " HNSW_BLK_R(13) "\n\t"   /* the code's word */ \
        "v_and_b32_e32 " HNSW_BLK_R(8) ", 31, %[nc]\n\t"                                                        \
        "v_lshlrev_b32_e64 " HNSW_BLK_R(11) ", " HNSW_BLK_R(8) ", 1\n\t"         /* ... and bit */               \
        "v_lshl_or_b32 " HNSW_BLK_R(9) ", %[tag], 8, %[tmp]\n\t"                 /* (block << 8) | now */        \
        "v_mov_b32_e32 " HNSW_BLK_R(10) ", 0\n\t"                                                               \
        "s_mov_b64 exec, %[um1]\n\t"                                                                            \
        "ds_read_b32 " HNSW_BLK_R(10) ", " HNSW_BLK_R(12) "\n\t"                 /* 1 */                         \
        "s_mov_b64 exec, vcc\n\t"                                                                               \
        "ds_write_b32 %[vw], " HNSW_BLK_R(9) "\n\t"                              /* 2 */                         \
        "ds_read_b32 " HNSW_BLK_R(14) ", %[vw]\n\t"                              /* 3 */                         \
        "s_mov_b64 exec, -1\n\t"                                                                                \
        HNSW_BLK_ZERO                                                                                           \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
        "v_and_b32_e32 " HNSW_BLK_R(10) ", " HNSW_BLK_R(10) ", " HNSW_BLK_R(11) "\n\t"                          \
        "v_lshrrev_b32_e32 " HNSW_BLK_R(14) ", 8, " HNSW_BLK_R(14) "\n\t"                                       \
        "v_cmp_ne_u32_e64 %[um2], 0, " HNSW_BLK_R(10) "\n\t"                     /* Visited.mem */               \
        "v_cmp_eq_u32_e64 %[um3], " HNSW_BLK_R(14) ", %[tag]\n\t"                /* the slot carries my block */ \
        "s_and_b64 %[um3], %[um3], vcc\n\t"                                      /* ... and I hold a neighbour: owner */ \
        "s_andn2_b64 %[um0], %[um3], %[um1]\n\t"                                 /* owners that missed */        \
        "s_mov_b64 exec, %[um0]\n\t"                                                                            \
        HNSW_BLK_CLEAR(HNSW_BLK_R(13))                                           /* 4 */                         \
        "s_andn2_b64 %[um0], %[um3], %[um2]\n\t"                                 /* owners whose bit was clear */ \
        "s_mov_b64 exec, %[um0]\n\t"                                                                            \
        "ds_or_b32 " HNSW_BLK_R(12) ", " HNSW_BLK_R(11) "\n\t"                   /* 5: Visited.add */            \
        "s_mov_b64 exec, -1\n\t"                                                                                \
        "s_andn2_b64 %[fresh], vcc, %[um2]\n\t"                                                                 \
        "s_cbranch_scc0 1b\n\t"  /* nothing fresh: next hop */                                                  \
  /* the list of fresh neighbours, in row order */                                                              \
        "s_bcnt1_i32_b64 %[cnt], %[fresh]\n\t"                                                                  \
        "s_mov_b64 exec, %[fresh]\n\t"                                                                          \
        "v_mbcnt_lo_u32_b32 %[t0], exec_lo, 0\n\t"                                                              \
        "v_mbcnt_hi_u32_b32 %[t0], exec_hi, %[t0]\n\t"                                                          \
        "v_lshl_add_u32 %[t0], %[t0], 2, %[cand]\n\t"                                                           \
        "ds_write_b32 %[t0], %[nb]\n\t"                                                                         \
        HNSW_SPLIT_COMPACT                                                                                      \
        "s_mov_b64 exec, -1\n\t"                                                                                \
        "s_add_u32 %[nd], %[nd], %[cnt]\n\t"                                                                    \
        "s_lshl2_add_u32 %[lastad], %[cnt], %[candm4]\n\t"                                                      \
        "s_mov_b32 %[sx], %[cand]\n"

// a list longer than one round (rare): the next candidate's address from what is left, then the next round
#define HNSW_HOP_NEXT_ROUND \
        "s_lshl_b32 %[tmp], %[cnt], 2\n\t"         \
        "s_sub_u32 %[sx], %[lastad], %[tmp]\n\t"   \
        "s_add_u32 %[sx], %[sx], 4\n\t"            \
        "s_branch 20b\n"

// One round of 4 / 8 / 16 rows -> ckey, cid, the accept mask in `fresh`, cnt reduced by the round's size, then the insertion
// (50).  Label 20 picks the shape; the 8-row round (5..8 candidates left: the usual case of an M = 16 graph) follows in
// line and falls into 50, the other two shapes (25, 40: HNSW_HOP_ROUNDS_RARE) sit behind the loop and branch back.
// Every shape in two halves: ISSUE (ids from LDS, row addresses, the loads) and CONSUME (dot products, the transposing
// reduction, keys, accept mask, cnt).  A round runs them back to back; a hop with MORE than one round (more than 16 fresh
// neighbours: four hops in ten on a hard set at ef 176) issues the next round's loads BEFORE the current round's insertions
// and consumes them behind -- HNSW_ASM_RPIPE, round 6: the next round's memory round trip (400 cycles idle, 1000 loaded) runs
// under the 1000-1500 cycles of insertions instead of behind them.  Nothing the insertions touch is live in the issued half
// (rows d0..d7, ids id0..id3, addresses ad0 / ad1; the select into cid belongs to CONSUME: cid still holds the current
// round's ids), cnt does not change in between (the shape picked at issue is the shape consumed), evaluation and accept
// order are those of the unpipelined loop: same bits, same counters.
#ifndef HNSW_ASM_RPIPE
#define HNSW_ASM_RPIPE 1
#endif
#define HNSW_B8_ISSUE_8                                                                                                               \
        HNSW_ID_READ0("%[id0]", 1)                                                                                                  \
        HNSW_ID_READN("%[id1]")                                                                                                  \
        "s_waitcnt lgkmcnt(1)\n\t"                                                                                                    \
        HNSW_ROW_LOAD("%[id0]", "%[ad0]", "%[d0]", "%[d1]")                                                                           \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                                    \
        HNSW_ROW_LOAD("%[id1]", "%[ad1]", "%[d2]", "%[d3]")
#define HNSW_B8_CONSUME_8                                                                                                             \
        "v_cndmask_b32_e64 %[cid], %[id0], %[id1], %[b3m]\n\t"                                                                        \
        "s_waitcnt vmcnt(2)\n\t"                                                                                                      \
        HNSW_B8_DOTS("%[d0]", "%[d1]", "%[ta]")                                                                                          \
        "s_waitcnt vmcnt(0)\n\t"                                                                                                      \
        HNSW_B8_DOTS("%[d2]", "%[d3]", "%[tb]")                                                                                          \
        HNSW_B8_COMBINE("%[d0]", "%[ta]")                                                                                                \
        "s_nop 2\n\t"                                                                                                                 \
        HNSW_B8_COMBINE("%[d2]", "%[tb]")                                                                                                \
        "v_cndmask_b32_e64 %[ta], %[d0], %[d2], %[b3m]\n\t"  /* keep: the sum this half of the group is for */                        \
        "v_cndmask_b32_e64 %[tb], %[d2], %[d0], %[b3m]\n\t"  /* give: the other half's */                                             \
        HNSW_ACCEPT_EARLY(1)                                                                                                               \
        "v_add_u32_dpp %[ta], %[tb], %[ta] row_ror:8" HNSW_DPP_BC "\n\t"                                                              \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[ta], %[ta], %[ta] row_half_mirror" HNSW_DPP_BC "\n\t"                                                        \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[ta], %[ta], %[ta] quad_perm:[1,0,3,2]" HNSW_DPP_BC "\n\t"                                                    \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[ta], %[ta], %[ta] quad_perm:[2,3,0,1]" HNSW_DPP_BC "\n\t"                                                    \
        HNSW_B8_KEY("%[ta]")                                                                                        \
        HNSW_ACCEPT_LATE                                                                                                              \
        "s_sub_u32 %[cnt], %[cnt], 8\n"
#define HNSW_B8_ISSUE_4                                                                                                               \
        HNSW_ID_READ0("%[id0]", 0)                                                                                                  \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                                    \
        HNSW_ROW_LOAD("%[id0]", "%[ad0]", "%[d0]", "%[d1]")
#define HNSW_B8_CONSUME_4                                                                                                             \
        "s_waitcnt vmcnt(0)\n\t"                                                                                                      \
        HNSW_B8_DOTS("%[d0]", "%[d1]", "%[ta]")                                                                                          \
        "v_mov_b32_e32 %[cid], %[id0]\n\t"                                                                                            \
        "s_nop 2\n\t"                                                                                                                 \
        HNSW_B8_COMBINE("%[d0]", "%[ta]")                                                                                                \
        HNSW_ACCEPT_EARLY(0)                                                                                                   \
        "v_add_u32_dpp %[d0], %[d0], %[d0] row_ror:8" HNSW_DPP_BC "\n\t"                                                              \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[d0], %[d0], %[d0] row_ror:4" HNSW_DPP_BC "\n\t"                                                              \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[d0], %[d0], %[d0] row_ror:2" HNSW_DPP_BC "\n\t"                                                              \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[d0], %[d0], %[d0] row_ror:1" HNSW_DPP_BC "\n\t"                                                              \
        HNSW_B8_KEY("%[d0]")                                                                                        \
        HNSW_ACCEPT_LATE                                                                                                              \
        "s_sub_u32 %[cnt], %[cnt], 4\n\t"
#define HNSW_B8_ISSUE_16                                                                                                              \
        HNSW_ID_READ0("%[id0]", 2)                                                                                                  \
        HNSW_ID_READN("%[id1]")                                                                                                  \
        HNSW_ID_READN("%[id2]")                                                                                                  \
        HNSW_ID_READN("%[id3]")                                                                                                  \
        "s_waitcnt lgkmcnt(3)\n\t"                                                                                                    \
        HNSW_ROW_LOAD("%[id0]", "%[ad0]", "%[d0]", "%[d1]")                                                                           \
        "s_waitcnt lgkmcnt(2)\n\t"                                                                                                    \
        HNSW_ROW_LOAD("%[id1]", "%[ad1]", "%[d2]", "%[d3]")                                                                           \
        "s_waitcnt lgkmcnt(1)\n\t"                                                                                                    \
        HNSW_ROW_LOAD("%[id2]", "%[ad0]", "%[d4]", "%[d5]")                                                                           \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                                    \
        HNSW_ROW_LOAD("%[id3]", "%[ad1]", "%[d6]", "%[d7]")
#define HNSW_B8_CONSUME_16                                                                                                            \
        "v_cndmask_b32_e64 %[id0], %[id0], %[id1], %[b2m]\n\t"                                                                        \
        "v_cndmask_b32_e64 %[id2], %[id2], %[id3], %[b2m]\n\t"                                                                        \
        "v_cndmask_b32_e64 %[cid], %[id0], %[id2], %[b3m]\n\t"                                                                        \
        "s_waitcnt vmcnt(6)\n\t"                                                                                                      \
        HNSW_B8_DOTS("%[d0]", "%[d1]", "%[ta]")                                                                                          \
        "s_waitcnt vmcnt(4)\n\t"                                                                                                      \
        HNSW_B8_DOTS("%[d2]", "%[d3]", "%[tb]")                                                                                          \
        "s_waitcnt vmcnt(2)\n\t"                                                                                                      \
        HNSW_B8_COMBINE("%[d0]", "%[ta]")                                                                                                \
        HNSW_B8_DOTS("%[d4]", "%[d5]", "%[ta]")                                                                                          \
        "s_waitcnt vmcnt(0)\n\t"                                                                                                      \
        HNSW_B8_COMBINE("%[d2]", "%[tb]")                                                                                                \
        HNSW_B8_DOTS("%[d6]", "%[d7]", "%[tb]")                                                                                          \
        HNSW_B8_COMBINE("%[d4]", "%[ta]")                                                                                                \
        "s_nop 2\n\t"                                                                                                                 \
        HNSW_B8_COMBINE("%[d6]", "%[tb]")                                                                                                \
  /* sums of candidates 0..3 of the group in d0, d2, d4, d6 -> quads of the group's 16 lanes */                                       \
        "v_cndmask_b32_e64 %[ta], %[d0], %[d4], %[b3m]\n\t"                                                                           \
        "v_cndmask_b32_e64 %[d1], %[d4], %[d0], %[b3m]\n\t"                                                                           \
        "v_cndmask_b32_e64 %[tb], %[d2], %[d6], %[b3m]\n\t"                                                                           \
        "v_cndmask_b32_e64 %[d3], %[d6], %[d2], %[b3m]\n\t"                                                                           \
        "v_add_u32_dpp %[ta], %[d1], %[ta] row_ror:8" HNSW_DPP_BC "\n\t"  /* candidates 0 | 2 (d1 written three instructions ago) */  \
        "s_nop 0\n\t"                                                                                                                 \
        "v_add_u32_dpp %[tb], %[d3], %[tb] row_ror:8" HNSW_DPP_BC "\n\t"  /* candidates 1 | 3 */                                      \
        "v_cndmask_b32_e64 %[d0], %[ta], %[tb], %[b2m]\n\t"                                                                           \
        "v_cndmask_b32_e64 %[d1], %[tb], %[ta], %[b2m]\n\t"                                                                           \
        HNSW_ACCEPT_EARLY(2)                                                                                                                \
        "v_add_u32_dpp %[d0], %[d1], %[d0] row_half_mirror" HNSW_DPP_BC "\n\t"                                                        \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[d0], %[d0], %[d0] quad_perm:[1,0,3,2]" HNSW_DPP_BC "\n\t"                                                    \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[d0], %[d0], %[d0] quad_perm:[2,3,0,1]" HNSW_DPP_BC "\n\t"                                                    \
        HNSW_B8_KEY("%[d0]")                                                                                        \
        HNSW_ACCEPT_LATE                                                                                                              \
        "s_sub_u32 %[cnt], %[cnt], 16\n\t"

#define HNSW_HOP_ROUND_SELECT \
        "20:\n\t"                                                                                                                     \
        "s_cmp_gt_u32 %[cnt], 8\n\t"                                                                                                  \
        "s_cbranch_scc1 40f\n\t"                                                                                                      \
        "s_cmp_lt_u32 %[cnt], 5\n\t"                                                                                                  \
        "s_cbranch_scc1 25f\n\t"
// behind the insertions of a round: nothing left -> next hop (1b); else the next round
#define HNSW_HOP_AFTER_INSERT_PLAIN \
        "s_cmp_gt_i32 %[cnt], 0\n\t"  \
        "s_cbranch_scc0 1b\n\t"       \
        HNSW_HOP_NEXT_ROUND
#if HNSW_ASM_RPIPE
// (an 8-row or a 4-row round never leaves candidates behind: 5..8 / 1..4 were left when it was picked; only the 16-row round
// does, and it sits behind the loop: the in-line path is the unpipelined text)
#define HNSW_HOP_ROUND_COMMON \
        HNSW_HOP_ROUND_SELECT                                                                                                         \
  /* ---- 8 rows: two batches */                                                                                                      \
        "30:\n\t"                                                                                                                     \
        HNSW_B8_ISSUE_8                                                                                                               \
        "31:\n\t"                                                                                                                     \
        HNSW_B8_CONSUME_8
// ... the next round's loads are in flight (45): the shape they were issued for, from the unchanged cnt
#define HNSW_HOP_AFTER_INSERT \
        "s_cmp_gt_i32 %[cnt], 0\n\t"  \
        "s_cbranch_scc0 1b\n\t"       \
        "s_cmp_gt_u32 %[cnt], 8\n\t"  \
        "s_cbranch_scc1 41f\n\t"      \
        "s_cmp_lt_u32 %[cnt], 5\n\t"  \
        "s_cbranch_scc1 26f\n\t"      \
        "s_branch 31b\n"
#define HNSW_HOP_ROUNDS_RARE \
        "25:\n\t"                                                                                                                     \
  /* ---- 4 rows: one batch */                                                                                                        \
        HNSW_B8_ISSUE_4                                                                                                               \
        "26:\n\t"                                                                                                                     \
        HNSW_B8_CONSUME_4                                                                                                             \
        "s_branch 50b\n"                                                                                                              \
  /* ---- 16 rows: four batches (a list of 9..12 re-reads its last row in the groups past the end) */                                 \
        "40:\n\t"                                                                                                                     \
        HNSW_B8_ISSUE_16                                                                                                              \
        "41:\n\t"                                                                                                                     \
        HNSW_B8_CONSUME_16                                                                                                            \
        "s_cmp_gt_i32 %[cnt], 0\n\t"                                                                                                  \
        "s_cbranch_scc0 50b\n\t"                                                                                                      \
  /* ---- more than 16 fresh neighbours: the next round's loads go out now, its sums are taken behind this round's insertions */     \
        "s_lshl_b32 %[tmp], %[cnt], 2\n\t"                                                                                            \
        "s_sub_u32 %[sx], %[lastad], %[tmp]\n\t"                                                                                      \
        "s_add_u32 %[sx], %[sx], 4\n\t"                                                                                               \
        "s_cmp_gt_u32 %[cnt], 8\n\t"                                                                                                  \
        "s_cbranch_scc1 46f\n\t"                                                                                                      \
        "s_cmp_lt_u32 %[cnt], 5\n\t"                                                                                                  \
        "s_cbranch_scc1 47f\n\t"                                                                                                      \
        HNSW_B8_ISSUE_8                                                                                                               \
        "s_branch 50b\n"                                                                                                              \
        "46:\n\t"                                                                                                                     \
        HNSW_B8_ISSUE_16                                                                                                              \
        "s_branch 50b\n"                                                                                                              \
        "47:\n\t"                                                                                                                     \
        HNSW_B8_ISSUE_4                                                                                                               \
        "s_branch 50b\n"
#else
#define HNSW_HOP_ROUND_COMMON \
        HNSW_HOP_ROUND_SELECT                                                                                                         \
  /* ---- 8 rows: two batches */                                                                                                      \
        "30:\n\t"                                                                                                                     \
        HNSW_B8_ISSUE_8                                                                                                               \
        HNSW_B8_CONSUME_8
#define HNSW_HOP_AFTER_INSERT HNSW_HOP_AFTER_INSERT_PLAIN
#define HNSW_HOP_ROUNDS_RARE \
        "25:\n\t"                                                                                                                     \
  /* ---- 4 rows: one batch */                                                                                                        \
        HNSW_B8_ISSUE_4                                                                                                               \
        HNSW_B8_CONSUME_4                                                                                                             \
        "s_branch 50b\n"                                                                                                              \
  /* ---- 16 rows: four batches (a list of 9..12 re-reads its last row in the groups past the end) */                                 \
        "40:\n\t"                                                                                                                     \
        HNSW_B8_ISSUE_16                                                                                                              \
        HNSW_B8_CONSUME_16                                                                                                            \
        "s_branch 50b\n"
#endif

// ---- the rounds of byte rows of 129..256 dimensions (NCH = 4: four dwords per lane and row at +0, +64, +128, +192; batch b in
// d[4b .. 4b+3], its sum in d[4b]): the text above with twice the loads and dot products per batch (derived from it mechanically)
#define HNSW_ROW_LOAD_N4(ID, AD, DA, DB, DC, DD)                                         \
    "v_mad_u64_u32 " AD ", vcc, " ID ", %[st8], %[xl]\n\t"                               \
    "global_load_dword " DA ", " AD ", off\n\t"                                          \
    "global_load_dword " DB ", " AD ", off offset:64\n\t"                                \
    "global_load_dword " DC ", " AD ", off offset:128\n\t"                               \
    "global_load_dword " DD ", " AD ", off offset:192\n\t"
#define HNSW_DOTS_N4(DA, DB, DC, DD, TA)                                                 \
    "v_dot4_u32_u8 " TA ", " DA ", %[qb0], 0\n\t"                                        \
    "v_dot4_u32_u8 " DA ", " DA ", " DA ", %[q2v]\n\t"                                   \
    "v_dot4_u32_u8 " TA ", " DB ", %[qb1], " TA "\n\t"                                   \
    "v_dot4_u32_u8 " DA ", " DB ", " DB ", " DA "\n\t"                                   \
    "v_dot4_u32_u8 " TA ", " DC ", %[qb2], " TA "\n\t"                                   \
    "v_dot4_u32_u8 " DA ", " DC ", " DC ", " DA "\n\t"                                   \
    "v_dot4_u32_u8 " TA ", " DD ", %[qb3], " TA "\n\t"                                   \
    "v_dot4_u32_u8 " DA ", " DD ", " DD ", " DA "\n\t"
#define HNSW_DOTS_IP_N4(DA, DB, DC, DD, TA)                                              \
    "v_dot4_u32_u8 " TA ", " DA ", %[qb0], 0\n\t"                                        \
    "v_dot4_u32_u8 " TA ", " DB ", %[qb1], " TA "\n\t"                                   \
    "v_dot4_u32_u8 " TA ", " DC ", %[qb2], " TA "\n\t"                                   \
    "v_dot4_u32_u8 " DA ", " DD ", %[qb3], " TA "\n\t"
#define HNSW_HOP_ROUND_COMMON_N4 \
        "20:\n\t"                                                                                                                     \
        "s_cmp_gt_u32 %[cnt], 8\n\t"                                                                                                  \
        "s_cbranch_scc1 40f\n\t"                                                                                                      \
        "s_cmp_lt_u32 %[cnt], 5\n\t"                                                                                                  \
        "s_cbranch_scc1 25f\n\t"                                                                                                      \
  /* ---- 8 rows: two batches */                                                                                                      \
        "30:\n\t"                                                                                                                     \
        HNSW_ID_READ0("%[id0]", 1)                                                                                                  \
        HNSW_ID_READN("%[id1]")                                                                                                  \
        "s_waitcnt lgkmcnt(1)\n\t"                                                                                                    \
        HNSW_ROW_LOAD_N4("%[id0]", "%[ad0]", "%[d0]", "%[d1]", "%[d2]", "%[d3]")                                                                           \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                                    \
        HNSW_ROW_LOAD_N4("%[id1]", "%[ad1]", "%[d4]", "%[d5]", "%[d6]", "%[d7]")                                                                           \
        "v_cndmask_b32_e64 %[cid], %[id0], %[id1], %[b3m]\n\t"                                                                        \
        "s_waitcnt vmcnt(4)\n\t"                                                                                                      \
        HNSW_B8_DOTS_N4("%[d0]", "%[d1]", "%[d2]", "%[d3]", "%[ta]")                                                                                          \
        "s_waitcnt vmcnt(0)\n\t"                                                                                                      \
        HNSW_B8_DOTS_N4("%[d4]", "%[d5]", "%[d6]", "%[d7]", "%[tb]")                                                                                          \
        HNSW_B8_COMBINE("%[d0]", "%[ta]")                                                                                                \
        "s_nop 2\n\t"                                                                                                                 \
        HNSW_B8_COMBINE("%[d4]", "%[tb]")                                                                                                \
        "v_cndmask_b32_e64 %[ta], %[d0], %[d4], %[b3m]\n\t"  /* keep: the sum this half of the group is for */                        \
        "v_cndmask_b32_e64 %[tb], %[d4], %[d0], %[b3m]\n\t"  /* give: the other half's */                                             \
        HNSW_ACCEPT_EARLY(1)                                                                                                               \
        "v_add_u32_dpp %[ta], %[tb], %[ta] row_ror:8" HNSW_DPP_BC "\n\t"                                                              \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[ta], %[ta], %[ta] row_half_mirror" HNSW_DPP_BC "\n\t"                                                        \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[ta], %[ta], %[ta] quad_perm:[1,0,3,2]" HNSW_DPP_BC "\n\t"                                                    \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[ta], %[ta], %[ta] quad_perm:[2,3,0,1]" HNSW_DPP_BC "\n\t"                                                    \
        HNSW_B8_KEY("%[ta]")                                                                                        \
        HNSW_ACCEPT_LATE                                                                                                              \
        "s_sub_u32 %[cnt], %[cnt], 8\n"

#define HNSW_HOP_ROUNDS_RARE_N4 \
        "25:\n\t"                                                                                                                     \
  /* ---- 4 rows: one batch */                                                                                                        \
        HNSW_ID_READ0("%[id0]", 0)                                                                                                  \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                                    \
        HNSW_ROW_LOAD_N4("%[id0]", "%[ad0]", "%[d0]", "%[d1]", "%[d2]", "%[d3]")                                                                           \
        "s_waitcnt vmcnt(0)\n\t"                                                                                                      \
        HNSW_B8_DOTS_N4("%[d0]", "%[d1]", "%[d2]", "%[d3]", "%[ta]")                                                                                          \
        "v_mov_b32_e32 %[cid], %[id0]\n\t"                                                                                            \
        "s_nop 2\n\t"                                                                                                                 \
        HNSW_B8_COMBINE("%[d0]", "%[ta]")                                                                                                \
        HNSW_ACCEPT_EARLY(0)                                                                                                   \
        "v_add_u32_dpp %[d0], %[d0], %[d0] row_ror:8" HNSW_DPP_BC "\n\t"                                                              \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[d0], %[d0], %[d0] row_ror:4" HNSW_DPP_BC "\n\t"                                                              \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[d0], %[d0], %[d0] row_ror:2" HNSW_DPP_BC "\n\t"                                                              \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[d0], %[d0], %[d0] row_ror:1" HNSW_DPP_BC "\n\t"                                                              \
        HNSW_B8_KEY("%[d0]")                                                                                        \
        HNSW_ACCEPT_LATE                                                                                                              \
        "s_sub_u32 %[cnt], %[cnt], 4\n\t"                                                                                           \
        "s_branch 50b\n"                                                                                                              \
  /* ---- 16 rows: four batches (a list of 9..12 re-reads its last row in the groups past the end) */                                 \
        "40:\n\t"                                                                                                                     \
        HNSW_ID_READ0("%[id0]", 2)                                                                                                  \
        HNSW_ID_READN("%[id1]")                                                                                                  \
        HNSW_ID_READN("%[id2]")                                                                                                  \
        HNSW_ID_READN("%[id3]")                                                                                                  \
        "s_waitcnt lgkmcnt(3)\n\t"                                                                                                    \
        HNSW_ROW_LOAD_N4("%[id0]", "%[ad0]", "%[d0]", "%[d1]", "%[d2]", "%[d3]")                                                                           \
        "s_waitcnt lgkmcnt(2)\n\t"                                                                                                    \
        HNSW_ROW_LOAD_N4("%[id1]", "%[ad1]", "%[d4]", "%[d5]", "%[d6]", "%[d7]")                                                                           \
        "s_waitcnt lgkmcnt(1)\n\t"                                                                                                    \
        HNSW_ROW_LOAD_N4("%[id2]", "%[ad0]", "%[d8]", "%[d9]", "%[d10]", "%[d11]")                                                                           \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                                    \
        HNSW_ROW_LOAD_N4("%[id3]", "%[ad1]", "%[d12]", "%[d13]", "%[d14]", "%[d15]")                                                                           \
        "v_cndmask_b32_e64 %[id0], %[id0], %[id1], %[b2m]\n\t"                                                                        \
        "v_cndmask_b32_e64 %[id2], %[id2], %[id3], %[b2m]\n\t"                                                                        \
        "v_cndmask_b32_e64 %[cid], %[id0], %[id2], %[b3m]\n\t"                                                                        \
        "s_waitcnt vmcnt(12)\n\t"                                                                                                      \
        HNSW_B8_DOTS_N4("%[d0]", "%[d1]", "%[d2]", "%[d3]", "%[ta]")                                                                                          \
        "s_waitcnt vmcnt(8)\n\t"                                                                                                      \
        HNSW_B8_DOTS_N4("%[d4]", "%[d5]", "%[d6]", "%[d7]", "%[tb]")                                                                                          \
        "s_waitcnt vmcnt(4)\n\t"                                                                                                      \
        HNSW_B8_COMBINE("%[d0]", "%[ta]")                                                                                                \
        HNSW_B8_DOTS_N4("%[d8]", "%[d9]", "%[d10]", "%[d11]", "%[ta]")                                                                                          \
        "s_waitcnt vmcnt(0)\n\t"                                                                                                      \
        HNSW_B8_COMBINE("%[d4]", "%[tb]")                                                                                                \
        HNSW_B8_DOTS_N4("%[d12]", "%[d13]", "%[d14]", "%[d15]", "%[tb]")                                                                                          \
        HNSW_B8_COMBINE("%[d8]", "%[ta]")                                                                                                \
        "s_nop 2\n\t"                                                                                                                 \
        HNSW_B8_COMBINE("%[d12]", "%[tb]")                                                                                                \
  /* sums of candidates 0..3 of the group in d0, d2, d4, d6 -> quads of the group's 16 lanes */                                       \
        "v_cndmask_b32_e64 %[ta], %[d0], %[d8], %[b3m]\n\t"                                                                           \
        "v_cndmask_b32_e64 %[d1], %[d8], %[d0], %[b3m]\n\t"                                                                           \
        "v_cndmask_b32_e64 %[tb], %[d4], %[d12], %[b3m]\n\t"                                                                           \
        "v_cndmask_b32_e64 %[d5], %[d12], %[d4], %[b3m]\n\t"                                                                           \
        "v_add_u32_dpp %[ta], %[d1], %[ta] row_ror:8" HNSW_DPP_BC "\n\t"  /* candidates 0 | 2 (d1 written three instructions ago) */  \
        "s_nop 0\n\t"                                                                                                                 \
        "v_add_u32_dpp %[tb], %[d5], %[tb] row_ror:8" HNSW_DPP_BC "\n\t"  /* candidates 1 | 3 */                                      \
        "v_cndmask_b32_e64 %[d0], %[ta], %[tb], %[b2m]\n\t"                                                                           \
        "v_cndmask_b32_e64 %[d1], %[tb], %[ta], %[b2m]\n\t"                                                                           \
        HNSW_ACCEPT_EARLY(2)                                                                                                                \
        "v_add_u32_dpp %[d0], %[d1], %[d0] row_half_mirror" HNSW_DPP_BC "\n\t"                                                        \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[d0], %[d0], %[d0] quad_perm:[1,0,3,2]" HNSW_DPP_BC "\n\t"                                                    \
        "s_nop 1\n\t"                                                                                                                 \
        "v_add_u32_dpp %[d0], %[d0], %[d0] quad_perm:[2,3,0,1]" HNSW_DPP_BC "\n\t"                                                    \
        HNSW_B8_KEY("%[d0]")                                                                                        \
        HNSW_ACCEPT_LATE                                                                                                              \
        "s_sub_u32 %[cnt], %[cnt], 16\n\t"                                                                                            \
        "s_branch 50b\n"

// labels 90 / 99: no unexpanded member of W: entries evicted while tied with max(W) are still candidates (:568), else done
#define HNSW_HOP_TAIL \
        "90:\n\t"                                                                   \
        "s_cmp_eq_u32 %[oc], 0\n\t"                                                 \
        "s_cbranch_scc1 99f\n\t"                                                    \
        "s_cmp_lg_u32 %[od], %[wmax]\n\t"  /* a list from a larger max(W): dead */  \
        "s_cbranch_scc1 99f\n\t"                                                    \
        "s_sub_u32 %[oc], %[oc], 1\n\t"                                             \
        "s_lshl_b32 %[tmp], %[oc], 2\n\t"                                           \
        "s_add_u32 %[tmp], %[tmp], %[cand]\n\t"                                     \
        "v_mov_b32_e32 %[t0], %[tmp]\n\t"                                           \
        "ds_read_b32 %[t0], %[t0] offset:768\n\t"                                   \
        "s_waitcnt lgkmcnt(0)\n\t"                                                  \
        "v_readfirstlane_b32 %[kd], %[t0]\n\t"                                      \
        "s_add_u32 %[kd], %[kd], 1\n\t"  /* the list holds node ids */                \
        "s_mov_b64 %[um0], 0\n\t"                                                   \
        "s_mov_b64 %[um1], 0\n\t"                                                   \
        "s_branch 4b\n"                                                             \
        "99:\n\t"                                                                   \
        "s_waitcnt vmcnt(0)"  /* a speculative row fetch may still be in flight */  

// per-lane constants of the hop: the candidate a lane's sum belongs to in a round of 1 / 2 / 4 batches (bytes 0 / 1 / 2 of co; 0xff
// on the lanes that hold no candidate's sum), the lane-bit masks of the transposing reduction (D0..D2 are scratch here;
// Q2V: the byte rows' q.q accumulator start, while vcc still marks lane 0 of every 16)
#define HNSW_HOP_CONSTANTS_X(D0, D1, D2, Q2V) \
        "v_lshrrev_b32_e32 %[r4], 4, %[lane]\n\t"  /* r */                                             \
        "v_bfe_u32 %[t0], %[lane], 3, 1\n\t"                                                           \
        "v_lshl_add_u32 " D1 ", %[r4], 1, %[t0]\n\t"  /* NB 2: 2 r + bit 3 of the lane */               \
        "v_bfe_u32 %[t0], %[lane], 2, 2\n\t"                                                           \
        "v_lshl_add_u32 " D2 ", %[r4], 2, %[t0]\n\t"  /* NB 4: 4 r + bits 3:2 */                        \
        "v_mov_b32_e32 " D0 ", %[r4]\n\t"  /* NB 1: candidate r */                                      \
        "v_mov_b32_e32 %[t1], 0xff\n\t"  /* lanes that hold no candidate's sum: never below cnt */     \
        "v_and_b32_e32 %[t0], 15, %[lane]\n\t"                                                         \
        "v_cmp_eq_u32_e32 vcc, 0, %[t0]\n\t"                                                           \
        "s_nop 1\n\t"  /* gfx950: a vector write of vcc, then 2 wait states before a vector read */    \
        "v_cndmask_b32_e32 " D0 ", %[t1], " D0 ", vcc\n\t"                                               \
        Q2V                                                                                            \
        "v_and_b32_e32 %[t0], 7, %[lane]\n\t"                                                          \
        "v_cmp_eq_u32_e32 vcc, 0, %[t0]\n\t"                                                           \
        "s_nop 1\n\t"                                                                                  \
        "v_cndmask_b32_e32 " D1 ", %[t1], " D1 ", vcc\n\t"                                               \
        "v_and_b32_e32 %[t0], 3, %[lane]\n\t"                                                          \
        "v_cmp_eq_u32_e32 vcc, 0, %[t0]\n\t"                                                           \
        "s_nop 1\n\t"                                                                                  \
        "v_cndmask_b32_e32 " D2 ", %[t1], " D2 ", vcc\n\t"                                               \
        "v_lshl_or_b32 %[co], " D1 ", 8, " D0 "\n\t"  /* one byte per round shape */                       \
        "v_lshl_or_b32 %[co], " D2 ", 16, %[co]\n\t"                                                      \
        "v_lshlrev_b32_e32 %[r4], 2, %[r4]\n\t"  /* 4 r: byte offset of candidate r in the id list */  \
        "v_and_b32_e32 %[t0], 8, %[lane]\n\t"                                                          \
        "v_cmp_ne_u32_e64 %[b3m], 0, %[t0]\n\t"                                                        \
        "v_and_b32_e32 %[t0], 4, %[lane]\n\t"                                                          \
        "v_cmp_ne_u32_e64 %[b2m], 0, %[t0]\n\t"                                                        
#define HNSW_HOP_CONSTANTS HNSW_HOP_CONSTANTS_X("%[d0]", "%[d1]", "%[d2]",                                                  \
        "v_mov_b32_e32 %[q2v], %[q2]\n\t"  /* q.q, once per 16 lanes: the accumulator x.x starts from */                   \
        "v_cndmask_b32_e32 %[q2v], 0, %[q2v], vcc\n\t")

// labels 8 / 9: pref holds the low key half of the next nearest unexpanded member of W: its node, and its adjacency row
// requested into pnb beside this hop's vectors
#define HNSW_HOP_PREFETCH_LOAD \
        "8:\n\t"                                                  \
        "s_mul_i32 %[tmp], %[pref], %[rowb]\n\t"                  \
        "v_lshl_add_u32 %[t1], %[lane], 2, %[tmp]\n\t"            \
        "v_mov_b32_e32 %[pnb], -1\n\t"                            \
        "s_mov_b64 exec, %[rowm]\n\t"                             \
        "global_load_dword %[pnb], %[t1], %[nbrm]\n\t"             \
        "s_mov_b64 exec, -1\n"                                    \
        "9:\n\t"

// Runs the layer-0 search to completion.  On entry W holds the start node (unexpanded) and the visited cache knows it.

// =====================================================================================================================
// The same loop for W in FOUR key registers per lane (ef 129..256) and in ONE (ef <= 64).  Shared with the two-slot
// loop: everything between the pop and the insertion (HNSW_HOP_* above).  Different: where the flags are looked for
// (one mask per slot) and the insertion.
//
// Four slots: the rank is found in two steps, as in wlist_insert's NSLOT > 2 path: the slot first, from the slots'
// maxima (mx0..mx2 = the keys' distance halves in lane 63 of slots 0..2, kept in scalar registers: max(W) is slot 3's),
// then the position inside that slot (one compare + count).  The shift cascades: every slot above the rank's slot
// moves by one whole and takes its lower neighbour's last key in lane 0, the rank's slot moves from the rank on.
// A member of W at exactly the candidate's distance (ids decide; the run of equal keys may cross a slot boundary)
// takes the general path, which ranks over all four slots.
// =====================================================================================================================

// pop from slot S (key halves LS) behind the loop: label LBL; continues at NEXT when the slot has no unexpanded member; back at 3
#define HNSW_POP_SLOT(LBL, UM, LS, NEXT)                            \
    LBL ":\n\t"                                                     \
    "s_cmp_eq_u64 " UM ", 0\n\t"                                    \
    "s_cbranch_scc1 " NEXT "\n\t"                                   \
    "s_ff1_i32_b64 %[i], " UM "\n\t"                                \
    "v_readlane_b32 %[kd], " LS ", %[i]\n\t"                        \
    "s_bitset0_b64 " UM ", %[i]\n\t"                                \
    "s_mov_b32 m0, %[i]\n\t"                                        \
    "s_or_b32 %[t], %[kd], 0x80000000\n\t"                          \
    "v_writelane_b32 " LS ", %[t], m0\n\t"                          \
    "s_branch 3b\n"
// the lowest slot's pop, in line: falls into 3
#define HNSW_POP_SLOT0(UM, LS, NEXT)                                \
    "s_cmp_eq_u64 " UM ", 0\n\t"                                    \
    "s_cbranch_scc1 " NEXT "\n\t"                                   \
    "s_ff1_i32_b64 %[i], " UM "\n\t"                                \
    "v_readlane_b32 %[kd], " LS ", %[i]\n\t"                        \
    "s_bitset0_b64 " UM ", %[i]\n\t"                                \
    "s_mov_b32 m0, %[i]\n\t"                                        \
    "s_or_b32 %[t], %[kd], 0x80000000\n\t"                          \
    "v_writelane_b32 " LS ", %[t], m0\n"
// the same walk for the speculative row fetch: the first remaining unexpanded member's low half -> pref, then 8; none: 9
#define HNSW_PEEK_SLOT(LBL, UM, LS, NEXT)                           \
    LBL ":\n\t"                                                     \
    "s_cmp_eq_u64 " UM ", 0\n\t"                                    \
    "s_cbranch_scc1 " NEXT "\n\t"                                   \
    "s_ff1_i32_b64 %[i], " UM "\n\t"                                \
    "v_readlane_b32 %[pref], " LS ", %[i]\n\t"                      \
    "s_branch 8b\n"
#define HNSW_PEEK_SLOT0(UM, LS, NEXT)                               \
    "s_cmp_eq_u64 " UM ", 0\n\t"                                    \
    "s_cbranch_scc1 " NEXT "\n\t"                                   \
    "s_ff1_i32_b64 %[i], " UM "\n\t"                                \
    "v_readlane_b32 %[pref], " LS ", %[i]\n"
// general path, per slot: keys below by distance, plus the members at this distance with a smaller id; the node itself
// (flag bit either way) in W -> dup
#define HNSW_RANK_GENERAL_SLOT(HS, LS)                              \
    "v_cmp_gt_u32_e32 vcc, %[kd], " HS "\n\t"                       \
    "s_bcnt1_i32_b64 %[t], vcc\n\t"                                 \
    "s_add_u32 %[P], %[P], %[t]\n\t"                                \
    "v_cmp_eq_u32_e64 %[um0], %[kd], " HS "\n\t"                    \
    "v_and_b32_e32 %[t0], 0x7fffffff, " LS "\n\t"                   \
    "v_cmp_gt_u32_e32 vcc, %[klo], %[t0]\n\t"                       \
    "s_and_b64 vcc, vcc, %[um0]\n\t"                                \
    "s_bcnt1_i32_b64 %[t], vcc\n\t"                                 \
    "s_add_u32 %[P], %[P], %[t]\n\t"                                \
    "v_cmp_eq_u32_e32 vcc, %[klo], %[t0]\n\t"                       \
    "s_and_b64 vcc, vcc, %[um0]\n\t"                                \
    "s_or_b64 %[um1], %[um1], vcc\n\t"

// Four slots (ef 129..256).  The slot of the rank comes from the slots' maxima (scalar registers), then everything is
// per slot: tie check, position (one compare + count), the cascade (slots above move whole and take their lower
// neighbour's last key), the slot itself from the position on (v_cndmask_b32_dpp under an s_bfm mask, no EXEC writes).
// The top slot's way is in line; slots 2, 1, 0, the general rank for distance ties (14; it re-enters at the shifts 8s1
// with the full rank P) and the tie-list push are HNSW_INSERT_RARE4, behind the hop loop.
#define HNSW_INSERT_LOOP4                                                                                                   \
    "10:\n\t"                                                                                                               \
    "s_cmp_eq_u64 %[fresh], 0\n\t"                                                                                          \
    "s_cbranch_scc1 19f\n"                                                                                                  \
    "110:\n\t"                                                                                                              \
    "s_ff1_i32_b64 %[i], %[fresh]\n\t"                                                                                      \
    "v_readlane_b32 %[kd], %[ckey], %[i]\n\t"                                                                               \
    "v_readlane_b32 %[klo], %[cid], %[i]\n\t"                                                                               \
    "v_readlane_b32 %[nw], %[h3], 62\n\t"                                                                                   \
    "s_cmp_ge_u32 %[kd], %[wmax]\n\t"                                                                                       \
    "s_cbranch_scc1 " HNSW_SEM_REJECT "\n\t"                                              /* no longer below max(W): rejected, :574 */      \
    "s_cmp_lt_u32 %[mx2], %[kd]\n\t"                                      /* the rank's slot, from the slots' maxima */     \
    "s_cbranch_scc0 82f\n\t"                                                                                                \
    "v_cmp_eq_u32_e32 vcc, %[kd], %[h3]\n\t"                                                                                \
    "v_cmp_gt_u32_e64 %[um0], %[kd], %[h3]\n\t"                                                                             \
    "s_cmp_lg_u64 vcc, 0\n\t"                                                                                               \
    "s_cbranch_scc1 14f\n\t"                                              /* a member of W at this very distance */         \
    "s_bcnt1_i32_b64 m0, %[um0]\n"                                        /* rank inside the slot */                        \
    "s_max_u32 %[nw], %[nw], %[kd]\n\t"                                   /* the new max(W).d (this key if it ranks last) */ \
    "s_cmp_eq_u32 %[nw], %[wmax]\n\t"                                                                                       \
    "s_cbranch_scc1 153f\n"                                               /* the entry falling off ties with it */          \
    "831:\n\t"                                                                                                              \
    "s_bfm_b64 vcc, m0, 0\n\t"                                            /* lanes below m0 keep their keys */               \
    "v_cndmask_b32_dpp %[h3], %[h3], %[h3], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_cndmask_b32_dpp %[l3], %[l3], %[l3], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_writelane_b32 %[h3], %[kd], m0\n\t"                                                                                  \
    "v_writelane_b32 %[l3], %[klo], m0\n\t"                                                                                 \
    "s_mov_b32 %[wmax], %[nw]\n"                                                                                            \
    "18:\n\t"                                                                                                               \
    "s_bitset0_b64 %[fresh], %[i]\n\t"                                                                                      \
    "s_cmp_lg_u64 %[fresh], 0\n\t"                                                                                          \
    "s_cbranch_scc1 110b\n"                                               /* the next accepted candidate */                 \
    "19:\n\t"

#define HNSW_INSERT_RARE4                                                                                                   \
    "82:\n\t"                                                                                                               \
    "s_cmp_lt_u32 %[mx1], %[kd]\n\t"                                                                                        \
    "s_cbranch_scc0 81f\n\t"                                                                                                \
    "v_cmp_eq_u32_e32 vcc, %[kd], %[h2]\n\t"                                                                                \
    "v_cmp_gt_u32_e64 %[um0], %[kd], %[h2]\n\t"                                                                             \
    "s_cmp_lg_u64 vcc, 0\n\t"                                                                                               \
    "s_cbranch_scc1 14f\n\t"                                              /* a member of W at this very distance */         \
    "s_bcnt1_i32_b64 m0, %[um0]\n"                                        /* rank inside the slot */                        \
    "s_max_u32 %[nw], %[nw], %[kd]\n\t"                                   /* the new max(W).d (this key if it ranks last) */ \
    "s_cmp_eq_u32 %[nw], %[wmax]\n\t"                                                                                       \
    "s_cbranch_scc1 152f\n"                                               /* the entry falling off ties with it */          \
    "821:\n\t"                                                                                                              \
    "v_readlane_b32 %[sx], %[h2], 63\n\t"                                                                                   \
    "v_readlane_b32 %[tmp], %[l2], 63\n\t"                                                                                  \
    "v_mov_b32_dpp %[h3], %[h3] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_mov_b32_dpp %[l3], %[l3] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_writelane_b32 %[h3], %[sx], 0\n\t"                                                                                   \
    "v_writelane_b32 %[l3], %[tmp], 0\n\t"                                                                                  \
    "s_bfm_b64 vcc, m0, 0\n\t"                                            /* lanes below m0 keep their keys */               \
    "v_cndmask_b32_dpp %[h2], %[h2], %[h2], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_cndmask_b32_dpp %[l2], %[l2], %[l2], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_writelane_b32 %[h2], %[kd], m0\n\t"                                                                                  \
    "v_writelane_b32 %[l2], %[klo], m0\n\t"                                                                                 \
    "v_readlane_b32 %[mx2], %[h2], 63\n\t"                                                                                  \
    "s_mov_b32 %[wmax], %[nw]\n\t"                                                                                          \
    "s_bitset0_b64 %[fresh], %[i]\n\t"                                                                                      \
    "s_cmp_lg_u64 %[fresh], 0\n\t"                                                                                          \
    "s_cbranch_scc1 110b\n\t"                                                                                               \
    "s_branch 19b\n"                                                                                                        \
    "81:\n\t"                                                                                                               \
    "s_cmp_lt_u32 %[mx0], %[kd]\n\t"                                                                                        \
    "s_cbranch_scc0 80f\n\t"                                                                                                \
    "v_cmp_eq_u32_e32 vcc, %[kd], %[h1]\n\t"                                                                                \
    "v_cmp_gt_u32_e64 %[um0], %[kd], %[h1]\n\t"                                                                             \
    "s_cmp_lg_u64 vcc, 0\n\t"                                                                                               \
    "s_cbranch_scc1 14f\n\t"                                              /* a member of W at this very distance */         \
    "s_bcnt1_i32_b64 m0, %[um0]\n"                                        /* rank inside the slot */                        \
    "s_max_u32 %[nw], %[nw], %[kd]\n\t"                                   /* the new max(W).d (this key if it ranks last) */ \
    "s_cmp_eq_u32 %[nw], %[wmax]\n\t"                                                                                       \
    "s_cbranch_scc1 151f\n"                                               /* the entry falling off ties with it */          \
    "811:\n\t"                                                                                                              \
    "v_readlane_b32 %[sx], %[h2], 63\n\t"                                                                                   \
    "v_readlane_b32 %[tmp], %[l2], 63\n\t"                                                                                  \
    "v_mov_b32_dpp %[h3], %[h3] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_mov_b32_dpp %[l3], %[l3] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_writelane_b32 %[h3], %[sx], 0\n\t"                                                                                   \
    "v_writelane_b32 %[l3], %[tmp], 0\n\t"                                                                                  \
    "v_readlane_b32 %[sx], %[h1], 63\n\t"                                                                                   \
    "v_readlane_b32 %[tmp], %[l1], 63\n\t"                                                                                  \
    "v_mov_b32_dpp %[h2], %[h2] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_mov_b32_dpp %[l2], %[l2] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_writelane_b32 %[h2], %[sx], 0\n\t"                                                                                   \
    "v_writelane_b32 %[l2], %[tmp], 0\n\t"                                                                                  \
    "s_bfm_b64 vcc, m0, 0\n\t"                                            /* lanes below m0 keep their keys */               \
    "v_cndmask_b32_dpp %[h1], %[h1], %[h1], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_cndmask_b32_dpp %[l1], %[l1], %[l1], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_writelane_b32 %[h1], %[kd], m0\n\t"                                                                                  \
    "v_writelane_b32 %[l1], %[klo], m0\n\t"                                                                                 \
    "v_readlane_b32 %[mx1], %[h1], 63\n\t"                                                                                  \
    "v_readlane_b32 %[mx2], %[h2], 63\n\t"                                                                                  \
    "s_mov_b32 %[wmax], %[nw]\n\t"                                                                                          \
    "s_bitset0_b64 %[fresh], %[i]\n\t"                                                                                      \
    "s_cmp_lg_u64 %[fresh], 0\n\t"                                                                                          \
    "s_cbranch_scc1 110b\n\t"                                                                                               \
    "s_branch 19b\n"                                                                                                        \
    "80:\n\t"                                                                                                               \
    "v_cmp_eq_u32_e32 vcc, %[kd], %[h0]\n\t"                                                                                \
    "v_cmp_gt_u32_e64 %[um0], %[kd], %[h0]\n\t"                                                                             \
    "s_cmp_lg_u64 vcc, 0\n\t"                                                                                               \
    "s_cbranch_scc1 14f\n\t"                                              /* a member of W at this very distance */         \
    "s_bcnt1_i32_b64 m0, %[um0]\n"                                        /* rank inside the slot */                        \
    "s_max_u32 %[nw], %[nw], %[kd]\n\t"                                   /* the new max(W).d (this key if it ranks last) */ \
    "s_cmp_eq_u32 %[nw], %[wmax]\n\t"                                                                                       \
    "s_cbranch_scc1 150f\n"                                               /* the entry falling off ties with it */          \
    "801:\n\t"                                                                                                              \
    "v_readlane_b32 %[sx], %[h2], 63\n\t"                                                                                   \
    "v_readlane_b32 %[tmp], %[l2], 63\n\t"                                                                                  \
    "v_mov_b32_dpp %[h3], %[h3] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_mov_b32_dpp %[l3], %[l3] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_writelane_b32 %[h3], %[sx], 0\n\t"                                                                                   \
    "v_writelane_b32 %[l3], %[tmp], 0\n\t"                                                                                  \
    "v_readlane_b32 %[sx], %[h1], 63\n\t"                                                                                   \
    "v_readlane_b32 %[tmp], %[l1], 63\n\t"                                                                                  \
    "v_mov_b32_dpp %[h2], %[h2] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_mov_b32_dpp %[l2], %[l2] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_writelane_b32 %[h2], %[sx], 0\n\t"                                                                                   \
    "v_writelane_b32 %[l2], %[tmp], 0\n\t"                                                                                  \
    "v_readlane_b32 %[sx], %[h0], 63\n\t"                                                                                   \
    "v_readlane_b32 %[tmp], %[l0], 63\n\t"                                                                                  \
    "v_mov_b32_dpp %[h1], %[h1] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_mov_b32_dpp %[l1], %[l1] wave_shr:1" HNSW_DPP_ALL "\n\t"                                                             \
    "v_writelane_b32 %[h1], %[sx], 0\n\t"                                                                                   \
    "v_writelane_b32 %[l1], %[tmp], 0\n\t"                                                                                  \
    "s_bfm_b64 vcc, m0, 0\n\t"                                            /* lanes below m0 keep their keys */               \
    "v_cndmask_b32_dpp %[h0], %[h0], %[h0], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_cndmask_b32_dpp %[l0], %[l0], %[l0], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_writelane_b32 %[h0], %[kd], m0\n\t"                                                                                  \
    "v_writelane_b32 %[l0], %[klo], m0\n\t"                                                                                 \
    "v_readlane_b32 %[mx0], %[h0], 63\n\t"                                                                                  \
    "v_readlane_b32 %[mx1], %[h1], 63\n\t"                                                                                  \
    "v_readlane_b32 %[mx2], %[h2], 63\n\t"                                                                                  \
    "s_mov_b32 %[wmax], %[nw]\n\t"                                                                                          \
    "s_bitset0_b64 %[fresh], %[i]\n\t"                                                                                      \
    "s_cmp_lg_u64 %[fresh], 0\n\t"                                                                                          \
    "s_cbranch_scc1 110b\n\t"                                                                                               \
    "s_branch 19b\n"                                                                                                        \
    /* rare: a member of W at exactly this distance: rank over all four slots, ids decide; the node itself in W: ignored */ \
    "14:\n\t"                                                                                                               \
    "s_mov_b32 %[P], 0\n\t"                                                                                                 \
    "s_mov_b64 %[um1], 0\n\t"                                                                                               \
    HNSW_RANK_GENERAL_SLOT("%[h0]", "%[l0]")                                                                               \
    HNSW_RANK_GENERAL_SLOT("%[h1]", "%[l1]")                                                                               \
    HNSW_RANK_GENERAL_SLOT("%[h2]", "%[l2]")                                                                               \
    HNSW_RANK_GENERAL_SLOT("%[h3]", "%[l3]")                                                                               \
    "s_cmp_lg_u64 %[um1], 0\n\t"                                                                                            \
    "s_cbranch_scc1 18b\n\t"                                              /* already in W */                                \
    "s_max_u32 %[nw], %[nw], %[kd]\n\t"                                                                                     \
    "s_cmp_eq_u32 %[nw], %[wmax]\n\t"                                                                                       \
    "s_cbranch_scc1 154f\n"                                                                                                 \
    "141:\n\t"                                                                                                              \
    "s_and_b32 m0, %[P], 63\n\t"                                                                                            \
    "s_cmp_ge_u32 %[P], 192\n\t"                                                                                            \
    "s_cbranch_scc1 831b\n\t"                                                                                               \
    "s_cmp_ge_u32 %[P], 128\n\t"                                                                                            \
    "s_cbranch_scc1 821b\n\t"                                                                                               \
    "s_cmp_ge_u32 %[P], 64\n\t"                                                                                             \
    "s_cbranch_scc1 811b\n\t"                                                                                               \
    "s_branch 801b\n"                                                                                                       \
    /* rare: the entry that falls off is at the new maximum's distance (see the two-slot loop); one copy per way back */     \
    HNSW_EVICT_TIE("153", "831b", "%[l3]", "163")                                                                                                       \
    HNSW_EVICT_TIE("152", "821b", "%[l3]", "162")                                                                                                       \
    HNSW_EVICT_TIE("151", "811b", "%[l3]", "161")                                                                                                       \
    HNSW_EVICT_TIE("150", "801b", "%[l3]", "160")                                                                                                       \
    HNSW_EVICT_TIE("154", "141b", "%[l3]", "164")

// Three, six and eight slots (ef 129..192, 257..384, 385..512): the same structure, written by tools/gen_hop_slots.py (the generator
// reproduces the four-slot text above instruction for instruction; a test compares them), together with everything else in
// hnsw_hop_loop.inc that depends on the slot count (pop / peek chains, declarations, operand lists: HNSW_NSX_*_<N>)
#include "hnsw_hop_slots.inc"
#ifndef HNSW_ASM_ALIGN_PAD8
#define HNSW_ASM_ALIGN_PAD8 6
#endif
#ifndef HNSW_ASM_ALIGN_PAD3
#define HNSW_ASM_ALIGN_PAD3 6
#endif
#ifndef HNSW_ASM_ALIGN_PAD6
#define HNSW_ASM_ALIGN_PAD6 6
#endif


// ---- one slot (ef <= 64): no cascade; the same steps as the two-slot loop's upper slot --------------------------------
#define HNSW_INSERT_LOOP1                                                                                                   \
    "10:\n\t"                                                                                                               \
    "s_cmp_eq_u64 %[fresh], 0\n\t"                                                                                          \
    "s_cbranch_scc1 19f\n"                                                                                                  \
    "110:\n\t"                                                                                                              \
    "s_ff1_i32_b64 %[i], %[fresh]\n\t"                                                                                      \
    "v_readlane_b32 %[kd], %[ckey], %[i]\n\t"                                                                               \
    "v_readlane_b32 %[klo], %[cid], %[i]\n\t"                                                                               \
    "v_readlane_b32 %[nw], %[h0], 62\n\t"                                                                                   \
    "v_cmp_eq_u32_e64 %[um0], %[kd], %[h0]\n\t"                                                                             \
    "v_cmp_gt_u32_e32 vcc, %[kd], %[h0]\n\t"                                                                                \
    "s_cmp_ge_u32 %[kd], %[wmax]\n\t"                                                                                       \
    "s_cbranch_scc1 " HNSW_SEM_REJECT "\n\t"                                              /* no longer below max(W): rejected, :574 */      \
    "s_cmp_lg_u64 %[um0], 0\n\t"                                                                                            \
    "s_cbranch_scc1 14f\n\t"                                              /* members of W at this very distance */          \
    "s_bcnt1_i32_b64 m0, vcc\n"                                           /* rank = keys at a smaller distance */           \
    "11:\n\t"                                                                                                               \
    "s_max_u32 %[nw], %[nw], %[kd]\n\t"                                   /* the new max(W).d (this key if it ranks last) */ \
    "s_cmp_eq_u32 %[nw], %[wmax]\n\t"                                                                                       \
    "s_cbranch_scc1 15f\n"                                                /* the entry falling off ties with it */          \
    "12:\n\t"                                                                                                               \
    "s_bfm_b64 vcc, m0, 0\n\t"                                            /* lanes below m0 keep their keys */               \
    "v_cndmask_b32_dpp %[h0], %[h0], %[h0], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_cndmask_b32_dpp %[l0], %[l0], %[l0], vcc wave_shr:1" HNSW_DPP_ALL "\n\t"                                             \
    "v_writelane_b32 %[h0], %[kd], m0\n\t"                                                                                  \
    "v_writelane_b32 %[l0], %[klo], m0\n\t"                                                                                 \
    "s_mov_b32 %[wmax], %[nw]\n"                                                                                            \
    "18:\n\t"                                                                                                               \
    "s_bitset0_b64 %[fresh], %[i]\n\t"                                                                                      \
    "s_cmp_lg_u64 %[fresh], 0\n\t"                                                                                          \
    "s_cbranch_scc1 110b\n"                                               /* the next accepted candidate */                 \
    "19:\n\t"

#define HNSW_INSERT_RARE1                                                                                                   \
    "14:\n\t"                                                                                                               \
    "s_bcnt1_i32_b64 %[p], vcc\n\t"                                                                                         \
    "v_and_b32_e32 %[t0], 0x7fffffff, %[l0]\n\t"                                                                            \
    "v_cmp_gt_u32_e32 vcc, %[klo], %[t0]\n\t"                                                                               \
    "s_and_b64 vcc, vcc, %[um0]\n\t"                                                                                        \
    "s_bcnt1_i32_b64 %[t], vcc\n\t"                                                                                         \
    "s_add_u32 %[p], %[p], %[t]\n\t"                                                                                        \
    "v_cmp_eq_u32_e32 vcc, %[klo], %[t0]\n\t"                                                                               \
    "s_and_b64 vcc, vcc, %[um0]\n\t"                                                                                        \
    "s_cbranch_scc1 18b\n\t"                                              /* already in W */                                \
    "s_mov_b32 m0, %[p]\n\t"                                                                                               \
    "s_branch 11b\n"                                                                                                        \
    HNSW_EVICT_TIE("15", "12b", "%[l0]", "16")


// byte rows of 129..256 dimensions (NCH = 4)

// =====================================================================================================================
// The same loops over FLOAT32 rows of 65..128 dimensions (two float4 chunks per lane of a 16-lane group): the shape data
// that is not byte-valued takes (C2's general format, C3, C5).  Pop, adjacency, visited filter, compaction and insertion
// are the text above; the round differs: two global_load_dwordx4 per row and lane, the distance in hop_round's operation
// order -- per lane x, y, z, w of chunk 0, then of chunk 1 (v_sub_f32 + v_fmac_f32 for L2, v_fmac_f32 for the inner
// product), then the 16 lanes of the group summed pairwise at lane distance 8, 4, 2, 1 (reduce16) -- so every key is the
// one hop_round computes, bit for bit.  Float sums depend on their order, so the transposing reduction of the integer
// rounds (which pairs lanes by mirroring) is not available as it is: here a lane distance d always pairs lane l with
// lane l ^ d -- row_ror:8 for 8, TWO bank-masked v_add_f32_dpp for 4 (row_ror:4 into the lanes with bit 2 set, row_ror:12
// into the others: neither crosses into the other half of the row, which belongs to another candidate by then),
// quad_perm for 2 and 1 -- and the round's candidates are still folded into one register on the way (keep / give selects
// by lane bit 3, then bit 2), which leaves the keys exactly where the integer rounds leave them: the accept mask, the
// ids and the insertion do not know the difference.
// The 32 row registers are addressed BY NAME (v[HNSW_F32_BASE] ..: inline assembly cannot name a component of a 128-bit
// operand) and declared clobbered; chunk c, component k of batch b is HNSW_FX(b, c, k).
// Ragged rows (d not a multiple of 4 x 16 lanes: 17..31 chunks): the lanes whose second chunk lies past the row end are
// switched off (EXEC = cvm) around that chunk's load and arithmetic, which is hop_round's "the chunk's contribution is
// dropped whole".
// =====================================================================================================================
#define HNSW_F32_BASE 40
#define HNSW_F32_CLOBBER , "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
                           "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71"
// (HNSW_FX_BS: registers per batch, "8" for rows of two chunks per lane, "16" for four: set per instantiation in hnsw_hop_loop.inc)
#define HNSW_FX(B, C, K) "v[" HNSW_STR(HNSW_F32_BASE) "+" #B "*" HNSW_FX_BS "+" #C "*4+" #K "]"
#define HNSW_FX4(B, C) "v[" HNSW_STR(HNSW_F32_BASE) "+" #B "*" HNSW_FX_BS "+" #C "*4:" HNSW_STR(HNSW_F32_BASE) "+" #B "*" HNSW_FX_BS "+" #C "*4+3]"
#define HNSW_F32_CONSTANTS HNSW_HOP_CONSTANTS_X(HNSW_FX(0, 0, 0), HNSW_FX(0, 0, 1), HNSW_FX(0, 0, 2), "")
// row address (one 64-bit multiply-add) and the row's two float4 per lane
#define HNSW_F32_ROW_LOAD_PLAIN(ID, PJ, AD, B)                                           \
    "v_mad_u64_u32 " AD ", vcc, " ID ", %[st8], %[xl]\n\t"                               \
    "global_load_dwordx4 " HNSW_FX4(B, 0) ", " AD ", off\n\t"                            \
    HNSW_F32_RAG_ON                                                                      \
    "global_load_dwordx4 " HNSW_FX4(B, 1) ", " AD ", off offset:256\n\t"                 \
    HNSW_F32_RAG_OFF
// Split rows (hnsw_rows_split.hip: whole-line main rows in Xm, the last one or two chunks of node nbr0[c][j] beside slot (c, j)
// of the layer-0 adjacency): chunk 0 always lies in the main row; of chunk 1 the lanes below the main row's end read it at
// +256, the one or two tail lanes read the expanded node's tail row at the candidate's slot PJ (trow = that row's address per
// lane, less the 256 of the shared offset; computed once per hop, HNSW_SPLIT_HOP_ON), the rest are off.  The address pair is
// switched per lane between the two tables AFTER the first load has been issued (a load has read its address by then); the
// pairs are registers named by number (v32..v37: two alternating main pairs and the tail pair), because inline assembly cannot
// name the halves of a 64-bit operand.
#define HNSW_SPLIT_AD(B) "v[32+(" #B "&1)*2:33+(" #B "&1)*2]"
#define HNSW_SPLIT_AD_LO(B) "v[32+(" #B "&1)*2]"
#define HNSW_SPLIT_AD_HI(B) "v[33+(" #B "&1)*2]"
#define HNSW_F32_ROW_LOAD_SPLIT(ID, PJ, AD, B)                                           \
    "v_mad_u64_u32 " HNSW_SPLIT_AD(B) ", vcc, " ID ", %[st8], %[xl]\n\t"                 \
    "global_load_dwordx4 " HNSW_FX4(B, 0) ", " HNSW_SPLIT_AD(B) ", off\n\t"              \
    "v_mad_u64_u32 v[36:37], vcc, " PJ ", %[c16t], %[trow]\n\t"                          \
    "v_cndmask_b32_e64 " HNSW_SPLIT_AD_LO(B) ", " HNSW_SPLIT_AD_LO(B) ", v36, %[tlm]\n\t" \
    "v_cndmask_b32_e64 " HNSW_SPLIT_AD_HI(B) ", " HNSW_SPLIT_AD_HI(B) ", v37, %[tlm]\n\t" \
    "s_mov_b64 exec, %[cvm]\n\t"                                                         \
    "global_load_dwordx4 " HNSW_FX4(B, 1) ", " HNSW_SPLIT_AD(B) ", off offset:256\n\t"   \
    "s_mov_b64 exec, -1\n\t"
// ... split rows of 129..256 dimensions: chunks 0 and 1 always lie in the main row; the one or two tail chunks are the last of
// the row, so they fall into ONE of the chunk columns 2 and 3 -- tlm2 / tlm3 are the tail lanes of either (one of them empty);
// a lane may be a main lane of column 2 and a tail lane of column 3, so its address is switched in front of the column it is a
// tail lane of (trow already contains the lane's chunk offset less 512 or 768)
#define HNSW_F32_ROW_LOAD_N4_SPLIT(ID, PJ, AD, B)                                        \
    "v_mad_u64_u32 " HNSW_SPLIT_AD(B) ", vcc, " ID ", %[st8], %[xl]\n\t"                 \
    "global_load_dwordx4 " HNSW_FX4(B, 0) ", " HNSW_SPLIT_AD(B) ", off\n\t"              \
    "global_load_dwordx4 " HNSW_FX4(B, 1) ", " HNSW_SPLIT_AD(B) ", off offset:256\n\t"   \
    "v_mad_u64_u32 v[36:37], vcc, " PJ ", %[c16t], %[trow]\n\t"                          \
    "v_cndmask_b32_e64 " HNSW_SPLIT_AD_LO(B) ", " HNSW_SPLIT_AD_LO(B) ", v36, %[tlm2]\n\t" \
    "v_cndmask_b32_e64 " HNSW_SPLIT_AD_HI(B) ", " HNSW_SPLIT_AD_HI(B) ", v37, %[tlm2]\n\t" \
    "s_mov_b64 exec, %[cvm2]\n\t"                                                        \
    "global_load_dwordx4 " HNSW_FX4(B, 2) ", " HNSW_SPLIT_AD(B) ", off offset:512\n\t"   \
    "s_mov_b64 exec, -1\n\t"                                                             \
    "v_cndmask_b32_e64 " HNSW_SPLIT_AD_LO(B) ", " HNSW_SPLIT_AD_LO(B) ", v36, %[tlm3]\n\t" \
    "v_cndmask_b32_e64 " HNSW_SPLIT_AD_HI(B) ", " HNSW_SPLIT_AD_HI(B) ", v37, %[tlm3]\n\t" \
    "s_mov_b64 exec, %[cvm3]\n\t"                                                        \
    "global_load_dwordx4 " HNSW_FX4(B, 3) ", " HNSW_SPLIT_AD(B) ", off offset:768\n\t"   \
    "s_mov_b64 exec, -1\n\t"
#define HNSW_SPLIT_CLOBBER , "v32", "v33", "v34", "v35", "v36", "v37"
// the hop's node (kd = its id + 1) -> the address of its tail row per lane; kept for the hand-over of the functor rule
#define HNSW_SPLIT_HOP_ON                                                                \
    "v_mad_u64_u32 %[trow], vcc, %[kd], %[s0tv], %[tbase]\n\t"                           \
    "s_mov_b32 %[hopn], %[kd]\n\t"
#define HNSW_SPLIT_COMPACT_ON "ds_write_b32 %[t0], %[lane] offset:256\n\t"   /* cand_key[]: the candidate's slot in the node's row */
// ids (and, split rows, slots) of a round's batches; lgkmcnt values: one LDS read per batch, two with the slots
#define HNSW_F32_ID_READ0_PLAIN(ID, PJ, SH) HNSW_ID_READ0(ID, SH)
#define HNSW_F32_ID_READN_PLAIN(ID, PJ) HNSW_ID_READN(ID)
#define HNSW_F32_ID_READ0_SPLIT(ID, PJ, SH) HNSW_ID_READ0(ID, SH) "ds_read_b32 " PJ ", %[t0] offset:256\n\t"
#define HNSW_F32_ID_READN_SPLIT(ID, PJ) HNSW_ID_READN(ID) "ds_read_b32 " PJ ", %[t0] offset:256\n\t"
// ... and for rows of 129..256 dimensions (four float4 per lane: +0, +256, +512, +768; the third and the fourth chunk may lie
// past a ragged row's end: cvm2 / cvm3)
#define HNSW_F32_ROW_LOAD_N4(ID, PJ, AD, B)                                              \
    "v_mad_u64_u32 " AD ", vcc, " ID ", %[st8], %[xl]\n\t"                               \
    "global_load_dwordx4 " HNSW_FX4(B, 0) ", " AD ", off\n\t"                            \
    "global_load_dwordx4 " HNSW_FX4(B, 1) ", " AD ", off offset:256\n\t"                 \
    HNSW_F32_RAG2_ON                                                                     \
    "global_load_dwordx4 " HNSW_FX4(B, 2) ", " AD ", off offset:512\n\t"                 \
    HNSW_F32_RAG3_ON                                                                     \
    "global_load_dwordx4 " HNSW_FX4(B, 3) ", " AD ", off offset:768\n\t"                 \
    HNSW_F32_RAG_OFF
// the lane's share of batch B's distance -> HNSW_FX(B, 0, 0)
#define HNSW_F32_SUB(B, C, K, Q) "v_sub_f32_e32 " HNSW_FX(B, C, K) ", " HNSW_FX(B, C, K) ", " Q "\n\t"
#define HNSW_F32_SQ(B, C, K) "v_fmac_f32_e32 " HNSW_FX(B, 0, 0) ", " HNSW_FX(B, C, K) ", " HNSW_FX(B, C, K) "\n\t"
#define HNSW_F32_DIST_L2(B)                                                              \
    HNSW_F32_SUB(B, 0, 0, "%[qf0]") HNSW_F32_SUB(B, 0, 1, "%[qf1]") HNSW_F32_SUB(B, 0, 2, "%[qf2]") HNSW_F32_SUB(B, 0, 3, "%[qf3]") \
    "v_mul_f32_e32 " HNSW_FX(B, 0, 0) ", " HNSW_FX(B, 0, 0) ", " HNSW_FX(B, 0, 0) "\n\t" /* fma(dx, dx, +0): a square is never -0 */ \
    HNSW_F32_SQ(B, 0, 1) HNSW_F32_SQ(B, 0, 2) HNSW_F32_SQ(B, 0, 3)                       \
    HNSW_F32_RAG_ON                                                                      \
    HNSW_F32_SUB(B, 1, 0, "%[qf4]") HNSW_F32_SUB(B, 1, 1, "%[qf5]") HNSW_F32_SUB(B, 1, 2, "%[qf6]") HNSW_F32_SUB(B, 1, 3, "%[qf7]") \
    HNSW_F32_SQ(B, 1, 0) HNSW_F32_SQ(B, 1, 1) HNSW_F32_SQ(B, 1, 2) HNSW_F32_SQ(B, 1, 3)  \
    HNSW_F32_RAG_OFF
#define HNSW_F32_MAC(B, C, K, Q) "v_fmac_f32_e32 " HNSW_FX(B, 0, 0) ", " HNSW_FX(B, C, K) ", " Q "\n\t"
#define HNSW_F32_DIST_IP(B)                                                              \
    "v_fma_f32 " HNSW_FX(B, 0, 0) ", " HNSW_FX(B, 0, 0) ", %[qf0], 0\n\t"  /* (a product may be -0: the sum starts from +0) */ \
    HNSW_F32_MAC(B, 0, 1, "%[qf1]") HNSW_F32_MAC(B, 0, 2, "%[qf2]") HNSW_F32_MAC(B, 0, 3, "%[qf3]") \
    HNSW_F32_RAG_ON                                                                      \
    HNSW_F32_MAC(B, 1, 0, "%[qf4]") HNSW_F32_MAC(B, 1, 1, "%[qf5]") HNSW_F32_MAC(B, 1, 2, "%[qf6]") HNSW_F32_MAC(B, 1, 3, "%[qf7]") \
    HNSW_F32_RAG_OFF
#define HNSW_F32_DIST_L2_N4(B)                                                           \
    HNSW_F32_SUB(B, 0, 0, "%[qf0]") HNSW_F32_SUB(B, 0, 1, "%[qf1]") HNSW_F32_SUB(B, 0, 2, "%[qf2]") HNSW_F32_SUB(B, 0, 3, "%[qf3]") \
    "v_mul_f32_e32 " HNSW_FX(B, 0, 0) ", " HNSW_FX(B, 0, 0) ", " HNSW_FX(B, 0, 0) "\n\t" \
    HNSW_F32_SQ(B, 0, 1) HNSW_F32_SQ(B, 0, 2) HNSW_F32_SQ(B, 0, 3)                       \
    HNSW_F32_SUB(B, 1, 0, "%[qf4]") HNSW_F32_SUB(B, 1, 1, "%[qf5]") HNSW_F32_SUB(B, 1, 2, "%[qf6]") HNSW_F32_SUB(B, 1, 3, "%[qf7]") \
    HNSW_F32_SQ(B, 1, 0) HNSW_F32_SQ(B, 1, 1) HNSW_F32_SQ(B, 1, 2) HNSW_F32_SQ(B, 1, 3)  \
    HNSW_F32_RAG2_ON                                                                     \
    HNSW_F32_SUB(B, 2, 0, "%[qf8]") HNSW_F32_SUB(B, 2, 1, "%[qf9]") HNSW_F32_SUB(B, 2, 2, "%[qf10]") HNSW_F32_SUB(B, 2, 3, "%[qf11]") \
    HNSW_F32_SQ(B, 2, 0) HNSW_F32_SQ(B, 2, 1) HNSW_F32_SQ(B, 2, 2) HNSW_F32_SQ(B, 2, 3)  \
    HNSW_F32_RAG3_ON                                                                     \
    HNSW_F32_SUB(B, 3, 0, "%[qf12]") HNSW_F32_SUB(B, 3, 1, "%[qf13]") HNSW_F32_SUB(B, 3, 2, "%[qf14]") HNSW_F32_SUB(B, 3, 3, "%[qf15]") \
    HNSW_F32_SQ(B, 3, 0) HNSW_F32_SQ(B, 3, 1) HNSW_F32_SQ(B, 3, 2) HNSW_F32_SQ(B, 3, 3)  \
    HNSW_F32_RAG_OFF
#define HNSW_F32_DIST_IP_N4(B)                                                           \
    "v_fma_f32 " HNSW_FX(B, 0, 0) ", " HNSW_FX(B, 0, 0) ", %[qf0], 0\n\t"                \
    HNSW_F32_MAC(B, 0, 1, "%[qf1]") HNSW_F32_MAC(B, 0, 2, "%[qf2]") HNSW_F32_MAC(B, 0, 3, "%[qf3]") \
    HNSW_F32_MAC(B, 1, 0, "%[qf4]") HNSW_F32_MAC(B, 1, 1, "%[qf5]") HNSW_F32_MAC(B, 1, 2, "%[qf6]") HNSW_F32_MAC(B, 1, 3, "%[qf7]") \
    HNSW_F32_RAG2_ON                                                                     \
    HNSW_F32_MAC(B, 2, 0, "%[qf8]") HNSW_F32_MAC(B, 2, 1, "%[qf9]") HNSW_F32_MAC(B, 2, 2, "%[qf10]") HNSW_F32_MAC(B, 2, 3, "%[qf11]") \
    HNSW_F32_RAG3_ON                                                                     \
    HNSW_F32_MAC(B, 3, 0, "%[qf12]") HNSW_F32_MAC(B, 3, 1, "%[qf13]") HNSW_F32_MAC(B, 3, 2, "%[qf14]") HNSW_F32_MAC(B, 3, 3, "%[qf15]") \
    HNSW_F32_RAG_OFF
// lane l <- KEEP[l] + GIVE[l ^ 4] (DST may be KEEP; TMP is scratch).  Both rotations are computed for every lane and the lane's
// bit 2 picks: merging them under bank masks instead would make the second instruction read what the first has just written
// (gfx950 wants two wait states in front of ANY register a DPP instruction reads)
#define HNSW_F32_XOR4(DST, GIVE, KEEP, TMP)                                              \
    "v_add_f32_dpp " TMP ", " GIVE ", " KEEP " row_ror:12" HNSW_DPP_ALL "\n\t"  /* l <- l + 4: the lanes with bit 2 clear */ \
    "v_add_f32_dpp " DST ", " GIVE ", " KEEP " row_ror:4" HNSW_DPP_ALL "\n\t"   /* l <- l - 4: the lanes with bit 2 set */   \
    "v_cndmask_b32_e64 " DST ", " TMP ", " DST ", %[b2m]\n\t"
#define HNSW_ACCEPT_EARLY_A "v_add_u32_e32 %[cid], 1, %[cid]\n\t"           /* HNSW_ACCEPT_EARLY's two halves */
#define HNSW_ACCEPT_EARLY_B(SHAPE) "v_cmp_gt_u32_sdwa vcc, %[cnt], %[co] src0_sel:DWORD src1_sel:BYTE_" #SHAPE "\n\t"
// the group's sum -> key (dist_to_key): L2 the sum's bits (HNSW_F32_SUM is %[ckey] itself); inner product 1 - sum, sign-flipped
#define HNSW_F32_KEY_IP                                                                  \
    "v_sub_f32_e32 %[t1], 1.0, %[t1]\n\t"                                                \
    "v_ashrrev_i32_e32 %[t0], 31, %[t1]\n\t"                                             \
    "v_or_b32_e32 %[t0], 0x80000000, %[t0]\n\t"                                          \
    "v_xor_b32_e32 %[ckey], %[t1], %[t0]\n\t"

#define HNSW_F32_ROUND_PICK16                                                                                                         \
        "s_cmp_gt_u32 %[cnt], 8\n\t"                                                                                                  \
        "s_cbranch_scc1 40f\n\t"
// The float32 rounds in two halves, ISSUE (ids -- and, split rows, slots -- from LDS, row addresses, the loads) and CONSUME (distances, the
// folding reduction, keys, accept mask, cnt), as the byte-row rounds (HNSW_B8_ISSUE_* / HNSW_B8_CONSUME_*, HNSW_ASM_RPIPE): a hop of rows of
// 65..128 dimensions with more than 16 fresh neighbours -- C5's and C3's graphs have rows of 64 -- issues its next round's loads before the
// current round's insertions.  (The row registers are named and clobbered, the address pairs of the split rows too: nothing the insertions
// use.)  Rows of 129..256 dimensions keep the unpipelined flow (their rounds are 4 and 8 rows: HNSW_F32_ROUND_PICK is empty there).
#define HNSW_F32_ISSUE_8 \
        HNSW_F32_ID_READ0("%[id0]", "%[pj0]", 1) \
        HNSW_F32_ID_READN("%[id1]", "%[pj1]") \
        "s_waitcnt lgkmcnt(" HNSW_F32_LG1 ")\n\t" \
        HNSW_F32_ROW_LOAD("%[id0]", "%[pj0]", "%[ad0]", 0) \
        "s_waitcnt lgkmcnt(" HNSW_F32_LG0 ")\n\t" \
        HNSW_F32_ROW_LOAD("%[id1]", "%[pj1]", "%[ad1]", 1)
#define HNSW_F32_CONSUME_8 \
        "v_cndmask_b32_e64 %[cid], %[id0], %[id1], %[b3m]\n\t" \
        "s_waitcnt vmcnt(" HNSW_F32_VM1B ")\n\t" \
        HNSW_F32_DIST(0) \
        "s_waitcnt vmcnt(0)\n\t" \
        HNSW_F32_DIST(1) \
        "v_cndmask_b32_e64 %[ta], " HNSW_FX(0, 0, 0) ", " HNSW_FX(1, 0, 0) ", %[b3m]\n\t"  /* keep: the sum this half of the group is for */ \
        "v_cndmask_b32_e64 %[tb], " HNSW_FX(1, 0, 0) ", " HNSW_FX(0, 0, 0) ", %[b3m]\n\t"  /* give: the other half's */ \
        HNSW_ACCEPT_EARLY(1) \
        "v_add_f32_dpp %[ta], %[tb], %[ta] row_ror:8" HNSW_DPP_ALL "\n\t" \
        "s_nop 1\n\t" \
        HNSW_F32_XOR4("%[tb]", "%[ta]", "%[ta]", HNSW_FX(0, 0, 1)) \
        "s_nop 1\n\t" \
        "v_add_f32_dpp %[tb], %[tb], %[tb] quad_perm:[2,3,0,1]" HNSW_DPP_ALL "\n\t" \
        "s_nop 1\n\t" \
        "v_add_f32_dpp " HNSW_F32_SUM ", %[tb], %[tb] quad_perm:[1,0,3,2]" HNSW_DPP_ALL "\n\t" \
        HNSW_F32_KEY \
        HNSW_ACCEPT_LATE \
        "s_sub_u32 %[cnt], %[cnt], 8\n"
#define HNSW_F32_ISSUE_4 \
        HNSW_F32_ID_READ0("%[id0]", "%[pj0]", 0) \
        "s_waitcnt lgkmcnt(" HNSW_F32_LG0 ")\n\t" \
        HNSW_F32_ROW_LOAD("%[id0]", "%[pj0]", "%[ad0]", 0)
#define HNSW_F32_CONSUME_4 \
        "s_waitcnt vmcnt(0)\n\t" \
        HNSW_F32_DIST(0) \
        "v_mov_b32_e32 %[cid], %[id0]\n\t" \
        HNSW_ACCEPT_EARLY(0) \
        "v_add_f32_dpp %[ta], " HNSW_FX(0, 0, 0) ", " HNSW_FX(0, 0, 0) " row_ror:8" HNSW_DPP_ALL "\n\t" \
        "s_nop 1\n\t" \
        "v_add_f32_dpp %[ta], %[ta], %[ta] row_ror:4" HNSW_DPP_ALL "\n\t" \
        "s_nop 1\n\t" \
        "v_add_f32_dpp %[ta], %[ta], %[ta] row_ror:2" HNSW_DPP_ALL "\n\t" \
        "s_nop 1\n\t" \
        "v_add_f32_dpp " HNSW_F32_SUM ", %[ta], %[ta] row_ror:1" HNSW_DPP_ALL "\n\t" \
        HNSW_F32_KEY \
        HNSW_ACCEPT_LATE \
        "s_sub_u32 %[cnt], %[cnt], 4\n\t"
#define HNSW_F32_ISSUE_16 \
        HNSW_F32_ID_READ0("%[id0]", "%[pj0]", 2) \
        HNSW_F32_ID_READN("%[id1]", "%[pj1]") \
        HNSW_F32_ID_READN("%[id2]", "%[pj2]") \
        HNSW_F32_ID_READN("%[id3]", "%[pj3]") \
        "s_waitcnt lgkmcnt(" HNSW_F32_LG3 ")\n\t" \
        HNSW_F32_ROW_LOAD("%[id0]", "%[pj0]", "%[ad0]", 0) \
        "s_waitcnt lgkmcnt(" HNSW_F32_LG2 ")\n\t" \
        HNSW_F32_ROW_LOAD("%[id1]", "%[pj1]", "%[ad1]", 1) \
        "s_waitcnt lgkmcnt(" HNSW_F32_LG1 ")\n\t" \
        HNSW_F32_ROW_LOAD("%[id2]", "%[pj2]", "%[ad0]", 2) \
        "s_waitcnt lgkmcnt(" HNSW_F32_LG0 ")\n\t" \
        HNSW_F32_ROW_LOAD("%[id3]", "%[pj3]", "%[ad1]", 3)
#define HNSW_F32_CONSUME_16 \
        "v_cndmask_b32_e64 %[id0], %[id0], %[id1], %[b2m]\n\t" \
        "v_cndmask_b32_e64 %[id2], %[id2], %[id3], %[b2m]\n\t" \
        "v_cndmask_b32_e64 %[cid], %[id0], %[id2], %[b3m]\n\t" \
        "s_waitcnt vmcnt(6)\n\t" \
        HNSW_F32_DIST(0) \
        "s_waitcnt vmcnt(4)\n\t" \
        HNSW_F32_DIST(1) \
        "s_waitcnt vmcnt(2)\n\t" \
        HNSW_F32_DIST(2) \
        "s_waitcnt vmcnt(0)\n\t" \
        HNSW_F32_DIST(3) \
  /* sums of candidates 0..3 of the group in batch registers 0..3 -> quads of the group's 16 lanes (scratch: two row registers) */ \
        "v_cndmask_b32_e64 %[ta], " HNSW_FX(0, 0, 0) ", " HNSW_FX(2, 0, 0) ", %[b3m]\n\t"  /* keep 0 | 2 */ \
        "v_cndmask_b32_e64 " HNSW_FX(0, 0, 1) ", " HNSW_FX(2, 0, 0) ", " HNSW_FX(0, 0, 0) ", %[b3m]\n\t"  /* give */ \
        "v_cndmask_b32_e64 %[tb], " HNSW_FX(1, 0, 0) ", " HNSW_FX(3, 0, 0) ", %[b3m]\n\t"  /* keep 1 | 3 */ \
        "v_cndmask_b32_e64 " HNSW_FX(1, 0, 1) ", " HNSW_FX(3, 0, 0) ", " HNSW_FX(1, 0, 0) ", %[b3m]\n\t"  /* give */ \
        HNSW_ACCEPT_EARLY_A \
        "v_add_f32_dpp %[ta], " HNSW_FX(0, 0, 1) ", %[ta] row_ror:8" HNSW_DPP_ALL "\n\t" \
        "v_add_f32_dpp %[tb], " HNSW_FX(1, 0, 1) ", %[tb] row_ror:8" HNSW_DPP_ALL "\n\t" \
        "v_cndmask_b32_e64 " HNSW_FX(0, 0, 1) ", %[ta], %[tb], %[b2m]\n\t"  /* keep */ \
        "v_cndmask_b32_e64 " HNSW_FX(1, 0, 1) ", %[tb], %[ta], %[b2m]\n\t"  /* give */ \
        HNSW_ACCEPT_EARLY_B(2) \
        "s_nop 0\n\t" \
        HNSW_F32_XOR4(HNSW_FX(0, 0, 1), HNSW_FX(1, 0, 1), HNSW_FX(0, 0, 1), "%[ta]") \
        "s_nop 1\n\t" \
        "v_add_f32_dpp %[ta], " HNSW_FX(0, 0, 1) ", " HNSW_FX(0, 0, 1) " quad_perm:[2,3,0,1]" HNSW_DPP_ALL "\n\t" \
        "s_nop 1\n\t" \
        "v_add_f32_dpp " HNSW_F32_SUM ", %[ta], %[ta] quad_perm:[1,0,3,2]" HNSW_DPP_ALL "\n\t" \
        HNSW_F32_KEY \
        HNSW_ACCEPT_LATE \
        "s_sub_u32 %[cnt], %[cnt], 16\n\t"
#define HNSW_F32_ROUND_COMMON \
        "20:\n\t"                                                                                                                     \
        HNSW_F32_ROUND_PICK                                                                                                           \
        "s_cmp_lt_u32 %[cnt], 5\n\t"                                                                                                  \
        "s_cbranch_scc1 25f\n\t"                                                                                                      \
  /* ---- 8 rows: two batches */                                                                                                      \
        "30:\n\t"                                                                                                                     \
        HNSW_F32_ISSUE_8                                                                                                              \
        HNSW_F32_LABEL_31                                                                                                             \
        HNSW_F32_CONSUME_8

#define HNSW_F32_ROUND_4ROWS \
        "25:\n\t"                                                                                                                     \
  /* ---- 4 rows: one batch */                                                                                                        \
        HNSW_F32_ISSUE_4                                                                                                              \
        HNSW_F32_LABEL_26                                                                                                             \
        HNSW_F32_CONSUME_4                                                                                                            \
        "s_branch 50b\n"
#if HNSW_ASM_RPIPE
#define HNSW_F32_LABEL_31 "31:\n\t"
#define HNSW_F32_LABEL_26 "26:\n\t"
#define HNSW_F32_ROUND_16ROWS \
  /* ---- 16 rows: four batches (a list of 9..12 re-reads its last row in the groups past the end) */                                 \
        "40:\n\t"                                                                                                                     \
        HNSW_F32_ISSUE_16                                                                                                             \
        "41:\n\t"                                                                                                                     \
        HNSW_F32_CONSUME_16                                                                                                           \
        "s_cmp_gt_i32 %[cnt], 0\n\t"                                                                                                  \
        "s_cbranch_scc0 50b\n\t"                                                                                                      \
  /* ---- more than 16 fresh neighbours: the next round's loads go out now, its sums are taken behind this round's insertions */     \
        "s_lshl_b32 %[tmp], %[cnt], 2\n\t"                                                                                            \
        "s_sub_u32 %[sx], %[lastad], %[tmp]\n\t"                                                                                      \
        "s_add_u32 %[sx], %[sx], 4\n\t"                                                                                               \
        "s_cmp_gt_u32 %[cnt], 8\n\t"                                                                                                  \
        "s_cbranch_scc1 46f\n\t"                                                                                                      \
        "s_cmp_lt_u32 %[cnt], 5\n\t"                                                                                                  \
        "s_cbranch_scc1 47f\n\t"                                                                                                      \
        HNSW_F32_ISSUE_8                                                                                                              \
        "s_branch 50b\n"                                                                                                              \
        "46:\n\t"                                                                                                                     \
        HNSW_F32_ISSUE_16                                                                                                             \
        "s_branch 50b\n"                                                                                                              \
        "47:\n\t"                                                                                                                     \
        HNSW_F32_ISSUE_4                                                                                                              \
        "s_branch 50b\n"
#else
#define HNSW_F32_LABEL_31
#define HNSW_F32_LABEL_26
#define HNSW_F32_ROUND_16ROWS \
  /* ---- 16 rows: four batches (a list of 9..12 re-reads its last row in the groups past the end) */                                 \
        "40:\n\t"                                                                                                                     \
        HNSW_F32_ISSUE_16                                                                                                             \
        HNSW_F32_CONSUME_16                                                                                                           \
        "s_branch 50b\n"
#endif
#define HNSW_F32_ROUNDS_RARE HNSW_F32_ROUND_4ROWS HNSW_F32_ROUND_16ROWS


// =====================================================================================================================
// The instantiations: HopLoop<NCH, NSLOT, METRIC, ROWS, SEM, BLK>::run, one explicit specialisation per shape, each made by
// including hnsw_hop_loop.inc with the HNSW_LOOP_* macros set.  The table is GENERATED (tools/gen_hop_slots.py ->
// hnsw_hop_instances.inc): row families bytes (NCH 2) / bytes of 129..256 dimensions (NCH 4) / float32 full, ragged, split
// (NCH 2 and 4) x metric x accept rule x W in 1, 2, 3, 4, 6, 8 registers x visited structure (bitmap blocks: three or more
// registers, not the NCH 4 byte rows), under the feature switches at the top of this file, and in a translation unit of
// hnsw_search_variants.hip only the shapes of that unit's (metric, rule, row format).  search_layer asks
// HopLoop<...>::available and calls run(); a shape without an instantiation (the primary template) keeps the C++ loop.
// =====================================================================================================================
#include "hnsw_hop_instances.inc"

// =====================================================================================================================
// The descent through the upper layers (greedy_descend; Ohnsw.search_one, lib/ohnsw.ml:492-508) for the same shape:
// byte rows of 65..128 dimensions, byte-valued query, L2, upper rows of at most 16 neighbours.  One hop = the node's
// {row offset, level} by two scalar loads, its row (one dword per lane), the 16 vectors as ONE round of four batches --
// no compaction: candidate j is lane j of the row, a hole or padding (-1) is evaluated on row 0 and masked out -- then
// the nearest of the candidates strictly below the current key, the first in row order among equals (:502).  n_dist
// counts the valid neighbours of every row read, as the C++ loop does.  C2's 10 k batch: ordering pre-pass 52 -> 45 us,
// of which the descent below the top layer is 20 us, two launches and the sort the rest.  (Fetching every candidate's
// {row offset, level} together with its vector -- two dependent loads per hop instead of three -- measured no different.)
// =====================================================================================================================
#define HNSW_DESCENT_ROUND16 \
        "v_lshl_add_u32 %[t0], %[r4], 2, %[cand]\n\t"                              \
        "ds_read_b32 %[id0], %[t0]\n\t"                                            \
        "ds_read_b32 %[id1], %[t0] offset:4\n\t"                                   \
        "ds_read_b32 %[id2], %[t0] offset:8\n\t"                                   \
        "ds_read_b32 %[id3], %[t0] offset:12\n\t"                                  \
        "s_waitcnt lgkmcnt(3)\n\t"                                                 \
        "v_max_i32_e32 %[t1], 0, %[id0]\n\t"                                       \
        HNSW_ROW_LOAD("%[t1]", "%[ad0]", "%[d0]", "%[d1]")                         \
        "s_waitcnt lgkmcnt(2)\n\t"                                                 \
        "v_max_i32_e32 %[t1], 0, %[id1]\n\t"                                       \
        HNSW_ROW_LOAD("%[t1]", "%[ad1]", "%[d2]", "%[d3]")                         \
        "s_waitcnt lgkmcnt(1)\n\t"                                                 \
        "v_max_i32_e32 %[t1], 0, %[id2]\n\t"                                       \
        HNSW_ROW_LOAD("%[t1]", "%[ad0]", "%[d4]", "%[d5]")                         \
        "s_waitcnt lgkmcnt(0)\n\t"                                                 \
        "v_max_i32_e32 %[t1], 0, %[id3]\n\t"                                       \
        HNSW_ROW_LOAD("%[t1]", "%[ad1]", "%[d6]", "%[d7]")                         \
        "v_cndmask_b32_e64 %[id0], %[id0], %[id1], %[b2m]\n\t"                     \
        "v_cndmask_b32_e64 %[id2], %[id2], %[id3], %[b2m]\n\t"                     \
        "v_cndmask_b32_e64 %[cid], %[id0], %[id2], %[b3m]\n\t"                     \
        "s_waitcnt vmcnt(6)\n\t"                                                   \
        HNSW_DOTS("%[d0]", "%[d1]", "%[ta]")                                       \
        "s_waitcnt vmcnt(4)\n\t"                                                   \
        HNSW_DOTS("%[d2]", "%[d3]", "%[tb]")                                       \
        "s_waitcnt vmcnt(2)\n\t"                                                   \
        HNSW_COMBINE("%[d0]", "%[ta]")                                             \
        HNSW_DOTS("%[d4]", "%[d5]", "%[ta]")                                       \
        "s_waitcnt vmcnt(0)\n\t"                                                   \
        HNSW_COMBINE("%[d2]", "%[tb]")                                             \
        HNSW_DOTS("%[d6]", "%[d7]", "%[tb]")                                       \
        HNSW_COMBINE("%[d4]", "%[ta]")                                             \
        "s_nop 2\n\t"                                                              \
        HNSW_COMBINE("%[d6]", "%[tb]")                                             \
        "v_cndmask_b32_e64 %[ta], %[d0], %[d4], %[b3m]\n\t"                        \
        "v_cndmask_b32_e64 %[d1], %[d4], %[d0], %[b3m]\n\t"                        \
        "v_cndmask_b32_e64 %[tb], %[d2], %[d6], %[b3m]\n\t"                        \
        "v_cndmask_b32_e64 %[d3], %[d6], %[d2], %[b3m]\n\t"                        \
        "v_add_u32_dpp %[ta], %[d1], %[ta] row_ror:8" HNSW_DPP_BC "\n\t"           \
        "s_nop 0\n\t"                                                              \
        "v_add_u32_dpp %[tb], %[d3], %[tb] row_ror:8" HNSW_DPP_BC "\n\t"           \
        "v_cndmask_b32_e64 %[d0], %[ta], %[tb], %[b2m]\n\t"                        \
        "v_cndmask_b32_e64 %[d1], %[tb], %[ta], %[b2m]\n\t"                        \
        "v_cmp_lt_i32_e64 %[um0], -1, %[cid]\n\t"          /* a real neighbour */  \
        "v_cmp_gt_u32_sdwa %[um1], %[c16], %[co] src0_sel:DWORD src1_sel:BYTE_2\n\t"  /* a lane that holds a sum */ \
        "v_add_u32_dpp %[d0], %[d1], %[d0] row_half_mirror" HNSW_DPP_BC "\n\t"     \
        "s_nop 1\n\t"                                                              \
        "v_add_u32_dpp %[d0], %[d0], %[d0] quad_perm:[1,0,3,2]" HNSW_DPP_BC "\n\t" \
        "s_nop 1\n\t"                                                              \
        "v_add_u32_dpp %[d0], %[d0], %[d0] quad_perm:[2,3,0,1]" HNSW_DPP_BC "\n\t" \
        "v_cvt_f32_i32_e32 %[ckey], %[d0]\n\t"

__device__ __forceinline__ void greedy_descend_bytes_l2_asm(const IndexView &iv, int from, int to, int &cur_io, uint32_t &cur_key,
                                                            const WaveCtx &cx, uint32_t &n_dist) {
    const uint64_t xl = (uint64_t)(uintptr_t)iv.X8 + 4u * (uint32_t)cx.l16;
    const uint64_t nbrU = (uint64_t)(uintptr_t)iv.nbrU, uref = (uint64_t)(uintptr_t)iv.upper_ref;
    const uint64_t rowm = iv.SU >= 64 ? ~0ull : ((1ull << iv.SU) - 1ull);
    const uint32_t su4 = (uint32_t)iv.SU * 4u, st8 = (uint32_t)iv.stride8, cand = lds_offset(cx.cand_id);
    const uint32_t q2 = (uint32_t)uniform(cx.q2), c16 = 16u;
    uint32_t cur = (uint32_t)uniform(cur_io), kcur = (uint32_t)uniform((int)cur_key), nd = (uint32_t)uniform((int)n_dist);
    uint32_t layer = (uint32_t)uniform(from), lto = (uint32_t)uniform(to);
    uint32_t nb, r4, co, q2v, id0, id1, id2, id3, d0, d1, d2, d3, d4, d5, d6, d7, ta, tb, ckey, cid, t0, t1;
    uint64_t ad0, ad1;
    uint64_t um0, um1, fresh, b3m, b2m;
    uint32_t lm1, o8, off, lvl, cnt, i, kd, best, bi;
    asm volatile(
        HNSW_HOP_CONSTANTS
        "s_cmp_lt_i32 %[layer], %[lto]\n\t"
        "s_cbranch_scc1 9f\n"
        // ---- one layer
        "1:\n\t"
        "s_sub_u32 %[lm1], %[layer], 1\n"
        // ---- one hop: the node's rows start at off, it has lvl of them
        "2:\n\t"
        "s_lshl_b32 %[o8], %[cur], 3\n\t"
        "s_load_dword %[off], %[uref], %[o8]\n\t"
        "s_add_u32 %[o8], %[o8], 4\n\t"
        "s_load_dword %[lvl], %[uref], %[o8]\n\t"
        "v_mov_b32_e32 %[nb], -1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_cmp_gt_i32 %[layer], %[lvl]\n\t"                               // not on this layer (never in a consistent graph): an empty row
        "s_cbranch_scc1 8f\n\t"
        "s_add_u32 %[off], %[off], %[lm1]\n\t"
        "s_mul_i32 %[off], %[off], %[su4]\n\t"                            // byte offset of the row (the caller checked the table's size)
        "v_lshl_add_u32 %[t0], %[lane], 2, %[off]\n\t"
        "s_mov_b64 exec, %[rowm]\n\t"
        "global_load_dword %[nb], %[t0], %[nbrU]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "v_lshl_add_u32 %[t1], %[lane], 2, %[cand]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "ds_write_b32 %[t1], %[nb]\n\t"                                   // candidate j = lane j of the row
        "v_cmp_lt_i32_e32 vcc, -1, %[nb]\n\t"
        "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
        "s_cbranch_scc0 8f\n\t"                                           // no neighbour at all
        "s_add_u32 %[nd], %[nd], %[cnt]\n\t"
        HNSW_DESCENT_ROUND16
        "v_cmp_gt_u32_e32 vcc, %[kcur], %[ckey]\n\t"                      // strictly closer than the current node (:502)
        "s_and_b64 %[um0], %[um0], %[um1]\n\t"
        "s_and_b64 %[fresh], %[um0], vcc\n\t"
        "s_cbranch_scc0 8f\n\t"                                           // none: this layer is done
        "s_mov_b32 %[best], %[kcur]\n"
        "3:\n\t"                                                          // the nearest of them, the first in row order among equals
        "s_ff1_i32_b64 %[i], %[fresh]\n\t"
        "v_readlane_b32 %[kd], %[ckey], %[i]\n\t"
        "s_bitset0_b64 %[fresh], %[i]\n\t"
        "s_cmp_lt_u32 %[kd], %[best]\n\t"
        "s_cbranch_scc0 4f\n\t"
        "s_mov_b32 %[best], %[kd]\n\t"
        "s_mov_b32 %[bi], %[i]\n"
        "4:\n\t"
        "s_cmp_lg_u64 %[fresh], 0\n\t"
        "s_cbranch_scc1 3b\n\t"
        "v_readlane_b32 %[cur], %[cid], %[bi]\n\t"
        "s_mov_b32 %[kcur], %[best]\n\t"
        "s_branch 2b\n"
        "8:\n\t"                                                          // next layer down
        "s_sub_u32 %[layer], %[layer], 1\n\t"
        "s_cmp_ge_i32 %[layer], %[lto]\n\t"
        "s_cbranch_scc1 1b\n"
        "9:\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [cur] "+&s"(cur), [kcur] "+&s"(kcur), [nd] "+&s"(nd), [layer] "+&s"(layer),
          [nb] "=&v"(nb), [r4] "=&v"(r4), [co] "=&v"(co), [q2v] "=&v"(q2v),
          [id0] "=&v"(id0), [id1] "=&v"(id1), [id2] "=&v"(id2), [id3] "=&v"(id3),
          [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [d4] "=&v"(d4), [d5] "=&v"(d5), [d6] "=&v"(d6), [d7] "=&v"(d7),
          [ta] "=&v"(ta), [tb] "=&v"(tb), [ckey] "=&v"(ckey), [cid] "=&v"(cid), [t0] "=&v"(t0), [t1] "=&v"(t1),
          [ad0] "=&v"(ad0), [ad1] "=&v"(ad1),
          [um0] "=&s"(um0), [um1] "=&s"(um1), [fresh] "=&s"(fresh), [b3m] "=&s"(b3m), [b2m] "=&s"(b2m),
          [lm1] "=&s"(lm1), [o8] "=&s"(o8), [off] "=&s"(off), [lvl] "=&s"(lvl), [cnt] "=&s"(cnt), [i] "=&s"(i), [kd] "=&s"(kd),
          [best] "=&s"(best), [bi] "=&s"(bi)
        : [qb0] "v"(cx.qb[0]), [qb1] "v"(cx.qb[1]), [xl] "v"(xl), [lane] "v"(cx.lane),
          [nbrU] "s"(nbrU), [uref] "s"(uref), [rowm] "s"(rowm), [su4] "s"(su4), [st8] "s"(st8), [cand] "s"(cand), [q2] "s"(q2),
          [c16] "s"(c16), [lto] "s"(lto)
        : "vcc", "scc", "memory");
    cur_io = (int)cur; cur_key = kcur; n_dist = nd;
}

} // namespace hnsw_dev
