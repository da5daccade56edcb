// hnsw_device.hip.h -- gfx950 device code of the HNSW search path (one wavefront per query).
//
// Replaces the bodies of Ohnsw.search_one_simple / search_k / knn (lib/ohnsw.ml:492-508,
// 543-588, 859-875) and the distance stub (lib/ohnsw.ml:899, lib/hnsw.ml:814).
//
// Execution model
//   * one 64-lane wavefront = one query; a workgroup is one wave (no cross-wave sync anywhere);
//   * a vector row is read by a 16-lane group (= one DPP row): lane l16 loads float4 chunks
//     l16, l16+16, ... so a wave-instruction fetches 4 rows x 256 contiguous bytes; the 16
//     partial sums are reduced with DPP row_ror:8/4/2/1 adds (no LDS, no ds_bpermute);
//   * all rows of a hop are requested before the first one is consumed (up to RB*4 rows and
//     RB*NCH global_load_dwordx4 in flight per wave);
//   * W (the ef nearest so far, lib/ohnsw.ml:574-577) lives in registers as a sorted list of
//     64-bit keys (ordered distance bits << 32 | id << 1 | expanded), striped slot-major across
//     the wave; an insertion is ballot-rank + DPP wave_shr:1 shift.  The candidate queue C of
//     the reference is the set of unexpanded members of W plus a small stack of entries that
//     were evicted while tied with max(W) (they are still expandable, lib/ohnsw.ml:568);
//   * the visited set (lib/ohnsw.ml:256-268) is an LDS cache of ids with false negatives only:
//     a re-evaluated node can never enter W again (it is either still in W -- detected as a
//     duplicate key -- or was rejected/evicted with d >= max(W).d, which never grows), so the
//     result equals the exact-visited-set result while LDS stays small enough for 16+ waves/CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hnsw_dev {

constexpr uint32_t KEY_INF = 0xFFFFFFFFu;

// experiment switches (tools/mkvariant.sh builds variants with -D...=0)
#ifndef HNSW_INT_TRANSPOSE
#define HNSW_INT_TRANSPOSE 1     // byte rows + byte query: the NB integer reductions of a round as one transposing reduction
#endif
#ifndef HNSW_ASM_LOOP
#define HNSW_ASM_LOOP 1          // the headline shape's layer-0 loop instruction by instruction (hnsw_hop_asm.hip.h)
#endif
#ifndef HNSW_INSERT_ISLAND
#define HNSW_INSERT_ISLAND 1     // two-slot Ohnsw lists: the accept-and-insert loop of a round as one hand-scheduled block
#endif

struct IndexView {
    const float *X;          // [n][stride] zero-padded rows
    int64_t stride;          // floats, multiple of 4
    int64_t n;
    int32_t d;
    int32_t nchunks;         // ceil(d/4)
    const int32_t *nbr0;     // [n][S0], -1 padded, 0-based ids, reference iteration order
    int32_t S0;
    int32_t SU;
    const int32_t *nbrU;     // [rowsU][SU]
    int64_t rowsU;           // rows of nbrU (what the 32-bit row offsets of the hand-scheduled descent must reach)
    const int32_t *upper_off;// [n] first upper row of the node (layer 1), -1 if none
    const uint8_t *upper_lvl;// [n] number of upper rows of the node
    const int2 *upper_ref;   // [n] {upper_off, upper_lvl} side by side: ONE dependent load per upper-layer hop instead of two
    int32_t max_layer;
    int32_t entry_point;     // 0-based, -1 = empty
    int32_t id_base;
    // optional lossless copy of X for data whose every value is an integer in 0..255 (SIFT descriptors): rows of
    // 64*NCH bytes, zero padded; chunk c of a row (dims 4c..4c+3) is one dword.  nullptr = not available / switched off
    const uint8_t *X8;
    int32_t stride8;         // bytes per byte row
    // optional split copy of X for rows that end a little past a 128-byte line (4d mod 128 in 1..32, e.g. d = 100: 400 B =
    // three lines + 16 B; hnsw_rows_split.hip): the first main_chunks float4 chunks of every row in a table of 128-byte
    // multiples (a row costs exactly main_chunks / 8 lines), and the remaining tail_chunks chunks stored beside the
    // neighbour in layer 0's adjacency: tail0[c][j] = the tail of node nbr0[c][j], so the tails of a hop's candidates
    // share the S0 * 16 * tail_chunks contiguous bytes of the expanded node.  nullptr = not available / switched off
    const float *Xm;
    const float *tail0;      // [n][S0][4 * tail_chunks]
    int32_t stride_m;        // bytes per main row = 16 * main_chunks
    int32_t main_chunks;     // multiple of 8
    int32_t tail_chunks;     // 1 or 2; main_chunks + tail_chunks == nchunks
    // optional locality codes (hnsw_locality.hip): a bijection node -> position in an order that keeps graph-close nodes
    // together, and the same for every slot of layer 0's adjacency (lcode0[c][j] = lcode[nbr0[c][j]]): what the visited
    // set's bitmap blocks are keyed by (visited_blocks_*).  nullptr = not built
    const int32_t *lcode;    // [n]
    const int32_t *lcode0;   // [n][S0]
};

struct SearchArgs {
    const float *Q;
    int64_t q_stride;
    int64_t nq;
    int32_t ef, k, fill;
    int32_t sem;             // 0 = Ohnsw accept rule, 1 = functor (Nearest.insert_distance) rule, 2 = 1 + nearest_k's output
    int32_t vt_bits;         // log2 of the LDS visited-cache entries
    int32_t blk_bits;        // 0: the visited set is the tag cache; else log2 of its bitmap-block slots (visited_blocks_*: needs iv.lcode0)
    int32_t lds_pad;         // bytes of LDS requested beyond wave_lds_words(vt_bits): never touched; the host uses it to choose how many waves a CU holds (balanced_lds_pad)
    int32_t *out_ids;
    float *out_dist;
    uint32_t *out_ndist, *out_nhops, *out_status;
    const int32_t *qmap;     // optional: block b searches query qmap[b] (re-run of flagged queries, ordered launches)
    int64_t q_limit;         // with qmap: indices outside [0, q_limit) are skipped (0 = not checked)
    uint32_t *ovf_g;         // optional: [grid][ovf_gcap] global overflow slabs
    int32_t ovf_gcap;
    // optional: the descent was done by hnsw_descent_kernel (longest-first ordering of a large batch):
    // per query its layer-0 entry node, that node's key, and the evaluations spent so far
    const int32_t *pre_entry;
    const uint32_t *pre_key;
    const uint32_t *pre_nd;
    int32_t pre_layer;       // the pre-pass descended through layers max_layer..pre_layer; this kernel continues below
    int32_t prio_head;       // ordered launches: blocks below prio_head and blocks from prio_tail on run at issue priority 3
    int32_t prio_tail;       // (0 / INT32_MAX: every block at the default priority)
    uint32_t *any_flag;      // optional: one word, bit 0 set when ANY query of the launch carries status bit 0 (the host-buffer
                             // entry points read this word back with the results instead of scanning nq status words)
};

// ---- distance keys -------------------------------------------------------------------------
// L2: the key is the fp32 squared distance (>= +0, so its bit pattern is monotone); sqrt is
// monotone and injective on floats when taken in double (lib/ohnsw.ml:899), so ordering squared
// distances == ordering the reference's distances.  IP: distance 1 - <a,b> may be negative:
// standard order-preserving bit flip.
template <int METRIC> __device__ __forceinline__ uint32_t dist_to_key(float acc) {
    if (METRIC == 0) return __float_as_uint(acc);
    float dist = 1.0f - acc;
    uint32_t b = __float_as_uint(dist);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
template <int METRIC> __device__ __forceinline__ float key_to_dist(uint32_t key) {
    if (METRIC == 0) return sqrtf(__uint_as_float(key)); // correctly rounded == (float)sqrt((double)x)
    uint32_t b = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;
    return __uint_as_float(b);
}

// ---- DPP helpers ---------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
    return v + __int_as_float(t);
}
// sum over the 16 lanes of a DPP row; every lane ends with the same bits:
// p[j] += p[j+8]; p[j] += p[j+4]; p[j] += p[j+2]; p[j] += p[j+1]  (oracle: OG_TREE16)
__device__ __forceinline__ float reduce16(float v) {
    v = dpp_add<0x128>(v); // row_ror:8
    v = dpp_add<0x124>(v); // row_ror:4
    v = dpp_add<0x122>(v); // row_ror:2
    v = dpp_add<0x121>(v); // row_ror:1
    return v;
}
template <int CTRL> __device__ __forceinline__ int32_t dpp_add_i32(int32_t v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true) + v;
}
__device__ __forceinline__ int32_t reduce16_i32(int32_t v) {
    v = dpp_add_i32<0x128>(v);
    v = dpp_add_i32<0x124>(v);
    v = dpp_add_i32<0x122>(v);
    v = dpp_add_i32<0x121>(v);
    return v;
}
// sums of NBP values per lane over the 16 lanes of a DPP row, transposed: on return lane l16 holds the row's sum of
// value (l16 * NBP) >> 4 (NBP = 1: every lane the one sum).  Halving steps at lane distance 8, 4, 2: a lane keeps the
// value its half is for and adds the other half's partial of the same value (row_ror:8 swaps the 8-lane halves,
// row_half_mirror pairs lanes across bit 2, quad_perm [2,3,0,1] across bit 1); the rest is a plain butterfly.
template <int CTRL> __device__ __forceinline__ int32_t dpp_i32(int32_t v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int NBP>
__device__ __forceinline__ int32_t transpose_reduce16_i32(int32_t (&t)[NBP], int l16) {
    constexpr int ROR8 = 0x128, HALF_MIRROR = 0x141, QP_1032 = 0xB1, QP_2301 = 0x4E;
    if constexpr (NBP == 1) {
        return reduce16_i32(t[0]);
    } else {
        const bool b3 = (l16 & 8) != 0;
        int32_t u[NBP / 2];
#pragma unroll
        for (int j = 0; j < NBP / 2; ++j) {
            const int32_t keep = b3 ? t[j + NBP / 2] : t[j], give = b3 ? t[j] : t[j + NBP / 2];
            u[j] = keep + dpp_i32<ROR8>(give);
        }
        if constexpr (NBP == 2) {
            int32_t v = u[0];
            v += dpp_i32<HALF_MIRROR>(v);
            v += dpp_i32<QP_1032>(v);
            v += dpp_i32<QP_2301>(v);
            return v;
        } else {
            const bool b2 = (l16 & 4) != 0;
            int32_t w[NBP / 4];
#pragma unroll
            for (int j = 0; j < NBP / 4; ++j) {
                const int32_t keep = b2 ? u[j + NBP / 4] : u[j], give = b2 ? u[j] : u[j + NBP / 4];
                w[j] = keep + dpp_i32<HALF_MIRROR>(give);
            }
            if constexpr (NBP == 4) {
                int32_t v = w[0];
                v += dpp_i32<QP_1032>(v);
                v += dpp_i32<QP_2301>(v);
                return v;
            } else {
                const bool b1 = (l16 & 2) != 0;
                const int32_t keep = b1 ? w[1] : w[0], give = b1 ? w[0] : w[1];
                int32_t v = keep + dpp_i32<QP_2301>(give);
                v += dpp_i32<QP_1032>(v);
                return v;
            }
        }
    }
}
// minimum over the wave (wave-uniform result): row_ror mins inside the four 16-lane rows, then the
// four row results through scalar registers -- no LDS round trips
template <int CTRL> __device__ __forceinline__ uint32_t dpp_min_u32(uint32_t v) {
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
    return t < v ? t : v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = dpp_min_u32<0x128>(v);
    v = dpp_min_u32<0x124>(v);
    v = dpp_min_u32<0x122>(v);
    v = dpp_min_u32<0x121>(v);
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
// lane i <- lane i-1; lane 0 <- carry (wave_shr:1, bound_ctrl off keeps `old` in lane 0)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t carry) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// v_writelane_b32: lane `lane` of v <- val (both wave-uniform).  This clang has no builtin for it; the
// LLVM intrinsic is reached through its name, so the compiler still sees it (hazards, scheduling).
extern "C" __device__ int hnsw_llvm_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t wrlane(uint32_t v, uint32_t val, int lane) {
    return (uint32_t)hnsw_llvm_writelane((int)val, lane, (int)v);
}

// ---- row evaluation ------------------------------------------------------------------------
// Distances of the cnt candidates listed in LDS cand_id[] to the query held in qv; the ordered
// key of candidate ci is written to LDS cand_key[ci].
//
// NB batches of 4 rows, straight-line: all ids are read from LDS first, then all NB*NCH
// global_load_dwordx4 are issued, then consumed.  A group whose candidate index is past cnt
// re-reads the row of group 0 of its batch (same addresses: coalesced, no extra traffic) and its
// result is dropped, so no load sits behind an exec-mask branch.
template <int NCH, int NB, int METRIC, bool FULL>
__device__ __forceinline__ void eval_nb(const IndexView &iv, const float4 (&qv)[NCH],
                                        const int32_t *cand_id, uint32_t *cand_key,
                                        uint32_t *trash, int base, int cnt, int r, int l16) {
    const uint32_t stride_b = (uint32_t)iv.stride * 4u;
    const char *row[NB];     // 64-bit row addresses (base + id * stride, one v_mad_u64_u32): no table-size limit
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int ci = base + 4 * b + r;
        const int ce = ci < cnt ? ci : base + 4 * b;
        row[b] = reinterpret_cast<const char *>(iv.X) + (uint64_t)(uint32_t)cand_id[ce] * stride_b;
    }
    float4 v[NB][NCH];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 16 + l16;
            v[b][i] = *reinterpret_cast<const float4 *>(row[b] + 16u * (uint32_t)((FULL || c < iv.nchunks) ? c : 0));
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            float4 z = v[b][i];
            if (!FULL) {  // lanes past the row end: make them add exactly 0 (their qv is 0)
                const bool cv = (i * 16 + l16) < iv.nchunks;
                z.x = cv ? z.x : 0.f; z.y = cv ? z.y : 0.f; z.z = cv ? z.z : 0.f; z.w = cv ? z.w : 0.f;
            }
            if (METRIC == 0) {
                float dx = z.x - qv[i].x; acc = __builtin_fmaf(dx, dx, acc);
                float dy = z.y - qv[i].y; acc = __builtin_fmaf(dy, dy, acc);
                float dz = z.z - qv[i].z; acc = __builtin_fmaf(dz, dz, acc);
                float dw = z.w - qv[i].w; acc = __builtin_fmaf(dw, dw, acc);
            } else {
                acc = __builtin_fmaf(z.x, qv[i].x, acc);
                acc = __builtin_fmaf(z.y, qv[i].y, acc);
                acc = __builtin_fmaf(z.z, qv[i].z, acc);
                acc = __builtin_fmaf(z.w, qv[i].w, acc);
            }
        }
        acc = reduce16(acc);
        const int ci = base + 4 * b + r;
        // one lane per group stores; the others (and groups past cnt) hit scratch entries
        *((l16 == 0 && ci < cnt) ? &cand_key[ci] : &trash[(l16 << 2) + r]) = dist_to_key<METRIC>(acc);
    }
}

template <int NCH, int RB, int METRIC, bool FULL>
__device__ __forceinline__ void eval_full(const IndexView &iv, const float4 (&qv)[NCH],
                                          const int32_t *cand_id, uint32_t *cand_key,
                                          uint32_t *trash, int cnt, int r, int l16) {
    for (int base = 0; base < cnt;) {
        const int nbb = (cnt - base + 3) >> 2;   // wave-uniform
        if (RB >= 8 && nbb >= 8) { eval_nb<NCH, (RB >= 8 ? 8 : 1), METRIC, FULL>(iv, qv, cand_id, cand_key, trash, base, cnt, r, l16); base += 32; }
        else if (RB >= 4 && nbb >= 4) { eval_nb<NCH, (RB >= 4 ? 4 : 1), METRIC, FULL>(iv, qv, cand_id, cand_key, trash, base, cnt, r, l16); base += 16; }
        else if (RB >= 3 && nbb >= 3) { eval_nb<NCH, (RB >= 3 ? 3 : 1), METRIC, FULL>(iv, qv, cand_id, cand_key, trash, base, cnt, r, l16); base += 12; }
        else if (RB >= 2 && nbb >= 2) { eval_nb<NCH, (RB >= 2 ? 2 : 1), METRIC, FULL>(iv, qv, cand_id, cand_key, trash, base, cnt, r, l16); base += 8; }
        else { eval_nb<NCH, 1, METRIC, FULL>(iv, qv, cand_id, cand_key, trash, base, cnt, r, l16); base += 4; }
    }
}

template <int NCH, int RB, int METRIC>
__device__ __forceinline__ void eval_candidates(const IndexView &iv, const float4 (&qv)[NCH],
                                                const int32_t *cand_id, uint32_t *cand_key,
                                                uint32_t *trash, int cnt, int r, int l16) {
    if (iv.nchunks == 16 * NCH) eval_full<NCH, RB, METRIC, true>(iv, qv, cand_id, cand_key, trash, cnt, r, l16);
    else eval_full<NCH, RB, METRIC, false>(iv, qv, cand_id, cand_key, trash, cnt, r, l16);
}

// ---- W: sorted register-resident list ------------------------------------------------------
// Keys are (hi, lo) = (ordered distance bits, (id + 1) << 1 | expanded), compared as one 64-bit number,
// ascending, slot-major across the wave (position j = slot j / 64, lane j % 64).  The halves live in
// separate 32-bit registers: every update (DPP shift, v_writelane, flag bit) is a native 32-bit
// operation on one half, and the rank of a new key needs the low halves only on an exact distance tie.
// W always holds exactly ef entries in the TOP ef positions of its NSLOT*64-position capacity: before ef
// real nodes have been found the upper ones are +inf dummies (flagged expanded), so "|W| < ef or
// d < max(W).d" (lib/ohnsw.ml:574) is the single test d < max(W).d, the maximum always sits in the last
// lane of the last slot, and an insertion is always "shift right from the rank position, the old maximum
// falls off".  Positions below the window hold the pad key (0, 1): smaller than every real key (the id
// field stores id + 1, so a real low half is >= 2), flagged expanded.
constexpr uint32_t DUMMY_HI = 0xFFFFFFFEu;
constexpr uint32_t DUMMY_LO = 0xFFFFFFFFu;
constexpr uint32_t PAD_HI = 0u, PAD_LO = 1u;
__device__ __forceinline__ uint32_t key_id(uint32_t lo) { return (lo >> 1) - 1u; }
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int popc(uint64_t m) { return __builtin_popcountll(m); }

template <int NSLOT> struct WList {
    uint32_t hi[NSLOT], lo[NSLOT];
    uint32_t smax_hi, smax_lo;   // NSLOT > 2 only: lane s holds the key of slot s, lane 63 (the slot's maximum)
    uint32_t wmax;               // distance part of the top entry = max(W): the accept threshold (DUMMY_HI while |W| < ef)
    // entries of C that are not in W (see below).  Invariant: ovf_cnt > 0 only while every listed node is
    // at distance max(W).d (wlist_insert, the only place max(W) changes, empties the list when it drops)
    int ovf_cnt;
};

constexpr int OVF_CAP = 64; // LDS entries
// OVF_CAP entries in LDS, then an optional global slab (the exact fallback the host entry point uses
// for the rare queries that need more)
struct OvfStore {
    uint32_t *lds;
    uint32_t *g;
    int gcap;
};
__device__ __forceinline__ bool ovf_push(const OvfStore &ov, int at, uint32_t id, int lane) {
    if (at < OVF_CAP) { if (lane == 0) ov.lds[at] = id; return true; }
    if (ov.g && at - OVF_CAP < ov.gcap) { if (lane == 0) ov.g[at - OVF_CAP] = id; return true; }
    return false;
}
__device__ __forceinline__ uint32_t ovf_get(const OvfStore &ov, int at) {
    if (at < OVF_CAP) return ov.lds[at];
    return ov.g ? ov.g[at - OVF_CAP] : 0u;
}

template <int NSLOT>
__device__ __forceinline__ void wlist_init(WList<NSLOT> &w, int ef, int lane) {
    const int base = NSLOT * 64 - ef;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const bool pad = (s * 64 + lane) < base;
        w.hi[s] = pad ? PAD_HI : DUMMY_HI;
        w.lo[s] = pad ? PAD_LO : DUMMY_LO;
    }
    // lanes >= NSLOT of the slot maxima hold +inf (never "below K")
    const bool sm_pad = lane < NSLOT && (lane + 1) * 64 <= base;
    w.smax_hi = sm_pad ? PAD_HI : (lane < NSLOT ? DUMMY_HI : 0xFFFFFFFFu);
    w.smax_lo = sm_pad ? PAD_LO : 0xFFFFFFFFu;
    w.wmax = DUMMY_HI; w.ovf_cnt = 0;
}
template <int NSLOT> __device__ __forceinline__ bool wlist_full(const WList<NSLOT> &w) { return w.wmax != DUMMY_HI; }
// number of real entries
template <int NSLOT> __device__ __forceinline__ int wlist_count(const WList<NSLOT> &w) {
    int c = 0;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) c += popc(ballot(w.lo[s] != PAD_LO && w.hi[s] < DUMMY_HI));
    return c;
}

// ---- entries of C that are not in W ---------------------------------------------------------------
// SEM 0 (Ohnsw): entries evicted from W while tied with the new max(W) and not yet expanded; they are
// still in the reference's candidate queue and "c.d > max(W).d" (lib/ohnsw.ml:568) is false for them.
// Evictions come in descending key order, so last-in-first-out IS the canonical (d, id) pop order.
// SEM 1/2 (functor): additionally every neighbour evaluated while tied with max(W): Nearest.insert_distance
// answers Inserted without changing W (lib/hnsw.ml:501-504 with the merge of lib/hnsw_algo.ml:25-31), so
// the node enters VisitMe (lib/hnsw_algo.ml:360-364) and is expanded later although it never sat in W.
// Here the list is a SET (it doubles as the exact visited check for those nodes -- the LDS visited cache
// may forget them): entry = id | expanded << 31; evicted entries are kept whether expanded or not.
// In both modes the list only lives while max(W).d stays what it was when the entries were listed:
// max(W).d never grows, so once it drops every listed node is farther than max(W) for good.
__device__ __forceinline__ bool tie_contains(const OvfStore &ov, int cnt, uint32_t id, int lane) {
    for (int base = 0; base < cnt; base += 64) {
        const int at = base + lane;
        const uint32_t e = at < cnt ? ovf_get(ov, at) : 0xFFFFFFFFu;
        if (ballot((e & 0x7FFFFFFFu) == id)) return true;
    }
    return false;
}
template <int NSLOT>
__device__ __forceinline__ void tie_add(WList<NSLOT> &w, const OvfStore &ov, uint32_t entry, int lane, uint32_t &status) {
    if (ovf_push(ov, w.ovf_cnt, entry, lane)) w.ovf_cnt++;
    else status |= 1u;
}
// SEM 1: the unexpanded entry with the smallest id (canonical (d, id) order of VisitMe among equal
// distances): returns its id and list position, or -1
__device__ __forceinline__ int tie_min_unexpanded(const OvfStore &ov, int cnt, int lane, int &pos) {
    uint32_t best = 0xFFFFFFFFu; int bpos = -1;
    for (int base = 0; base < cnt; base += 64) {
        const int at = base + lane;
        const uint32_t e = at < cnt ? ovf_get(ov, at) : 0xFFFFFFFFu;     // expanded entries have bit 31 set: never the minimum
        const uint32_t m = wave_min_u32(e);
        if (m < best && m < 0x80000000u) { best = m; bpos = base + __builtin_ctzll(ballot(e == m)); }
    }
    pos = bpos;
    return bpos >= 0 ? (int)best : -1;
}
__device__ __forceinline__ void tie_mark_expanded(const OvfStore &ov, int pos, int lane) {
    if (lane == 0) {
        if (pos < OVF_CAP) ov.lds[pos] |= 0x80000000u;
        else ov.g[pos - OVF_CAP] |= 0x80000000u;
    }
}

// Insert (kd, kid) -- both wave-uniform, kd < w.wmax.  Mirrors lib/ohnsw.ml:575-577: push W, pop
// the farthest.  Duplicates (same id already in W: a re-evaluated node) are ignored.
// Rank = number of keys below (kd, klo): one "hi < kd" ballot per slot; only when some member of W has
// exactly the distance kd (one more ballot says so) are the low halves compared -- that is also the only
// case in which the node itself can already be in W.  The shift is a DPP wave_shr:1 per half and slot,
// the new key then lands in its lane with v_writelane (no per-lane equality selects).
// (Measured alternative, same results: deciding "my key is above K" per lane and selecting
// mine / left neighbour / K without any scalar rank -- fewer scalar hand-offs but more vector
// instructions; 4 % slower on an idle chip, 3 % on the 10 k batch.)
template <int NSLOT, int SEM = 0>
__device__ __forceinline__ void wlist_insert(WList<NSLOT> &w, uint32_t kd, uint32_t kid, int lane,
                                             const OvfStore &ov, uint32_t &status) {
    const uint32_t klo = (kid + 1u) << 1;
    int p = 0;
    if (NSLOT <= 2) {
        uint64_t eq[NSLOT];
        bool tie = false;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            p += popc(ballot(w.hi[s] < kd));
            eq[s] = ballot(w.hi[s] == kd);
            tie = tie || eq[s] != 0ull;
        }
        if (tie) {
            int q = p;
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                p += popc(eq[s] & ballot(w.lo[s] < klo));
                q += popc(eq[s] & ballot(w.lo[s] < klo + 2u));   // counts the node itself, expanded or not
            }
            if (p != q) return;                   // already in W
        }
    } else {
        // the slot holding the rank position: the first whose maximum is not below K (slot maxima, one per lane)
        const int first = popc(ballot(w.smax_hi < kd || (w.smax_hi == kd && w.smax_lo < klo)));   // lanes >= NSLOT hold +inf
        int q = 0;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            if (s == first) {
                const uint64_t eq = ballot(w.hi[s] == kd);
                p = s * 64 + popc(ballot(w.hi[s] < kd));
                q = p + popc(eq & ballot(w.lo[s] < klo + 2u));
                p += popc(eq & ballot(w.lo[s] < klo));
            }
        }
        if (p != q) return;                       // already in W
    }
    const uint32_t ev_hi = w.wmax, ev_lo = rdlane(w.lo[NSLOT - 1], 63);   // the entry that falls off
#pragma unroll
    for (int s = NSLOT - 1; s >= 0; --s) {
        if (s * 64 + 63 < p) continue;            // slots wholly below the rank stay (uniform)
        uint32_t sh_hi, sh_lo;
        if (s > 0) { sh_hi = wave_shr1(w.hi[s], rdlane(w.hi[s - 1], 63)); sh_lo = wave_shr1(w.lo[s], rdlane(w.lo[s - 1], 63)); }
        else { sh_hi = wave_shr1(w.hi[s], 0); sh_lo = wave_shr1(w.lo[s], 0); }
        const bool mv = (s * 64 + lane) >= p;     // lane p takes its left neighbour's key here, K below
        w.hi[s] = mv ? sh_hi : w.hi[s];
        w.lo[s] = mv ? sh_lo : w.lo[s];
        if (s * 64 <= p) { w.hi[s] = wrlane(w.hi[s], kd, p - s * 64); w.lo[s] = wrlane(w.lo[s], klo, p - s * 64); }
        if (NSLOT > 2) {
            const uint32_t mh = rdlane(w.hi[s], 63), ml = rdlane(w.lo[s], 63);
            w.smax_hi = (lane == s) ? mh : w.smax_hi;
            w.smax_lo = (lane == s) ? ml : w.smax_lo;
        }
    }
    w.wmax = rdlane(w.hi[NSLOT - 1], 63);
    if (ev_hi != w.wmax) {
        w.ovf_cnt = 0;                            // max(W).d dropped: every listed entry is dead
    } else if (ev_hi == DUMMY_HI) {               // W still holds dummies: nothing real fell off
    } else if (SEM != 0) {                        // evicted while tied with the new max(W): stays in C, and visited
        tie_add(w, ov, key_id(ev_lo) | ((ev_lo & 1u) << 31), lane, status);
    } else if (!(ev_lo & 1u)) {                   // rare: evicted while tied with the new max(W), unexpanded
        tie_add(w, ov, key_id(ev_lo), lane, status);
    }
}

// Unexpanded-member masks of W, one ballot per slot.
template <int NSLOT>
__device__ __forceinline__ void wlist_unexpanded_masks(const WList<NSLOT> &w, uint64_t (&m)[NSLOT]) {
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) m[s] = ballot((w.lo[s] & 1u) == 0u);
}
// First set position over the slot masks: returns the node id there (or -1) and clears that bit.
// A chain of wave-uniform branches, one per slot (template recursion keeps every slot index a constant, so
// the masks and keys stay in registers): the taken path is a count-trailing-zeros, one v_readlane and a
// few scalar operations.
template <int NSLOT, int S> struct WFirst {
    static __device__ __forceinline__ int run(const WList<NSLOT> &w, uint64_t (&m)[NSLOT], int &index) {
        if (m[S] != 0ull) {
            const int L = __builtin_ctzll(m[S]);
            m[S] &= m[S] - 1;
            index = S * 64 + L;
            return (int)key_id(rdlane(w.lo[S], L));
        }
        return WFirst<NSLOT, S + 1>::run(w, m, index);
    }
};
template <int NSLOT> struct WFirst<NSLOT, NSLOT> {
    static __device__ __forceinline__ int run(const WList<NSLOT> &, uint64_t (&)[NSLOT], int &index) { index = -1; return -1; }
};
template <int NSLOT>
__device__ __forceinline__ int wlist_take_first(const WList<NSLOT> &w, uint64_t (&m)[NSLOT], int &index) {
    return WFirst<NSLOT, 0>::run(w, m, index);
}
// mark entry `index` expanded (pop_min, lib/ohnsw.ml:565); index -1 marks nothing.  Branch-free on
// purpose: a compare and an OR per slot keep the key registers out of control-flow merges.
template <int NSLOT>
__device__ __forceinline__ void wlist_mark_expanded(WList<NSLOT> &w, int index, int lane) {
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) w.lo[s] |= (uint32_t)((s * 64 + lane) == index);
}

// ---- per-wave scratch in LDS -------------------------------------------------------------------
// Visited cache: 2^(vt_bits-1) sets of two 16-bit tags (one 32-bit word per set, newest tag in the
// low half).  set = id mod #sets, tag = id div #sets: (set, tag) identifies the id exactly, so a hit
// is never a false positive (node ids are insertion order, i.e. unrelated to the geometry: the low
// bits spread as well as a hash would, and no multiply sits in front of the LDS read).  The host
// sizes the cache so that (n-1) div #sets < 0xFFFF: the value 0xFFFF marks an empty way and never
// equals a real tag.  A miss for a node already evaluated is harmless -- see the header comment.
#ifndef HNSW_VT_THREE_WAYS
#define HNSW_VT_THREE_WAYS 1
#endif
struct WaveCtx {
    int lane, r, l16;
    uint32_t *vt;        // visited cache, 1 << (vt_bits - 1) words
    uint32_t set_mask;   // #sets - 1
    int set_bits;        // vt_bits - 1
    uint32_t tag_shift;  // 16: two 16-bit tags per word; 10: three 10-bit tags (visited_three_ways)
    int vt_words;
    uint32_t blk_set_mask;   // visited as bitmap blocks (visited_blocks_*): #sets - 1 of the block directory ...
    uint32_t *blk_bm;        // ... and the slots' bitmaps (BLK_WORDS words each); nullptr: the tag cache is in use
    int32_t *cand_id;    // [64]
    uint32_t *cand_key;  // [64]
    uint32_t *trash;     // [64] write-only sink shared by every masked-off store
    OvfStore ovf;        // lds: [OVF_CAP]
    // byte rows only (query_bytes): the query itself is byte-valued and short enough for exact integer arithmetic
    bool qint;
    uint32_t qb[4];      // this lane's chunks of the query as packed bytes
    int32_t q2;          // sum of the squares of the whole query
};
// 4 KiB of tags at vt_bits = 11 plus 1 KiB: 28 waves per CU fit the 160 KiB LDS
__host__ __device__ inline size_t wave_lds_words(int vt_bits) { return (((size_t)1 << vt_bits) >> 1) + 192 + OVF_CAP; }
// Visited as a cache of BITMAP BLOCKS over the locality codes (IndexView::lcode0): 2^blk_bits slots, each a block of
// 2^BLK_SHIFT consecutive codes = BLK_WORDS words of bits, in sets of BLK_WAYS slots; a slot's directory word is
// (block number << 8) | stamp of its last use.  See visited_blocks_mem_add.
constexpr int BLK_SHIFT = 8, BLK_WORDS = (1 << BLK_SHIFT) / 32, BLK_WAYS = 8;
__host__ __device__ inline size_t wave_lds_words_blocks(int blk_bits) { return ((size_t)(1 + BLK_WORDS) << blk_bits) + 192 + OVF_CAP; }
// LDS words of one search wave (the launch's dynamic LDS, before any padding)
__host__ __device__ inline size_t search_lds_words(int vt_bits, int blk_bits) {
    const size_t t = wave_lds_words(vt_bits), b = blk_bits > 0 ? wave_lds_words_blocks(blk_bits) : 0;
    return t > b ? t : b;
}
__device__ __forceinline__ WaveCtx make_ctx(uint32_t *lds, int vt_bits, int lane, int blk_bits = 0) {
    WaveCtx cx;
    cx.qint = false; cx.q2 = 0;
    cx.lane = lane; cx.r = lane >> 4; cx.l16 = lane & 15;
    cx.vt = lds;
    cx.set_bits = vt_bits - 1;
    cx.tag_shift = 16;
    cx.set_mask = (1u << (vt_bits - 1)) - 1u;
    cx.vt_words = 1 << (vt_bits - 1);
    cx.blk_set_mask = 0; cx.blk_bm = nullptr;
    if (blk_bits > 0) {      // directory words first (what visited_clear resets), the bitmaps behind them
        cx.blk_set_mask = ((1u << blk_bits) / BLK_WAYS) - 1u;
        cx.blk_bm = lds + (1 << blk_bits);
        cx.vt_words = (1 + BLK_WORDS) << blk_bits;
    }
    uint32_t *rest = lds + cx.vt_words;
    cx.cand_id = reinterpret_cast<int32_t *>(rest);
    cx.cand_key = rest + 64;
    cx.trash = rest + 128;
    cx.ovf.lds = rest + 192; cx.ovf.g = nullptr; cx.ovf.gcap = 0;
    return cx;
}
__device__ __forceinline__ void visited_clear(const WaveCtx &cx) { // Visited.clear, lib/ohnsw.ml:262
    const int words = cx.blk_bm ? (int)(cx.blk_bm - cx.vt) : cx.vt_words;   // blocks: empty directory words; a bitmap is cleared when its slot is claimed
    const uint32_t empty = cx.blk_bm ? 0xFFFFFF00u : 0xFFFFFFFFu;
    for (int i = cx.lane; i < words; i += 64) cx.vt[i] = empty;
}
// Visited.mem (lib/ohnsw.ml:259); `word` returns the set's current content for the add that follows
__device__ __forceinline__ bool visited_mem(const WaveCtx &cx, uint32_t id, uint32_t &word) {
    word = cx.vt[id & cx.set_mask];
    const uint32_t tag = id >> cx.set_bits;
    if (cx.tag_shift == 16) return (word & 0xFFFFu) == tag || (word >> 16) == tag;
    return (word & 0x3FFu) == tag || ((word >> 10) & 0x3FFu) == tag || ((word >> 20) & 0x3FFu) == tag;
}
// Three ways per set where the tags are short enough: with n nodes and 2^set_bits sets a tag has bits(n - 1) - set_bits bits; at most
// 10 of them (and never the all-ones value, which marks an empty way) -> three tags per word, the same LDS holds half as many
// tags again.  Worth it where a walk visits several times the cache's capacity (large ef): on clustered data a 2-way cache of
// 2^12 tags re-evaluates 31 % (ef 256, 5 k nodes visited; tools/visited_cache_sim.py), three ways 12 %.  The knn kernels with
// W in four or eight registers -- on float32 rows also those with two -- switch it on when the index is small enough; everything
// else keeps two ways (one SDWA compare per way in the hand-scheduled loops).
__device__ __forceinline__ void visited_three_ways(WaveCtx &cx, int32_t n) {
    if (n > 0 && ((uint32_t)(n - 1) >> cx.set_bits) < 0x3FFu) cx.tag_shift = 10;
}
// Visited.add (lib/ohnsw.ml:260) for all lanes at once: the new tag enters way 0, way 0 moves to
// way 1; lanes with on == false store to the scratch sink (no branch)
__device__ __forceinline__ void visited_add_masked(const WaveCtx &cx, uint32_t id, uint32_t word, bool on) {
    uint32_t *slot = on ? &cx.vt[id & cx.set_mask] : &cx.trash[cx.lane];
    *slot = (word << cx.tag_shift) | (id >> cx.set_bits);
}

// ---- Visited as bitmap blocks over locality codes -------------------------------------------------------------------
// The tag cache above remembers 2^12 nodes in 8 KiB; a walk at ef 512 visits four times that, and on clustered data (where
// neighbourhoods overlap) a forgotten node comes back: 40 % of the evaluations of C5's shape were repeats (round 4).  No
// replacement policy fixes that (tools/visited_policy_sim.py: anything implementable stays above 30 %), and 2 bytes per
// remembered node is within a quarter of the information-theoretic minimum for an arbitrary subset of 10 M ids.  What the
// ids lack is LOCALITY: they are insertion order.  Under a code that numbers graph-close nodes consecutively (lcode: nodes
// grouped by the layer-2 node their own descent reaches, those groups by their layer-3 node, ... -- hnsw_locality.hip) the
// nodes a walk visits fill a few hundred blocks of 256 consecutive codes, a third of each: one BIT per node.  The same
// 9 KiB then hold 256 blocks = 65 536 codes and the walk's re-evaluations fall to 1-4 % (same simulator, same traces).
// Exact as the tag cache is: (block number, bit) identifies the code, the code identifies the node (a bijection); a slot's
// bits are cleared before its directory word can match anybody.  What is lost is lost in whole blocks (the least recently
// touched of the set's eight), never invented.
//
// One call handles all 64 neighbours of a hop: `code` the lane's neighbour's code, `valid` whether the lane holds a
// neighbour at all, `now` the hop's stamp.  Returns Visited.mem (lib/ohnsw.ml:259) and performs Visited.add (:260) for the
// lanes that answer false.  LDS operations of one wave execute in order, lane conflicts inside one store are decided by
// the hardware (one lane wins); the steps below only rely on that:
//   1 read the set's eight directory words; a lane whose block is there (hit) reads its bit -> the answer;
//   2 every valid lane writes (block, now) to its slot -- a hit refreshes the stamp, a miss claims the set's least recently
//     used slot (several misses on one set claim the same slot: one wins);
//   3 every lane reads its slot's word back: it OWNS the slot iff the word carries its block;
//   4 owners that missed clear the slot's bitmap (all of them the same words, before any bit is set);
//   5 owners whose answer was false set their bit (atomic OR: several lanes may share a word).
// A lane that lost its slot in step 2 evaluates its neighbour without remembering it -- harmless, like any forgetting.
__device__ __forceinline__ bool visited_blocks_mem_add(const WaveCtx &cx, bool valid, uint32_t code, uint32_t now) {
    const uint32_t b = code >> BLK_SHIFT, bit = code & ((1u << BLK_SHIFT) - 1u);
    const uint32_t set = b & cx.blk_set_mask;
    uint32_t *dir = cx.vt + set * BLK_WAYS;
    uint32_t wd[BLK_WAYS];
    {
        const uint4 a0 = *reinterpret_cast<const uint4 *>(dir), a1 = *reinterpret_cast<const uint4 *>(dir + 4);
        wd[0] = a0.x; wd[1] = a0.y; wd[2] = a0.z; wd[3] = a0.w; wd[4] = a1.x; wd[5] = a1.y; wd[6] = a1.z; wd[7] = a1.w;
    }
    int hit = -1;
    uint32_t oldest = 0;
#pragma unroll
    for (int i = 0; i < BLK_WAYS; ++i) {
        if ((wd[i] >> 8) == b) hit = i;
        // age modulo 256 (a block untouched for 512 hops looks young again: a worse victim choice, nothing else); an empty
        // way is block 0xFFFFFF (no code reaches it) last touched at stamp 0: the oldest there is while the stamps have not wrapped
        const uint32_t age = (now - wd[i]) & 255u;
        const uint32_t key = (age << 3) | (uint32_t)i;
        oldest = key > oldest ? key : oldest;
    }
    const int way = hit >= 0 ? hit : (int)(oldest & 7u);
    uint32_t *slot_dir = dir + way;
    uint32_t *bm = cx.blk_bm + (set * BLK_WAYS + (uint32_t)way) * BLK_WORDS;
    bool seen = false;
    if (hit >= 0) seen = (bm[bit >> 5] >> (bit & 31u)) & 1u;                       // 1
    __syncthreads();
    *(valid ? slot_dir : &cx.trash[cx.lane]) = (b << 8) | (now & 255u);            // 2
    __syncthreads();
    const bool own = valid && (*slot_dir >> 8) == b;                                // 3
    __syncthreads();
    if (own && hit < 0) {                                                           // 4
        *reinterpret_cast<uint4 *>(bm) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4 *>(bm + 4) = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    if (own && !seen) atomicOr(&bm[bit >> 5], 1u << (bit & 31u));                   // 5
    __syncthreads();
    return seen;
}
// the stamp of a hop (one step per two hops: 512 hops before it wraps)
__device__ __forceinline__ uint32_t visited_blocks_stamp(uint32_t n_hops) { return (n_hops >> 1) & 255u; }

// neighbour row of node c on `layer` (Graph.adjacent, lib/ohnsw.ml:171); -1 = hole / padding
__device__ __forceinline__ int adj_entry(const IndexView &iv, int layer, int c, int lane) {
    if (layer == 0) return lane < iv.S0 ? iv.nbr0[(int64_t)c * iv.S0 + lane] : -1;
    const int2 ref = iv.upper_ref[c];
    const int off = ref.x, lvl = ref.y;
    return (lane < iv.SU && layer <= lvl) ? iv.nbrU[((int64_t)off + (layer - 1)) * iv.SU + lane] : -1;
}

// ---- one round of a hop: up to 4*NB fresh neighbours evaluated ------------------------------------
// The 16-lane group r evaluates candidates base + NB*r + b, b = 0..NB-1 (batch b = one wave-instruction
// per 256 row bytes: rows of candidates b, NB+b, 2NB+b, 3NB+b).  Straight-line: the NB ids come from LDS
// (every lane of a group reads its group's entry), the row address is one 64-bit multiply-add per batch
// (no 32-bit offset limit), all NB*NCH global_load_dwordx4 are issued before the first is consumed, and
// the keys never travel through LDS: after the DPP reduction every lane of group r holds the key of
// candidate (r, b); lane 16*r + b keeps it (and the id), so one register pair holds the round's
// candidates and ASCENDING LANE = ASCENDING ROW ORDER (lane 16*r + b <-> candidate base + NB*r + b).
// A group whose candidate is past the end of the list re-reads the row of candidate base + b (group 0 of
// the same batch: same addresses, coalesced, no extra traffic) and its key is forced to +inf.
// ROWS: 0 = fp32 rows, ragged (lanes past the row end are masked); 1 = fp32 rows, every chunk of the lane grid inside
// the row; 2 = byte rows (IndexView::X8): one dword per chunk, converted to the same four floats the fp32 row holds, so
// the arithmetic and its result are those of the fp32 row bit for bit, for a quarter of the bytes.  (The zero padding
// of a byte row meets the zero padding of the query: it adds exact zeros.)
// 3 = split fp32 rows (IndexView::Xm / tail0; layer-0 hops only): the lane grid and the arithmetic are those of ROWS = 0 --
// chunk c of a row is still evaluated by lane c % 16 in its (c / 16)-th step -- but the lanes whose chunk lies in the
// tail take it from the expanded node's tail row (`tail_row`, at the candidate's position in that node's adjacency row,
// which the compaction left in cand_key[]) instead of from a fourth 128-byte line of the vector's own row.
template <int NCH, int NB, int METRIC, int ROWS>
__device__ __forceinline__ void hop_round(const IndexView &iv, const float4 (&qv)[NCH], const WaveCtx &cx,
                                          int base, int cnt, uint32_t &out_key, uint32_t &out_id,
                                          const char *tail_row = nullptr) {
    constexpr bool FULL = ROWS == 1 || ROWS == 2;
    const int r = cx.r, l16 = cx.l16;
    const uint32_t stride_b = ROWS == 2 ? (uint32_t)iv.stride8 : (uint32_t)iv.stride * 4u;
    const int mine = base + NB * r;
    uint32_t id[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int ci = (mine + b < cnt) ? mine + b : base + b;   // base + b < cnt: a round of NB batches has > 4*(NB-1) candidates
        id[b] = (uint32_t)cx.cand_id[ci];
    }
    float4 v[ROWS == 2 ? 1 : NB][ROWS == 2 ? 1 : NCH];
    uint32_t v8[ROWS == 2 ? NB : 1][ROWS == 2 ? NCH : 1];
    if constexpr (ROWS == 2) {
        const char *xlane = reinterpret_cast<const char *>(iv.X8) + 4 * l16;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const uint32_t *row = reinterpret_cast<const uint32_t *>(xlane + (uint64_t)id[b] * stride_b);
#pragma unroll
            for (int i = 0; i < NCH; ++i) v8[b][i] = row[i * 16];
        }
    } else if constexpr (FULL) {
        const char *xlane = reinterpret_cast<const char *>(iv.X) + 16 * l16;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 *row = reinterpret_cast<const float4 *>(xlane + (uint64_t)id[b] * stride_b);
#pragma unroll
            for (int i = 0; i < NCH; ++i) v[b][i] = row[i * 16];
        }
    } else if constexpr (ROWS == 3) {
        // per lane and step: base address and multiplier of whichever table the lane's chunk lives in (lanes past the
        // row end re-read chunk 0 of the main row and are dropped below, as in the ragged path)
        const char *lbase[NCH];
        uint32_t lmul[NCH];
        bool ltail[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 16 + l16;
            ltail[i] = c >= iv.main_chunks && c < iv.nchunks;
            lbase[i] = ltail[i] ? tail_row + 16 * (c - iv.main_chunks)
                                : reinterpret_cast<const char *>(iv.Xm) + (c < iv.main_chunks ? 16 * c : 0);
            lmul[i] = ltail[i] ? 16u * (uint32_t)iv.tail_chunks : (uint32_t)iv.stride_m;
        }
        uint32_t pj[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int ci = (mine + b < cnt) ? mine + b : base + b;
            pj[b] = cx.cand_key[ci];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                v[b][i] = *reinterpret_cast<const float4 *>(lbase[i] + (uint64_t)(ltail[i] ? pj[b] : id[b]) * lmul[i]);
        }
    } else {
        uint32_t coff[NCH];   // lanes past the row end re-read chunk 0 (coalesced) and are zeroed below
#pragma unroll
        for (int i = 0; i < NCH; ++i) coff[i] = ((i * 16 + l16) < iv.nchunks) ? (uint32_t)(i * 16 + l16) * 16u : 0u;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const char *row = reinterpret_cast<const char *>(iv.X) + (uint64_t)id[b] * stride_b;
#pragma unroll
            for (int i = 0; i < NCH; ++i) v[b][i] = *reinterpret_cast<const float4 *>(row + coff[i]);
        }
    }
    // every load above is issued before the first is consumed: without this fence the scheduler may sink
    // the later loads below the first batches' arithmetic to save registers, which turns one memory round
    // trip per round into several (measured on the ragged-row inner-product variant: +50 % per query)
    __builtin_amdgcn_sched_barrier(0);
    uint32_t kk = KEY_INF, ii = 0;
    if constexpr (ROWS == 2 && NCH <= 4) {
        // Byte row AND byte-valued query, d <= 256: every product and every partial sum of the float arithmetic below is
        // an integer below 2^24, so that arithmetic never rounds and its result is the exact integer sum whatever the
        // order -- computed here with 4-byte dot products (|x - q|^2 = x.x - 2 x.q + q.q): 4 instructions per 8
        // dimensions where the float path needs 24.
        if (cx.qint) {
#if !HNSW_INT_TRANSPOSE
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                uint32_t sxq = 0, sxx = 0;
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    sxq = __builtin_amdgcn_udot4(v8[b][i], cx.qb[i], sxq, false);
                    if (METRIC == 0) sxx = __builtin_amdgcn_udot4(v8[b][i], v8[b][i], sxx, false);
                }
                int32_t t = METRIC == 0 ? (int32_t)sxx - 2 * (int32_t)sxq : (int32_t)sxq;
                t = reduce16_i32(t);
                if (METRIC == 0) t += cx.q2;
                const uint32_t key = dist_to_key<METRIC>((float)t);
                const bool here = l16 == b;
                kk = here ? key : kk;
                ii = here ? id[b] : ii;
            }
            out_key = (mine + l16 < cnt) ? kk : KEY_INF;
            out_id = ii;
            return;
#endif
            constexpr int NBP = NB <= 1 ? 1 : NB <= 2 ? 2 : NB <= 4 ? 4 : 8;   // batches padded to a power of two
            int32_t t[NBP];
#pragma unroll
            for (int b = 0; b < NBP; ++b) {
                uint32_t sxq = 0, sxx = 0;
                if (b < NB) {
#pragma unroll
                    for (int i = 0; i < NCH; ++i) {
                        sxq = __builtin_amdgcn_udot4(v8[b][i], cx.qb[i], sxq, false);
                        if (METRIC == 0) sxx = __builtin_amdgcn_udot4(v8[b][i], v8[b][i], sxx, false);
                    }
                }
                t[b] = METRIC == 0 ? (int32_t)sxx - 2 * (int32_t)sxq : (int32_t)sxq;
            }
            // Integer sums are exact in any order, so the NB reductions over the group's 16 lanes are done as ONE
            // transposing reduction: lanes l16 of 16/NBP consecutive lanes end with candidate b = l16 * NBP / 16's sum
            // (NBP = 2: 6 instructions instead of 8, and no per-batch "is this my lane" selects afterwards).
            int32_t tsum = transpose_reduce16_i32<NBP>(t, l16);
            if (METRIC == 0) tsum += cx.q2;
            constexpr int SH = NBP == 1 ? 4 : NBP == 2 ? 3 : NBP == 4 ? 2 : 1;   // lanes per candidate = 1 << SH
            const int b = l16 >> SH;
            uint32_t myid = id[0];
#pragma unroll
            for (int j = 1; j < NB; ++j) myid = (b == j) ? id[j] : myid;
            const bool rep = (l16 & ((1 << SH) - 1)) == 0 && b < NB && mine + b < cnt;
            out_key = rep ? dist_to_key<METRIC>((float)tsum) : KEY_INF;
            out_id = myid;
            return;
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            float4 z;
            if constexpr (ROWS == 2) {
                const uint32_t u = v8[b][i];
                z = make_float4((float)(u & 0xFFu), (float)((u >> 8) & 0xFFu), (float)((u >> 16) & 0xFFu), (float)(u >> 24));
            } else {
                z = v[b][i];
            }
            float t = acc;
            if (METRIC == 0) {
                float dx = z.x - qv[i].x; t = __builtin_fmaf(dx, dx, t);
                float dy = z.y - qv[i].y; t = __builtin_fmaf(dy, dy, t);
                float dz = z.z - qv[i].z; t = __builtin_fmaf(dz, dz, t);
                float dw = z.w - qv[i].w; t = __builtin_fmaf(dw, dw, t);
            } else {
                t = __builtin_fmaf(z.x, qv[i].x, t);
                t = __builtin_fmaf(z.y, qv[i].y, t);
                t = __builtin_fmaf(z.z, qv[i].z, t);
                t = __builtin_fmaf(z.w, qv[i].w, t);
            }
            // lanes past the row end re-read chunk 0: their chunk's contribution is dropped whole (one select
            // per chunk; the sum a lane keeps is exactly the sum over its valid chunks)
            acc = (FULL || (i * 16 + l16) < iv.nchunks) ? t : acc;
        }
        const uint32_t key = dist_to_key<METRIC>(reduce16(acc));
        const bool here = l16 == b;
        kk = here ? key : kk;
        ii = here ? id[b] : ii;
    }
    out_key = (mine + l16 < cnt) ? kk : KEY_INF;   // lanes l16 >= NB kept KEY_INF
    out_id = ii;
}

// one round with as many 4-row batches as the list still needs (at most RB): returns the candidates consumed
template <int NCH, int RB, int METRIC, int ROWS>
__device__ __forceinline__ int eval_round(const IndexView &iv, const float4 (&qv)[NCH], const WaveCtx &cx,
                                          int base, int cnt, uint32_t &ckey, uint32_t &cid, const char *tail_row = nullptr) {
    const int nbb = (cnt - base + 3) >> 2;   // wave-uniform
    if (RB >= 8 && nbb >= 8) { hop_round<NCH, (RB >= 8 ? 8 : 1), METRIC, ROWS>(iv, qv, cx, base, cnt, ckey, cid, tail_row); return 32; }
    if (RB >= 4 && nbb >= 4) { hop_round<NCH, (RB >= 4 ? 4 : 1), METRIC, ROWS>(iv, qv, cx, base, cnt, ckey, cid, tail_row); return 16; }
    if (RB >= 3 && nbb >= 3) { hop_round<NCH, (RB >= 3 ? 3 : 1), METRIC, ROWS>(iv, qv, cx, base, cnt, ckey, cid, tail_row); return 12; }
    if (RB >= 2 && nbb >= 2) { hop_round<NCH, (RB >= 2 ? 2 : 1), METRIC, ROWS>(iv, qv, cx, base, cnt, ckey, cid, tail_row); return 8; }
    hop_round<NCH, 1, METRIC, ROWS>(iv, qv, cx, base, cnt, ckey, cid, tail_row);
    return 4;
}

// Ohnsw.search_one_simple (lib/ohnsw.ml:492-508) on layers `from` down to `to` (inclusive):
// scan ALL neighbours of the current best, move to the first-in-row-order strictly closer one,
// repeat until no change.  No visited set (argument ignored, :493).  Same row evaluation as the layer
// search (keys stay in registers: lane order = row order, so the first lane holding the minimum is the
// first neighbour in row order that attains it); the upper-row lookup is one 8-byte load.
#if HNSW_ASM_LOOP
// the same descent instruction by instruction for C2's shape (hnsw_hop_asm.hip.h)
__device__ __forceinline__ void greedy_descend_bytes_l2_asm(const IndexView &iv, int from, int to, int &cur, uint32_t &cur_key,
                                                            const WaveCtx &cx, uint32_t &n_dist);
#endif
template <int NCH, int RB, int METRIC, int ROWS = -1>
__device__ __forceinline__ void greedy_descend(const IndexView &iv, const float4 (&qv)[NCH], int from,
                                               int to, int &cur, uint32_t &cur_key, const WaveCtx &cx,
                                               uint32_t &n_dist) {
#if HNSW_ASM_LOOP && !defined(HNSW_PHASE_TIMING)
    if constexpr (NCH == 2 && METRIC == 0 && ROWS == 2) {
        // upper rows of at most 16 neighbours (M <= 16), a byte-valued query, tables the 32-bit offsets reach: the last row
        // ends at byte (rowsU * SU) * 4, and the block multiplies (row + layer - 1) by SU * 4 in 32 bits
        if (cx.qint && iv.SU <= 16 && ((uint64_t)iv.rowsU + 1) * (uint64_t)iv.SU < (1ull << 28)) {
            greedy_descend_bytes_l2_asm(iv, from, to, cur, cur_key, cx, n_dist);
            return;
        }
    }
#endif
    const bool full_rows = ROWS < 0 ? iv.nchunks == 16 * NCH : ROWS == 1;   // (split rows, ROWS = 3, serve layer-0 hops only: the descent reads X)
    for (int layer = from; layer >= to; --layer) {
        for (;;) {
            const int nb = adj_entry(iv, layer, cur, cx.lane);
            const bool valid = nb >= 0;
            const uint64_t m = ballot(valid);
            const int cnt = popc(m);
            if (cnt == 0) break;
            const int pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            __syncthreads();
            if (valid) cx.cand_id[pos] = nb;
            __syncthreads();
            n_dist += cnt;
            uint32_t bkey = KEY_INF;
            int bid = -1;
            for (int base = 0; base < cnt;) {
                uint32_t ckey, cid;
                if (ROWS == 2) base += eval_round<NCH, RB, METRIC, 2>(iv, qv, cx, base, cnt, ckey, cid);
                else if (ROWS == 1 || (ROWS < 0 && full_rows)) base += eval_round<NCH, RB, METRIC, 1>(iv, qv, cx, base, cnt, ckey, cid);
                else base += eval_round<NCH, RB, METRIC, 0>(iv, qv, cx, base, cnt, ckey, cid);
                const uint32_t mk = wave_min_u32(ckey);
                if (mk < bkey) {                                             // strict: an earlier round keeps a tie
                    bkey = mk;
                    bid = (int)rdlane(cid, __builtin_ctzll(ballot(ckey == mk)));
                }
            }
            if (bkey < cur_key) { cur = bid; cur_key = bkey; }                // strict, :502
            else break;
        }
    }
}

// ---- the accept-and-insert loop of one round for a two-slot Ohnsw list, hand-scheduled -------------------------------
// What hop_eval's loop does per accepted candidate (lib/ohnsw.ml:574-577), as ONE block of gfx950 instructions: next
// candidate of `pass` -> its key and id to scalar registers -> accept test against the CURRENT max(W) -> rank (two compare
// + count pairs) -> new maximum -> shift (EXEC narrowed to the lanes from the rank up around two in-place v_mov_b32_dpp
// wave_shr:1 per slot; the lane at the rank keeps its value -- its source lane is switched off -- until the v_writelane
// pair puts the new key there) -> next.  About 38 instructions and 3 branches per insertion where hipcc's code for the
// C++ loop above needs 55 to 67 and 8; both the scalar unit (shared by the CU's four SIMDs) and the vector unit are
// close to saturated in a loaded launch, and a lone wave pays for every taken branch.
// The block handles the COMMON insertion only and leaves the loop, with the candidate still in `pass`, when
//   * some member of W is at exactly the candidate's distance (ids decide the rank, and the node itself may be in W), or
//   * the entry that falls off is a real one at the new maximum's distance (it stays in C: the tie list);
// the caller then runs the general wlist_insert for that candidate and comes back.  Returns whether anything was
// inserted (every insertion here lowers max(W).d or replaces a dummy: the tie list is dead or empty).
__device__ __forceinline__ bool insert_island2(WList<2> &w, uint32_t ckey, uint32_t cid, uint64_t &pass) {
    // wave-uniform by construction; the readfirstlanes only matter where the compiler cannot see it (a caller whose
    // loop bounds come from memory) and fold away elsewhere
    uint32_t any = 0, wmax = (uint32_t)uniform((int)w.wmax);
    pass = ((uint64_t)(uint32_t)uniform((int)(pass >> 32)) << 32) | (uint32_t)uniform((int)(uint32_t)pass);
    uint32_t i, kd, klo, p, t, nw, ca, cb, sm0;
    uint64_t e0, e1, m, sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b32 %[sm0], m0\n"                                  // m0 (lane select of v_writelane) is handed back as it was
        "1:\n\t"
        "s_cmp_eq_u64 %[pass], 0\n\t"
        "s_cbranch_scc1 9f\n\t"
        "s_ff1_i32_b64 %[i], %[pass]\n\t"
        "v_readlane_b32 %[kd], %[ck], %[i]\n\t"
        "v_readlane_b32 %[klo], %[ci], %[i]\n\t"
        "s_cmp_ge_u32 %[kd], %[wmax]\n\t"
        "s_cbranch_scc1 8f\n\t"                                  // no longer below max(W): rejected (:574)
        "v_cmp_eq_u32_e64 %[e0], %[kd], %[h0]\n\t"
        "v_cmp_eq_u32_e64 %[e1], %[kd], %[h1]\n\t"
        "v_cmp_gt_u32_e32 vcc, %[kd], %[h0]\n\t"
        "s_bcnt1_i32_b64 %[p], vcc\n\t"
        "v_cmp_gt_u32_e32 vcc, %[kd], %[h1]\n\t"
        "s_bcnt1_i32_b64 %[t], vcc\n\t"
        "s_or_b64 %[e0], %[e0], %[e1]\n\t"
        "s_cbranch_scc1 9f\n\t"                                  // a member of W at this very distance: general path
        "s_add_u32 %[p], %[p], %[t]\n\t"                         // rank = keys below
        "v_readlane_b32 %[nw], %[h1], 62\n\t"
        "s_cmp_eq_u32 %[p], 127\n\t"
        "s_cselect_b32 %[nw], %[kd], %[nw]\n\t"                  // the new max(W).d
        "s_cmp_eq_u32 %[nw], %[wmax]\n\t"
        "s_cbranch_scc1 4f\n"                                     // the entry falling off ties with it: dummy or general path
        "3:\n\t"
        "s_lshl_b32 %[klo], %[klo], 1\n\t"
        "s_add_u32 %[klo], %[klo], 2\n\t"                        // low half: (id + 1) << 1, unexpanded
        "s_cmp_lt_u32 %[p], 64\n\t"
        "s_cbranch_scc1 5f\n\t"
        "s_sub_u32 %[p], %[p], 64\n\t"                           // rank in the upper slot
        "s_lshl_b64 %[m], -1, %[p]\n\t"
        "s_mov_b64 exec, %[m]\n\t"
        "v_mov_b32_dpp %[h1], %[h1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[l1], %[l1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_mov_b32 m0, %[p]\n\t"
        "v_writelane_b32 %[h1], %[kd], m0\n\t"
        "v_writelane_b32 %[l1], %[klo], m0\n\t"
        "s_branch 7f\n"
        "4:\n\t"
        "s_cmp_eq_u32 %[wmax], -2\n\t"                           // DUMMY_HI: W still holds dummies, nothing real falls off
        "s_cbranch_scc1 3b\n\t"
        "s_branch 9f\n"
        "5:\n\t"                                                 // rank in the lower slot: the upper slot moves as a whole
        "v_readlane_b32 %[ca], %[h0], 63\n\t"
        "v_readlane_b32 %[cb], %[l0], 63\n\t"
        "v_mov_b32_dpp %[h1], %[h1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[l1], %[l1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_lshl_b64 %[m], -1, %[p]\n\t"
        "s_mov_b64 exec, %[m]\n\t"
        "v_mov_b32_dpp %[h0], %[h0] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[l0], %[l0] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "v_writelane_b32 %[h1], %[ca], 0\n\t"
        "v_writelane_b32 %[l1], %[cb], 0\n\t"
        "s_mov_b32 m0, %[p]\n\t"
        "v_writelane_b32 %[h0], %[kd], m0\n\t"
        "v_writelane_b32 %[l0], %[klo], m0\n"
        "7:\n\t"
        "s_mov_b32 %[wmax], %[nw]\n\t"
        "s_mov_b32 %[any], 1\n"
        "8:\n\t"
        "s_bitset0_b64 %[pass], %[i]\n\t"
        "s_branch 1b\n"
        "9:\n\t"
        "s_mov_b32 m0, %[sm0]"
        : [h0] "+&v"(w.hi[0]), [h1] "+&v"(w.hi[1]), [l0] "+&v"(w.lo[0]), [l1] "+&v"(w.lo[1]),
          [pass] "+&s"(pass), [wmax] "+&s"(wmax), [any] "+&s"(any),
          [i] "=&s"(i), [kd] "=&s"(kd), [klo] "=&s"(klo), [p] "=&s"(p), [t] "=&s"(t), [nw] "=&s"(nw),
          [ca] "=&s"(ca), [cb] "=&s"(cb), [e0] "=&s"(e0), [e1] "=&s"(e1), [m] "=&s"(m), [sv] "=&s"(sv), [sm0] "=&s"(sm0)
        : [ck] "v"(ckey), [ci] "v"(cid)
        : "vcc", "scc");
    w.wmax = wmax;
    return any != 0;
}

// The rounds of one hop.  Accept test of lib/ohnsw.ml:574 in row order (= ascending lane), each candidate
// against the CURRENT W.
#ifdef HNSW_PHASE_TIMING   // measurement build: shader-clock cycles per phase, summed over the query's hops
struct PhaseClock { uint64_t t[6] = {0, 0, 0, 0, 0, 0}, mark = 0; uint32_t n_ins = 0; };
#define HNSW_PHASE(pc, i) do { const uint64_t tn__ = clock64(); (pc).t[i] += tn__ - (pc).mark; (pc).mark = tn__; } while (0)
#else
struct PhaseClock {};
#define HNSW_PHASE(pc, i) do { } while (0)
#endif

// the accepted candidates of one round (`pass`: lanes of ckey / cid), in row order (= ascending lane), each against the CURRENT W
template <int NSLOT, int SEM>
__device__ __forceinline__ void accept_candidates(WList<NSLOT> &w, const WaveCtx &cx, uint64_t pass, uint32_t ckey, uint32_t cid,
                                                  uint32_t &status, PhaseClock &pc) {
    const int lane = cx.lane;
    while (pass) {
        const int i = __builtin_ctzll(pass);
        pass &= pass - 1;
        const uint32_t kd = rdlane(ckey, i);
        if (kd < w.wmax) {
#ifdef HNSW_PHASE_TIMING
            pc.n_ins++;
#endif
            wlist_insert<NSLOT, SEM>(w, kd, rdlane(cid, i), lane, cx.ovf, status);     // :575-577
        } else if (SEM && kd == w.wmax && wlist_full(w)) {
            // Nearest.insert_distance on a tie with max(W): Inserted, W unchanged (lib/hnsw.ml:501-504):
            // the node joins C only.  It may be a node the visited cache forgot: still in W, or listed.
            const uint32_t kid = rdlane(cid, i);
            const uint32_t klo1 = ((kid + 1u) << 1) | 1u;
            bool known = false;
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) known = known || (ballot(w.hi[s] == kd && (w.lo[s] | 1u) == klo1) != 0ull);
            if (!known && w.ovf_cnt > 0) { __syncthreads(); known = tie_contains(cx.ovf, w.ovf_cnt, kid, lane); }
            if (!known) tie_add(w, cx.ovf, kid, lane, status);
        }
    }
}

// The rounds of one hop from candidate `base0` of the hop's list on.
template <int NCH, int RB, int NSLOT, int METRIC, int SEM, int ROWS>
__device__ __forceinline__ void hop_eval(const IndexView &iv, const float4 (&qv)[NCH], WList<NSLOT> &w,
                                         const WaveCtx &cx, int cnt, uint32_t &status, PhaseClock &pc,
                                         const char *tail_row = nullptr, int base0 = 0) {
    const int lane = cx.lane;
    for (int base = base0; base < cnt;) {
        uint32_t ckey, cid;
        base += eval_round<NCH, RB, METRIC, ROWS>(iv, qv, cx, base, cnt, ckey, cid, tail_row);
        uint64_t pass = ballot(SEM ? ckey <= w.wmax : ckey < w.wmax);
#ifdef HNSW_PHASE_TIMING
        asm volatile("" :: "s"(pass));
#endif
        HNSW_PHASE(pc, 2);                                               // ids from LDS, row loads, arithmetic, keys, accept ballot
#if HNSW_INSERT_ISLAND && !defined(HNSW_PHASE_TIMING)
        // (not in the kernel that has the hand-written loop: this is then its seldom-taken path, and the block's scalar
        // temporaries would push that kernel past the 96 SGPRs that eight waves per SIMD allow)
        if constexpr (NSLOT == 2 && SEM == 0 && !(HNSW_ASM_LOOP && NCH == 2 && METRIC == 0 && ROWS == 2)) {
            while (pass) {
                if (insert_island2(w, ckey, cid, pass)) w.ovf_cnt = 0;
                if (pass == 0ull) break;
                // the lowest remaining candidate needs the general insertion (see insert_island2)
                const int i = __builtin_ctzll(pass);
                pass &= pass - 1;
                const uint32_t kd = rdlane(ckey, i);
                if (kd < w.wmax) wlist_insert<NSLOT, SEM>(w, kd, rdlane(cid, i), lane, cx.ovf, status);
            }
            continue;
        }
#endif
        accept_candidates<NSLOT, SEM>(w, cx, pass, ckey, cid, status, pc);
    }
}

// What a hand-scheduled loop instantiated for the functor rule (hnsw_hop_loop.inc, HNSW_LOOP_SEM 1) hands back when it leaves a
// hop in the middle: the candidates of the current round not yet dealt with (`pass`: lanes of ckey / cid1 = id + 1), how
// many of the hop's `total` fresh neighbours have not been evaluated yet, and (split rows) the node being expanded.
struct HopResume { uint64_t pass; uint32_t ckey, cid1; int total, remaining, node; };

} // namespace hnsw_dev
#include "hnsw_hop_asm.hip.h"
namespace hnsw_dev {

// Ohnsw.search_k (lib/ohnsw.ml:543-588) on one layer.  On entry W holds the start nodes
// (all unexpanded = the start queue, :555-559); on exit W is the ef nearest found.
// While the rows of a hop are in flight, the adjacency row of the then-nearest unexpanded
// candidate is fetched too: if it is still the nearest after this hop's insertions (the common
// case once the search has converged) the next hop starts without a dependent round trip.
//
// SEM 0: Ohnsw, accept iff d < max(W).d (lib/ohnsw.ml:574).
// SEM 1: the functor path, Hnsw_algo.Search.search (lib/hnsw_algo.ml:350-391) with
// Nearest.insert_distance (lib/hnsw.ml:494-506): d < max(W).d replaces the maximum; d == max(W).d is
// answered Inserted but leaves W as it is (the merge of lib/hnsw_algo.ml:25-31 puts the new element on
// top and remove_max takes it off again), and the node is still pushed to VisitMe (:360-364) and expanded
// later; d > max(W).d is Too_far.  Order among equal distances (the in-tree pairing heap's, restated only
// in the oracle's TIES_HEAP mode) is fixed to (d, id) here: VisitMe pops the smallest id first, a
// replacement evicts the largest id of the farthest class.
// ROWS: 1 = every float4 chunk of the 16 x NCH lane grid lies inside a row (d in 64*NCH-3 .. 64*NCH: no
// masking), 0 = ragged rows, 2 = byte rows (see hop_round), -1 = fp32 rows, shape decided at run time (the builder and
// the layer operators; the knn kernel is instantiated per case so that neither pays for the other's registers).
// BLK 1 (the knn kernel's layer-0 search only): Visited is the bitmap-block cache over the neighbours' locality codes
// (visited_blocks_mem_add) instead of the tag cache.
template <int NCH, int RB, int NSLOT, int METRIC, int SEM = 0, int ROWS = -1, int BLK = 0>
__device__ __forceinline__ void search_layer(const IndexView &iv, const float4 (&qv)[NCH], int layer,
                                             WList<NSLOT> &w, int ef, const WaveCtx &cx,
                                             uint32_t &n_dist, uint32_t &n_hops, uint32_t &status) {
    const int lane = cx.lane;
    const bool full_rows = ROWS < 0 ? iv.nchunks == 16 * NCH : ROWS == 1;
#if HNSW_ASM_LOOP && !defined(HNSW_PHASE_TIMING)
    // The hand-scheduled loops (hnsw_hop_asm.hip.h: HopLoop<NCH, NSLOT, METRIC, ROWS, SEM, BLK>, one instantiation per shape from the
    // generated table hnsw_hop_instances.inc), same results.  Shapes: byte rows and a byte-valued query, or float32 rows (full,
    // ragged, split), of 65..128 (NCH 2) or 129..256 (NCH 4) dimensions; W in 1, 2, 3, 4, 6 or 8 registers; either metric, either
    // rule; Visited as the tag cache or (W in three or more registers; not the NCH 4 byte rows) as bitmap blocks.
    // (The blocks form the byte offset (id + 1) * S0 * 4 + lane * 4 of an adjacency row in 32 bits, one row ahead of the node
    // they fetch: the row AFTER the last node's must still be below 2^32 bytes whatever S0 is; and they restore EXEC with
    // s_mov_b64 exec, -1: they are entered with all 64 lanes on -- the kernel runs one full wave per query and reaches this
    // point through wave-uniform branches only.  A query whose tie list went to the global slab keeps the C++ loop.)
    constexpr int SEMK = SEM != 0 ? 1 : 0;
    constexpr bool ASM_ANY = HopLoop<NCH, NSLOT, METRIC, ROWS, SEMK, BLK>::available;
    bool asm_ok = false;
    if constexpr (ASM_ANY) {
        asm_ok = layer == 0 && cx.ovf.g == nullptr && ((uint64_t)iv.n + 1) * (uint64_t)iv.S0 < (1ull << 30) &&
                 (ROWS == 2 ? cx.qint != 0 : iv.nchunks > (NCH == 4 ? 32 : 16));
    }
    if constexpr (ASM_ANY && SEM == 0) {
        if (asm_ok) {
            HopResume rs0;
#ifdef HNSW_ASM_DEBUG
            if constexpr (NCH == 2 && NSLOT == 2 && METRIC == 0 && ROWS == 2 && BLK == 0) {
                // debugging: hops to run in the block, then the C++ loop goes on from there (the high half of ef, or a build constant)
#ifdef HNSW_ASM_DEBUG_HOPS
                HopLoop<NCH, NSLOT, METRIC, ROWS, 0, BLK>::run(iv, w, cx, rs0, qv, n_dist, n_hops, status, HNSW_ASM_DEBUG_HOPS);
#else
                HopLoop<NCH, NSLOT, METRIC, ROWS, 0, BLK>::run(iv, w, cx, rs0, qv, n_dist, n_hops, status, (uint32_t)ef >> 16);
#endif
            } else {
                HopLoop<NCH, NSLOT, METRIC, ROWS, 0, BLK>::run(iv, w, cx, rs0, qv, n_dist, n_hops, status);
                return;
            }
#else
            HopLoop<NCH, NSLOT, METRIC, ROWS, 0, BLK>::run(iv, w, cx, rs0, qv, n_dist, n_hops, status);
            return;
#endif
        }
    }
#endif
    int pref_id = -1, pref_nb = -1;
    uint32_t pref_code = 0;
    constexpr bool blocks = BLK != 0;                              // Visited as bitmap blocks over the neighbours' locality codes
    PhaseClock pc;
#ifdef HNSW_PHASE_TIMING
    pc.mark = clock64();
#endif
    for (;;) {
#if HNSW_ASM_LOOP && HNSW_ASM_LOOP_SEM1 && !defined(HNSW_PHASE_TIMING)
        if constexpr (ASM_ANY && SEM != 0) {
            // The functor rule on the same hand-scheduled loops: while the tie set is empty the two rules differ only in the
            // moments an entry would ENTER the set (a neighbour evaluated AT max(W).d with W full; an entry evicted while tied
            // with the new maximum); the loop instantiated for this rule leaves at exactly those moments, in the middle of the
            // hop, and the hop is finished here.  The hops below then run while the set is alive (it dies when max(W).d drops).
            if (asm_ok && w.ovf_cnt == 0) {
                HopResume rs;
                const bool left = HopLoop<NCH, NSLOT, METRIC, ROWS, 1, BLK>::run(iv, w, cx, rs, qv, n_dist, n_hops, status);
                if (!left) break;
                pref_id = -1;
                if constexpr (NSLOT > 2) {        // the loop keeps the slots' maxima in scalar registers: wlist_insert's copy is stale
#pragma unroll
                    for (int s = 0; s < NSLOT; ++s) {
                        const uint32_t mh = rdlane(w.hi[s], 63), ml = rdlane(w.lo[s], 63);
                        w.smax_hi = (lane == s) ? mh : w.smax_hi;
                        w.smax_lo = (lane == s) ? ml : w.smax_lo;
                    }
                }
                accept_candidates<NSLOT, SEM>(w, cx, rs.pass, rs.ckey, rs.cid1 - 1u, status, pc);
                const char *tail_row = nullptr;
                if constexpr (ROWS == 3)
                    tail_row = reinterpret_cast<const char *>(iv.tail0) + (uint64_t)(uint32_t)rs.node * (uint32_t)(iv.S0 * 16 * iv.tail_chunks);
                hop_eval<NCH, RB, NSLOT, METRIC, SEM, ROWS>(iv, qv, w, cx, rs.total, status, pc, tail_row, rs.total - rs.remaining);
                continue;
            }
        }
#endif
        uint64_t um[NSLOT];
        wlist_unexpanded_masks(w, um);
        int cidx;
        int c = wlist_take_first(w, um, cidx);                           // pop_min, :565
        if (SEM == 0) {
            wlist_mark_expanded(w, cidx, lane);
            if (c < 0) {
                // no unexpanded member of W: only entries evicted while tied with max(W) can still
                // satisfy "not (c.d > max(W).d)" (:568)
                if (w.ovf_cnt > 0) { __syncthreads(); c = uniform((int)ovf_get(cx.ovf, --w.ovf_cnt)); }
                else break;
            }
        } else {
            // VisitMe.pop_nearest in (d, id) order: members of W below max(W).d first; among the nodes AT
            // max(W).d the smallest id, whether it sits in W or only in the list
            bool from_w = c >= 0;
            if (w.ovf_cnt > 0) {
                uint32_t chi = 0;
                if (c >= 0) {
#pragma unroll
                    for (int s = 0; s < NSLOT; ++s) if ((cidx >> 6) == s) chi = rdlane(w.hi[s], cidx & 63);
                }
                if (c < 0 || chi == w.wmax) {
                    __syncthreads();
                    int tpos;
                    const int t = tie_min_unexpanded(cx.ovf, w.ovf_cnt, lane, tpos);
                    if (t >= 0 && (c < 0 || t < c)) {
                        tie_mark_expanded(cx.ovf, tpos, lane);
                        __syncthreads();
                        c = t; from_w = false;
                        wlist_unexpanded_masks(w, um);                   // the W member was not taken
                    }
                }
            }
            if (c < 0) break;
            wlist_mark_expanded(w, from_w ? cidx : -1, lane);
        }
        n_hops++;
        int nb;
        uint32_t code = 0;
        if (c == pref_id) { nb = pref_nb; code = pref_code; status += 256u; }   // Graph.adjacent, :570 (bits 8..: prefetch hits)
        else {
            nb = adj_entry(iv, layer, c, lane);
            if (blocks) code = lane < iv.S0 ? (uint32_t)iv.lcode0[(int64_t)c * iv.S0 + lane] : 0u;
        }
        uint32_t vword = 0;
        bool seen;
        if constexpr (blocks) seen = visited_blocks_mem_add(cx, nb >= 0, code, visited_blocks_stamp(n_hops));   // Visited.mem and .add, :571-572
        else seen = visited_mem(cx, (uint32_t)nb, vword);                // Visited.mem, :571 (a hole reads some set: harmless)
        const bool fresh = (nb >= 0) & !seen;
        const uint64_t m = ballot(fresh);
        const int cnt = popc(m);
#ifdef HNSW_PHASE_TIMING
        asm volatile("" :: "s"(m));
#endif
        HNSW_PHASE(pc, 0);                                               // pop + adjacency row + visited filter
        // issued only now so that it shares its flight with this hop's rows (loads return in order)
        int pidx;
        pref_id = wlist_take_first(w, um, pidx);                         // the next nearest unexpanded
        if (pref_id >= 0) {
            pref_nb = adj_entry(iv, layer, pref_id, lane);
            if (blocks) pref_code = lane < iv.S0 ? (uint32_t)iv.lcode0[(int64_t)pref_id * iv.S0 + lane] : 0u;
        }
        if (cnt != 0) {
            const int pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            __syncthreads();
            if (fresh) {                                                 // Visited.add, :572
                if (!blocks) cx.vt[(uint32_t)nb & cx.set_mask] = (vword << cx.tag_shift) | ((uint32_t)nb >> cx.set_bits);
                cx.cand_id[pos] = nb;
                if (ROWS == 3) cx.cand_key[pos] = (uint32_t)lane;        // split rows: where in c's row (= in its tail row) the candidate sits
            }
            __syncthreads();
            n_dist += cnt;
            HNSW_PHASE(pc, 1);                                           // prefetch issue + compaction through LDS
            if (ROWS == 3)      // layer 0 only (the knn kernel): the tails of c's neighbours lie beside its adjacency row
                hop_eval<NCH, RB, NSLOT, METRIC, SEM, 3>(iv, qv, w, cx, cnt, status, pc,
                                                         reinterpret_cast<const char *>(iv.tail0) + (uint64_t)(uint32_t)c * (uint32_t)(iv.S0 * 16 * iv.tail_chunks));
            else if (ROWS == 2) hop_eval<NCH, RB, NSLOT, METRIC, SEM, 2>(iv, qv, w, cx, cnt, status, pc);                                  // :573-577
            else if (ROWS == 1 || (ROWS < 0 && full_rows)) hop_eval<NCH, RB, NSLOT, METRIC, SEM, 1>(iv, qv, w, cx, cnt, status, pc);
            else hop_eval<NCH, RB, NSLOT, METRIC, SEM, 0>(iv, qv, w, cx, cnt, status, pc);
        }
#ifdef HNSW_PHASE_TIMING
        if (!wlist_full(w)) HNSW_PHASE(pc, 4); else                        // insertions while W still holds dummies
#endif
        HNSW_PHASE(pc, 3);                                               // insertions
    }
#ifdef HNSW_PHASE_TIMING   // reported through the counters: n_dist = phase 0, n_hops = phase 2, status = phase 3 (20 bits) | phase 1 / 64 (12 bits)
#if HNSW_PHASE_TIMING == 2   // second set: insert time while W fills / once it is full, and the number of insertions
    n_dist = (uint32_t)pc.t[4]; n_hops = (uint32_t)pc.t[3]; status = pc.n_ins;
#else
    n_dist = (uint32_t)pc.t[0]; n_hops = (uint32_t)pc.t[2];
    { const uint64_t t1 = pc.t[1] >> 6; const uint64_t t3 = pc.t[3] + pc.t[4]; status = ((uint32_t)t3 & 0xFFFFFu) | ((uint32_t)(t1 > 4095 ? 4095 : t1) << 20); }
#endif
#endif
}

template <int NCH>
__device__ __forceinline__ void load_query(float4 (&qv)[NCH], const float *qp, int d, int l16) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {   // zero beyond d
        const int e0 = 4 * (i * 16 + l16);
        qv[i].x = (e0 + 0 < d) ? qp[e0 + 0] : 0.f;
        qv[i].y = (e0 + 1 < d) ? qp[e0 + 1] : 0.f;
        qv[i].z = (e0 + 2 < d) ? qp[e0 + 2] : 0.f;
        qv[i].w = (e0 + 3 < d) ? qp[e0 + 3] : 0.f;
    }
}
// the query as loaded, written to a device-resident copy by one 16-lane group (the pre-pass of a batch whose queries are read
// straight from the caller's registered host matrix: the search kernel then reads the copy, not the PCIe bus again)
template <int NCH>
__device__ __forceinline__ void store_query(const float4 (&qv)[NCH], float *qp, int d, int l16) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int e0 = 4 * (i * 16 + l16);
        if (e0 + 0 < d) qp[e0 + 0] = qv[i].x;
        if (e0 + 1 < d) qp[e0 + 1] = qv[i].y;
        if (e0 + 2 < d) qp[e0 + 2] = qv[i].z;
        if (e0 + 3 < d) qp[e0 + 3] = qv[i].w;
    }
}
// byte rows: is the query byte-valued too (SIFT queries are)?  Then hop_round may use exact integer arithmetic.
template <int NCH>
__device__ __forceinline__ void query_bytes(WaveCtx &cx, const float4 (&qv)[NCH], int d) {
    cx.qint = false; cx.q2 = 0;
    if constexpr (NCH <= 4) {
        bool ok = d <= 256;               // 256 * 255^2 < 2^24
        uint32_t sq = 0;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const float c[4] = {qv[i].x, qv[i].y, qv[i].z, qv[i].w};
            uint32_t u = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ok = ok && c[j] >= 0.0f && c[j] <= 255.0f && c[j] == truncf(c[j]);
                u |= ((uint32_t)c[j] & 0xFFu) << (8 * j);
            }
            cx.qb[i] = u;
            sq = __builtin_amdgcn_udot4(u, u, sq, false);
        }
        cx.qint = ballot(ok) == ~0ull;
        cx.q2 = reduce16_i32((int32_t)sq);
    }
}
// a database row as the query (rows are zero padded to the stride)
template <int NCH>
__device__ __forceinline__ void load_row(float4 (&qv)[NCH], const IndexView &iv, int node, int l16) {
    const float4 *row = reinterpret_cast<const float4 *>(iv.X) + (int64_t)node * (iv.stride >> 2);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = i * 16 + l16;
        qv[i] = (c < iv.nchunks) ? row[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ---- the search kernel: Ohnsw.knn (lib/ohnsw.ml:859-875) per query -------------------------------
// waves per SIMD the register allocator must leave room for: the L2 variants for d <= 128 with W in at most two key
// registers per lane and unmasked rows fit 72 VGPRs (7 waves/SIMD, 7168 resident queries) without
// spilling (8 waves would spill); the kernel with the hand-written loop (hnsw_hop_asm.hip.h) is held to 8 waves/SIMD: the
// loop itself needs 57 VGPRs and ~45 SGPRs, but left alone the allocator spreads the C++ paths around it over all 102
// SGPRs, which costs the eighth wave (it now parks ~20 scalars of the prologue / epilogue in VGPR lanes instead); the
// other variants are left to the allocator
#ifndef HNSW_SEM1_8WAVES       /* the functor-rule kernels of the headline shape at eight waves per SIMD too */
#define HNSW_SEM1_8WAVES 1
#endif
#ifndef HNSW_SEARCH_MIN_WAVES
#define HNSW_SEARCH_MIN_WAVES(NCH, NSLOT, METRIC, ROWS, SEMF) \
    (((NCH) <= 2 && (NSLOT) <= 2 && (METRIC) == 0 && (ROWS) == 1) ? 7 : \
     (HNSW_ASM_LOOP && (NCH) == 2 && (NSLOT) <= 4 && (METRIC) == 0 && (ROWS) == 2 && ((SEMF) == 0 || HNSW_SEM1_8WAVES)) ? 8 : 1)
#endif
// SEMF: 0 = Ohnsw accept rule, 1 = the functor path's (a.sem 1 and 2); ROWS: 0 | 1 | 2 | 3, see hop_round
// BLK: 1 = Visited as bitmap blocks (a.blk_bits slots, iv.lcode / lcode0 present), see search_layer
template <int NCH, int RB, int NSLOT, int METRIC, int SEMF, int ROWS, int BLK = 0>
__global__ void __launch_bounds__(64, HNSW_SEARCH_MIN_WAVES(NCH, NSLOT, METRIC, ROWS, SEMF))
hnsw_search_kernel(const IndexView iv, const SearchArgs a) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    if ((int64_t)blockIdx.x >= a.nq) return;
    const int64_t q = a.qmap ? a.qmap[blockIdx.x] : (int64_t)blockIdx.x;
    if (a.q_limit && (q < 0 || q >= a.q_limit)) return;     // never follow a bad map entry into memory
    WaveCtx cx = make_ctx(lds, a.vt_bits, lane, BLK ? a.blk_bits : 0);
    if (a.ovf_g) { cx.ovf.g = a.ovf_g + (int64_t)blockIdx.x * a.ovf_gcap; cx.ovf.gcap = a.ovf_gcap; }
#if HNSW_VT_THREE_WAYS
    // ef > 128: the walk visits several times what the cache holds; float32 rows from ef 65 on (a re-evaluation costs four times a
    // byte row's bytes there, and the two-slot byte-row loop -- the headline -- keeps its branch-free two-way filter)
    if constexpr (NSLOT >= 3 || (NSLOT == 2 && ROWS != 2)) visited_three_ways(cx, iv.n);
#endif
    // Issue priority inside an ordered launch (blocks run the walks predicted longest first): the launch ends with its
    // longest walk or with the last of the late starters (the blocks that had to wait for a free slot), so those two ends
    // of the order are issued ahead of the waves they share a SIMD with; the middle has slack (C2, 10 k queries: 0.38 ->
    // 0.36 ms per call, nearly all of it from the late starters; priorities without the ordering, or raised by a walk itself
    // once it has grown long, gained nothing).  Results do not depend on it.
    if ((int)blockIdx.x < a.prio_head || (int)blockIdx.x >= a.prio_tail) __builtin_amdgcn_s_setprio(3);
#ifdef HNSW_TIMING
    const uint64_t t_start__ = wall_clock64();
#endif

    float4 qv[NCH];
    load_query<NCH>(qv, a.Q + q * a.q_stride, iv.d, cx.l16);
    if (ROWS == 2) query_bytes<NCH>(cx, qv, iv.d);
    visited_clear(cx);

    uint32_t n_dist = 0, n_hops = 0, status = 0;
    constexpr int DROWS = ROWS == 3 ? 0 : ROWS;   // split rows serve the layer-0 hops; the entry point and the upper layers read X

    int cur;
    uint32_t cur_key;
    if (a.pre_entry) {                       // descent already done (hnsw_descent_kernel)
        cur = a.pre_entry[q]; cur_key = a.pre_key[q]; n_dist = a.pre_nd[q];
        if (a.pre_layer > 1) greedy_descend<NCH, RB, METRIC, DROWS>(iv, qv, a.pre_layer - 1, 1, cur, cur_key, cx, n_dist);
    } else {
        // entry point
        cur = iv.entry_point;
        if (lane == 0) cx.cand_id[0] = cur;
        __syncthreads();
        { uint32_t ck, ci; hop_round<NCH, 1, METRIC, DROWS>(iv, qv, cx, 0, 1, ck, ci); cur_key = rdlane(ck, 0); }
        n_dist += 1;
        greedy_descend<NCH, RB, METRIC, DROWS>(iv, qv, iv.max_layer, 1, cur, cur_key, cx, n_dist);   // :865-867
    }

    WList<NSLOT> w;
    wlist_init(w, a.ef, lane);
    wlist_insert<NSLOT, SEMF>(w, cur_key, (uint32_t)cur, lane, cx.ovf, status);         // :871, seeds W :555-557
    if constexpr (BLK != 0) (void)visited_blocks_mem_add(cx, lane == 0, (uint32_t)iv.lcode[cur], 0u);
    else { uint32_t hw; (void)visited_mem(cx, (uint32_t)cur, hw); visited_add_masked(cx, (uint32_t)cur, hw, lane == 0); }
    __syncthreads();
#ifdef HNSW_ASM_DEBUG
    search_layer<NCH, RB, NSLOT, METRIC, SEMF, ROWS, BLK>(iv, qv, 0, w, a.ef | (a.lds_pad << 16), cx, n_dist, n_hops, status);
#else
    search_layer<NCH, RB, NSLOT, METRIC, SEMF, ROWS, BLK>(iv, qv, 0, w, a.ef, cx, n_dist, n_hops, status); // :872-874
#endif

    // results: W[0..k) ascending (lib/ohnsw.ml:886-893); sem 2: nearest_k's k farthest of W, lib/hnsw.ml:522-525
    int wbase = NSLOT * 64 - a.ef;
    if (a.sem == 2) { const int cnt = wlist_count(w); wbase += cnt > a.k ? cnt - a.k : 0; }
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int idx = s * 64 + lane - wbase;
        if (idx >= 0 && idx < a.k) {
            int32_t oid = -1;
            float od = a.fill == 0 ? __uint_as_float(0x7FC00000u) : __uint_as_float(0x7F800000u);
            const uint32_t hi = w.hi[s];
            if (hi < DUMMY_HI) { oid = (int32_t)key_id(w.lo[s]) + iv.id_base; od = key_to_dist<METRIC>(hi); }
            a.out_ids[q * a.k + idx] = oid;
            a.out_dist[q * a.k + idx] = od;
        }
    }
    if (lane == 0) {
#ifdef HNSW_TIMING   // measurement build (tools/sweep.py TIMELINE=1): each query's start / end in 10 ns ticks
        if (a.out_ndist) a.out_ndist[q] = (uint32_t)t_start__;
        if (a.out_nhops) a.out_nhops[q] = (uint32_t)wall_clock64();
#else
        if (a.out_ndist) a.out_ndist[q] = n_dist;
        if (a.out_nhops) a.out_nhops[q] = n_hops;
#endif
        if (a.out_status) a.out_status[q] = status;
#ifndef HNSW_PHASE_TIMING
        // (a plain store: the word carries this one bit, and it may live in pinned host memory, where a device atomic would
        // need PCIe atomics)
        if ((status & 1u) && a.any_flag) *reinterpret_cast<volatile uint32_t *>(a.any_flag) = 1u;
#endif
    }
}

// ---- the descent alone (lib/ohnsw.ml:865-867), for the longest-first ordering of a large batch --
// A launch of more queries than the chip holds at once ends with a drain: the queries that start
// last run on a nearly empty chip.  How long a query's layer-0 walk is correlates with how far its
// layer-0 entry node is from it (Spearman 0.82 with the hop count on the C2 workload), so the host
// runs this kernel first, sorts the queries by that distance, farthest first, and launches the
// search kernel in that order (qmap) with the descent result handed over (pre_*): the long walks
// start first and the drain is made of short ones.  Per-query results do not depend on the order.
template <int NCH, int RB, int METRIC, int ROWS = -1>
__global__ void __launch_bounds__(64)
hnsw_descent_kernel(const IndexView iv, const float *Q, int64_t q_stride, int64_t nq, int32_t to_layer,
                    int32_t *out_entry, uint32_t *out_key, uint32_t *out_nd,
                    uint32_t *out_sortkey, int32_t *out_index, float *stage) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    if (q >= nq) return;
    WaveCtx cx = make_ctx(lds, 4, lane);
    float4 qv[NCH];
    load_query<NCH>(qv, Q + q * q_stride, iv.d, cx.l16);
    if (stage && cx.r == 0) store_query<NCH>(qv, stage + q * q_stride, iv.d, cx.l16);   // Q may be host memory: keep a device copy
    if (ROWS == 2) query_bytes<NCH>(cx, qv, iv.d);
    int cur = iv.entry_point;
    if (lane == 0) cx.cand_id[0] = cur;
    __syncthreads();
    uint32_t cur_key;
    if (ROWS == 2) {
        uint32_t ck, ci;
        hop_round<NCH, 1, METRIC, ROWS == 2 ? 2 : 1>(iv, qv, cx, 0, 1, ck, ci);
        cur_key = rdlane(ck, 0);
    } else {
        eval_candidates<NCH, RB, METRIC>(iv, qv, cx.cand_id, cx.cand_key, cx.trash, 1, cx.r, cx.l16);
        __syncthreads();
        cur_key = cx.cand_key[0];
    }
    uint32_t n_dist = 1;
    greedy_descend<NCH, RB, METRIC, ROWS>(iv, qv, iv.max_layer, to_layer, cur, cur_key, cx, n_dist);
    if (lane == 0) {
        out_entry[q] = cur; out_key[q] = cur_key; out_nd[q] = n_dist;
        out_sortkey[q] = ~cur_key;           // ascending sort of this = farthest entry first
        out_index[q] = (int32_t)q;
    }
}

// ---- gathered distances (bench_dist/bench_dist.ml counterpart) --------------------------------
// One 16-lane group per (query, id) pair, UB x 4 pairs of the same query in flight per wave; same
// arithmetic and summation order as the search kernel.
// Rows of more than 512 dimensions (NCH 16, 8: bench_dist's d = 784 is 3136 bytes a row): the wave's 64 lanes load the query ONCE and
// hand it to the four 16-lane groups through LDS (rounds 1-5: every group loaded all of it from global memory), and the next batch's
// ids are requested before the current batch's rows are consumed.  The compiler then keeps the query in registers across the loop
// (hnsw_distance_kernel<16>: 166 VGPRs, three waves per SIMD; pinning the LDS reads inside the loop -- 128 VGPRs, four waves -- was
// measured: no faster, profiles/r06_dist_ab.txt): a gather kernel with 12.5 KB of rows in flight per wave is at the memory
// system's random-row ceiling at either occupancy.
template <int NCH, int METRIC>
__global__ void __launch_bounds__(64)
hnsw_distance_kernel(const IndexView iv, const float *Q, int64_t q_stride, int64_t nq,
                     const int32_t *ids, int32_t m, float *out) {
    constexpr int UB = NCH <= 2 ? 4 : (NCH <= 4 ? 2 : 1);
    constexpr bool QLDS = NCH >= 8;
    __shared__ float4 qs[QLDS ? 16 * NCH : 1];
    const int lane = threadIdx.x;
    const int r = lane >> 4, l16 = lane & 15;
    const int64_t q = blockIdx.x;
    if (q >= nq) return;
    float4 qv[QLDS ? 1 : NCH];
    if constexpr (QLDS) {
        float4 t[NCH / 4];                       // the wave's 64 lanes load the 16 * NCH chunks once (zero beyond d)
#pragma unroll
        for (int j = 0; j < NCH / 4; ++j) {
            const int c = j * 64 + lane, e0 = 4 * c;
            const float *qp = Q + q * q_stride;
            t[j].x = (e0 + 0 < iv.d) ? qp[e0 + 0] : 0.f; t[j].y = (e0 + 1 < iv.d) ? qp[e0 + 1] : 0.f;
            t[j].z = (e0 + 2 < iv.d) ? qp[e0 + 2] : 0.f; t[j].w = (e0 + 3 < iv.d) ? qp[e0 + 3] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NCH / 4; ++j) qs[j * 64 + lane] = t[j];
        __syncthreads();
    } else {
        load_query<NCH>(qv, Q + q * q_stride, iv.d, l16);
    }
    const uint32_t stride_b = (uint32_t)iv.stride * 4u;
    const int step = 4 * UB * gridDim.y;
    int base = blockIdx.y * 4 * UB;
    int32_t idn[UB];                             // the ids of the batch about to be loaded
#pragma unroll
    for (int u = 0; u < UB; ++u) { const int j = base + 4 * u + r; idn[u] = base < m ? ids[q * m + (j < m ? j : base)] : iv.id_base; }
    for (; base < m; base += step) {
        float4 v[UB][NCH];
#pragma unroll
        for (int u = 0; u < UB; ++u) {           // past the end: a row already in flight is read again (see idn)
            const char *row = reinterpret_cast<const char *>(iv.X) + (uint64_t)(uint32_t)(idn[u] - iv.id_base) * stride_b;   // 64-bit: no table-size limit
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = i * 16 + l16;
                v[u][i] = *reinterpret_cast<const float4 *>(row + 16u * (uint32_t)(c < iv.nchunks ? c : 0));
            }
        }
        const int nbase = base + step;           // the next batch's ids travel with this batch's rows
        if (nbase < m) {
#pragma unroll
            for (int u = 0; u < UB; ++u) { const int j = nbase + 4 * u + r; idn[u] = ids[q * m + (j < m ? j : nbase)]; }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                float4 z = v[u][i];
                const float4 qi = QLDS ? qs[i * 16 + l16] : qv[QLDS ? 0 : i];
                const bool cv = (i * 16 + l16) < iv.nchunks;
                z.x = cv ? z.x : 0.f; z.y = cv ? z.y : 0.f; z.z = cv ? z.z : 0.f; z.w = cv ? z.w : 0.f;
                if (METRIC == 0) {
                    float dx = z.x - qi.x; acc = __builtin_fmaf(dx, dx, acc);
                    float dy = z.y - qi.y; acc = __builtin_fmaf(dy, dy, acc);
                    float dz = z.z - qi.z; acc = __builtin_fmaf(dz, dz, acc);
                    float dw = z.w - qi.w; acc = __builtin_fmaf(dw, dw, acc);
                } else {
                    acc = __builtin_fmaf(z.x, qi.x, acc);
                    acc = __builtin_fmaf(z.y, qi.y, acc);
                    acc = __builtin_fmaf(z.z, qi.z, acc);
                    acc = __builtin_fmaf(z.w, qi.w, acc);
                }
            }
            acc = reduce16(acc);
            const int j = base + 4 * u + r;
            if (j < m && l16 == 0) out[q * m + j] = key_to_dist<METRIC>(dist_to_key<METRIC>(acc));
        }
    }
}

} // namespace hnsw_dev
