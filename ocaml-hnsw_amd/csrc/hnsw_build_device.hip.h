// hnsw_build_device.hip.h -- gfx950 kernels of the batched graph builder.
//
// The reference builds one node at a time (Ohnsw.insert, lib/ohnsw.ml:766-837; Hnsw_algo
// insert_knowing_node, lib/hnsw_algo.ml:629-713).  Here a batch of nodes is inserted against the
// graph as it stood at batch start, with the same per-node steps:
//   K1 build_search   descent with search_one to level+1 (:785-789), then search_k with
//                     efConstruction on every layer <= level, each seeded with the whole W of the
//                     layer above (:806-816)
//   K2 build_select   select_neighbours (lib/ohnsw.ml:647-663; M on upper layers, 2M on layer 0,
//                     :818-819) and set_connections_for_new_node's forward half (:198-199)
//   (sort)            the back-links (neighbour, new node), grouped per neighbour
//   K4 build_merge    back-links (:200-202) and, for a neighbour pushed past its cap, the shrink
//                     by select_neighbours with the neighbour as base (:821-829)
//   K5 build_unlink   the symmetric removals of Graph.set_connections (:190-192)
// Rows may carry -1 holes between batches (the search kernels skip them); build_compact closes
// them at the end.  Everything is deterministic: no result depends on atomic arrival order.
#pragma once

#include "hnsw_device.hip.h"

namespace hnsw_dev {

struct BuildView {
    IndexView iv;            // tables being built (nbr0 / nbrU are written through the pointers below)
    int32_t *nbr0_w;
    int32_t *nbrU_w;
    int32_t efc;             // efConstruction
    int32_t cand_stride;     // >= efc
    int32_t vt_bits;
};

struct BatchView {
    const int32_t *nodes;    // [B] node ids of the batch
    const int32_t *rec_of;   // [B][lcap] record index of (batch slot, layer) or -1
    int32_t B, lcap;
    int32_t cur_max_layer;   // graph state at batch start
    int32_t entry;
    int32_t *cand_id;        // [nrec][cand_stride]
    uint32_t *cand_key;      // [nrec][cand_stride]
    int32_t *cand_cnt;       // [nrec]
};

__device__ __forceinline__ int32_t *row_ptr_w(const BuildView &bv, int layer, int node, int &width) {
    if (layer == 0) { width = bv.iv.S0; return bv.nbr0_w + (int64_t)node * bv.iv.S0; }
    width = bv.iv.SU;
    return bv.nbrU_w + ((int64_t)bv.iv.upper_off[node] + (layer - 1)) * bv.iv.SU;
}

// ---- K1: construction search ---------------------------------------------------------------------
template <int NCH, int RB, int NSLOT, int METRIC>
__global__ void __launch_bounds__(64)
build_search_kernel(const BuildView bv, const BatchView bt) {
    extern __shared__ uint32_t lds[];
    const IndexView &iv = bv.iv;
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    if (i >= bt.B) return;
    const WaveCtx cx = make_ctx(lds, bv.vt_bits, lane);
    const int q = bt.nodes[i];
    const int lvl = iv.upper_lvl[q];

    float4 qv[NCH];
    load_row<NCH>(qv, iv, q, cx.l16);
    visited_clear(cx);                                         // lib/ohnsw.ml:780
    uint32_t n_dist = 0, n_hops = 0, status = 0;

    int cur = bt.entry;
    if (lane == 0) cx.cand_id[0] = cur;
    __syncthreads();
    eval_candidates<NCH, RB, METRIC>(iv, qv, cx.cand_id, cx.cand_key, cx.trash, 1, cx.r, cx.l16);
    __syncthreads();
    uint32_t cur_key = cx.cand_key[0];

    const int top = bt.cur_max_layer;
    if (lvl < top) greedy_descend<NCH, RB, METRIC>(iv, qv, top, lvl + 1, cur, cur_key, cx, n_dist); // :785-789

    WList<NSLOT> w;
    wlist_init(w, bv.efc, lane);
    wlist_insert(w, cur_key, (uint32_t)cur, lane, cx.ovf, status);                         // :802
    { uint32_t hw; (void)visited_mem(cx, (uint32_t)cur, hw); visited_add_masked(cx, (uint32_t)cur, hw, lane == 0); }
    __syncthreads();

    for (int layer = (lvl < top ? lvl : top); layer >= 0; --layer) {                         // :806
        // the start queue of this layer is the whole W of the layer above (:814-816): every
        // member becomes unexpanded again.  The visited cache is NOT cleared: a node evaluated
        // on an upper layer is either in W or rejected for good (max(W) never grows).
#pragma unroll
        for (int s = 0; s < NSLOT; ++s)
            if (w.hi[s] < DUMMY_HI && w.lo[s] != PAD_LO) w.lo[s] &= ~1u;
        w.ovf_cnt = 0;
        search_layer<NCH, RB, NSLOT, METRIC>(iv, qv, layer, w, bv.efc, cx, n_dist, n_hops, status); // :811
        const int rec = bt.rec_of[(int64_t)i * bt.lcap + layer];
        if (rec >= 0) {
            const int wbase = NSLOT * 64 - bv.efc;
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                const int idx = s * 64 + lane - wbase;
                const uint32_t hi = w.hi[s];
                if (idx >= 0 && hi < DUMMY_HI) {
                    bt.cand_id[(int64_t)rec * bv.cand_stride + idx] = (int32_t)key_id(w.lo[s]);
                    bt.cand_key[(int64_t)rec * bv.cand_stride + idx] = hi;
                }
            }
            const int cntw = wlist_count(w);
            if (lane == 0) bt.cand_cnt[rec] = cntw;
        }
    }
}

// ---- select_neighbours heuristic (lib/ohnsw.ml:647-663) -------------------------------------------
// Candidates (LDS c_id / c_key, ascending by (distance to the base, id)) are popped nearest first;
// e is kept iff e.d < distance(value kept, value e) for every kept neighbour (:657-659); stops at R
// kept (:660).  Kept ids go to LDS k_id in selection order.  Returns the count.
template <int NCH, int METRIC>
__device__ __forceinline__ int select_heuristic(const IndexView &iv, const int32_t *c_id,
                                                const uint32_t *c_key, int nc, int R, int32_t *k_id,
                                                int lane, int kept0 = 0) {
    const int r = lane >> 4, l16 = lane & 15;
    const float4 *X4 = reinterpret_cast<const float4 *>(iv.X);
    const int64_t stride4 = iv.stride >> 2;
    int kept = kept0;   // k_id[0..kept0) already holds forced neighbours (do_not_isolate)
    for (int i = 0; i < nc && kept < R; ++i) {
        const int c = c_id[i];
        const uint32_t ckey = c_key[i];
        float4 cv[NCH];
        load_row<NCH>(cv, iv, c, l16);
        bool reject = false;
        for (int base = 0; base < kept && !reject; base += 4) {
            const int j = base + r;
            const bool valid = j < kept;
            const int s = valid ? k_id[j] : 0;
            const float4 *row = X4 + (int64_t)s * stride4;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < NCH; ++t) {
                const int ch = t * 16 + l16;
                float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid && ch < iv.nchunks) z = row[ch];
                if (METRIC == 0) {
                    float dx = z.x - cv[t].x; acc = __builtin_fmaf(dx, dx, acc);
                    float dy = z.y - cv[t].y; acc = __builtin_fmaf(dy, dy, acc);
                    float dz = z.z - cv[t].z; acc = __builtin_fmaf(dz, dz, acc);
                    float dw = z.w - cv[t].w; acc = __builtin_fmaf(dw, dw, acc);
                } else {
                    acc = __builtin_fmaf(z.x, cv[t].x, acc);
                    acc = __builtin_fmaf(z.y, cv[t].y, acc);
                    acc = __builtin_fmaf(z.z, cv[t].z, acc);
                    acc = __builtin_fmaf(z.w, cv[t].w, acc);
                }
            }
            acc = reduce16(acc);
            const uint32_t key_cs = dist_to_key<METRIC>(acc);
            reject = ballot(valid && !(ckey < key_cs)) != 0ull;      // :657-659, strict
        }
        if (!reject) {
            __syncthreads();
            if (lane == 0) k_id[kept] = c;
            __syncthreads();
            kept++;
        }
    }
    return kept;
}

// ---- K2: select for the new nodes, write their rows, emit back-link keys ---------------------------
struct SelectArgs {
    const int32_t *rec_node; // [nrec] node of each record
    int32_t rec_begin, rec_end, layer, R;
    uint64_t *edges;         // [(rec_end-rec_begin)][R] keys (neighbour << 32 | new node), ~0 = none
};

template <int NCH, int METRIC>
__global__ void __launch_bounds__(64)
build_select_kernel(const BuildView bv, const BatchView bt, const SelectArgs sa) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    const int rec = sa.rec_begin + blockIdx.x;
    if (rec >= sa.rec_end) return;
    int32_t *c_id = reinterpret_cast<int32_t *>(lds);                 // [cand_stride]
    uint32_t *c_key = lds + bv.cand_stride;                           // [cand_stride]
    int32_t *k_id = reinterpret_cast<int32_t *>(lds + 2 * bv.cand_stride); // [64]
    const int q = sa.rec_node[rec];
    const int nc = bt.cand_cnt[rec];
    for (int j = lane; j < nc; j += 64) {
        c_id[j] = bt.cand_id[(int64_t)rec * bv.cand_stride + j];
        c_key[j] = bt.cand_key[(int64_t)rec * bv.cand_stride + j];
    }
    __syncthreads();
    const int kept = select_heuristic<NCH, METRIC>(bv.iv, c_id, c_key, nc, sa.R, k_id, lane);
    __syncthreads();
    int width;
    int32_t *row = row_ptr_w(bv, sa.layer, q, width);
    // the reference conses: Neighbours.iter order is the reverse of the selection order
    if (lane < width) row[lane] = lane < kept ? k_id[kept - 1 - lane] : -1;
    uint64_t *e = sa.edges + (int64_t)(rec - sa.rec_begin) * sa.R;
    if (lane < sa.R) e[lane] = lane < kept ? (((uint64_t)(uint32_t)k_id[lane] << 32) | (uint32_t)q) : ~0ull;
}

// ---- K4: back-links and shrink ----------------------------------------------------------------------
struct MergeArgs {
    const uint64_t *edges;   // sorted ascending; ~0 entries last
    int32_t n_edges, layer, R;
    uint64_t *removals;      // (node << 32 | neighbour to drop from node's row)
    uint32_t *rem_cnt;
    uint32_t rem_cap;
};

// A neighbour's pending new links are merged 64 at a time (a hub can collect more than 64 in one batch of
// up to 8192 insertions): each chunk is consed onto the row as the sequential builder would have consed
// those nodes one after the other, and the row is re-selected whenever it outgrows its cap.
// Removals that do not fit the buffer are counted in rem_cnt[1]: the host then repeats the build with a
// larger buffer (nothing is ever dropped silently).
template <int NCH, int RB, int METRIC>
__global__ void __launch_bounds__(64)
build_merge_kernel(const BuildView bv, const MergeArgs ma) {
    __shared__ int32_t u_id[128];     // union: pending new nodes + old row
    __shared__ uint32_t u_key[128];
    __shared__ int32_t s_id[128];     // sorted union
    __shared__ uint32_t s_key[128];
    __shared__ int32_t k_id[64];
    __shared__ uint32_t trash[64];
    const IndexView &iv = bv.iv;
    const int lane = threadIdx.x;
    const int e = blockIdx.x;
    if (e >= ma.n_edges) return;
    const uint64_t key = ma.edges[e];
    if (key == ~0ull) return;
    const int nb = (int)(key >> 32);
    if (e > 0 && (int)(ma.edges[e - 1] >> 32) == nb) return;          // not the head of its segment

    // the segment of nb: its pending new links, ascending new-node id
    int np_total = 0;
    for (int base = e;; base += 64) {
        const int j = base + lane;
        const bool mine = j < ma.n_edges && ma.edges[j] != ~0ull && (int)(ma.edges[j] >> 32) == nb;
        const int c = popc(ballot(mine));
        np_total += c;
        if (c < 64) break;
    }
    int width;
    int32_t *row = row_ptr_w(bv, ma.layer, nb, width);
    float4 qv[NCH];
    bool have_q = false;

    for (int c0 = 0; c0 < np_total; c0 += 64) {
        const int np = np_total - c0 < 64 ? np_total - c0 : 64;
        __syncthreads();
        if (lane < np) u_id[lane] = (int)(ma.edges[e + c0 + lane] & 0xFFFFFFFFu);
        const int old = lane < width ? row[lane] : -1;
        const bool ovalid = old >= 0;
        const uint64_t om = ballot(ovalid);
        const int no = popc(om);
        const int opos = popc(om & ((1ull << lane) - 1ull));
        __syncthreads();
        if (ovalid) u_id[np + opos] = old;
        __syncthreads();
        const int nu = np + no;

        if (nu <= ma.R) {
            // Neighbours.add conses (lib/ohnsw.ml:116-118): the latest link comes first
            int v = -1;
            if (lane < np) v = u_id[np - 1 - lane];
            else if (lane < nu) v = u_id[lane];
            __syncthreads();
            if (lane < width) row[lane] = v;
        } else {
            // shrink (lib/ohnsw.ml:823-828): distances of the union to nb, ascending (d, id), heuristic
            if (!have_q) { load_row<NCH>(qv, iv, nb, lane & 15); have_q = true; }
            for (int base = 0; base < nu; base += 64) {
                const int m = nu - base < 64 ? nu - base : 64;
                eval_candidates<NCH, RB, METRIC>(iv, qv, u_id + base, u_key + base, trash, m, lane >> 4, lane & 15);
            }
            __syncthreads();
            for (int j = lane; j < nu; j += 64) {                               // rank sort, nu <= 128
                const uint64_t mk = ((uint64_t)u_key[j] << 32) | (uint32_t)u_id[j];
                int rank = 0;
                for (int t = 0; t < nu; ++t) {
                    const uint64_t ok = ((uint64_t)u_key[t] << 32) | (uint32_t)u_id[t];
                    rank += ok < mk;
                }
                s_id[rank] = u_id[j];
                s_key[rank] = u_key[j];
            }
            __syncthreads();
            const int kept = select_heuristic<NCH, METRIC>(iv, s_id, s_key, nu, ma.R, k_id, lane);
            __syncthreads();
            if (lane < width) row[lane] = lane < kept ? k_id[kept - 1 - lane] : -1;
            // dropped members lose their link to nb (Graph.set_connections step 2, lib/ohnsw.ml:190-192)
            for (int j = lane; j < nu; j += 64) {
                const int x = s_id[j];
                bool in = false;
                for (int t = 0; t < kept; ++t) in = in || (k_id[t] == x);
                if (!in) {
                    const uint32_t slot = atomicAdd(ma.rem_cnt, 1u);
                    if (slot < ma.rem_cap) ma.removals[slot] = ((uint64_t)(uint32_t)x << 32) | (uint32_t)nb;
                    else atomicMax(ma.rem_cnt + 1, slot + 1u);                   // the host repeats the build with more room
                }
            }
        }
        if (c0 + 64 < np_total) { __threadfence(); __syncthreads(); }           // the next chunk reads the row just written
    }
}

// ---- sequential linking: the link step of Ohnsw.insert exactly (lib/ohnsw.ml:820-829) ---------------
// Used for batches of ONE node (max_batch = 1, and the first insertions of every build): one wave walks
// the new node's neighbours in list order, as the reference does, so the result is the reference's graph
// link for link, list order included:
//   * set_connections_for_new_node (:198-202): the new node q is consed onto every neighbour's list FIRST.
//     A row has no room for a (cap + 1)-th entry, so that cons is kept virtual (vq[i]) until the neighbour
//     is visited or one of its links is removed -- every reader below sees [q] + row;
//   * Neighbours.iter neighbours (:821, a snapshot of q's list): a neighbour whose list is longer than the
//     cap is re-selected from the min-queue of its neighbours (:791-798, :823-827) and set_connections
//     (:182-196) stores the reduced list and removes the neighbour from every dropped node's list;
//   * Neighbours.remove (:119-124) rebuilds the list by consing: the survivors end up in REVERSED order.
// Rows stay compacted (no holes) in this mode.
struct LinkArgs {
    int32_t q, layer, R;
};

template <int NCH, int RB, int METRIC>
__global__ void __launch_bounds__(64)
build_link_sequential_kernel(const BuildView bv, const LinkArgs la) {
    __shared__ int32_t it[64];        // snapshot of q's list
    __shared__ int32_t vq[64];        // 1 while q still heads it[i]'s list only virtually
    __shared__ int32_t u_id[128];
    __shared__ uint32_t u_key[128];
    __shared__ int32_t s_id[128];
    __shared__ uint32_t s_key[128];
    __shared__ int32_t k_id[64];
    __shared__ uint32_t trash[64];
    const IndexView &iv = bv.iv;
    const int lane = threadIdx.x;
    const int q = la.q, R = la.R;
    int wq;
    int32_t *qrow = row_ptr_w(bv, la.layer, q, wq);
    const int qv0 = lane < wq ? qrow[lane] : -1;
    const int len = popc(ballot(qv0 >= 0));                       // K2 wrote the list compacted, in list order
    it[lane] = qv0;
    vq[lane] = qv0 >= 0 ? 1 : 0;
    __syncthreads();

    // list of node x as the reference sees it now -> u_id[0..n): [q] (if still virtual) + its row
    auto read_list = [&](int x, int &virt_at) -> int {
        const uint64_t vm = ballot(it[lane] == x && vq[lane] != 0);
        virt_at = vm ? __builtin_ctzll(vm) : -1;
        const int hasq = vm ? 1 : 0;
        int w;
        const int32_t *row = row_ptr_w(bv, la.layer, x, w);
        const int old = lane < w ? row[lane] : -1;
        const uint64_t om = ballot(old >= 0);
        const int opos = popc(om & ((1ull << lane) - 1ull));
        __syncthreads();
        if (old >= 0) u_id[hasq + opos] = old;
        if (hasq && lane == 0) u_id[0] = q;
        __syncthreads();
        return hasq + popc(om);
    };
    auto write_row = [&](int x, const int32_t *src, int n, bool reversed) {
        int w;
        int32_t *row = row_ptr_w(bv, la.layer, x, w);
        const int val = lane < n ? src[reversed ? n - 1 - lane : lane] : -1;
        __syncthreads();
        if (lane < w) row[lane] = val;
    };

    for (int i = 0; i < len; ++i) {
        const int nb = it[i];
        int virt_at;
        const int nu = read_list(nb, virt_at);
        if (nu <= R) {                                              // :823 not over the cap
            if (virt_at >= 0) {                                     // the cons of :202 becomes physical
                write_row(nb, u_id, nu, false);
                if (lane == 0) vq[virt_at] = 0;
            }
            __threadfence();
            __syncthreads();
            continue;
        }
        // :824-827 min_queue_of_neighbours + select_neighbours: distances of the list to nb, ascending (d, id)
        float4 nv[NCH];
        load_row<NCH>(nv, iv, nb, lane & 15);
        for (int base = 0; base < nu; base += 64) {
            const int m = nu - base < 64 ? nu - base : 64;
            eval_candidates<NCH, RB, METRIC>(iv, nv, u_id + base, u_key + base, trash, m, lane >> 4, lane & 15);
        }
        __syncthreads();
        for (int j = lane; j < nu; j += 64) {                       // rank sort, nu <= 65
            const uint64_t mk = ((uint64_t)u_key[j] << 32) | (uint32_t)u_id[j];
            int rank = 0;
            for (int t = 0; t < nu; ++t) rank += ((((uint64_t)u_key[t] << 32) | (uint32_t)u_id[t]) < mk);
            s_id[rank] = u_id[j];
            s_key[rank] = u_key[j];
        }
        __syncthreads();
        const int kept = select_heuristic<NCH, METRIC>(iv, s_id, s_key, nu, R, k_id, lane);
        __syncthreads();
        write_row(nb, k_id, kept, true);                            // the reference conses: list order = reverse selection order
        if (lane == 0 && virt_at >= 0) vq[virt_at] = 0;
        __threadfence();
        __syncthreads();
        // :190-192 every dropped node loses its link to nb; Neighbours.remove reverses what is left
        for (int j = 0; j < nu; ++j) {
            const int x = s_id[j];
            bool in = false;
            for (int t = 0; t < kept; ++t) in = in || (k_id[t] == x);
            if (in) continue;                                       // uniform
            int xv;
            const int nx = read_list(x, xv);
            // survivors, in list order, then reversed.  The list holds up to 65 entries (the virtual cons of q in
            // front of a full 64-entry row): entry 64 is taken separately, it is the last in list order
            const int e = lane < nx ? u_id[lane] : -1;
            const bool keep = lane < nx && e != nb;
            const uint64_t sm = ballot(keep);
            const int spos = popc(sm & ((1ull << lane) - 1ull));
            int ns = popc(sm);
            const int e64 = nx > 64 ? u_id[64] : -1;                // wave-uniform
            const bool keep64 = nx > 64 && e64 != nb;
            __syncthreads();
            if (keep) s_key[spos] = (uint32_t)e;                    // s_key is free here: reuse it as the survivor list
            if (keep64 && lane == 0) s_key[ns] = (uint32_t)e64;
            ns += keep64 ? 1 : 0;
            __syncthreads();
            int w;
            int32_t *xrow = row_ptr_w(bv, la.layer, x, w);
            const int val = lane < ns ? (int32_t)s_key[ns - 1 - lane] : -1;
            __syncthreads();
            if (lane < w) xrow[lane] = val;
            if (lane == 0 && xv >= 0) vq[xv] = 0;
            __threadfence();
            __syncthreads();
        }
    }
}

// ---- select_neighbours as an operator (lib/ohnsw.ml:647-663; lib/hnsw_algo.ml:572-609) ------------
// One wave per base value: distances of the candidates to the target, ascending (distance, id)
// order (the MinQueue pop order under the canonical tie rule), then the heuristic.
struct SelectOpArgs {
    const float *targets;   // [nb][t_stride]
    int64_t t_stride;
    const int32_t *cand;    // [nb][cand_stride] 0-based ids
    const int32_t *cand_cnt;// [nb]
    int32_t cand_stride, nb, R;
    int32_t keep_all_if_few; // functor path: #candidates <= R returns them all (hnsw_algo.ml:596-599)
    const int32_t *cand_deg; // optional [nb][cand_stride]: do_not_isolate forces candidates whose
                             // degree is <= 1 into the result (hnsw_algo.ml:591-592)
    int32_t *out;           // [nb][R] selection order, -1 padded (0-based)
    int32_t *out_cnt;       // [nb]
};

template <int NCH, int RB, int METRIC>
__global__ void __launch_bounds__(64)
select_neighbours_kernel(const IndexView iv, const SelectOpArgs sa) {
    extern __shared__ uint32_t lds[];
    const int cs = sa.cand_stride;
    int32_t *c_id = reinterpret_cast<int32_t *>(lds);
    uint32_t *c_key = lds + cs;
    int32_t *s_id = reinterpret_cast<int32_t *>(lds + 2 * cs);
    uint32_t *s_key = lds + 3 * cs;
    int32_t *k_id = reinterpret_cast<int32_t *>(lds + 4 * cs);
    uint32_t *trash = lds + 4 * cs + 64;
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (b >= sa.nb) return;
    const int nall = sa.cand_cnt[b];
    // do_not_isolate: candidates with degree <= 1 go straight to the result, in candidate order;
    // the others form the heap.  (Serial over the list: this is a cold, build-time operator.)
    int nc = 0, forced = 0;
    if (sa.cand_deg) {
        for (int j = 0; j < nall; ++j) {
            const int id = sa.cand[(int64_t)b * cs + j];
            const bool f = sa.cand_deg[(int64_t)b * cs + j] <= 1;
            if (lane == 0) { if (f) k_id[forced] = id; else c_id[nc] = id; }
            forced += f; nc += !f;
        }
    } else {
        nc = nall;
        for (int j = lane; j < nc; j += 64) c_id[j] = sa.cand[(int64_t)b * cs + j];
    }
    float4 qv[NCH];
    load_query<NCH>(qv, sa.targets + (int64_t)b * sa.t_stride, iv.d, lane & 15);
    __syncthreads();
    for (int base = 0; base < nc; base += 64) {   // eval works on lists of <= 64
        const int m = nc - base < 64 ? nc - base : 64;
        eval_candidates<NCH, RB, METRIC>(iv, qv, c_id + base, c_key + base, trash, m, lane >> 4, lane & 15);
    }
    __syncthreads();
    for (int j = lane; j < nc; j += 64) {         // rank sort by (key, id)
        const uint64_t mk = ((uint64_t)c_key[j] << 32) | (uint32_t)c_id[j];
        int rank = 0;
        for (int t = 0; t < nc; ++t) rank += ((((uint64_t)c_key[t] << 32) | (uint32_t)c_id[t]) < mk);
        s_id[rank] = c_id[j];
        s_key[rank] = c_key[j];
    }
    __syncthreads();
    int kept;
    if (sa.keep_all_if_few && nall <= sa.R) {
        kept = forced + nc;
        if (lane < nc) k_id[forced + lane] = s_id[lane];
    } else {
        kept = select_heuristic<NCH, METRIC>(iv, s_id, s_key, nc, sa.R, k_id, lane, forced);
    }
    __syncthreads();
    if (lane < sa.R) sa.out[(int64_t)b * sa.R + lane] = lane < kept ? k_id[lane] : -1;
    if (lane == 0) sa.out_cnt[b] = kept;
}

// ---- K5: symmetric removals -------------------------------------------------------------------------
__global__ void build_unlink_kernel(const BuildView bv, const uint64_t *removals, const uint32_t *rem_cnt,
                                    uint32_t rem_cap, int layer) {
    uint32_t n = *rem_cnt;
    if (n > rem_cap) n = rem_cap;      // the excess was counted in rem_cnt[1]: the host discards this attempt
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {   // the count is only known here
        const uint64_t rk = removals[t];
        const int x = (int)(rk >> 32), nb = (int)(rk & 0xFFFFFFFFu);
        int width;
        int32_t *row = row_ptr_w(bv, layer, x, width);
        for (int j = 0; j < width; ++j)
            if (row[j] == nb) row[j] = -1;
    }
}

// ---- final: close the holes, keep row order ----------------------------------------------------------
__global__ void __launch_bounds__(256) build_compact_kernel(int32_t *rows, int64_t nrows, int width) {
    __shared__ int32_t tmp[4][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t rrow = (int64_t)blockIdx.x * 4 + wv;
    if (rrow >= nrows) return;
    int32_t *row = rows + rrow * width;
    const int v = lane < width ? row[lane] : -1;
    const uint64_t m = ballot(v >= 0);
    const int pos = popc(m & ((1ull << lane) - 1ull));
    const int cnt = popc(m);
    if (v >= 0) tmp[wv][pos] = v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < width) row[lane] = lane < cnt ? tmp[wv][lane] : -1;
}

} // namespace hnsw_dev
