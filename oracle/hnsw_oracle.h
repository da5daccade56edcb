/*
 * hnsw_oracle.h -- CPU restatement of ocaml-hnsw's search path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for the MI355X HNSW search path.  It restates, in plain C, the
 * algorithm of the reference (lehy/ocaml-hnsw) -- it is NOT part of the product and nothing in
 * ocaml-hnsw_amd/ may include, link or call it.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it.
 *
 * Pinning status:
 *   - search_one / search_k / select_neighbours: PINNED by the reference's 34 inline known-answer
 *     tests (lib/ohnsw.ml:514-534, 593-644, 665-764), transcribed in tests/golden/.
 *   - L2 arithmetic: UNPINNED.  The reference computes sqrt(Lacaml.S.Vec.ssqr_diff a b)
 *     (lib/hnsw.ml:814, lib/ohnsw.ml:899); lacaml is an un-vendored, un-versioned dependency and
 *     no reference test exercises it.  Contract = north-star tolerance 1e-5 relative.
 *   - tie order on the imperative path: UNPINNED (Core_kernel.Heap, un-vendored).  The functor
 *     path's tie order is pinned by the in-tree pairing heap (lib/hnsw_algo.ml:17-66).
 *   - The reference itself (OCaml + jbuilder + ppx_jane + core_kernel + lacaml) cannot be built
 *     here: no OCaml toolchain in the image.  There is no oracle/_ref.
 *
 * Node ids are 0-based everywhere in the oracle (Ohnsw convention, lib/ohnsw.ml:159-161); the
 * 1-based ids of the functor path (lib/hnsw.ml:325) are an API-boundary offset (id_base).
 */
#ifndef HNSW_ORACLE_H
#define HNSW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- distance spaces ------------------------------------------------------------------ */
enum { OG_SCALAR_ABS = 0, /* |a-b| on scalars: the reference tests' distance, lib/ohnsw.ml:361 */
       OG_L2 = 1,         /* sqrt(sum (a_i-b_i)^2): lib/hnsw.ml:814, lib/ohnsw.ml:899         */
       OG_IP = 2 };       /* 1 - <a,b>: NEW (no inner product in the reference)               */

enum { OG_SEQ_F32 = 0,  /* sequential fp32 accumulation, sqrt in double (Lacaml-style stub)       */
       OG_F64 = 1,      /* everything in double                                                   */
       OG_TREE16 = 2 }; /* fp32, 16 strided partial sums of fmaf chains + stride-halving tree:
                           the summation order of the HIP kernel (bit-parity mode)              */

typedef struct og_space {
    int32_t kind;
    int32_t arith;
    const float *X;      /* [n][stride] vectors (OG_L2 / OG_IP)        */
    int64_t n;
    int32_t d;
    int64_t stride;      /* row stride in floats                       */
    const double *S;     /* [n] scalar values (OG_SCALAR_ABS)          */
    uint64_t n_calls;    /* distance-call counter, lib/hnsw.ml:732-735 */
} og_space;

double og_distance_raw(const og_space *sp, const void *a, const void *b);
/* squared-L2 / dot accumulations exposed for golden-vector generation */
float og_l2sq_tree16(const float *a, const float *b, int32_t d);
float og_dot_tree16(const float *a, const float *b, int32_t d);
float og_l2sq_seq(const float *a, const float *b, int32_t d);
double og_l2sq_f64(const float *a, const float *b, int32_t d);

/* ---- flattened graph (same layout as the C ABI of include/hnsw_mi355x.h) ---------------- */
typedef struct og_graph og_graph;

/* upper_* are arrays of length max_layer (entry l-1 describes layer l). nodes are 0-based. */
og_graph *og_graph_create(int64_t n, int32_t max_layer, int64_t entry_point,
                          int32_t stride0, const int32_t *deg0, const int32_t *nbr0,
                          int32_t strideU, const int64_t *upper_n,
                          const int64_t *const *upper_nodes, const int32_t *const *upper_deg,
                          const int32_t *const *upper_nbr);
void og_graph_destroy(og_graph *g);
int64_t og_graph_num_nodes(const og_graph *g);
int32_t og_graph_max_layer(const og_graph *g);
int64_t og_graph_entry_point(const og_graph *g);
/* returns degree, writes up to cap neighbour ids */
int32_t og_graph_adjacent(const og_graph *g, int32_t layer, int64_t node, int32_t *out, int32_t cap);

/* Hgraph.Stats.compute for one layer (lib/hnsw.ml:353-375): layer size, min / max / mean of the neighbour-list lengths
 * and the isolated nodes in the reference's list order (descending id: consed during an ascending fold).  Returns the
 * number of isolated nodes (-1: no such layer); at most isolated_cap ids are written.  An empty layer yields
 * min 1000000, max -1, mean nan, as the OCaml fold does. */
int64_t og_layer_stats(const og_graph *g, int32_t layer, int64_t *num_nodes, int32_t *min_degree,
                       int32_t *max_degree, double *mean_degree, int64_t *isolated, int64_t isolated_cap);

/* ---- tie modes --------------------------------------------------------------------------- */
enum { OG_TIES_HEAP = 0,       /* heap-implementation-defined (pairing heap of hnsw_algo.ml:17-66) */
       OG_TIES_CANONICAL = 1 };/* total order (distance, id) in both heaps                         */

typedef struct og_counters {
    uint64_t n_dist; /* distance evaluations (unit of work of the roofline) */
    uint64_t n_hops; /* candidates expanded on layer 0                                 */
    uint64_t n_hops_upper;
    uint64_t n_dist_upper; /* of n_dist, those spent before the layer-0 loop starts (descent + the
                              evaluation of the layer-0 start node, lib/ohnsw.ml:865-871)        */
} og_counters;

/* ---- imperative path: lib/ohnsw.ml -------------------------------------------------------- */
int64_t og_ohnsw_search_one(const og_graph *g, int32_t layer, og_space *sp, int64_t start,
                            const void *target, int32_t paper_variant);
/* search_k: returns number of results; results ascending by distance (result_minq order). */
int32_t og_ohnsw_search_k(const og_graph *g, int32_t layer, og_space *sp,
                          const int64_t *start_nodes, int32_t n_start, const void *target,
                          int32_t k, int32_t ties, int64_t *out_nodes, double *out_dist,
                          og_counters *ctr);
/* select_neighbours: candidates form a MinQueue keyed by distance to target; returns count,
 * out in selection order (nearest first). */
int32_t og_ohnsw_select_neighbours(og_space *sp, const int64_t *cand, int32_t n_cand,
                                   const void *target, int32_t num_neighbours, int32_t ties,
                                   int64_t *out_nodes);
/* knn: descent + search_k(ef) on layer 0; writes the first k of the ef results, ascending.
 * ef == k is the reference's own call shape (lib/ohnsw.ml:859-875). Returns count or -1 if
 * the graph is empty ("knn: empty hgraph", lib/ohnsw.ml:862). */
int32_t og_ohnsw_knn(const og_graph *g, og_space *sp, const void *target, int32_t ef, int32_t k,
                     int32_t ties, int64_t *out_nodes, double *out_dist, og_counters *ctr);
/* knn_batch_bigarray (lib/ohnsw.ml:877-897): ids [nq][k] filled -1, dist [nq][k] filled NaN. */
int32_t og_ohnsw_knn_batch(const og_graph *g, og_space *sp, const float *Q, int64_t nq,
                           int64_t q_stride, int32_t ef, int32_t k, int32_t ties,
                           int32_t *out_ids, float *out_dist, uint32_t *out_ndist,
                           uint32_t *out_nhops);

/* the same, with the per-query split of the evaluations: out_ndist_upper (optional) = those spent before
 * the layer-0 loop (what a separate descent pass does) */
int32_t og_ohnsw_knn_batch_split(const og_graph *g, og_space *sp, const float *Q, int64_t nq,
                                 int64_t q_stride, int32_t ef, int32_t k, int32_t ties,
                                 int32_t *out_ids, float *out_dist, uint32_t *out_ndist,
                                 uint32_t *out_nhops, uint32_t *out_ndist_upper);

/* the same over `nthreads` host threads (CPU baseline on all cores; the reference is 1 thread) */
int32_t og_ohnsw_knn_batch_mt(const og_graph *g, og_space *sp, const float *Q, int64_t nq,
                              int64_t q_stride, int32_t ef, int32_t k, int32_t ties, int32_t nthreads,
                              int32_t *out_ids, float *out_dist);

/* ---- functor path: lib/hnsw_algo.ml + lib/hnsw.ml ----------------------------------------- */
int64_t og_functor_search_one(const og_graph *g, int32_t layer, og_space *sp, int64_t start,
                              const void *target, int32_t ties, double *out_dist,
                              og_counters *ctr);
/* Search.search: returns |W|; outputs W sorted ascending by (distance, then heap order). */
int32_t og_functor_search(const og_graph *g, int32_t layer, og_space *sp,
                          const int64_t *start_nodes, int32_t n_start, const void *target,
                          int32_t ef, int32_t ties, int64_t *out_nodes, double *out_dist,
                          og_counters *ctr);
/* Knn.knn: bug_compat_farthest_k=1 reproduces Nearest.nearest_k (lib/hnsw.ml:522-525), which
 * returns the k FARTHEST of the ef results when ef > k. 0 returns W[0..k) ascending. */
int32_t og_functor_knn(const og_graph *g, og_space *sp, const void *target, int32_t ef,
                       int32_t k, int32_t ties, int32_t bug_compat_farthest_k,
                       int64_t *out_nodes, double *out_dist, og_counters *ctr);
/* MakeBatch.knn_batch (lib/hnsw.ml:769-777): distances only, [nq][k] filled +inf. */
int32_t og_functor_knn_batch(const og_graph *g, og_space *sp, const float *Q, int64_t nq,
                             int64_t q_stride, int32_t ef, int32_t k, int32_t ties,
                             int32_t bug_compat_farthest_k, float *out_dist, int32_t *out_ids);
/* SelectNeighbours.select_neighbours (lib/hnsw_algo.ml:572-609). cand_dist are the candidates'
 * distances to the base node; cand_degree their current degree (for do_not_isolate). */
int32_t og_functor_select_neighbours(og_space *sp, const int64_t *cand, const double *cand_dist,
                                     const int32_t *cand_degree, int32_t n_cand,
                                     int32_t num_neighbours, int32_t do_not_isolate,
                                     int32_t ties, int64_t *out_nodes);

/* ---- test-graph generator: restatement of Ohnsw.insert / build_batch_bigarray --------------
 * (lib/ohnsw.ml:766-857).  Own seeded RNG: graph identity with an OCaml build is neither
 * possible (OCaml Random) nor needed; search parity is defined GIVEN a graph. */
typedef struct og_builder og_builder;
og_builder *og_build_ohnsw(og_space *sp, int64_t n, int32_t num_connections,
                           int32_t num_nodes_search_construction, uint64_t seed, int32_t ties);
void og_builder_destroy(og_builder *b);
int32_t og_builder_max_layer(const og_builder *b);
int64_t og_builder_entry_point(const og_builder *b);
int64_t og_builder_layer_count(const og_builder *b, int32_t layer); /* nodes with deg>0 or level>=layer */
/* export layer 0 as fixed-stride table; returns max degree seen (error if > stride: -1) */
int32_t og_builder_export_layer0(const og_builder *b, int32_t stride, int32_t *deg0, int32_t *nbr0);
/* export upper layer l (>=1): nodes whose level >= l. returns n_l; arrays sized by layer_count */
int64_t og_builder_export_upper(const og_builder *b, int32_t layer, int32_t stride,
                                int64_t *nodes, int32_t *deg, int32_t *nbr);
/* symmetric-link invariant of lib/ohnsw.ml:217-225 on every layer; 1 = holds */
int32_t og_builder_invariant(const og_builder *b);

/* ---- brute force ground truth (benchmark/dataset.ml:15-30) and recall (:105-127) ---------- */
void og_brute_force_knn(og_space *sp, const float *Q, int64_t nq, int64_t q_stride, int32_t k,
                        int32_t *out_ids, float *out_dist);
double og_recall_distance_threshold(const float *expected, const float *got, int64_t nq,
                                    int32_t k, double epsilon);

/* Visited (lib/ohnsw.ml:256-268) exposed for the epoch-overflow fixture (:285-295). */
typedef struct og_visited og_visited;
og_visited *og_visited_create(int64_t n);
void og_visited_destroy(og_visited *v);
int32_t og_visited_mem(const og_visited *v, int64_t node);
void og_visited_add(og_visited *v, int64_t node);
void og_visited_clear(og_visited *v);
int64_t og_visited_card(const og_visited *v);
void og_visited_set_epoch(og_visited *v, int64_t epoch);

#ifdef __cplusplus
}
#endif
#endif
