/*
 * hnsw_mi355x.h -- C ABI of libhnsw_mi355x.so: the MI355X (gfx950) HNSW search path.
 *
 * Drop-in boundary for ONE path of lehy/ocaml-hnsw: batched k-NN search = greedy upper-layer
 * descent + ef-bounded best-first expansion on layer 0.  The reference has no FFI of its own
 * (it is 100 % OCaml); each entry point below names the OCaml function whose BODY it replaces
 * while the OCaml signature stays (see INTEGRATION.md for the ctypes binding):
 *
 *   hnsw_index_create        flatten + upload of Ohnsw.Hgraph.t (lib/ohnsw.ml:307-312, iterated
 *                            with Graph.iter_neighbours :176-180) or Hnsw.Ba.Hgraph.t
 *                            (lib/hnsw.ml:342-351, fold_layers) plus the vectors (value:
 *                            lib/ohnsw.ml:842, lib/hnsw.ml:313)
 *   hnsw_search_batch        Ohnsw.knn_batch_bigarray (lib/ohnsw.ml:877-897)
 *                            Hnsw.Ba.knn_batch / MakeBatch.knn_batch (lib/hnsw.ml:769-777)
 *   hnsw_knn                 Ohnsw.knn (lib/ohnsw.ml:859-875), Hnsw.Ba.knn (lib/hnsw.ml:763-767),
 *                            Hnsw_algo.Knn.knn (lib/hnsw_algo.ml:990-1011)
 *   hnsw_search_layer_batch  Ohnsw.search_k (lib/ohnsw.ml:543-588), Hnsw_algo.Search.search
 *                            (lib/hnsw_algo.ml:350-391): one layer, explicit start nodes
 *   hnsw_search_one_batch    Ohnsw.search_one (lib/ohnsw.ml:492-512), Search.search_one
 *                            (lib/hnsw_algo.ml:393-437)
 *   hnsw_search_batch_h2d    the same call with host queries in and DEVICE results out (per-rank step of a
 *                            one-process-per-GPU deployment: results are exchanged between devices first)
 *   hnsw_search_submit/_wait the same batch call in two halves (several batches in flight)
 *   hnsw_multi_*             the batch entry point over several GPUs from one host process (RCCL all-gather)
 *   hnsw_distance_batch      Ohnsw.distance_l2 / EuclideanBa.distance (lib/ohnsw.ml:899,
 *                            lib/hnsw.ml:809-815) as timed by bench_dist/bench_dist.ml:22-33
 *   hnsw_index_layer_stats / hnsw_index_layer_isolated   Hgraph.Stats.compute (lib/hnsw.ml:353-375)
 *   hnsw_select_neighbours_batch  Ohnsw.select_neighbours (lib/ohnsw.ml:647-663),
 *                            Hnsw_algo.SelectNeighbours.select_neighbours (lib/hnsw_algo.ml:572-609)
 *   hnsw_build               Ohnsw.build_batch_bigarray (lib/ohnsw.ml:840-857), batched on the device
 *   hnsw_host_alloc / hnsw_host_register   (nothing in the reference) page-locked query / result matrices,
 *                            which the entry points above read and write from the device in place
 *
 * Plain pointers and sizes only.  All host buffers stay owned by the caller and are not
 * retained after a call returns (OCaml Bigarrays are not moved by the GC, so they can be passed
 * for the duration of a call).  One thread at a time per handle.
 *
 * Search semantics (identical to the reference's imperative path, with its heap tie order --
 * which the reference leaves to an un-vendored library -- fixed to the total order
 * (distance, node id)):
 *   - descent on layers max_layer..1: Ohnsw.search_one_simple (lib/ohnsw.ml:492-508);
 *   - layer 0: Ohnsw.search_k (lib/ohnsw.ml:543-588) with W bounded by ef: a neighbour is
 *     accepted iff |W| < ef or d < max(W).d (strict, :574); neighbours of an expanded node are
 *     tested in adjacency-row order (:570); the search stops when the nearest unexpanded
 *     candidate is farther than max(W) (:568);
 *   - the result is W[0..k), ascending.  (Hnsw.Nearest.nearest_k, lib/hnsw.ml:522-525, returns
 *     the k FARTHEST of W when ef > k; that defect is not reproduced.)  k > ef is rejected.
 */
#ifndef HNSW_MI355X_H
#define HNSW_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an entry point, a structure or a documented behaviour changes; the Python and OCaml loaders refuse a library
 * that answers another number (hnsw_abi_version).
 *   2  (round 5) hnsw_index_locality_codes and the option "visited_blocks"; hnsw_host_unregister / hnsw_host_free wait for
 *      asynchronous readers and refuse pointers that are not theirs (HNSW_ERR_BAD_ARG); and, since version 1 was first cut:
 *      hnsw_search_batch_h2d, hnsw_host_alloc / hnsw_host_free, hnsw_index_layer_isolated, hnsw_multi_debug_counters,
 *      hnsw_index_info.row_format (was `reserved`), hnsw_index_layer_stats' values for a layer without nodes
 *      (min 1000000, max -1, mean nan: the reference's fold), Hgraph.stats()['isolated'] a list of ids
 *   3  (round 6) hnsw_index_desc.expected_ef / expected_semantics and hnsw_build_params.expected_ef / expected_semantics (the
 *      structures GROW: a caller compiled against version 2 passes too short a structure), hnsw_index_prepare; the index file
 *      (hnsw_index_save, format 2) also carries the locality codes and the prepared shapes' decisions, hnsw_index_load applies
 *      them (format 1 files still load); W in three and six key registers (ef 129..192, 257..384: no interface change);
 *      hnsw_multi_search_batch* abort their communicators when an exchange is refused part-way */
#define HNSW_ABI_VERSION 3

typedef struct hnsw_index hnsw_index;

/* Error convention: 0 = ok, negative = error; the message is in hnsw_last_error().
 * The OCaml shim maps BAD_ARG / EMPTY_INDEX / DEGREE_OVERFLOW to Invalid_argument (as
 * "knn: empty hgraph", lib/ohnsw.ml:862) and the rest to Failure. */
enum {
    HNSW_OK = 0,
    HNSW_ERR_BAD_ARG = -1,
    HNSW_ERR_EMPTY_INDEX = -2,     /* Invalid_argument "knn: empty hgraph" */
    HNSW_ERR_DEGREE_OVERFLOW = -3, /* a neighbour list longer than its row: never truncated */
    HNSW_ERR_NO_DEVICE = -4,
    HNSW_ERR_HIP = -5,
    HNSW_ERR_OOM = -6,
    HNSW_ERR_UNSUPPORTED = -7
};

enum { HNSW_METRIC_L2 = 0,  /* sqrt(sum (a_i-b_i)^2)                                     */
       HNSW_METRIC_IP = 1 };/* 1 - <a,b>  (new: the reference has Euclidean only,
                               benchmark/dataset.ml:6-13; extension point = DISTANCE sig) */

enum { HNSW_FILL_OHNSW = 0, /* missing results: id -1, distance NaN  (lib/ohnsw.ml:880-881) */
       HNSW_FILL_BA = 1 };  /* missing results: id -1, distance +inf (lib/hnsw.ml:771)      */

/* One upper layer (l >= 1) as a sparse list of the nodes present in it. */
typedef struct hnsw_layer_desc {
    int64_t n_nodes;
    const int64_t *nodes; /* [n_nodes] node ids (id_base-based), any order                  */
    const int32_t *deg;   /* [n_nodes]                                                      */
    const int32_t *nbr;   /* [n_nodes][max_degree] ids (id_base-based) in the reference's
                             iteration order (Neighbours.iter / fold); entries >= deg ignored */
} hnsw_layer_desc;

typedef struct hnsw_index_desc {
    const float *vectors; /* [n][row_stride] fp32: a Lacaml.S.mat (dim x n, Fortran layout)  */
    int64_t n;
    int32_t d;
    int64_t row_stride;   /* floats between consecutive vectors (>= d)                       */
    int32_t metric;       /* HNSW_METRIC_*                                                   */
    int32_t id_base;      /* 0 for Ohnsw (lib/ohnsw.ml:159), 1 for Hnsw.Ba (lib/hnsw.ml:325) */
    int32_t max_degree0;  /* row width of nbr0 = Mmax0 = 2M (lib/hnsw.ml:719, ohnsw.ml:818)  */
    int32_t max_degree;   /* row width of upper rows = Mmax = M                              */
    int32_t max_layer;    /* number of upper layers                                          */
    int64_t entry_point;  /* id_base-based; id_base-1 (or any value < id_base) = empty index */
    const int32_t *deg0;  /* [n]                                                             */
    const int32_t *nbr0;  /* [n][max_degree0], reference iteration order                     */
    const hnsw_layer_desc *upper; /* [max_layer]; upper[l-1] describes layer l               */
    /* Optional (0 = not known): the ef (~num_neighbours_search; Ohnsw: k) and accept rule (HNSW_SEM_*) the index will be searched
     * with.  When given, hnsw_index_create does NOW, once, what the first search with these parameters would otherwise do inside
     * the call (hnsw_index_prepare below): construction is where the reference's benchmark pays one-time costs too
     * (benchmark/benchmark.ml:66-80 before :89-96).  A value the library cannot serve (ef > 1024) is ignored here and
     * reported by the search. */
    int32_t expected_ef;
    int32_t expected_semantics;
} hnsw_index_desc;

/* Which accept rule W uses (they differ only when a neighbour is exactly as far as max(W)):
 * OHNSW   = accept iff |W| < ef or d < max(W).d (Ohnsw.search_k, lib/ohnsw.ml:574): a tied neighbour
 *           is dropped.
 * FUNCTOR = Nearest.insert_distance (lib/hnsw.ml:494-506) with the in-tree heap's merge
 *           (lib/hnsw_algo.ml:25-31): d < max(W).d replaces the maximum; d == max(W).d is answered
 *           Inserted -- the neighbour enters VisitMe and IS expanded later (lib/hnsw_algo.ml:360-364) --
 *           but W keeps the incumbent; d > max(W).d is Too_far.  Use it for Hnsw.Ba / Hnsw_algo.Knn.knn.
 * What neither mode takes from the reference is the ORDER a heap yields equal keys in (Core_kernel.Heap
 * on the imperative path: un-vendored; the in-tree pairing heap on the functor path: shape dependent):
 * equal distances are popped / evicted in node-id order.  Against the pairing-heap restatement the
 * functor mode's full ef-sized distance profile (and the ids up to the farthest class) agree on 2699 of
 * the 2700 query runs of the tie-heavy suites of tests/test_gpu_functor_ties.py; the residual (suite
 * "levels8", ef 16, query 82) is such an order effect (profiles/r02_functor_tie_agreement.txt). */
enum { HNSW_SEM_OHNSW = 0, HNSW_SEM_FUNCTOR = 1,
       /* FUNCTOR, and the result is what Hnsw.Nearest.nearest_k (lib/hnsw.ml:522-525) returns: when
        * ef > k that is the k FARTHEST members of W, nearest of those first -- the reference's
        * actual behaviour of Hnsw.Ba.knn / knn_batch with ~num_neighbours_search > ~num_neighbours,
        * a defect its author's notes acknowledge.  For callers that want that output reproduced;
        * knn entry points only. */
       HNSW_SEM_FUNCTOR_NEAREST_K = 2 };

typedef struct hnsw_search_params {
    int32_t ef;   /* ~num_neighbours_search (lib/hnsw.ml:763); Ohnsw: ef == k (ohnsw.ml:859) */
    int32_t k;    /* ~num_neighbours / ~k                                                    */
    int32_t fill; /* HNSW_FILL_*                                                             */
    int32_t semantics; /* HNSW_SEM_*                                                         */
} hnsw_search_params;

enum { HNSW_ROWS_F32 = 0,   /* the float32 rows as handed over                                              */
       HNSW_ROWS_BYTES = 2, /* lossless byte copy (every value an integer in 0..255)                         */
       HNSW_ROWS_SPLIT = 3 };/* float32 rows whose last 16 / 32 bytes past a 128-byte line are stored beside the
                               neighbour in the layer-0 adjacency (option "split_rows")                      */

typedef struct hnsw_index_info {
    int64_t n;
    int32_t d, metric, id_base, max_degree0, max_degree, max_layer;
    int64_t entry_point;
    int64_t device_bytes;      /* HBM held by the index                                       */
    int64_t row_stride_bytes;  /* padded vector row on the device                             */
    int32_t device;
    int32_t row_format;        /* HNSW_ROWS_*: what the knn searches read right now (options "byte_rows", "split_rows") */
} hnsw_index_info;

int32_t hnsw_abi_version(void);
const char *hnsw_last_error(void);
int32_t hnsw_device_count(int32_t *count);

/* Copies vectors and graph to `device` (HBM); validates ids and degrees (never truncates).
 * An empty graph (no entry point) creates a valid handle whose searches fail with
 * HNSW_ERR_EMPTY_INDEX, like the reference. */
int32_t hnsw_index_create(const hnsw_index_desc *desc, int32_t device, hnsw_index **out);
int32_t hnsw_index_destroy(hnsw_index *idx);
int32_t hnsw_index_get_info(const hnsw_index *idx, hnsw_index_info *info);
/* Knobs that never change results:
 *   "vt_bits"       log2 entries of the per-query LDS visited cache (0 = automatic);
 *   "order_queries" a batch of more than half the queries the device holds at once is searched longest
 *                   walk first (a descent pre-pass + a sort decide the order; per-query results are
 *                   unchanged, the long walks start first and the launch's drain phase gets shorter):
 *                   -1 = automatic (default), 0 = never, 1 = always;
 *   "time_kernels"  1 = bracket the launches of every hnsw_search_batch_device call with HIP events
 *                   on the caller's stream (read back with hnsw_index_kernel_times);
 *   "lds_pad"       LDS bytes a search workgroup asks for beyond its own (how many queries a CU holds
 *                   at once in a batch larger than the device holds): -1 = automatic (default);
 *   "byte_rows"     when every value of the vectors is an integer in 0..255 (SIFT descriptors stored as
 *                   float32) the index keeps a second, lossless copy of the rows as bytes and the knn
 *                   searches read that one: each byte is converted back to the float it came from and
 *                   the arithmetic is unchanged, so distances are bit-identical for a quarter of the
 *                   bytes gathered.  1 = use the copy where it exists (default), 0 = read the float32
 *                   rows.  (Environment HNSW_BYTE_ROWS=0 at creation: do not build the copy.)
 *   "split_rows"    a float32 row that ends 1..32 bytes past a 128-byte line (4d mod 128 in 1..32, e.g. d = 100: 400
 *                   bytes) costs one more 128-byte request per evaluation for those last bytes.  For such shapes
 *                   (when there are no byte rows) the index keeps the whole lines of every row in a table of its own
 *                   and the 16 / 32 remaining bytes of node nbr0[c][j] beside slot (c, j) of the layer-0 adjacency,
 *                   where one hop finds all its candidates' tails in a few contiguous lines; the knn searches read
 *                   those on layer 0 (same lanes, operands and order of arithmetic: bit-identical results).
 *                   1 = use the copy where it exists (default), 0 = read the plain rows, -1 = read the plain rows and FREE
 *                   the copy (it cannot come back).  The copy is not small: n * (128-byte lines of a row) + n * max_degree0 *
 *                   (16 or 32) bytes -- for a GloVe-shaped index (1.18 M x 100, M 32) 0.45 + 1.2 GB beside 0.47 GB of
 *                   vectors; hnsw_index_info.device_bytes counts it.  (Environment HNSW_SPLIT_ROWS=0 at creation: do not
 *                   build it.)
 *   "visited_blocks" how the knn kernels with W in four or more registers (ef > 128) remember visited nodes
 *                   (Visited, lib/ohnsw.ml:256-268): 0 = an LDS cache of node tags (all that rounds 1-4 had); 1 = an LDS
 *                   cache of BITMAP BLOCKS over "locality codes" -- a second numbering of the nodes, derived from the
 *                   index's own upper layers, under which graph-close nodes are consecutive, so that the nodes a walk
 *                   visits share blocks and cost one bit each (ocaml-hnsw_amd/csrc/hnsw_locality.hip); -1 (default) = the
 *                   handle decides per kernel shape by searching 256 probe queries (midpoints between a stored vector and
 *                   its first neighbour) both ways and counting evaluations, inside the first search call that needs the answer --
 *                   or ahead of it: hnsw_index_prepare, expected_ef -- (which then also builds the codes: one
 *                   small layer search per node, n * (1 + max_degree0) * 4 bytes of tables, a device synchronisation --
 *                   1.2 s for 10 M nodes; indices below 200 000 nodes are not measured).  MEMORY: the n * max_degree0 * 4 bytes of
 *                   per-slot codes (2.56 GB for 10 M nodes at M 32; counted in hnsw_index_info.device_bytes while they exist) are kept
 *                   only while some kernel shape of the handle uses the blocks: a measurement that chooses the tag cache frees
 *                   them again, the n * 4 bytes of per-node codes stay (a later measurement re-makes the table in milliseconds).
 *                   Clustered / embedding-like data:
 *                   the blocks end the repeated evaluations of forgotten nodes (DEEP10M shape, ef 512: 40 % -> 5 % of all
 *                   evaluations); structureless data: the tags win and are kept.  Results are the same bits either way.
 * and one that buys exactness for the device-pointer entry point:
 *   "device_fallback_slab_bytes"  the library allocates a slab of this many bytes (0 frees it); hnsw_search_batch_device
 *                   then lists the queries its launch flagged (d_status bit 0: tie list outgrew its LDS slots) ON THE
 *                   DEVICE and searches them again with the slab on the caller's stream -- no host round trip, two small
 *                   extra launches per call.  One flagged query needs 4 n bytes (a slot per node), so a slab repairs
 *                   bytes / (4 n) queries per call; those it could not take keep their flag.  Needs d_status.  The list and
 *                   the slab belong to the handle: with this option ONE hnsw_search_batch_device / hnsw_search_batch_h2d call
 *                   in flight per handle (calls on one stream are ordered and therefore fine; calls on different streams
 *                   must not overlap). */
int32_t hnsw_index_set_option(hnsw_index *idx, const char *name, int64_t value);
/* Bytes of one vector as the knn searches read it: d for byte rows, 4 * d for float32 rows. */
int32_t hnsw_index_row_bytes(const hnsw_index *idx, int64_t *row_bytes);
/* Mean durations (ms) over the device-entry calls recorded since the last call of this function
 * (option "time_kernels"): the search kernel itself, and the ordering pre-pass (descent kernel +
 * sort; 0 when the batch was searched in the given order).  Waits for the recorded calls. */
int32_t hnsw_index_kernel_times(hnsw_index *idx, double *search_ms, double *prepass_ms, int32_t *calls);

/* Batched search, host buffers.  queries: [nq][q_stride] fp32 (a Lacaml.S.mat d x nq).
 * out_ids [nq][k] int32 (id_base-based), out_dist [nq][k] fp32, ascending.
 * out_ndist / out_nhops (optional, [nq]): distance evaluations and expanded candidates on
 * layer 0 per query -- the unit the reference counts in lib/hnsw.ml:730-751. */
int32_t hnsw_search_batch(hnsw_index *idx, const float *queries, int64_t nq, int64_t q_stride,
                          const hnsw_search_params *params, int32_t *out_ids, float *out_dist,
                          uint32_t *out_ndist, uint32_t *out_nhops);

/* Optional: page-locked query / result matrices.  The host-buffer entry points (hnsw_search_batch, hnsw_search_batch_h2d,
 * hnsw_search_submit / hnsw_search_wait, hnsw_multi_search_batch) then do not copy such a matrix at all where they can: the
 * device reads the queries straight out of it (each query once, by the wave that searches it: the transfer runs under the
 * ordering pre-pass) and writes every query's results straight into the caller's result matrices as the query finishes.
 * Everything is complete when the call returns.  Two ways to get such memory:
 *   hnsw_host_alloc / hnsw_host_free        the library allocates page-locked memory (hipHostMalloc); the OCaml side wraps it
 *                                           as a Bigarray (Ctypes.bigarray_of_ptr; Hnsw_mi355x.alloc_mat).
 *   hnsw_host_register / hnsw_host_unregister  page-lock an array the caller already has (hipHostRegister).  The CALLER owns
 *                                           the lifetime: the array stays allocated (a Bigarray: reachable) until
 *                                           hnsw_host_unregister; the library never registers anything behind the caller's
 *                                           back.  Registering an array twice is not an error; an array of which only a
 *                                           part is registered already is refused (HNSW_ERR_BAD_ARG); an array somebody else
 *                                           has page-locked is accepted, left to its owner and never accessed in place.
 * Only ranges obtained through these four calls are accessed directly; any other pointer is staged through copies.
 * LIFETIME.  A range may be unregistered / freed at any time after the call that used it has RETURNED: hnsw_host_unregister and
 * hnsw_host_free wait for the asynchronous work that still reads it (hnsw_search_batch_h2d's kernels read a registered query
 * matrix in place after the call has returned; hnsw_search_submit's upload is a DMA out of it) -- the library keeps an event
 * behind the last such reader per range and stream.  What the library cannot defend against is the memory itself going away
 * while registered (munmap / free of an array that was never unregistered): unregister first.  Both calls answer
 * HNSW_ERR_BAD_ARG for a pointer that is not the start of a range THEY handed out (hnsw_host_unregister: registered through
 * hnsw_host_register; hnsw_host_free: allocated by hnsw_host_alloc). */
int32_t hnsw_host_register(void *p, int64_t bytes);
int32_t hnsw_host_unregister(void *p);
int32_t hnsw_host_alloc(void **out, int64_t bytes);
int32_t hnsw_host_free(void *p);

/* Same, device buffers, asynchronous on `stream` (a hipStream_t; NULL = default stream).
 * d_status (optional, [nq] uint32): bit 0 set if the query's list of tied, still expandable
 * candidates outgrew its 64 LDS slots.  WITHOUT THE OPTION "device_fallback_slab_bytes" THIS ENTRY POINT HAS NO EXACTNESS
 * FALLBACK: entries that did not fit were not expanded, so a flagged query's result may MISS neighbours the reference would return
 * (tests/test_gpu_parity.py::test_tie_overflow_beyond_lds_stack builds such a case), not merely order
 * ties differently.  A caller that needs the reference's result passes d_status and re-runs the flagged
 * queries through hnsw_search_batch (host buffers; it searches them again with a global slab), as
 * bench.py counts them (checks.tie_overflow_flagged).  Needs exact ties on a massive scale (e.g. > 64
 * nodes at exactly max(W).d): 0 of the 10 000 queries of the SIFT-shaped benchmark sets. */
int32_t hnsw_search_batch_device(hnsw_index *idx, const float *d_queries, int64_t nq,
                                 int64_t q_stride, const hnsw_search_params *params,
                                 int32_t *d_ids, float *d_dist, uint32_t *d_ndist,
                                 uint32_t *d_nhops, uint32_t *d_status, void *stream);

/* The batch call with its two ends apart: the queries come from HOST memory ([nq][q_stride], as hnsw_search_batch takes
 * them: read by the device directly when the caller registered the matrix with hnsw_host_register, staged through the
 * handle's scratch otherwise), the results are left in DEVICE buffers, everything asynchronous on `stream` -- for a caller
 * that exchanges per-shard results between devices (one process per GPU and an RCCL all-gather: bench.py --gpus N,
 * ocaml-hnsw_amd/sharding.py) before anything goes back to the host.  The query matrix must stay ALLOCATED until the stream has
 * passed the call (unregistering it earlier is safe: hnsw_host_unregister waits, see LIFETIME above); one such call in flight
 * per handle (it uses the handle's query scratch).  As hnsw_search_batch_device:
 * flags only (d_status), unless the option "device_fallback_slab_bytes" is set. */
int32_t hnsw_search_batch_h2d(hnsw_index *idx, const float *queries, int64_t nq, int64_t q_stride,
                              const hnsw_search_params *params, int32_t *d_ids, float *d_dist,
                              uint32_t *d_ndist, uint32_t *d_nhops, uint32_t *d_status, void *stream);

/* The same batch call in two halves, for callers that keep batches coming: a single batch ends with
 * a drain phase (its last queries run on a nearly empty chip at their serial latency, DESIGN.md
 * section 4), which the next batch can fill.  hnsw_search_submit copies the queries in and starts the
 * search on one of the handle's streams, then returns; hnsw_search_wait blocks until that request is
 * done, copies its results out (same arrays and meaning as hnsw_search_batch) and releases it.
 * Requests may be waited for in any order; every submitted request must be waited for (or the index
 * destroyed).  Still one host thread at a time per handle.
 *     submit(b1); submit(b2); wait(b1); submit(b3); wait(b2); ...                                  */
typedef struct hnsw_request hnsw_request;
int32_t hnsw_search_submit(hnsw_index *idx, const float *queries, int64_t nq, int64_t q_stride,
                           const hnsw_search_params *params, hnsw_request **out);
int32_t hnsw_search_wait(hnsw_request *req, int32_t *out_ids, float *out_dist,
                         uint32_t *out_ndist, uint32_t *out_nhops);

/* Single query (Ohnsw.knn / Hnsw.Ba.knn).  *out_count = number of results (<= k). */
int32_t hnsw_knn(hnsw_index *idx, const float *query, const hnsw_search_params *params,
                 int32_t *out_ids, float *out_dist, int32_t *out_count);

/* Gathered distances: out[q][j] = distance(query q, vector ids[q][j]); ids id_base-based.
 * Counterpart of bench_dist/bench_dist.ml (one distance call per pair). */
int32_t hnsw_distance_batch(hnsw_index *idx, const float *queries, int64_t nq, int64_t q_stride,
                            const int32_t *ids, int32_t m, float *out);
int32_t hnsw_distance_batch_device(hnsw_index *idx, const float *d_queries, int64_t nq,
                                   int64_t q_stride, const int32_t *d_ids, int32_t m,
                                   float *d_out, void *stream);

/* ---- the layer-level functions of the path, as batched operators -------------------------------
 * hnsw_search_layer_batch = Ohnsw.search_k (lib/ohnsw.ml:543-588; params->semantics = OHNSW) or
 * Hnsw_algo.Search.search (lib/hnsw_algo.ml:350-391; FUNCTOR) on ONE layer from explicit start
 * nodes -- what Ohnsw.knn calls on layer 0 (:872-874) and what the builder calls on every layer
 * (lib/ohnsw.ml:811, lib/hnsw_algo.ml:663).  For target q: W is seeded with start_nodes[q][*]
 * (id_base-based; entries < id_base are skipped, so lists may be ragged; distinct ids;
 * 1 <= n_start <= ef, the only shape the reference produces), the layer is searched with W
 * bounded by params->ef, and W[0..k) comes back ascending in out_ids/out_dist [nq][k] with
 * out_cnt[q] = min(|W|, k) (optional).  A start node that does not exist on `layer` has no
 * neighbours there (MapGraph.adjacent of a missing node is empty, lib/hnsw.ml:146-149). */
int32_t hnsw_search_layer_batch(hnsw_index *idx, int32_t layer, const float *targets, int64_t nq,
                                int64_t t_stride, const int64_t *start_nodes, int32_t n_start,
                                const hnsw_search_params *params, int32_t *out_ids, float *out_dist,
                                int32_t *out_cnt, uint32_t *out_ndist, uint32_t *out_nhops);

/* hnsw_search_one_batch = Ohnsw.search_one (= search_one_simple, lib/ohnsw.ml:492-512); also the
 * result of Hnsw_algo.Search.search_one (lib/hnsw_algo.ml:393-437), which reaches the same node by
 * an ef = 1 search.  Greedy walk on `layer` from start[q]: scan all neighbours of the current
 * node, move to the nearest one if it is strictly closer (:502), until no change.
 * out_node [nq] id_base-based; out_dist [nq] (optional) its distance (the value_distance the
 * functor version carries down, lib/hnsw_algo.ml:1005). */
int32_t hnsw_search_one_batch(hnsw_index *idx, int32_t layer, const float *targets, int64_t nq,
                              int64_t t_stride, const int64_t *start, int64_t *out_node,
                              float *out_dist);

/* ---- one host process, several GPUs ------------------------------------------------
 * An OCaml program is ONE process: this is the form of BASELINE.json's "replicated index, query batch
 * sharded across the GPUs of one node, RCCL all-gather of per-shard results over xGMI" it can reach.
 * The index is REPLICATED on every listed device; a query batch is split into n_devices contiguous
 * shards [g*nq/G, (g+1)*nq/G), shard g is uploaded to and searched on device g (all devices
 * concurrently, one HIP stream each), and ONE exchange -- ncclAllGather on communicators from
 * ncclCommInitAll (in-place; unequal shards: one ncclBroadcast per shard inside the same group) --
 * leaves the full [nq][k] ids and distances resident on EVERY device.  Results are bit-identical to
 * hnsw_search_batch on one device (each query is an independent traversal, lib/ohnsw.ml:883-895), the
 * exactness fallback for tie-list overflow runs per shard before the exchange.
 * RCCL is bound at first use (dlopen librccl.so.1).  A device may be listed more than once (several
 * replicas on one GPU: a test arrangement); RCCL refuses that, the exchange is then device-to-device
 * copies.  The library pins nothing of the caller's: the shard uploads run at PCIe speed when the caller registered its
 * query matrix with hnsw_host_register (its lifetime, not the library's) and are staged by the runtime otherwise.
 * ERRORS.  A search that fails before the exchange leaves nothing enqueued.  If the exchange itself is refused part-way (one
 * device's collective not accepted inside the group) the library ABORTS the handle's communicators (ncclCommAbort) before it
 * returns HNSW_ERR_HIP, so no device is left waiting for a peer that never joins; the result tables of that call are undefined
 * and the next search on the handle creates new communicators.
 * (One process PER GPU, each with its own RCCL rank, is the other deployment:
 * ocaml-hnsw_amd/sharding.py and bench.py.) */
typedef struct hnsw_multi hnsw_multi;
int32_t hnsw_multi_create(const hnsw_index_desc *desc, const int32_t *devices, int32_t n_devices,
                          hnsw_multi **out);
int32_t hnsw_multi_destroy(hnsw_multi *m);
int32_t hnsw_multi_num_replicas(const hnsw_multi *m, int32_t *n_devices);
/* replica g (borrowed: destroyed with the hnsw_multi) */
int32_t hnsw_multi_replica(hnsw_multi *m, int32_t g, hnsw_index **out);
/* Ohnsw.knn_batch_bigarray / Hnsw.Ba.knn_batch over all replicas, host arrays in and out: the host
 * receives the table from one device after the exchange. */
int32_t hnsw_multi_search_batch(hnsw_multi *m, const float *queries, int64_t nq, int64_t q_stride,
                                const hnsw_search_params *params, int32_t *out_ids, float *out_dist,
                                uint32_t *out_ndist, uint32_t *out_nhops);
/* The same with the results left on the devices: on return (synchronised) d_ids[g] / d_dist[g],
 * g < n_devices, point to device g's copy of the full [nq][k] table (library-owned, valid until the
 * next search on this handle).  d_ids / d_dist are caller arrays of n_devices pointers (either may be
 * NULL). */
int32_t hnsw_multi_search_batch_device(hnsw_multi *m, const float *queries, int64_t nq, int64_t q_stride,
                                       const hnsw_search_params *params, int32_t **d_ids, float **d_dist);
/* What the exchanges of this handle were made of so far (tests, debugging): out4 = { ncclAllGather calls, ncclBroadcast
 * calls (unequal shards, and the re-send of a repaired shard), device-to-device copies (replicas sharing a device),
 * shards searched again by the exactness fallback }, each summed over the devices. */
int32_t hnsw_multi_debug_counters(const hnsw_multi *m, int64_t *out4);
/* device g's copy of the last device-resident result, copied to host arrays [nq][k] (tests, debugging) */
int32_t hnsw_multi_copy_result(hnsw_multi *m, int32_t g, int32_t *out_ids, float *out_dist);

/* ---- graph construction on the device (next-row scope: the reference's builder stays OCaml;
 * this entry point exists so an index can also be produced where no OCaml build is at hand,
 * e.g. by bench.py).  Batched restatement of Ohnsw.build_batch_bigarray (lib/ohnsw.ml:840-857):
 * same level law, same per-node steps (search_one descent, search_k with efConstruction,
 * select_neighbours with M / 2M, symmetric links, shrink), nodes inserted in batches against
 * the graph as of batch start.  Deterministic for a given seed. */
typedef struct hnsw_build_params {
    int32_t num_connections;               /* M   (~num_connections, lib/ohnsw.ml:841)          */
    int32_t num_nodes_search_construction; /* efConstruction                                   */
    int32_t metric;
    int32_t id_base;
    uint64_t seed;                         /* level draws (own RNG)                            */
    int32_t max_batch;                     /* 0 = default (8192)                               */
    int32_t batch_div;                     /* batch <= nodes already inserted / batch_div; 0 = 16 */
    int32_t expected_ef;                   /* as hnsw_index_desc.expected_ef: 0 = not known      */
    int32_t expected_semantics;
} hnsw_build_params;

int32_t hnsw_build(const float *vectors, int64_t n, int32_t d, int64_t row_stride,
                   const hnsw_build_params *params, int32_t device, hnsw_index **out);

/* select_neighbours as a batched operator (Ohnsw.select_neighbours lib/ohnsw.ml:647-663;
 * keep_all_if_few = 1 adds the functor path's "#candidates <= M returns them all" shortcut,
 * lib/hnsw_algo.ml:596-599).  For each of nb target vectors: the candidates (ids, id_base-based)
 * are ordered by (distance to the target, id) and kept iff strictly closer to the target than
 * to every neighbour kept so far; at most num_neighbours are kept.  out [nb][num_neighbours] in
 * selection order (nearest first), -1 padded. */
int32_t hnsw_select_neighbours_batch(hnsw_index *idx, const float *targets, int64_t nb, int64_t t_stride,
                                     const int32_t *cand, const int32_t *cand_cnt, int32_t cand_stride,
                                     int32_t num_neighbours, int32_t keep_all_if_few,
                                     const int32_t *cand_degree /* NULL, or [nb][cand_stride]: ~do_not_isolate:true,
                                        candidates whose degree is <= 1 are kept unconditionally, hnsw_algo.ml:591-592 */,
                                     int32_t *out, int32_t *out_cnt);

/* Export of the flattened graph held by an index (inverse of hnsw_index_create): ids
 * id_base-based, rows compacted, -1 padded. */
int32_t hnsw_index_export_layer0(const hnsw_index *idx, int32_t *deg0, int32_t *nbr0);
int32_t hnsw_index_export_upper_count(const hnsw_index *idx, int32_t layer, int64_t *n_nodes);
int32_t hnsw_index_export_upper(const hnsw_index *idx, int32_t layer, int64_t *nodes, int32_t *deg,
                                int32_t *nbr);

/* The locality codes behind the option "visited_blocks" (built on first use or by this call): out[v] = position of node v
 * (0-based, whatever id_base is) in an order that keeps graph-close nodes together -- a permutation of 0 .. n-1 derived
 * from the index's own upper layers (ocaml-hnsw_amd/csrc/hnsw_locality.hip).  Introspection and tests; the search never
 * hands codes out.  HNSW_ERR_UNSUPPORTED when the index has no upper layer to derive an order from. */
int32_t hnsw_index_locality_codes(hnsw_index *idx, int32_t *out);
/* Which visited structure the knn kernel of this handle uses for searches with this ef and accept rule (params->semantics):
 * *log2_slots = 0: the tag cache; else the bitmap-block directory has 2^*log2_slots slots (256 codes each).  With the option
 * "visited_blocks" at -1 the first call for a kernel shape makes the measurement described there (as the first search would):
 * a program that must not meet a device synchronisation inside its first search (a latency-critical path, a stream capture)
 * calls this once per ef at set-up. */
int32_t hnsw_index_visited_blocks(hnsw_index *idx, const hnsw_search_params *params, int32_t *log2_slots);

/* Everything a handle does ONCE for searches with these parameters, done now instead of inside the first such search call:
 * the visited-structure decision of option "visited_blocks" at -1 (building the locality codes and the measurement: 0.1-1.3 s
 * and a device synchronisation for an eligible shape), the kernel variant's residency query, and loading the code object of
 * the variant's translation unit (one query searched and discarded: ~1.4 ms).  hnsw_index_create / hnsw_build call it for
 * desc->expected_ef / params->expected_ef; hnsw_index_load for every shape the saved handle had prepared.  Any number of
 * parameter sets may be prepared; preparing is never needed for correctness.  The measurement draws its 256 probe queries from
 * midpoints between a stored vector and its first layer-0 neighbour (in-distribution, not themselves stored); an index whose
 * real queries come from elsewhere can still be steered with option "visited_blocks" 0 / 1.  Results never depend on any of it. */
int32_t hnsw_index_prepare(hnsw_index *idx, const hnsw_search_params *params);

/* Per-layer degree statistics: Hgraph.Stats.compute (lib/hnsw.ml:353-375; printed by
 * benchmark/benchmark.ml:70-71): layer size (layer_sizes) and the layer's mima record -- min / max / mean of the
 * neighbour-list lengths, the nodes without a neighbour.  Computed on the device from the resident tables.  The keys of
 * layer l >= 1 are the nodes inserted at level >= l.  A layer without nodes yields what the reference's fold yields:
 * min 1000000, max -1, mean nan. */
typedef struct hnsw_layer_stats {
    int64_t num_nodes;     /* layer_sizes */
    int32_t min_degree, max_degree;
    double mean_degree;
    int64_t num_isolated;  /* length of mima.isolated */
} hnsw_layer_stats;
int32_t hnsw_index_layer_stats(const hnsw_index *idx, int32_t layer, hnsw_layer_stats *out);
/* mima.isolated of that layer: the node ids (id_base-based) in the reference's list order -- min_max_connectivity
 * conses a key onto the list as the ascending Map.fold meets it (lib/hnsw.ml:364-366), so the list is DESCENDING.
 * *count = length of the list; the first min(cap, *count) ids are written (cap = 0, ids = NULL: count only). */
int32_t hnsw_index_layer_isolated(const hnsw_index *idx, int32_t layer, int64_t *ids, int64_t cap, int64_t *count);

/* Flattened-index file (new: the reference has no persistence; its types derive sexp but values
 * are sexp_opaque and nothing reads one back, lib/hnsw.ml:348, lib/ohnsw.ml:312).  Little-endian
 * header + vectors + layer 0 + upper layers; hnsw_index_load re-validates everything through
 * hnsw_index_create.  Format 2 (this version writes it; format 1 files still load) appends what the handle had learnt:
 * the locality codes, if built (n int32), and for every prepared / measured kernel shape whether it took the bitmap blocks --
 * hnsw_index_load adopts the codes (a permutation check: a corrupt table is dropped and rebuilt on demand) and the decisions
 * instead of building and measuring again, and prepares the saved shapes, so the first search after a load runs at the
 * steady-state rate. */
int32_t hnsw_index_save(const hnsw_index *idx, const char *path);
int32_t hnsw_index_load(const char *path, int32_t device, hnsw_index **out);

#ifdef __cplusplus
}
#endif
#endif /* HNSW_MI355X_H */
