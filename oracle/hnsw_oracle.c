/*
 * hnsw_oracle.c -- CPU restatement of ocaml-hnsw's search path.  TEST INFRASTRUCTURE ONLY
 * (see hnsw_oracle.h for the pinning status).  Plain C, single thread, like the reference.
 *
 * Every function cites the reference lines (under /root/reference) it follows.
 */
#include "hnsw_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================== */
/* arena: the reference allocates persistent heap nodes on the OCaml GC heap; we bump-allocate
 * and reset per query.                                                                      */
/* ======================================================================================== */
typedef struct arena_chunk {
    struct arena_chunk *next;
    size_t cap, used;
    char data[];
} arena_chunk;

typedef struct arena {
    arena_chunk *first; /* retained across resets */
    arena_chunk *cur;   /* chunk being filled; NULL right after a reset */
} arena;

static void *arena_alloc(arena *a, size_t sz) {
    sz = (sz + 15u) & ~(size_t)15u;
    if (!a->cur && a->first) { a->cur = a->first; a->cur->used = 0; }
    while (a->cur && a->cur->used + sz > a->cur->cap && a->cur->next) { a->cur = a->cur->next; a->cur->used = 0; }
    if (!a->cur || a->cur->used + sz > a->cur->cap) {
        size_t cap = (size_t)1 << 20;
        if (cap < sz) cap = sz;
        arena_chunk *c = (arena_chunk *)malloc(sizeof(arena_chunk) + cap);
        if (!c) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
        c->cap = cap; c->used = 0; c->next = NULL;
        if (a->cur) a->cur->next = c; else a->first = c;
        a->cur = c;
    }
    void *p = a->cur->data + a->cur->used;
    a->cur->used += sz;
    return p;
}
static void arena_reset(arena *a) { a->cur = NULL; }
static void arena_free(arena *a) {
    arena_chunk *c = a->first;
    while (c) { arena_chunk *n = c->next; free(c); c = n; }
    a->first = a->cur = NULL;
}

/* ======================================================================================== */
/* distances                                                                                */
/* ======================================================================================== */

/* Lacaml-style stub: sequential accumulation in the storage type (fp32). */
float og_l2sq_seq(const float *a, const float *b, int32_t d) {
    float acc = 0.0f;
    for (int32_t i = 0; i < d; ++i) { float df = a[i] - b[i]; acc = acc + df * df; }
    return acc;
}
double og_l2sq_f64(const float *a, const float *b, int32_t d) {
    double acc = 0.0;
    for (int32_t i = 0; i < d; ++i) { double df = (double)a[i] - (double)b[i]; acc += df * df; }
    return acc;
}
/* The HIP kernel's order: a row is cut in float4 chunks; partial j (0..15) runs an fmaf chain
 * over chunks j, j+16, j+32, ... (elements x,y,z,w in order); then p[j] += p[j+s] for
 * s = 8,4,2,1 (DPP row_ror butterfly).  IEEE addition commutes, so every lane of the 16-lane
 * group ends with these exact bits. */
float og_l2sq_tree16(const float *a, const float *b, int32_t d) {
    float p[16];
    int32_t nch = (d + 3) / 4;
    for (int j = 0; j < 16; ++j) {
        float acc = 0.0f;
        for (int32_t c = j; c < nch; c += 16)
            for (int e = 0; e < 4; ++e) {
                int32_t i = 4 * c + e;
                if (i < d) { float df = a[i] - b[i]; acc = fmaf(df, df, acc); }
            }
        p[j] = acc;
    }
    for (int s = 8; s >= 1; s >>= 1)
        for (int j = 0; j < s; ++j) p[j] = p[j] + p[j + s];
    return p[0];
}
float og_dot_tree16(const float *a, const float *b, int32_t d) {
    float p[16];
    int32_t nch = (d + 3) / 4;
    for (int j = 0; j < 16; ++j) {
        float acc = 0.0f;
        for (int32_t c = j; c < nch; c += 16)
            for (int e = 0; e < 4; ++e) {
                int32_t i = 4 * c + e;
                if (i < d) acc = fmaf(a[i], b[i], acc);
            }
        p[j] = acc;
    }
    for (int s = 8; s >= 1; s >>= 1)
        for (int j = 0; j < s; ++j) p[j] = p[j] + p[j + s];
    return p[0];
}

/* distance (lib/hnsw.ml:814, lib/ohnsw.ml:899): Float.sqrt (ssqr_diff a b), an OCaml float
 * (double).  Scalar spaces: Float.abs (a -. b) (lib/ohnsw.ml:361). */
double og_distance_raw(const og_space *sp, const void *a, const void *b) {
    switch (sp->kind) {
    case OG_SCALAR_ABS:
        return fabs(*(const double *)a - *(const double *)b);
    case OG_L2: {
        const float *x = (const float *)a, *y = (const float *)b;
        switch (sp->arith) {
        case OG_F64: return sqrt(og_l2sq_f64(x, y, sp->d));
        case OG_TREE16: return sqrt((double)og_l2sq_tree16(x, y, sp->d));
        default: return sqrt((double)og_l2sq_seq(x, y, sp->d));
        }
    }
    case OG_IP: {
        const float *x = (const float *)a, *y = (const float *)b;
        switch (sp->arith) {
        case OG_F64: {
            double acc = 0.0;
            for (int32_t i = 0; i < sp->d; ++i) acc += (double)x[i] * (double)y[i];
            return 1.0 - acc;
        }
        case OG_TREE16: return (double)(1.0f - og_dot_tree16(x, y, sp->d));
        default: {
            float acc = 0.0f;
            for (int32_t i = 0; i < sp->d; ++i) acc = acc + x[i] * y[i];
            return (double)(1.0f - acc);
        }
        }
    }
    }
    return NAN;
}

/* value (lib/ohnsw.ml:842: Mat.col batch (i+1); lib/ohnsw.ml:362 for scalar tests) */
static inline const void *sp_value(const og_space *sp, int64_t node) {
    if (sp->kind == OG_SCALAR_ABS) return (const void *)(sp->S + node);
    return (const void *)(sp->X + node * sp->stride);
}
static inline double sp_dist(og_space *sp, const void *a, const void *b) {
    sp->n_calls++;
    return og_distance_raw(sp, a, b);
}

/* ======================================================================================== */
/* graph views                                                                              */
/* ======================================================================================== */
typedef struct graph_view {
    /* adjacency in the reference's iteration order (Neighbours.iter, lib/ohnsw.ml:127) */
    const int32_t *(*adj)(const struct graph_view *, int32_t layer, int64_t node, int32_t *len);
    int64_t n;
} graph_view;

struct og_graph {
    graph_view view;
    int32_t max_layer;
    int64_t entry_point;
    int32_t stride0;
    const int32_t *deg0, *nbr0;
    int32_t strideU;
    int64_t *upper_n;
    const int32_t **upper_deg, **upper_nbr;
    int32_t **slot_of; /* [layer-1][n] -> slot or -1 */
};

static const int32_t *flat_adj(const graph_view *v, int32_t layer, int64_t node, int32_t *len) {
    const og_graph *g = (const og_graph *)v;
    if (layer == 0) {
        *len = g->deg0[node];
        return g->nbr0 + node * g->stride0;
    }
    if (layer > g->max_layer) { *len = 0; return NULL; }
    int32_t s = g->slot_of[layer - 1][node];
    if (s < 0) { *len = 0; return NULL; } /* Ohnsw: empty Neighbours; functor: Map.find None */
    *len = g->upper_deg[layer - 1][s];
    return g->upper_nbr[layer - 1] + (int64_t)s * g->strideU;
}

og_graph *og_graph_create(int64_t n, int32_t max_layer, int64_t entry_point, int32_t stride0,
                          const int32_t *deg0, const int32_t *nbr0, int32_t strideU,
                          const int64_t *upper_n, const int64_t *const *upper_nodes,
                          const int32_t *const *upper_deg, const int32_t *const *upper_nbr) {
    og_graph *g = (og_graph *)calloc(1, sizeof(og_graph));
    g->view.adj = flat_adj;
    g->view.n = n;
    g->max_layer = max_layer;
    g->entry_point = entry_point;
    g->stride0 = stride0; g->deg0 = deg0; g->nbr0 = nbr0; g->strideU = strideU;
    if (max_layer > 0) {
        g->upper_n = (int64_t *)calloc(max_layer, sizeof(int64_t));
        g->upper_deg = (const int32_t **)calloc(max_layer, sizeof(void *));
        g->upper_nbr = (const int32_t **)calloc(max_layer, sizeof(void *));
        g->slot_of = (int32_t **)calloc(max_layer, sizeof(void *));
        for (int32_t l = 0; l < max_layer; ++l) {
            g->upper_n[l] = upper_n[l];
            g->upper_deg[l] = upper_deg[l];
            g->upper_nbr[l] = upper_nbr[l];
            g->slot_of[l] = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
            for (int64_t i = 0; i < n; ++i) g->slot_of[l][i] = -1;
            for (int64_t s = 0; s < upper_n[l]; ++s) g->slot_of[l][upper_nodes[l][s]] = (int32_t)s;
        }
    }
    return g;
}
void og_graph_destroy(og_graph *g) {
    if (!g) return;
    for (int32_t l = 0; l < g->max_layer; ++l) free(g->slot_of[l]);
    free(g->slot_of); free(g->upper_n); free((void *)g->upper_deg); free((void *)g->upper_nbr);
    free(g);
}
int64_t og_graph_num_nodes(const og_graph *g) { return g->view.n; }
int32_t og_graph_max_layer(const og_graph *g) { return g->max_layer; }
int64_t og_graph_entry_point(const og_graph *g) { return g->entry_point; }
int32_t og_graph_adjacent(const og_graph *g, int32_t layer, int64_t node, int32_t *out, int32_t cap) {
    int32_t len; const int32_t *p = g->view.adj(&g->view, layer, node, &len);
    for (int32_t i = 0; i < len && i < cap; ++i) out[i] = p[i];
    return len;
}

/* ======================================================================================== */
/* Hgraph.Stats: lib/hnsw.ml:353-375.  min_max_connectivity (:361-368) folds the layer's      */
/* `connections` map in ascending key order from (1000000, -1, 0, 0., []): min / max of the    */
/* neighbour-list lengths, their mean as a float sum over a float count, and the nodes with    */
/* no neighbour consed onto `isolated` as they are met -- so the list holds them in DESCENDING */
/* id order.  compute (:370-375): layer_sizes = Map.length of the layer's connections.  The   */
/* keys of a layer are the nodes inserted at that level or above (layer 0: every node).        */
/* ======================================================================================== */
int64_t og_layer_stats(const og_graph *g, int32_t layer, int64_t *num_nodes, int32_t *min_degree,
                       int32_t *max_degree, double *mean_degree, int64_t *isolated, int64_t isolated_cap) {
    int32_t mi = 1000000, ma = -1;
    int64_t cnt = 0, n_iso = 0;
    double sum = 0.0;
    const int64_t n = g->view.n;
    if (layer < 0 || layer > g->max_layer) return -1;
    for (int64_t key = 0; key < n; ++key) {              /* Map.fold: ascending keys */
        if (layer > 0 && g->slot_of[layer - 1][key] < 0) continue;   /* not a key of this layer's map */
        int32_t len; (void)g->view.adj(&g->view, layer, key, &len);  /* Neighbours.length */
        if (!(len > 0)) n_iso++;                          /* key :: isolated (stored below, newest first) */
        mi = len < mi ? len : mi; ma = len > ma ? len : ma;
        cnt++; sum += (double)len;
    }
    if (isolated) {                                       /* the consed list, head first = descending ids */
        int64_t w = 0;
        for (int64_t key = n - 1; key >= 0 && w < isolated_cap; --key) {
            if (layer > 0 && g->slot_of[layer - 1][key] < 0) continue;
            int32_t len; (void)g->view.adj(&g->view, layer, key, &len);
            if (!(len > 0)) isolated[w++] = key;
        }
    }
    if (num_nodes) *num_nodes = cnt;
    if (min_degree) *min_degree = mi;
    if (max_degree) *max_degree = ma;
    if (mean_degree) *mean_degree = sum / (double)cnt;    /* 0. /. 0. = nan on an empty layer, as in OCaml */
    return n_iso;
}

/* ======================================================================================== */
/* Visited: lib/ohnsw.ml:256-268 (epoch array; clear = epoch++ with reset on overflow).     */
/* The functor copy (lib/hnsw.ml:105-121) is the same minus the overflow reset.             */
/* ======================================================================================== */
struct og_visited { int64_t *visited; int64_t n; int64_t epoch; };
#define OCAML_MAX_INT ((int64_t)4611686018427387903LL) /* Int.max_value, 63-bit */

og_visited *og_visited_create(int64_t n) {
    og_visited *v = (og_visited *)malloc(sizeof(og_visited));
    v->visited = (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t));
    v->n = n; v->epoch = 1;
    return v;
}
void og_visited_destroy(og_visited *v) { if (v) { free(v->visited); free(v); } }
int32_t og_visited_mem(const og_visited *v, int64_t node) {
    if (node < 0 || node >= v->n) return -1; /* OCaml: Invalid_argument index out of bounds */
    return v->visited[node] >= v->epoch;
}
void og_visited_add(og_visited *v, int64_t node) { v->visited[node] = v->epoch; }
void og_visited_clear(og_visited *v) {
    if (v->epoch < OCAML_MAX_INT - 1) v->epoch++;
    else { memset(v->visited, 0, sizeof(int64_t) * (size_t)v->n); v->epoch = 1; }
}
int64_t og_visited_card(const og_visited *v) {
    int64_t c = 0;
    for (int64_t i = 0; i < v->n; ++i) if (v->visited[i] >= v->epoch) c++;
    return c;
}
void og_visited_set_epoch(og_visited *v, int64_t epoch) { v->epoch = epoch; }

/* ======================================================================================== */
/* PairingHeap: lib/hnsw_algo.ml:17-66 (persistent; merge: a wins iff compare a b < 0, else b
 * -- including ties, so among equal keys the most recently added is on top).              */
/* Used for both paths: on the imperative path the reference uses Core_kernel.Heap (also a  */
/* pairing heap, un-vendored: tie order there is UNPINNED).                                 */
/* ======================================================================================== */
typedef struct elt { int64_t node; double dist; } elt;
typedef struct ph_list ph_list;
typedef struct ph_node { elt e; ph_list *sub; } ph_node;
struct ph_list { ph_node *h; ph_list *next; };

enum { CMP_NEAREST = 0, CMP_FARTHEST = 1 };
typedef struct ph_ctx { arena *ar; int kind; int canonical; } ph_ctx;

/* Float.compare (HeapElt.compare_nearest/farthest lib/ohnsw.ml:10-11; MinHeap/MaxHeap
 * Element.compare lib/hnsw_algo.ml:91,126).  Canonical mode breaks ties by node id so the
 * order is total: nearest = ascending (d,id); farthest = descending (d,id). */
static inline int elt_compare(const ph_ctx *c, const elt *a, const elt *b) {
    int r = (a->dist < b->dist) ? -1 : (a->dist > b->dist) ? 1 : 0;
    if (r == 0 && c->canonical) r = (a->node < b->node) ? -1 : (a->node > b->node) ? 1 : 0;
    return c->kind == CMP_NEAREST ? r : -r;
}
static ph_node *ph_singleton(const ph_ctx *c, elt x) {
    ph_node *n = (ph_node *)arena_alloc(c->ar, sizeof(ph_node));
    n->e = x; n->sub = NULL;
    return n;
}
static ph_node *ph_merge(const ph_ctx *c, ph_node *a, ph_node *b) { /* :25-31 */
    if (!a) return b;
    if (!b) return a;
    ph_node *n = (ph_node *)arena_alloc(c->ar, sizeof(ph_node));
    ph_list *cell = (ph_list *)arena_alloc(c->ar, sizeof(ph_list));
    if (elt_compare(c, &a->e, &b->e) < 0) { n->e = a->e; cell->h = b; cell->next = a->sub; }
    else { n->e = b->e; cell->h = a; cell->next = b->sub; }
    n->sub = cell;
    return n;
}
static ph_node *ph_add(const ph_ctx *c, ph_node *h, elt x) { return ph_merge(c, h, ph_singleton(c, x)); } /* :32 */
static ph_node *ph_merge_pairs(const ph_ctx *c, ph_list *l) { /* :37-41 */
    if (!l) return NULL;
    if (!l->next) return l->h;
    /* merge (merge h1 h2) (merge_pairs q): iterative first pass, then fold right-to-left to
     * keep the exact association of the recursive definition without deep C recursion. */
    size_t npairs = 0;
    for (ph_list *p = l; p; p = p->next ? p->next->next : NULL) npairs++;
    ph_node **tmp = (ph_node **)arena_alloc(c->ar, sizeof(ph_node *) * npairs);
    size_t i = 0;
    for (ph_list *p = l; p;) {
        if (p->next) { tmp[i++] = ph_merge(c, p->h, p->next->h); p = p->next->next; }
        else { tmp[i++] = p->h; p = NULL; }
    }
    ph_node *acc = tmp[npairs - 1];
    for (size_t j = npairs - 1; j-- > 0;) acc = ph_merge(c, tmp[j], acc);
    return acc;
}
static inline ph_node *ph_remove_top(const ph_ctx *c, ph_node *h) { return h ? ph_merge_pairs(c, h->sub) : NULL; } /* :43-45 */
/* fold (:52-55): node first, then subheaps left to right, depth first. */
typedef void (*ph_fold_fn)(void *ud, const elt *e);
static void ph_fold(const ph_node *h, ph_fold_fn f, void *ud) {
    if (!h) return;
    f(ud, &h->e);
    for (const ph_list *l = h->sub; l; l = l->next) ph_fold(l->h, f, ud);
}

/* elements in fold order (node, then subheaps left to right), arena-allocated */
static elt *ph_collect(arena *ar, const ph_node *h, int32_t *count) {
    int32_t cap = 64, cnt = 0, top = 0, scap = 64;
    elt *buf = (elt *)malloc(sizeof(elt) * (size_t)cap);
    const ph_node **st = (const ph_node **)malloc(sizeof(void *) * (size_t)scap);
    if (h) st[top++] = h;
    while (top > 0) {
        const ph_node *x = st[--top];
        if (cnt == cap) { cap *= 2; buf = (elt *)realloc(buf, sizeof(elt) * (size_t)cap); }
        buf[cnt++] = x->e;
        int32_t nchild = 0;
        for (const ph_list *l = x->sub; l; l = l->next) nchild++;
        if (top + nchild > scap) { scap = (top + nchild) * 2; st = (const ph_node **)realloc(st, sizeof(void *) * (size_t)scap); }
        int32_t base = top, j = 0; top += nchild;
        for (const ph_list *l = x->sub; l; l = l->next, ++j) st[base + nchild - 1 - j] = l->h; /* leftmost popped first */
    }
    elt *out = (elt *)arena_alloc(ar, sizeof(elt) * (size_t)(cnt > 0 ? cnt : 1));
    memcpy(out, buf, sizeof(elt) * (size_t)cnt);
    free(buf); free(st);
    *count = cnt;
    return out;
}

/* a mutable heap handle = Core_kernel.Heap.t on the imperative path */
typedef struct mheap { ph_ctx c; ph_node *root; int32_t len; } mheap;
static void mheap_init(mheap *h, arena *ar, int kind, int canonical) {
    h->c.ar = ar; h->c.kind = kind; h->c.canonical = canonical; h->root = NULL; h->len = 0;
}
static void mheap_add(mheap *h, elt x) { h->root = ph_add(&h->c, h->root, x); h->len++; }
static int mheap_pop(mheap *h, elt *out) {
    if (!h->root) return 0;
    *out = h->root->e; h->root = ph_remove_top(&h->c, h->root); h->len--;
    return 1;
}

/* ======================================================================================== */
/* imperative path                                                                          */
/* ======================================================================================== */

/* search_one_simple, lib/ohnsw.ml:492-508.  No visited set (argument ignored, :493). */
static int64_t ohnsw_search_one_simple(const graph_view *g, int32_t layer, og_space *sp,
                                       int64_t start, const void *target, og_counters *ctr) {
    int changed = 1;
    int64_t best_node = start;
    double best_distance = sp_dist(sp, sp_value(sp, start), target); /* :496 */
    if (ctr) ctr->n_dist++;
    while (changed) {
        changed = 0;
        int32_t len; const int32_t *nb = g->adj(g, layer, best_node, &len); /* :499 bound once */
        if (ctr) ctr->n_hops_upper++;
        for (int32_t i = 0; i < len; ++i) {
            double d = sp_dist(sp, sp_value(sp, nb[i]), target); /* :501 */
            if (ctr) ctr->n_dist++;
            if (d < best_distance) { best_node = nb[i]; best_distance = d; changed = 1; } /* :502-505 */
        }
    }
    return best_node;
}

/* search_one_paper, lib/ohnsw.ml:445-489 (unused by the reference: :512). */
static int64_t ohnsw_search_one_paper(const graph_view *g, int32_t layer, og_space *sp,
                                      og_visited *visited, arena *ar, int64_t start,
                                      const void *target) {
    og_visited_clear(visited);
    og_visited_add(visited, start);
    mheap visit_me; mheap_init(&visit_me, ar, CMP_NEAREST, 0);
    elt s = { start, sp_dist(sp, target, sp_value(sp, start)) };
    mheap_add(&visit_me, s);
    elt nearest = s;
    elt c;
    while (mheap_pop(&visit_me, &c)) {
        if (c.dist > nearest.dist) return nearest.node; /* :472 */
        int32_t len; const int32_t *nb = g->adj(g, layer, c.node, &len);
        for (int32_t i = 0; i < len; ++i) {
            int64_t e = nb[i];
            if (!og_visited_mem(visited, e)) {
                og_visited_add(visited, e);
                elt ee = { e, sp_dist(sp, target, sp_value(sp, e)) };
                if (ee.dist < nearest.dist) { mheap_add(&visit_me, ee); nearest = ee; } /* :481-484 */
            }
        }
    }
    return nearest.node;
}

/* search_k, lib/ohnsw.ml:543-588.  visit_me = the start queue (destroyed); W = nearest_maxq. */
typedef struct fold_to_heap { mheap *dst; } fold_to_heap;
static void fold_add_cb(void *ud, const elt *e) { mheap_add(((fold_to_heap *)ud)->dst, *e); }

static void ohnsw_search_k(const graph_view *g, int32_t layer, og_space *sp, og_visited *visited,
                           mheap *start_nodes, const void *target, int32_t k,
                           mheap *nearest_maxq, mheap *result_minq, og_counters *ctr) {
    og_visited_clear(visited);                         /* :553 */
    nearest_maxq->root = NULL; nearest_maxq->len = 0;  /* :554 clear_heap */
    /* :555-557 MinQueue.iter start_nodes: visited + W (fold order of the heap) */
    {
        int32_t cnt; elt *buf = ph_collect(start_nodes->c.ar, start_nodes->root, &cnt);
        for (int32_t i = 0; i < cnt; ++i) { og_visited_add(visited, buf[i].node); mheap_add(nearest_maxq, buf[i]); }
    }
    mheap *visit_me = start_nodes; /* :559 */
    elt c;
    while (mheap_pop(visit_me, &c)) {                                  /* :565 */
        if (c.dist > nearest_maxq->root->e.dist) break;                /* :568 */
        if (ctr) ctr->n_hops++;
        int32_t len; const int32_t *nb = g->adj(g, layer, c.node, &len);
        for (int32_t i = 0; i < len; ++i) {                            /* :570 list order */
            int64_t e = nb[i];
            if (!og_visited_mem(visited, e)) {                         /* :571 */
                og_visited_add(visited, e);                            /* :572 */
                elt ee = { e, sp_dist(sp, target, sp_value(sp, e)) };  /* :573 -> :8-9 */
                if (ctr) ctr->n_dist++;
                if (nearest_maxq->len < k || ee.dist < nearest_maxq->root->e.dist) { /* :574 */
                    mheap_add(visit_me, ee);                           /* :575 */
                    mheap_add(nearest_maxq, ee);                       /* :576 */
                    if (nearest_maxq->len > k) { elt dummy; mheap_pop(nearest_maxq, &dummy); } /* :577 */
                }
            }
        }
    }
    result_minq->root = NULL; result_minq->len = 0;                    /* :586 */
    fold_to_heap ud = { result_minq };
    ph_fold(nearest_maxq->root, fold_add_cb, &ud);                     /* :587 Heap.iter */
}

/* select_neighbours, lib/ohnsw.ml:647-663.  Destroys the candidate queue. Returned list is in
 * selection order (the reference conses, so its Neighbours.iter order is the reverse). */
static int32_t ohnsw_select_neighbours(og_space *sp, mheap *queue, int32_t num_neighbours,
                                       int64_t *out) {
    int32_t nsel = 0;
    elt e;
    while (mheap_pop(queue, &e)) {                                     /* :654 */
        int ok = 1;
        /* Neighbours.for_all selected (list order = newest first); stops at first false */
        for (int32_t j = nsel - 1; j >= 0; --j) {
            double dn = sp_dist(sp, sp_value(sp, out[j]), sp_value(sp, e.node)); /* :658 */
            if (!(e.dist < dn)) { ok = 0; break; }
        }
        if (ok) out[nsel++] = e.node;                                  /* :659 */
        if (!(nsel < num_neighbours)) break;                           /* :660 */
    }
    return nsel;
}

/* ---- public wrappers ---- */
typedef struct scratch { arena ar; og_visited *visited; } scratch;
static scratch *scratch_create(int64_t n) {
    scratch *s = (scratch *)calloc(1, sizeof(scratch));
    s->visited = og_visited_create(n);
    return s;
}
static void scratch_destroy(scratch *s) { og_visited_destroy(s->visited); arena_free(&s->ar); free(s); }

int64_t og_ohnsw_search_one(const og_graph *g, int32_t layer, og_space *sp, int64_t start,
                            const void *target, int32_t paper_variant) {
    if (!paper_variant) return ohnsw_search_one_simple(&g->view, layer, sp, start, target, NULL);
    scratch *s = scratch_create(g->view.n);
    int64_t r = ohnsw_search_one_paper(&g->view, layer, sp, s->visited, &s->ar, start, target);
    scratch_destroy(s);
    return r;
}

static int32_t drain_ascending(mheap *q, int32_t limit, int64_t *out_nodes, double *out_dist) {
    int32_t i = 0; elt e;
    while (i < limit && mheap_pop(q, &e)) { out_nodes[i] = e.node; if (out_dist) out_dist[i] = e.dist; i++; }
    return i;
}

int32_t og_ohnsw_search_k(const og_graph *g, int32_t layer, og_space *sp,
                          const int64_t *start_nodes, int32_t n_start, const void *target,
                          int32_t k, int32_t ties, int64_t *out_nodes, double *out_dist,
                          og_counters *ctr) {
    scratch *s = scratch_create(g->view.n);
    mheap startq, maxq, resq;
    mheap_init(&startq, &s->ar, CMP_NEAREST, ties);
    mheap_init(&maxq, &s->ar, CMP_FARTHEST, ties);
    mheap_init(&resq, &s->ar, CMP_NEAREST, ties);
    for (int32_t i = 0; i < n_start; ++i) { /* MinQueue.add_node, :430 */
        elt e = { start_nodes[i], sp_dist(sp, target, sp_value(sp, start_nodes[i])) };
        if (ctr) ctr->n_dist++;
        mheap_add(&startq, e);
    }
    ohnsw_search_k(&g->view, layer, sp, s->visited, &startq, target, k, &maxq, &resq, ctr);
    int32_t cnt = drain_ascending(&resq, resq.len, out_nodes, out_dist);
    scratch_destroy(s);
    return cnt;
}

int32_t og_ohnsw_select_neighbours(og_space *sp, const int64_t *cand, int32_t n_cand,
                                   const void *target, int32_t num_neighbours, int32_t ties,
                                   int64_t *out_nodes) {
    arena ar = { 0 };
    mheap q; mheap_init(&q, &ar, CMP_NEAREST, ties);
    for (int32_t i = 0; i < n_cand; ++i) {
        elt e = { cand[i], sp_dist(sp, target, sp_value(sp, cand[i])) };
        mheap_add(&q, e);
    }
    int32_t r = ohnsw_select_neighbours(sp, &q, num_neighbours, out_nodes);
    arena_free(&ar);
    return r;
}

/* knn, lib/ohnsw.ml:859-875, with ef (search_k's k argument) and k (how many of the ascending
 * results are reported) separated; the reference calls it with ef == k. */
static int32_t ohnsw_knn_view(const graph_view *g, int32_t max_layer, int64_t entry_point,
                              og_space *sp, scratch *s, const void *target, int32_t ef, int32_t k,
                              int32_t ties, int64_t *out_nodes, double *out_dist, og_counters *ctr) {
    if (entry_point < 0) return -1; /* invalid_arg "knn: empty hgraph", :862 */
    arena_reset(&s->ar);
    int64_t node = entry_point;
    for (int32_t layer = max_layer; layer >= 1; --layer)          /* :865-867 */
        node = ohnsw_search_one_simple(g, layer, sp, node, target, ctr);
    mheap maxq, resq, wq;
    mheap_init(&maxq, &s->ar, CMP_FARTHEST, ties);                 /* :868 */
    mheap_init(&resq, &s->ar, CMP_NEAREST, ties);                  /* :869 */
    mheap_init(&wq, &s->ar, CMP_NEAREST, ties);                    /* :870 */
    elt e = { node, sp_dist(sp, target, sp_value(sp, node)) };     /* :871 */
    if (ctr) { ctr->n_dist++; ctr->n_dist_upper = ctr->n_dist; }
    mheap_add(&wq, e);
    ohnsw_search_k(g, 0, sp, s->visited, &wq, target, ef, &maxq, &resq, ctr); /* :872-874 */
    return drain_ascending(&resq, k, out_nodes, out_dist);
}

int32_t og_ohnsw_knn(const og_graph *g, og_space *sp, const void *target, int32_t ef, int32_t k,
                     int32_t ties, int64_t *out_nodes, double *out_dist, og_counters *ctr) {
    scratch *s = scratch_create(g->view.n);
    int32_t r = ohnsw_knn_view(&g->view, g->max_layer, g->entry_point, sp, s, target, ef, k, ties,
                               out_nodes, out_dist, ctr);
    scratch_destroy(s);
    return r;
}

/* knn_batch_bigarray, lib/ohnsw.ml:877-897: sequential loop over query columns, one Visited
 * for the whole batch (:882), results popped ascending into distances.{i,j} / ids.(j-1).(i-1). */
int32_t og_ohnsw_knn_batch_split(const og_graph *g, og_space *sp, const float *Q, int64_t nq,
                                 int64_t q_stride, int32_t ef, int32_t k, int32_t ties,
                                 int32_t *out_ids, float *out_dist, uint32_t *out_ndist,
                                 uint32_t *out_nhops, uint32_t *out_ndist_upper) {
    if (g->entry_point < 0) return -1;
    scratch *s = scratch_create(g->view.n);
    int64_t *nodes = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
    double *dists = (double *)malloc(sizeof(double) * (size_t)k);
    for (int64_t j = 0; j < nq; ++j) {
        for (int32_t i = 0; i < k; ++i) { out_dist[j * k + i] = NAN; out_ids[j * k + i] = -1; } /* :880-881 */
        og_counters ctr = { 0, 0, 0, 0 };
        int32_t cnt = ohnsw_knn_view(&g->view, g->max_layer, g->entry_point, sp, s,
                                     (const void *)(Q + j * q_stride), ef, k, ties, nodes, dists, &ctr);
        for (int32_t i = 0; i < cnt; ++i) { out_dist[j * k + i] = (float)dists[i]; out_ids[j * k + i] = (int32_t)nodes[i]; }
        if (out_ndist) out_ndist[j] = (uint32_t)ctr.n_dist;
        if (out_nhops) out_nhops[j] = (uint32_t)ctr.n_hops;
        if (out_ndist_upper) out_ndist_upper[j] = (uint32_t)ctr.n_dist_upper;
    }
    free(nodes); free(dists);
    scratch_destroy(s);
    return 0;
}
int32_t og_ohnsw_knn_batch(const og_graph *g, og_space *sp, const float *Q, int64_t nq,
                           int64_t q_stride, int32_t ef, int32_t k, int32_t ties,
                           int32_t *out_ids, float *out_dist, uint32_t *out_ndist,
                           uint32_t *out_nhops) {
    return og_ohnsw_knn_batch_split(g, sp, Q, nq, q_stride, ef, k, ties, out_ids, out_dist, out_ndist, out_nhops, NULL);
}

/* The same batch loop with the queries split over host threads (each with its own Visited and
 * arena).  The reference itself is single-threaded (lib/ohnsw.ml:883 is a sequential fold); this
 * variant only exists so the CPU baseline can also be quoted on all host cores. */
#include <pthread.h>
typedef struct mt_job {
    const og_graph *g; const og_space *sp; const float *Q; int64_t q0, q1, q_stride;
    int32_t ef, k, ties; int32_t *out_ids; float *out_dist;
} mt_job;
static void *mt_worker(void *arg) {
    mt_job *j = (mt_job *)arg;
    og_space sp = *j->sp;   /* private distance-call counter */
    scratch *s = scratch_create(j->g->view.n);
    int64_t *nodes = (int64_t *)malloc(sizeof(int64_t) * (size_t)j->k);
    double *dists = (double *)malloc(sizeof(double) * (size_t)j->k);
    for (int64_t q = j->q0; q < j->q1; ++q) {
        for (int32_t i = 0; i < j->k; ++i) { j->out_dist[q * j->k + i] = NAN; j->out_ids[q * j->k + i] = -1; }
        int32_t cnt = ohnsw_knn_view(&j->g->view, j->g->max_layer, j->g->entry_point, &sp, s,
                                     (const void *)(j->Q + q * j->q_stride), j->ef, j->k, j->ties, nodes, dists, NULL);
        for (int32_t i = 0; i < cnt; ++i) { j->out_dist[q * j->k + i] = (float)dists[i]; j->out_ids[q * j->k + i] = (int32_t)nodes[i]; }
    }
    free(nodes); free(dists);
    scratch_destroy(s);
    return NULL;
}
int32_t og_ohnsw_knn_batch_mt(const og_graph *g, og_space *sp, const float *Q, int64_t nq,
                              int64_t q_stride, int32_t ef, int32_t k, int32_t ties, int32_t nthreads,
                              int32_t *out_ids, float *out_dist) {
    if (g->entry_point < 0) return -1;
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * (size_t)nthreads);
    for (int32_t t = 0; t < nthreads; ++t) {
        mt_job j = { g, sp, Q, nq * t / nthreads, nq * (t + 1) / nthreads, q_stride, ef, k, ties, out_ids, out_dist };
        jobs[t] = j;
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    for (int32_t t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
    return 0;
}

/* ======================================================================================== */
/* functor path                                                                             */
/* ======================================================================================== */

/* Nearest (W): lib/hnsw.ml:470-526 -- bounded max-heap + size. */
typedef struct nearest_t { ph_ctx c; ph_node *heap; int32_t size, max_size; } nearest_t;

/* insert_distance, lib/hnsw.ml:494-506.  Returns 1 = Inserted, 0 = Too_far.
 * The reference compares DISTANCES only (MaxHeap.Element.compare = Float.compare on
 * distance_to_target, lib/hnsw_algo.ml:126-131): farther => Too_far; otherwise add |> remove_max.
 * On an exact tie with max(W) the merge of lib/hnsw_algo.ml:25-31 puts the new element on top (the
 * else-branch wins on equality) and remove_max takes it off again: W is unchanged, yet the answer is
 * Inserted, so the caller still pushes the node to VisitMe (lib/hnsw_algo.ml:360-364).
 * TIES_HEAP runs exactly that code on the in-tree pairing heap.  TIES_CANONICAL keeps the same
 * three-way rule on distances and only fixes what the pairing heap's shape decides -- WHICH of
 * several equally far elements remove_max takes -- to the largest node id. */
static int nearest_insert_distance(nearest_t *q, elt element) {
    if (q->size < q->max_size) { q->heap = ph_add(&q->c, q->heap, element); q->size++; return 1; } /* :496-497 */
    if (!q->heap) return 0;                                                                      /* :499 */
    if (q->c.canonical) {
        if (element.dist > q->heap->e.dist) return 0;                                            /* :501 is_further */
        if (element.dist == q->heap->e.dist) return 1;                                           /* :503 add |> remove_max = identity */
        q->heap = ph_remove_top(&q->c, ph_add(&q->c, q->heap, element));                         /* :503 */
        return 1;
    }
    /* is_further element max = MaxHeap compare element max < 0 (hnsw_algo.ml:131) */
    if (!(elt_compare(&q->c, &element, &q->heap->e) < 0)) {                                      /* :501 */
        q->heap = ph_remove_top(&q->c, ph_add(&q->c, q->heap, element));                         /* :503 */
        return 1;
    }
    return 0;
}
static inline double nearest_max_distance(const nearest_t *q) { /* :514-517 */
    return q->heap ? q->heap->e.dist : -INFINITY;
}

/* Search.search, lib/hnsw_algo.ml:350-391. */
static void functor_search(const graph_view *g, int32_t layer, og_space *sp, og_visited *visited,
                           arena *ar, const elt *start, int32_t n_start, const void *target,
                           int32_t ef, int32_t ties, nearest_t *nearest_out, og_counters *ctr) {
    ph_ctx minc = { ar, CMP_NEAREST, ties };
    ph_node *visit_me = NULL;
    for (int32_t i = 0; i < n_start; ++i) visit_me = ph_add(&minc, visit_me, start[i]);
    /* nearest_of_visit_me :342-348: fold start nodes, RECOMPUTING their distances */
    nearest_t nearest; nearest.c.ar = ar; nearest.c.kind = CMP_FARTHEST; nearest.c.canonical = ties;
    nearest.heap = NULL; nearest.size = 0; nearest.max_size = ef;
    {
        /* VisitMe.fold order = MinHeap.fold (node, then subheaps) */
        int32_t cnt; elt *buf = ph_collect(ar, visit_me, &cnt);
        og_visited_clear(visited);                                       /* :388 */
        for (int32_t i = 0; i < cnt; ++i) {
            elt e = { buf[i].node, sp_dist(sp, sp_value(sp, buf[i].node), target) }; /* :345 */
            if (ctr) ctr->n_dist++;
            nearest_insert_distance(&nearest, e);
        }
        for (int32_t i = 0; i < cnt; ++i) og_visited_add(visited, buf[i].node);      /* :387-388 */
    }
    for (;;) {                                                           /* aux :367-386 */
        if (!visit_me) break;                                            /* :373 */
        elt c = visit_me->e;
        visit_me = ph_remove_top(&minc, visit_me);                       /* :372 pop_nearest */
        if (c.dist > nearest_max_distance(&nearest)) break;              /* :377 */
        if (ctr) ctr->n_hops++;
        int32_t len; const int32_t *nb = g->adj(g, layer, c.node, &len);
        for (int32_t i = 0; i < len; ++i) {                              /* :381-383 fold */
            int64_t n = nb[i];
            if (og_visited_mem(visited, n)) continue;                    /* :354 */
            og_visited_add(visited, n);                                  /* :356 */
            elt nd = { n, sp_dist(sp, sp_value(sp, n), target) };        /* :357-359 */
            if (ctr) ctr->n_dist++;
            if (nearest_insert_distance(&nearest, nd))                   /* :360-362 */
                visit_me = ph_add(&minc, visit_me, nd);                  /* :363 */
        }
    }
    *nearest_out = nearest;
}

/* Search.search_one, lib/hnsw_algo.ml:393-437 (carries the start distance, :393,1005). */
static elt functor_search_one(const graph_view *g, int32_t layer, og_space *sp, og_visited *visited,
                              arena *ar, elt start, const void *target, int32_t ties,
                              og_counters *ctr) {
    ph_ctx minc = { ar, CMP_NEAREST, ties };
    ph_node *visit_me = ph_singleton(&minc, start);                      /* :436 */
    elt nearest = start;
    og_visited_clear(visited); og_visited_add(visited, start.node);      /* :433-434 */
    for (;;) {
        if (!visit_me) break;                                            /* :413 */
        elt c = visit_me->e;
        visit_me = ph_remove_top(&minc, visit_me);
        if (c.dist > nearest.dist) break;                                /* :419 */
        if (ctr) ctr->n_hops_upper++;
        int32_t len; const int32_t *nb = g->adj(g, layer, c.node, &len);
        for (int32_t i = 0; i < len; ++i) {
            int64_t n = nb[i];
            if (og_visited_mem(visited, n)) continue;                    /* :396 */
            og_visited_add(visited, n);
            elt nd = { n, sp_dist(sp, sp_value(sp, n), target) };        /* :399-401 */
            if (ctr) ctr->n_dist++;
            if (nd.dist >= nearest.dist) continue;                       /* :402 */
            visit_me = ph_add(&minc, visit_me, nd);                      /* :405 */
            nearest = nd;                                                /* :406 */
        }
    }
    return nearest;
}

/* W as a list: fold_far_to_near (lib/hnsw.ml:519-520) = repeated top/remove_top. */
static int32_t nearest_drain_far_to_near(nearest_t *q, elt *out, int32_t cap) {
    int32_t cnt = 0;
    ph_node *h = q->heap;
    while (h && cnt < cap) { out[cnt++] = h->e; h = ph_remove_top(&q->c, h); }
    return cnt;
}

static int32_t functor_knn_view(const graph_view *g, int32_t max_layer, int64_t entry_point,
                                og_space *sp, scratch *s, const void *target, int32_t ef,
                                int32_t k, int32_t ties, int32_t bug_compat, int64_t *out_nodes,
                                double *out_dist, og_counters *ctr) {
    if (entry_point < 0) return -1;
    arena_reset(&s->ar);
    /* Knn.knn, lib/hnsw_algo.ml:990-1011 */
    elt start = { entry_point, sp_dist(sp, sp_value(sp, entry_point), target) }; /* :1000-1006 */
    if (ctr) ctr->n_dist++;
    for (int32_t l = max_layer; l >= 1; --l)                                       /* :993-998 */
        start = functor_search_one(g, l, sp, s->visited, &s->ar, start, target, ties, ctr);
    nearest_t nearest;
    functor_search(g, 0, sp, s->visited, &s->ar, &start, 1, target, ef, ties, &nearest, ctr); /* :1007-1010 */
    elt *far = (elt *)arena_alloc(&s->ar, sizeof(elt) * (size_t)(ef > 0 ? ef : 1));
    int32_t cnt = nearest_drain_far_to_near(&nearest, far, ef);
    int32_t nout = 0;
    if (bug_compat) {
        /* nearest_k, lib/hnsw.ml:522-525: walks far->near, keeps the FIRST k it meets, consing
         * (so the returned list is near-first among those k farthest). */
        int32_t m = cnt < k ? cnt : k;
        for (int32_t i = m - 1; i >= 0; --i) { out_nodes[nout] = far[i].node; if (out_dist) out_dist[nout] = far[i].dist; nout++; }
    } else {
        int32_t m = cnt < k ? cnt : k;
        for (int32_t i = 0; i < m; ++i) { out_nodes[nout] = far[cnt - 1 - i].node; if (out_dist) out_dist[nout] = far[cnt - 1 - i].dist; nout++; }
    }
    return nout;
}

int64_t og_functor_search_one(const og_graph *g, int32_t layer, og_space *sp, int64_t start,
                              const void *target, int32_t ties, double *out_dist, og_counters *ctr) {
    scratch *s = scratch_create(g->view.n);
    elt st = { start, sp_dist(sp, sp_value(sp, start), target) };
    elt r = functor_search_one(&g->view, layer, sp, s->visited, &s->ar, st, target, ties, ctr);
    if (out_dist) *out_dist = r.dist;
    scratch_destroy(s);
    return r.node;
}

int32_t og_functor_search(const og_graph *g, int32_t layer, og_space *sp,
                          const int64_t *start_nodes, int32_t n_start, const void *target,
                          int32_t ef, int32_t ties, int64_t *out_nodes, double *out_dist,
                          og_counters *ctr) {
    scratch *s = scratch_create(g->view.n);
    elt *st = (elt *)malloc(sizeof(elt) * (size_t)(n_start > 0 ? n_start : 1));
    for (int32_t i = 0; i < n_start; ++i) { st[i].node = start_nodes[i]; st[i].dist = sp_dist(sp, sp_value(sp, start_nodes[i]), target); }
    nearest_t nearest;
    functor_search(&g->view, layer, sp, s->visited, &s->ar, st, n_start, target, ef, ties, &nearest, ctr);
    elt *far = (elt *)malloc(sizeof(elt) * (size_t)(ef > 0 ? ef : 1));
    int32_t cnt = nearest_drain_far_to_near(&nearest, far, ef);
    for (int32_t i = 0; i < cnt; ++i) { out_nodes[i] = far[cnt - 1 - i].node; if (out_dist) out_dist[i] = far[cnt - 1 - i].dist; }
    free(far); free(st);
    scratch_destroy(s);
    return cnt;
}

int32_t og_functor_knn(const og_graph *g, og_space *sp, const void *target, int32_t ef,
                       int32_t k, int32_t ties, int32_t bug_compat_farthest_k,
                       int64_t *out_nodes, double *out_dist, og_counters *ctr) {
    scratch *s = scratch_create(g->view.n);
    int32_t r = functor_knn_view(&g->view, g->max_layer, g->entry_point, sp, s, target, ef, k, ties,
                                 bug_compat_farthest_k, out_nodes, out_dist, ctr);
    scratch_destroy(s);
    return r;
}

/* MakeBatch.knn_batch, lib/hnsw.ml:769-777: distances only, +inf filled (:771). out_ids is an
 * oracle-side extra (may be NULL). */
int32_t og_functor_knn_batch(const og_graph *g, og_space *sp, const float *Q, int64_t nq,
                             int64_t q_stride, int32_t ef, int32_t k, int32_t ties,
                             int32_t bug_compat_farthest_k, float *out_dist, int32_t *out_ids) {
    if (g->entry_point < 0) return -1;
    scratch *s = scratch_create(g->view.n);
    int64_t *nodes = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
    double *dists = (double *)malloc(sizeof(double) * (size_t)k);
    for (int64_t j = 0; j < nq; ++j) {
        for (int32_t i = 0; i < k; ++i) { out_dist[j * k + i] = INFINITY; if (out_ids) out_ids[j * k + i] = -1; }
        int32_t cnt = functor_knn_view(&g->view, g->max_layer, g->entry_point, sp, s,
                                       (const void *)(Q + j * q_stride), ef, k, ties,
                                       bug_compat_farthest_k, nodes, dists, NULL);
        for (int32_t i = 0; i < cnt; ++i) { out_dist[j * k + i] = (float)dists[i]; if (out_ids) out_ids[j * k + i] = (int32_t)nodes[i]; }
    }
    free(nodes); free(dists);
    scratch_destroy(s);
    return 0;
}

/* SelectNeighbours.select_neighbours, lib/hnsw_algo.ml:572-609. */
int32_t og_functor_select_neighbours(og_space *sp, const int64_t *cand, const double *cand_dist,
                                     const int32_t *cand_degree, int32_t n_cand,
                                     int32_t num_neighbours, int32_t do_not_isolate,
                                     int32_t ties, int64_t *out_nodes) {
    arena ar = { 0 };
    ph_ctx minc = { &ar, CMP_NEAREST, ties };
    ph_node *min_heap = NULL;
    int32_t nsel = 0, num_candidates = 0;
    for (int32_t i = 0; i < n_cand; ++i) {                                    /* :587-594 */
        num_candidates++;
        if (do_not_isolate && cand_degree && cand_degree[i] <= 1) out_nodes[nsel++] = cand[i]; /* :591-592 */
        else { elt e = { cand[i], cand_dist[i] }; min_heap = ph_add(&minc, min_heap, e); }
    }
    if (num_candidates <= num_neighbours) {                                   /* :596-599 */
        /* MinHeap.fold order: node, then subheaps */
        int32_t cnt; elt *buf = ph_collect(&ar, min_heap, &cnt);
        for (int32_t i = 0; i < cnt; ++i) out_nodes[nsel++] = buf[i].node;
    } else {
        ph_node *h = min_heap;                                                /* :600-606 */
        while (h) {
            elt node = h->e;
            int closer = 1;                                                   /* is_closer :578-583 */
            for (int32_t j = nsel - 1; j >= 0; --j) {
                double dn = sp_dist(sp, sp_value(sp, node.node), sp_value(sp, out_nodes[j]));
                if (!(dn > node.dist)) { closer = 0; break; }
            }
            if (closer) {
                out_nodes[nsel++] = node.node;
                if (nsel >= num_neighbours) break;                            /* :603 */
            }
            h = ph_remove_top(&minc, h);
        }
    }
    arena_free(&ar);
    return nsel;
}

/* ======================================================================================== */
/* test-graph generator: Ohnsw.insert / build_batch_bigarray, lib/ohnsw.ml:766-857          */
/* ======================================================================================== */
typedef struct nlist { int32_t *items; int32_t len, cap; } nlist; /* items[0] = head of the OCaml list */
typedef struct blayer { nlist *nodes; int64_t n, cap; } blayer;   /* Graph.t = Neighbours.t Vector.t */

struct og_builder {
    graph_view view;
    blayer *layers; int32_t n_layers, cap_layers;
    int64_t entry_point;
    int32_t *level; /* level drawn for each node (0 for the first) */
    int64_t n;
};

static void nlist_cons(nlist *l, int32_t x) { /* Neighbours.add :116-118 */
    if (l->len == l->cap) { l->cap = l->cap ? 2 * l->cap : 8; l->items = (int32_t *)realloc(l->items, sizeof(int32_t) * (size_t)l->cap); }
    memmove(l->items + 1, l->items, sizeof(int32_t) * (size_t)l->len);
    l->items[0] = x; l->len++;
}
static void nlist_remove(nlist *l, int32_t node) { /* Neighbours.remove :119-124: rebuilds by consing => order REVERSED */
    int32_t w = 0;
    for (int32_t i = 0; i < l->len; ++i) if (l->items[i] != node) l->items[w++] = l->items[i];
    l->len = w;
    for (int32_t i = 0; i < w / 2; ++i) { int32_t t = l->items[i]; l->items[i] = l->items[w - 1 - i]; l->items[w - 1 - i] = t; }
}
static int nlist_mem(const nlist *l, int32_t x) { for (int32_t i = 0; i < l->len; ++i) if (l->items[i] == x) return 1; return 0; }

static const int32_t *builder_adj(const graph_view *v, int32_t layer, int64_t node, int32_t *len) {
    const og_builder *b = (const og_builder *)v;
    if (layer >= b->n_layers || node >= b->layers[layer].n) { *len = 0; return NULL; }
    const nlist *l = &b->layers[layer].nodes[node];
    *len = l->len;
    return l->items;
}
static void blayer_push(blayer *L) { /* Graph.add_node :173-174 */
    if (L->n == L->cap) {
        int64_t nc = L->cap ? 2 * L->cap + 10 : 128;
        L->nodes = (nlist *)realloc(L->nodes, sizeof(nlist) * (size_t)nc);
        memset(L->nodes + L->cap, 0, sizeof(nlist) * (size_t)(nc - L->cap));
        L->cap = nc;
    }
    memset(&L->nodes[L->n], 0, sizeof(nlist));
    L->n++;
}
static void builder_set_max_layer(og_builder *b, int32_t n) { /* Hgraph.set_max_layer :347-351 */
    int64_t num_nodes = b->layers[b->n_layers - 1].n;
    for (int32_t i = b->n_layers; i <= n; ++i) {
        if (b->n_layers == b->cap_layers) { b->cap_layers *= 2; b->layers = (blayer *)realloc(b->layers, sizeof(blayer) * (size_t)b->cap_layers); }
        blayer *L = &b->layers[b->n_layers++];
        L->n = L->cap = num_nodes;
        L->nodes = (nlist *)calloc((size_t)(num_nodes > 0 ? num_nodes : 1), sizeof(nlist));
    }
}
static int cmp_i32(const void *a, const void *b) { int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return (x > y) - (x < y); }

/* Graph.set_connections :182-196 (symmetric maintenance; Set.iter = ascending ids).
 * new_list is in OCaml list order (head first) and becomes the node's list as-is. */
static void graph_set_connections(blayer *L, int32_t node, const int32_t *new_list, int32_t new_len) {
    nlist *old = &L->nodes[node];
    int32_t *olds = (int32_t *)malloc(sizeof(int32_t) * (size_t)(old->len + 1));
    int32_t *news = (int32_t *)malloc(sizeof(int32_t) * (size_t)(new_len + 1));
    int32_t nold = old->len;
    memcpy(olds, old->items, sizeof(int32_t) * (size_t)nold);
    memcpy(news, new_list, sizeof(int32_t) * (size_t)new_len);
    qsort(olds, (size_t)nold, sizeof(int32_t), cmp_i32);
    qsort(news, (size_t)new_len, sizeof(int32_t), cmp_i32);
    /* 1. set the node's list */
    if (old->cap < new_len) { old->cap = new_len; old->items = (int32_t *)realloc(old->items, sizeof(int32_t) * (size_t)old->cap); }
    memcpy(old->items, new_list, sizeof(int32_t) * (size_t)new_len);
    old->len = new_len;
    /* 2. removed = old \ new */
    for (int32_t i = 0; i < nold; ++i) {
        if (i > 0 && olds[i] == olds[i - 1]) continue;
        if (!bsearch(&olds[i], news, (size_t)new_len, sizeof(int32_t), cmp_i32)) nlist_remove(&L->nodes[olds[i]], node);
    }
    /* 3. added = new \ old */
    for (int32_t i = 0; i < new_len; ++i) {
        if (i > 0 && news[i] == news[i - 1]) continue;
        if (!bsearch(&news[i], olds, (size_t)nold, sizeof(int32_t), cmp_i32)) nlist_cons(&L->nodes[news[i]], node);
    }
    free(olds); free(news);
}
/* set_connections_for_new_node :198-202 */
static void graph_set_connections_new(blayer *L, int32_t node, const int32_t *list, int32_t len) {
    nlist *l = &L->nodes[node];
    if (l->cap < len) { l->cap = len; l->items = (int32_t *)realloc(l->items, sizeof(int32_t) * (size_t)l->cap); }
    memcpy(l->items, list, sizeof(int32_t) * (size_t)len);
    l->len = len;
    for (int32_t i = 0; i < len; ++i) nlist_cons(&L->nodes[list[i]], node);
}

/* splitmix64 -> uniform (0,1) */
static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline double rng_unit(uint64_t *s) { return ((double)(splitmix64(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

static mheap mheap_copy(const mheap *h) { return *h; } /* MinQueue.copy: persistent structure => O(1) */

og_builder *og_build_ohnsw(og_space *sp, int64_t n, int32_t num_connections,
                           int32_t num_nodes_search_construction, uint64_t seed, int32_t ties) {
    og_builder *b = (og_builder *)calloc(1, sizeof(og_builder));
    b->view.adj = builder_adj; b->view.n = n;
    b->cap_layers = 16; b->layers = (blayer *)calloc((size_t)b->cap_layers, sizeof(blayer));
    b->n_layers = 1;               /* Hgraph.create pushes one empty layer :323 */
    b->entry_point = -1;
    b->level = (int32_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t));
    b->n = n;
    double level_mult = 1.0 / log((double)num_connections);               /* :844 */
    og_visited *visited = og_visited_create(n);                            /* :845 */
    arena ar = { 0 };
    uint64_t rng = seed;
    int32_t M = num_connections, efc = num_nodes_search_construction;
    int64_t *sel = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 * M + 2));
    int32_t *sel32 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(2 * M + 2));

    for (int64_t it = 0; it < n; ++it) {                                   /* fold_cols :848 */
        const void *target = sp_value(sp, it);
        /* insert, :766-837 */
        for (int32_t l = 0; l < b->n_layers; ++l) blayer_push(&b->layers[l]); /* Hgraph.add_node :328-330 */
        int32_t new_node = (int32_t)(b->layers[0].n - 1);
        if (b->entry_point < 0) { b->entry_point = new_node; continue; }   /* :774-778 */
        arena_reset(&ar);
        og_visited_clear(visited);                                         /* :780 */
        double u = rng_unit(&rng);
        int32_t level = (int32_t)floor(-log(u) * level_mult + 0.5);        /* :781 round_nearest */
        b->level[new_node] = level;
        int64_t node = b->entry_point;
        int32_t max_layer = b->n_layers - 1;
        for (int32_t layer = max_layer; layer >= level + 1; --layer)       /* :785-789 */
            node = ohnsw_search_one_simple(&b->view, layer, sp, node, target, NULL);
        mheap nearest_maxq, wq, nearest_minq;
        mheap_init(&nearest_maxq, &ar, CMP_FARTHEST, ties);                /* :800 */
        mheap_init(&wq, &ar, CMP_NEAREST, ties);                           /* :801 */
        { elt e = { node, sp_dist(sp, target, sp_value(sp, node)) }; mheap_add(&wq, e); } /* :802 */
        mheap_init(&nearest_minq, &ar, CMP_NEAREST, ties);                 /* :804 */
        mheap *w_queue = &wq, *n_minq = &nearest_minq;
        for (int32_t layer = (level < max_layer ? level : max_layer); layer >= 0; --layer) { /* :806 */
            blayer *L = &b->layers[layer];
            ohnsw_search_k(&b->view, layer, sp, visited, w_queue, target, efc, &nearest_maxq, n_minq, NULL); /* :811-812 */
            { mheap *t = w_queue; w_queue = n_minq; n_minq = t; }           /* :814-816 */
            int32_t nconn = layer == 0 ? 2 * M : M;                         /* :818 */
            mheap cq = mheap_copy(w_queue);
            int32_t nsel = ohnsw_select_neighbours(sp, &cq, nconn, sel);    /* :819 */
            /* OCaml list order = reverse selection order (consing) */
            for (int32_t i = 0; i < nsel; ++i) sel32[i] = (int32_t)sel[nsel - 1 - i];
            graph_set_connections_new(L, new_node, sel32, nsel);            /* :820 */
            /* :821 Neighbours.iter neighbours (the list object just stored: snapshot it) */
            int32_t *iter = (int32_t *)arena_alloc(&ar, sizeof(int32_t) * (size_t)(nsel + 1));
            memcpy(iter, sel32, sizeof(int32_t) * (size_t)nsel);
            for (int32_t i = 0; i < nsel; ++i) {
                int32_t neighbour = iter[i];
                nlist *nn = &L->nodes[neighbour];
                if (nn->len > nconn) {                                      /* :823 */
                    mheap nq; mheap_init(&nq, &ar, CMP_NEAREST, ties);      /* :791-798 */
                    const void *nv = sp_value(sp, neighbour);
                    for (int32_t j = 0; j < nn->len; ++j) { elt e = { nn->items[j], sp_dist(sp, nv, sp_value(sp, nn->items[j])) }; mheap_add(&nq, e); }
                    int64_t *rs = (int64_t *)arena_alloc(&ar, sizeof(int64_t) * (size_t)(nconn + 1));
                    int32_t nr = ohnsw_select_neighbours(sp, &nq, nconn, rs); /* :825-827 */
                    int32_t *rl = (int32_t *)arena_alloc(&ar, sizeof(int32_t) * (size_t)(nr + 1));
                    for (int32_t j = 0; j < nr; ++j) rl[j] = (int32_t)rs[nr - 1 - j];
                    graph_set_connections(L, neighbour, rl, nr);            /* :828 */
                }
            }
        }
        if (level > max_layer) {                                            /* :832-836 */
            builder_set_max_layer(b, level);
            b->entry_point = new_node;
        }
    }
    free(sel); free(sel32);
    og_visited_destroy(visited);
    arena_free(&ar);
    return b;
}

void og_builder_destroy(og_builder *b) {
    if (!b) return;
    for (int32_t l = 0; l < b->n_layers; ++l) {
        for (int64_t i = 0; i < b->layers[l].n; ++i) free(b->layers[l].nodes[i].items);
        free(b->layers[l].nodes);
    }
    free(b->layers); free(b->level); free(b);
}
int32_t og_builder_max_layer(const og_builder *b) { return b->n_layers - 1; }
int64_t og_builder_entry_point(const og_builder *b) { return b->entry_point; }
static int builder_in_layer(const og_builder *b, int32_t layer, int64_t node) {
    /* a node "is in" upper layer l if it was inserted there (level >= l) or is the entry point
     * of a newly created layer (it has no links there but searches start from it) */
    if (layer == 0) return 1;
    if (b->level[node] >= layer) return 1;
    return b->layers[layer].nodes[node].len > 0;
}
int64_t og_builder_layer_count(const og_builder *b, int32_t layer) {
    if (layer >= b->n_layers) return 0;
    int64_t c = 0;
    for (int64_t i = 0; i < b->layers[layer].n; ++i) c += builder_in_layer(b, layer, i);
    return c;
}
int32_t og_builder_export_layer0(const og_builder *b, int32_t stride, int32_t *deg0, int32_t *nbr0) {
    int32_t maxdeg = 0;
    for (int64_t i = 0; i < b->layers[0].n; ++i) {
        const nlist *l = &b->layers[0].nodes[i];
        if (l->len > stride) return -1; /* flatten must fail loudly, never truncate */
        if (l->len > maxdeg) maxdeg = l->len;
        deg0[i] = l->len;
        for (int32_t j = 0; j < stride; ++j) nbr0[i * stride + j] = j < l->len ? l->items[j] : -1;
    }
    return maxdeg;
}
int64_t og_builder_export_upper(const og_builder *b, int32_t layer, int32_t stride,
                                int64_t *nodes, int32_t *deg, int32_t *nbr) {
    int64_t s = 0;
    for (int64_t i = 0; i < b->layers[layer].n; ++i) {
        if (!builder_in_layer(b, layer, i)) continue;
        const nlist *l = &b->layers[layer].nodes[i];
        if (l->len > stride) return -1;
        nodes[s] = i; deg[s] = l->len;
        for (int32_t j = 0; j < stride; ++j) nbr[s * stride + j] = j < l->len ? l->items[j] : -1;
        s++;
    }
    return s;
}
int32_t og_builder_invariant(const og_builder *b) { /* Graph.Test.invariant :217-225 + Hgraph :354-359 */
    for (int32_t l = 0; l < b->n_layers; ++l) {
        if (b->layers[l].n != b->layers[0].n) return 0;
        for (int64_t i = 0; i < b->layers[l].n; ++i) {
            const nlist *li = &b->layers[l].nodes[i];
            for (int32_t j = 0; j < li->len; ++j)
                if (!nlist_mem(&b->layers[l].nodes[li->items[j]], (int32_t)i)) return 0;
        }
    }
    return b->entry_point < b->layers[0].n;
}

/* ======================================================================================== */
/* brute force + recall: benchmark/dataset.ml:15-30, 105-127                                */
/* ======================================================================================== */
void og_brute_force_knn(og_space *sp, const float *Q, int64_t nq, int64_t q_stride, int32_t k,
                        int32_t *out_ids, float *out_dist) {
    /* bounded insertion into a sorted (distance, id) list; the reference sorts all n distances */
    double *bd = (double *)malloc(sizeof(double) * (size_t)k);
    int32_t *bi = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    for (int64_t q = 0; q < nq; ++q) {
        int32_t cnt = 0;
        const void *t = (const void *)(Q + q * q_stride);
        for (int64_t j = 0; j < sp->n; ++j) {
            double d = og_distance_raw(sp, sp_value(sp, j), t);
            if (cnt == k && !(d < bd[k - 1])) continue;
            int32_t p = cnt < k ? cnt : k - 1;
            while (p > 0 && (bd[p - 1] > d)) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; p--; }
            bd[p] = d; bi[p] = (int32_t)j;
            if (cnt < k) cnt++;
        }
        for (int32_t i = 0; i < k; ++i) {
            out_dist[q * k + i] = i < cnt ? (float)bd[i] : NAN;
            out_ids[q * k + i] = i < cnt ? bi[i] : -1;
        }
    }
    free(bd); free(bi);
}
/* Recall.compute, benchmark/dataset.ml:107-126: fraction of returned distances <= the true
 * k-th distance + epsilon. expected/got are [nq][k]. */
double og_recall_distance_threshold(const float *expected, const float *got, int64_t nq,
                                    int32_t k, double epsilon) {
    double ret = 0.0;
    for (int64_t q = 0; q < nq; ++q) {
        int32_t ok = 0;
        for (int32_t i = 0; i < k; ++i)
            if ((double)got[q * k + i] <= (double)expected[q * k + (k - 1)] + epsilon) ok++;
        ret += (double)ok / (double)k;
    }
    return ret / (double)nq;
}
