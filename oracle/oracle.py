"""ctypes front for oracle/liboracle.so -- the CPU restatement of ocaml-hnsw's search path.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never from the product package (ocaml-hnsw_amd/).

Names follow the reference: Ohnsw.search_one / search_k / select_neighbours / knn /
knn_batch_bigarray (lib/ohnsw.ml) and Hnsw_algo.Search.search / search_one / Knn.knn,
Hnsw.Ba.knn_batch (lib/hnsw_algo.ml, lib/hnsw.ml).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_SANITIZE=1 (tests/test_oracle_sanitizers.py, in a child process with libasan preloaded): an AddressSanitizer +
# UndefinedBehaviorSanitizer build of the same restatement under another name
_SANITIZE = bool(os.environ.get("ORACLE_SANITIZE"))
_LIB_PATH = os.path.join(_HERE, "liboracle_san.so" if _SANITIZE else "liboracle.so")

SCALAR_ABS, L2, IP = 0, 1, 2
SEQ_F32, F64, TREE16 = 0, 1, 2
TIES_HEAP, TIES_CANONICAL = 0, 1


def build(force=False):
    """gcc the restatement (plain C; -ffp-contract=off so only the explicit fmaf calls fuse)."""
    src = os.path.join(_HERE, "hnsw_oracle.c")
    hdr = os.path.join(_HERE, "hnsw_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-fPIC", "-shared", "-mavx2", "-mfma",
           "-ffp-contract=off", src, "-o", _LIB_PATH, "-lm", "-lpthread"]
    if _SANITIZE:
        cmd[1:2] = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    subprocess.check_call(cmd)
    return _LIB_PATH


class _Space(C.Structure):
    _fields_ = [("kind", C.c_int32), ("arith", C.c_int32), ("X", C.c_void_p), ("n", C.c_int64),
                ("d", C.c_int32), ("stride", C.c_int64), ("S", C.c_void_p),
                ("n_calls", C.c_uint64)]


class _Counters(C.Structure):
    _fields_ = [("n_dist", C.c_uint64), ("n_hops", C.c_uint64), ("n_hops_upper", C.c_uint64),
                ("n_dist_upper", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.og_graph_create.restype = vp
    L.og_graph_create.argtypes = [i64, i32, i64, i32, vp, vp, i32, vp, vp, vp, vp]
    L.og_graph_destroy.argtypes = [vp]
    L.og_layer_stats.restype = i64
    L.og_layer_stats.argtypes = [vp, i32, vp, vp, vp, vp, vp, i64]
    L.og_ohnsw_search_one.restype = i64
    L.og_ohnsw_search_one.argtypes = [vp, i32, vp, i64, vp, i32]
    L.og_ohnsw_search_k.restype = i32
    L.og_ohnsw_search_k.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, vp, vp, vp]
    L.og_ohnsw_select_neighbours.restype = i32
    L.og_ohnsw_select_neighbours.argtypes = [vp, vp, i32, vp, i32, i32, vp]
    L.og_ohnsw_knn.restype = i32
    L.og_ohnsw_knn.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp]
    L.og_ohnsw_knn_batch.restype = i32
    L.og_ohnsw_knn_batch.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, vp]
    L.og_ohnsw_knn_batch_split.restype = i32
    L.og_ohnsw_knn_batch_split.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, vp]
    L.og_ohnsw_knn_batch_mt.restype = i32
    L.og_ohnsw_knn_batch_mt.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, i32, vp, vp]
    L.og_functor_search_one.restype = i64
    L.og_functor_search_one.argtypes = [vp, i32, vp, i64, vp, i32, vp, vp]
    L.og_functor_search.restype = i32
    L.og_functor_search.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, vp, vp, vp]
    L.og_functor_knn.restype = i32
    L.og_functor_knn.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
    L.og_functor_knn_batch.restype = i32
    L.og_functor_knn_batch.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, i32, vp, vp]
    L.og_functor_select_neighbours.restype = i32
    L.og_functor_select_neighbours.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.og_build_ohnsw.restype = vp
    L.og_build_ohnsw.argtypes = [vp, i64, i32, i32, C.c_uint64, i32]
    L.og_builder_destroy.argtypes = [vp]
    L.og_builder_max_layer.restype = i32
    L.og_builder_max_layer.argtypes = [vp]
    L.og_builder_entry_point.restype = i64
    L.og_builder_entry_point.argtypes = [vp]
    L.og_builder_layer_count.restype = i64
    L.og_builder_layer_count.argtypes = [vp, i32]
    L.og_builder_export_layer0.restype = i32
    L.og_builder_export_layer0.argtypes = [vp, i32, vp, vp]
    L.og_builder_export_upper.restype = i64
    L.og_builder_export_upper.argtypes = [vp, i32, i32, vp, vp, vp]
    L.og_builder_invariant.restype = i32
    L.og_builder_invariant.argtypes = [vp]
    L.og_brute_force_knn.argtypes = [vp, vp, i64, i64, i32, vp, vp]
    L.og_recall_distance_threshold.restype = f64
    L.og_recall_distance_threshold.argtypes = [vp, vp, i64, i32, f64]
    L.og_l2sq_tree16.restype = C.c_float
    L.og_l2sq_tree16.argtypes = [vp, vp, i32]
    L.og_dot_tree16.restype = C.c_float
    L.og_dot_tree16.argtypes = [vp, vp, i32]
    L.og_l2sq_seq.restype = C.c_float
    L.og_l2sq_seq.argtypes = [vp, vp, i32]
    L.og_l2sq_f64.restype = f64
    L.og_l2sq_f64.argtypes = [vp, vp, i32]
    L.og_visited_create.restype = vp
    L.og_visited_create.argtypes = [i64]
    L.og_visited_destroy.argtypes = [vp]
    L.og_visited_mem.restype = i32
    L.og_visited_mem.argtypes = [vp, i64]
    L.og_visited_add.argtypes = [vp, i64]
    L.og_visited_clear.argtypes = [vp]
    L.og_visited_card.restype = i64
    L.og_visited_card.argtypes = [vp]
    L.og_visited_set_epoch.argtypes = [vp, i64]
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Space:
    """DISTANCE + VALUE of the reference (lib/hnsw_algo.ml:75-84): a distance and a node->value map."""

    def __init__(self, kind, arith=SEQ_F32, vectors=None, scalars=None):
        self.kind, self.arith = kind, arith
        self.vectors = None
        self.scalars = None
        s = _Space()
        s.kind, s.arith = kind, arith
        if kind == SCALAR_ABS:
            self.scalars = np.ascontiguousarray(scalars, dtype=np.float64)
            s.S = self.scalars.ctypes.data
            s.n = self.scalars.shape[0]
            s.d, s.stride = 1, 1
        else:
            self.vectors = np.ascontiguousarray(vectors, dtype=np.float32)
            s.X = self.vectors.ctypes.data
            s.n, s.d = self.vectors.shape
            s.stride = self.vectors.shape[1]
        self._s = s

    @classmethod
    def scalar(cls, values):
        return cls(SCALAR_ABS, scalars=values)

    @classmethod
    def l2(cls, vectors, arith=SEQ_F32):
        return cls(L2, arith, vectors=vectors)

    @classmethod
    def ip(cls, vectors, arith=SEQ_F32):
        return cls(IP, arith, vectors=vectors)

    @property
    def n(self):
        return int(self._s.n)

    @property
    def n_calls(self):
        return int(self._s.n_calls)

    def ref(self):
        return C.byref(self._s)

    def target(self, t):
        """Box a query (scalar or vector) the way the C side expects it; keeps it alive."""
        if self.kind == SCALAR_ABS:
            a = np.array([t], dtype=np.float64)
        else:
            a = np.ascontiguousarray(t, dtype=np.float32)
        return a


class Graph:
    """Flattened layered graph, 0-based ids: layer 0 as a fixed-stride table, upper layers sparse.
    Rows keep the reference's iteration order (Neighbours.iter, lib/ohnsw.ml:127)."""

    def __init__(self, n, entry_point, deg0, nbr0, upper=()):
        self.n = int(n)
        self.entry_point = int(entry_point)
        self.deg0 = np.ascontiguousarray(deg0, dtype=np.int32)
        self.nbr0 = np.ascontiguousarray(nbr0, dtype=np.int32).reshape(self.n, -1) if self.n else \
            np.zeros((0, max(1, np.asarray(nbr0).shape[-1] if np.asarray(nbr0).ndim > 1 else 1)), np.int32)
        self.stride0 = int(self.nbr0.shape[1])
        self.upper = []
        for (nodes, deg, nbr) in upper:
            nodes = np.ascontiguousarray(nodes, dtype=np.int64)
            deg = np.ascontiguousarray(deg, dtype=np.int32)
            nbr = np.ascontiguousarray(nbr, dtype=np.int32).reshape(len(nodes), -1)
            self.upper.append((nodes, deg, nbr))
        self.max_layer = len(self.upper)
        self.strideU = int(self.upper[0][2].shape[1]) if self.upper else 1
        L = lib()
        nl = self.max_layer
        self._un = (C.c_int64 * max(nl, 1))(*[len(u[0]) for u in self.upper])
        self._unodes = (C.c_void_p * max(nl, 1))(*[u[0].ctypes.data for u in self.upper])
        self._udeg = (C.c_void_p * max(nl, 1))(*[u[1].ctypes.data for u in self.upper])
        self._unbr = (C.c_void_p * max(nl, 1))(*[u[2].ctypes.data for u in self.upper])
        self._h = L.og_graph_create(self.n, nl, self.entry_point, self.stride0, _ptr(self.deg0),
                                    _ptr(self.nbr0), self.strideU, self._un, self._unodes,
                                    self._udeg, self._unbr)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().og_graph_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @classmethod
    def from_lists(cls, adjacency, entry_point=-1, stride=None):
        """Single-layer graph from python lists (list order = iteration order)."""
        n = len(adjacency)
        stride = stride or max([len(a) for a in adjacency] + [1])
        deg0 = np.array([len(a) for a in adjacency], dtype=np.int32)
        nbr0 = np.full((n, stride), -1, dtype=np.int32)
        for i, a in enumerate(adjacency):
            nbr0[i, :len(a)] = a
        return cls(n, entry_point, deg0, nbr0)

    @classmethod
    def ring(cls, n, entry_point=-1):
        """Graph.Test.create_loop (lib/ohnsw.ml:205-212): set_connections g i [i-1; i+1] for
        i = 0..n-1, with the back-links Graph.set_connections maintains (:182-196)."""
        lists = [[] for _ in range(n)]

        def remove(lst, x):  # Neighbours.remove reverses the survivors (:119-124)
            out = []
            for e in lst:
                if e != x:
                    out.insert(0, e)
            return out

        for i in range(n):
            new = [(i - 1) % n, (i + 1) % n]
            old = lists[i]
            added = sorted(set(new) - set(old))
            removed = sorted(set(old) - set(new))
            lists[i] = list(new)
            for r in removed:
                lists[r] = remove(lists[r], i)
            for a in added:
                lists[a].insert(0, i)
        return cls.from_lists(lists, entry_point)


def _ctr():
    return _Counters()


class Stats:
    """Hgraph.Stats (lib/hnsw.ml:353-375)"""

    @staticmethod
    def compute(graph):
        """Stats.compute hgraph -> {num_nodes, layer_sizes {layer: size}, layer_connectivity {layer: mima}} with
        mima = {min, max, mean, isolated}; `isolated` is the reference's list (node ids, descending: consed during the
        ascending Map.fold of min_max_connectivity, :361-368)."""
        out = {"num_nodes": graph.n, "layer_sizes": {}, "layer_connectivity": {}}
        for layer in range(graph.max_layer + 1):
            nn, mi, ma, mean = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_double(0.0)
            iso = np.empty(max(graph.n, 1), np.int64)
            c = lib().og_layer_stats(graph._h, layer, C.byref(nn), C.byref(mi), C.byref(ma), C.byref(mean), _ptr(iso), graph.n)
            assert c >= 0
            out["layer_sizes"][layer] = int(nn.value)
            out["layer_connectivity"][layer] = {"min": int(mi.value), "max": int(ma.value), "mean": float(mean.value),
                                                "isolated": iso[:c].tolist()}
        return out


class Ohnsw:
    """lib/ohnsw.ml"""

    @staticmethod
    def search_one(graph, space, start_node, target, layer=0, paper=False):
        t = space.target(target)
        return int(lib().og_ohnsw_search_one(graph._h, layer, space.ref(), start_node, _ptr(t), int(paper)))

    @staticmethod
    def search_k(graph, space, start_nodes, target, k, layer=0, ties=TIES_HEAP, counters=False):
        t = space.target(target)
        st = np.ascontiguousarray(start_nodes, dtype=np.int64)
        cap = graph.n + len(st) + 1
        nodes = np.empty(cap, np.int64)
        dist = np.empty(cap, np.float64)
        c = _ctr()
        cnt = lib().og_ohnsw_search_k(graph._h, layer, space.ref(), _ptr(st), len(st), _ptr(t), k,
                                      ties, _ptr(nodes), _ptr(dist), C.byref(c))
        res = list(zip(nodes[:cnt].tolist(), dist[:cnt].tolist()))
        return (res, c) if counters else res

    @staticmethod
    def select_neighbours(space, candidates, target, num_neighbours, ties=TIES_HEAP):
        t = space.target(target)
        cand = np.ascontiguousarray(candidates, dtype=np.int64)
        out = np.empty(max(len(cand), 1), np.int64)
        cnt = lib().og_ohnsw_select_neighbours(space.ref(), _ptr(cand), len(cand), _ptr(t),
                                               num_neighbours, ties, _ptr(out))
        return out[:cnt].tolist()

    @staticmethod
    def knn(graph, space, target, k, ef=None, ties=TIES_HEAP, counters=False):
        ef = k if ef is None else ef
        t = space.target(target)
        nodes = np.empty(max(k, 1), np.int64)
        dist = np.empty(max(k, 1), np.float64)
        c = _ctr()
        cnt = lib().og_ohnsw_knn(graph._h, space.ref(), _ptr(t), ef, k, ties, _ptr(nodes),
                                 _ptr(dist), C.byref(c))
        if cnt < 0:
            raise ValueError("knn: empty hgraph")  # Invalid_argument, lib/ohnsw.ml:862
        res = list(zip(nodes[:cnt].tolist(), dist[:cnt].tolist()))
        return (res, c) if counters else res

    @staticmethod
    def knn_batch_bigarray(graph, space, batch, k, ef=None, ties=TIES_HEAP, counters=False, split=False):
        """-> (ids [nq][k] int32, -1 filled; distances [nq][k] float32, NaN filled)
        counters: + per-query evaluations and layer-0 hops; split: + the evaluations spent before the
        layer-0 loop (descent + start node)"""
        ef = k if ef is None else ef
        Q = np.ascontiguousarray(batch, dtype=np.float32)
        nq = Q.shape[0]
        ids = np.empty((nq, k), np.int32)
        dist = np.empty((nq, k), np.float32)
        nd = np.zeros(nq, np.uint32)
        nh = np.zeros(nq, np.uint32)
        nu = np.zeros(nq, np.uint32)
        r = lib().og_ohnsw_knn_batch_split(graph._h, space.ref(), _ptr(Q), nq, Q.shape[1] if nq else 0,
                                           ef, k, ties, _ptr(ids), _ptr(dist), _ptr(nd), _ptr(nh), _ptr(nu))
        if r < 0:
            raise ValueError("knn: empty hgraph")
        if split:
            return ids, dist, nd, nh, nu
        return (ids, dist, nd, nh) if counters else (ids, dist)


def knn_batch_all_cores(graph, space, batch, k, ef, nthreads, ties=TIES_CANONICAL):
    """Ohnsw.knn_batch_bigarray with the query loop split over host threads (baseline only)."""
    Q = np.ascontiguousarray(batch, dtype=np.float32)
    nq = Q.shape[0]
    ids = np.empty((nq, k), np.int32)
    dist = np.empty((nq, k), np.float32)
    r = lib().og_ohnsw_knn_batch_mt(graph._h, space.ref(), _ptr(Q), nq, Q.shape[1], ef, k, ties, nthreads,
                                    _ptr(ids), _ptr(dist))
    if r < 0:
        raise ValueError("knn: empty hgraph")
    return ids, dist


class Functor:
    """lib/hnsw_algo.ml Search/Knn/SelectNeighbours instantiated as in lib/hnsw.ml (Hnsw.Ba)."""

    @staticmethod
    def search_one(graph, space, start_node, target, layer=0, ties=TIES_HEAP):
        t = space.target(target)
        d = C.c_double()
        n = lib().og_functor_search_one(graph._h, layer, space.ref(), start_node, _ptr(t), ties,
                                        C.byref(d), None)
        return int(n), d.value

    @staticmethod
    def search(graph, space, start_nodes, target, ef, layer=0, ties=TIES_HEAP, counters=False):
        t = space.target(target)
        st = np.ascontiguousarray(start_nodes, dtype=np.int64)
        nodes = np.empty(max(ef, 1), np.int64)
        dist = np.empty(max(ef, 1), np.float64)
        c = _ctr()
        cnt = lib().og_functor_search(graph._h, layer, space.ref(), _ptr(st), len(st), _ptr(t), ef,
                                      ties, _ptr(nodes), _ptr(dist), C.byref(c))
        res = list(zip(nodes[:cnt].tolist(), dist[:cnt].tolist()))
        return (res, c) if counters else res

    @staticmethod
    def knn(graph, space, target, num_neighbours_search, num_neighbours, ties=TIES_HEAP,
            bug_compat_farthest_k=False, counters=False):
        t = space.target(target)
        k = num_neighbours
        nodes = np.empty(max(k, 1), np.int64)
        dist = np.empty(max(k, 1), np.float64)
        c = _ctr()
        cnt = lib().og_functor_knn(graph._h, space.ref(), _ptr(t), num_neighbours_search, k, ties,
                                   int(bug_compat_farthest_k), _ptr(nodes), _ptr(dist), C.byref(c))
        if cnt < 0:
            raise ValueError("knn: empty hgraph")
        res = list(zip(nodes[:cnt].tolist(), dist[:cnt].tolist()))
        return (res, c) if counters else res

    @staticmethod
    def knn_batch(graph, space, batch, num_neighbours_search, num_neighbours, ties=TIES_HEAP,
                  bug_compat_farthest_k=False, with_ids=False):
        """Hnsw.Ba.knn_batch: distances [nq][k], +inf filled (lib/hnsw.ml:769-777)."""
        Q = np.ascontiguousarray(batch, dtype=np.float32)
        nq, k = Q.shape[0], num_neighbours
        dist = np.empty((nq, k), np.float32)
        ids = np.empty((nq, k), np.int32)
        r = lib().og_functor_knn_batch(graph._h, space.ref(), _ptr(Q), nq, Q.shape[1] if nq else 0,
                                       num_neighbours_search, k, ties, int(bug_compat_farthest_k),
                                       _ptr(dist), _ptr(ids))
        if r < 0:
            raise ValueError("knn: empty hgraph")
        return (dist, ids) if with_ids else dist

    @staticmethod
    def select_neighbours(space, candidates, cand_dist, num_neighbours, cand_degree=None,
                          do_not_isolate=False, ties=TIES_HEAP):
        cand = np.ascontiguousarray(candidates, dtype=np.int64)
        cd = np.ascontiguousarray(cand_dist, dtype=np.float64)
        deg = None if cand_degree is None else np.ascontiguousarray(cand_degree, dtype=np.int32)
        out = np.empty(max(len(cand), 1), np.int64)
        cnt = lib().og_functor_select_neighbours(space.ref(), _ptr(cand), _ptr(cd), _ptr(deg),
                                                 len(cand), num_neighbours, int(do_not_isolate),
                                                 ties, _ptr(out))
        return out[:cnt].tolist()


def build_ohnsw(space, num_connections, num_nodes_search_construction, seed=0, ties=TIES_HEAP,
                n=None):
    """Ohnsw.build_batch_bigarray restated (lib/ohnsw.ml:840-857) -> flattened Graph.
    Test-graph generator only (own RNG; graph identity with an OCaml build is not a goal)."""
    L = lib()
    n = space.n if n is None else n
    M = num_connections
    b = L.og_build_ohnsw(space.ref(), n, M, num_nodes_search_construction, seed, ties)
    try:
        assert L.og_builder_invariant(b) == 1, "symmetric-link invariant violated"
        max_layer = L.og_builder_max_layer(b)
        deg0 = np.empty(n, np.int32)
        nbr0 = np.empty((n, 2 * M), np.int32)
        md = L.og_builder_export_layer0(b, 2 * M, _ptr(deg0), _ptr(nbr0))
        if md < 0:
            raise RuntimeError("layer-0 degree exceeds 2M: flatten refuses to truncate")
        upper = []
        for l in range(1, max_layer + 1):
            cnt = L.og_builder_layer_count(b, l)
            nodes = np.empty(cnt, np.int64)
            deg = np.empty(cnt, np.int32)
            nbr = np.empty((cnt, M), np.int32)
            got = L.og_builder_export_upper(b, l, M, _ptr(nodes), _ptr(deg), _ptr(nbr))
            if got < 0:
                raise RuntimeError("upper-layer degree exceeds M")
            assert got == cnt
            upper.append((nodes, deg, nbr))
        ep = L.og_builder_entry_point(b)
    finally:
        L.og_builder_destroy(b)
    return Graph(n, ep, deg0, nbr0, upper)


def brute_force_knn(space, queries, k):
    Q = np.ascontiguousarray(queries, dtype=np.float32)
    nq = Q.shape[0]
    ids = np.empty((nq, k), np.int32)
    dist = np.empty((nq, k), np.float32)
    lib().og_brute_force_knn(space.ref(), _ptr(Q), nq, Q.shape[1], k, _ptr(ids), _ptr(dist))
    return ids, dist


def recall_distance_threshold(expected, got, epsilon=1e-8):
    e = np.ascontiguousarray(expected, dtype=np.float32)
    g = np.ascontiguousarray(got, dtype=np.float32)
    return float(lib().og_recall_distance_threshold(_ptr(e), _ptr(g), e.shape[0], e.shape[1], epsilon))


def l2sq_tree16(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return float(lib().og_l2sq_tree16(_ptr(a), _ptr(b), a.shape[0]))


def dot_tree16(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return float(lib().og_dot_tree16(_ptr(a), _ptr(b), a.shape[0]))


class Visited:
    """lib/ohnsw.ml:256-268"""

    def __init__(self, n):
        self._h = lib().og_visited_create(n)

    def __del__(self):
        try:
            lib().og_visited_destroy(self._h)
        except Exception:
            pass

    def mem(self, node):
        r = lib().og_visited_mem(self._h, node)
        if r < 0:
            raise IndexError("index out of bounds")
        return bool(r)

    def add(self, node):
        lib().og_visited_add(self._h, node)

    def clear(self):
        lib().og_visited_clear(self._h)

    def card(self):
        return int(lib().og_visited_card(self._h))

    def set_epoch(self, e):
        lib().og_visited_set_epoch(self._h, e)
