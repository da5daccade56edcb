"""ocaml-hnsw_amd -- MI355X-native HNSW search behind ocaml-hnsw's own API surface.

Host-side mirror (Python, over the C ABI of include/hnsw_mi355x.h) of the reference modules whose
search bodies move to the GPU:

    Ohnsw.knn / Ohnsw.knn_batch_bigarray / Ohnsw.distance_l2      lib/ohnsw.ml:859-899
    Ba.knn / Ba.knn_batch (= Hnsw.Ba, MakeBatch(EuclideanBa))     lib/hnsw.ml:763-777, 817-819

Same names, argument meaning and error behaviour (Invalid_argument -> InvalidArgument,
Failure -> Failure).  The OCaml graph builder stays OCaml: this side takes the flattened graph
(`Hgraph`) that the OCaml shim produces.  There is no CPU path: every compute call goes to
libhnsw_mi355x.so and fails loudly if the library or a device is missing.
"""
import ctypes as _C
import os as _os

import numpy as _np

_PKG_DIR = _os.path.dirname(_os.path.abspath(__file__)) if "__file__" in globals() and \
    _os.path.basename(_os.path.dirname(_os.path.abspath(__file__))) == "ocaml-hnsw_amd" else \
    _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "ocaml-hnsw_amd")
LIB_PATH = _os.environ.get("HNSW_LIB_PATH") or _os.path.join(_PKG_DIR, "libhnsw_mi355x.so")

OK, ERR_BAD_ARG, ERR_EMPTY_INDEX, ERR_DEGREE_OVERFLOW = 0, -1, -2, -3
ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_UNSUPPORTED = -4, -5, -6, -7
METRIC_L2, METRIC_IP = 0, 1
FILL_OHNSW, FILL_BA = 0, 1
SEM_OHNSW, SEM_FUNCTOR, SEM_FUNCTOR_NEAREST_K = 0, 1, 2

# every symbol include/hnsw_mi355x.h declares (tests check the .so exports all of them)
ABI_VERSION = 3          # HNSW_ABI_VERSION of include/hnsw_mi355x.h this mirror was written against

ABI_SYMBOLS = [
    "hnsw_abi_version", "hnsw_last_error", "hnsw_device_count", "hnsw_index_create",
    "hnsw_index_destroy", "hnsw_index_get_info", "hnsw_index_set_option", "hnsw_index_row_bytes", "hnsw_search_batch",
    "hnsw_search_batch_device", "hnsw_search_batch_h2d", "hnsw_knn", "hnsw_distance_batch", "hnsw_distance_batch_device",
    "hnsw_build", "hnsw_select_neighbours_batch", "hnsw_index_export_layer0", "hnsw_index_export_upper_count", "hnsw_index_export_upper",
    "hnsw_index_layer_stats", "hnsw_index_layer_isolated", "hnsw_index_locality_codes", "hnsw_index_visited_blocks", "hnsw_index_prepare", "hnsw_index_save", "hnsw_index_load",
    "hnsw_search_layer_batch", "hnsw_search_one_batch",
    "hnsw_search_submit", "hnsw_search_wait", "hnsw_index_kernel_times",
    "hnsw_multi_create", "hnsw_multi_destroy", "hnsw_multi_num_replicas", "hnsw_multi_replica", "hnsw_multi_search_batch",
    "hnsw_multi_search_batch_device", "hnsw_multi_copy_result", "hnsw_multi_debug_counters",
    "hnsw_host_register", "hnsw_host_unregister", "hnsw_host_alloc", "hnsw_host_free",
]


class InvalidArgument(ValueError):
    """OCaml Invalid_argument (e.g. "knn: empty hgraph", lib/ohnsw.ml:862)."""


class Failure(RuntimeError):
    """OCaml Failure: device / runtime errors."""


class _LayerDesc(_C.Structure):
    _fields_ = [("n_nodes", _C.c_int64), ("nodes", _C.c_void_p), ("deg", _C.c_void_p),
                ("nbr", _C.c_void_p)]


class _IndexDesc(_C.Structure):
    _fields_ = [("vectors", _C.c_void_p), ("n", _C.c_int64), ("d", _C.c_int32),
                ("row_stride", _C.c_int64), ("metric", _C.c_int32), ("id_base", _C.c_int32),
                ("max_degree0", _C.c_int32), ("max_degree", _C.c_int32), ("max_layer", _C.c_int32),
                ("entry_point", _C.c_int64), ("deg0", _C.c_void_p), ("nbr0", _C.c_void_p),
                ("upper", _C.c_void_p), ("expected_ef", _C.c_int32), ("expected_semantics", _C.c_int32)]


class _SearchParams(_C.Structure):
    _fields_ = [("ef", _C.c_int32), ("k", _C.c_int32), ("fill", _C.c_int32), ("semantics", _C.c_int32)]


class _BuildParams(_C.Structure):
    _fields_ = [("num_connections", _C.c_int32), ("num_nodes_search_construction", _C.c_int32),
                ("metric", _C.c_int32), ("id_base", _C.c_int32), ("seed", _C.c_uint64),
                ("max_batch", _C.c_int32), ("batch_div", _C.c_int32), ("expected_ef", _C.c_int32), ("expected_semantics", _C.c_int32)]


class LayerStats(_C.Structure):
    _fields_ = [("num_nodes", _C.c_int64), ("min_degree", _C.c_int32), ("max_degree", _C.c_int32),
                ("mean_degree", _C.c_double), ("num_isolated", _C.c_int64)]


class IndexInfo(_C.Structure):
    _fields_ = [("n", _C.c_int64), ("d", _C.c_int32), ("metric", _C.c_int32), ("id_base", _C.c_int32),
                ("max_degree0", _C.c_int32), ("max_degree", _C.c_int32), ("max_layer", _C.c_int32),
                ("entry_point", _C.c_int64), ("device_bytes", _C.c_int64),
                ("row_stride_bytes", _C.c_int64), ("device", _C.c_int32), ("row_format", _C.c_int32)]


_lib = None


def load():
    """dlopen the in-tree libhnsw_mi355x.so (built by build.py / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not _os.path.exists(LIB_PATH):
        raise Failure("%s is missing: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950); "
                      "there is no CPU fallback" % LIB_PATH)
    L = _C.CDLL(LIB_PATH)
    vp, i32, i64 = _C.c_void_p, _C.c_int32, _C.c_int64
    L.hnsw_abi_version.restype = i32
    if L.hnsw_abi_version() != ABI_VERSION:
        raise Failure("%s speaks ABI version %d, this binding %d: rebuild it (python __graft_entry__.py)" % (LIB_PATH, L.hnsw_abi_version(), ABI_VERSION))
    L.hnsw_last_error.restype = _C.c_char_p
    L.hnsw_device_count.argtypes = [vp]
    L.hnsw_index_create.argtypes = [vp, i32, vp]
    L.hnsw_index_destroy.argtypes = [vp]
    L.hnsw_index_get_info.argtypes = [vp, vp]
    L.hnsw_index_set_option.argtypes = [vp, _C.c_char_p, i64]
    L.hnsw_index_row_bytes.argtypes = [vp, vp]
    L.hnsw_index_row_bytes.restype = i32
    L.hnsw_search_batch.argtypes = [vp, vp, i64, i64, vp, vp, vp, vp, vp]
    L.hnsw_search_batch_device.argtypes = [vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    L.hnsw_search_batch_h2d.argtypes = [vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    L.hnsw_knn.argtypes = [vp, vp, vp, vp, vp, vp]
    L.hnsw_distance_batch.argtypes = [vp, vp, i64, i64, vp, i32, vp]
    L.hnsw_distance_batch_device.argtypes = [vp, vp, i64, i64, vp, i32, vp, vp]
    L.hnsw_build.argtypes = [vp, i64, i32, i64, vp, i32, vp]
    L.hnsw_select_neighbours_batch.argtypes = [vp, vp, i64, i64, vp, vp, i32, i32, i32, vp, vp, vp]
    L.hnsw_select_neighbours_batch.restype = i32
    L.hnsw_index_layer_stats.argtypes = [vp, i32, vp]
    L.hnsw_index_layer_isolated.argtypes = [vp, i32, vp, i64, vp]
    L.hnsw_index_locality_codes.argtypes = [vp, vp]
    L.hnsw_index_locality_codes.restype = i32
    L.hnsw_index_visited_blocks.argtypes = [vp, vp, vp]
    L.hnsw_index_visited_blocks.restype = i32
    L.hnsw_index_prepare.argtypes = [vp, vp]
    L.hnsw_index_prepare.restype = i32
    L.hnsw_index_save.argtypes = [vp, _C.c_char_p]
    L.hnsw_index_load.argtypes = [_C.c_char_p, i32, vp]
    for f in ("hnsw_index_layer_stats", "hnsw_index_layer_isolated", "hnsw_index_save", "hnsw_index_load"):
        getattr(L, f).restype = i32
    L.hnsw_index_export_layer0.argtypes = [vp, vp, vp]
    L.hnsw_index_export_upper_count.argtypes = [vp, i32, vp]
    L.hnsw_index_export_upper.argtypes = [vp, i32, vp, vp, vp]
    for f in ("hnsw_build", "hnsw_select_neighbours_batch", "hnsw_index_export_layer0", "hnsw_index_export_upper_count",
              "hnsw_index_export_upper"):
        getattr(L, f).restype = i32
    for f in ("hnsw_device_count", "hnsw_index_create", "hnsw_index_destroy", "hnsw_index_get_info",
              "hnsw_index_set_option", "hnsw_search_batch", "hnsw_search_batch_device", "hnsw_search_batch_h2d", "hnsw_knn",
              "hnsw_distance_batch", "hnsw_distance_batch_device"):
        getattr(L, f).restype = i32
    L.hnsw_search_layer_batch.argtypes = [vp, i32, vp, i64, i64, vp, i32, vp, vp, vp, vp, vp, vp]
    L.hnsw_search_one_batch.argtypes = [vp, i32, vp, i64, i64, vp, vp, vp]
    L.hnsw_index_kernel_times.argtypes = [vp, vp, vp, vp]
    L.hnsw_index_kernel_times.restype = i32
    L.hnsw_search_submit.argtypes = [vp, vp, i64, i64, vp, vp]
    L.hnsw_search_wait.argtypes = [vp, vp, vp, vp, vp]
    L.hnsw_search_submit.restype = L.hnsw_search_wait.restype = i32
    L.hnsw_multi_create.argtypes = [vp, vp, i32, vp]
    L.hnsw_multi_destroy.argtypes = [vp]
    L.hnsw_multi_num_replicas.argtypes = [vp, vp]
    L.hnsw_multi_replica.argtypes = [vp, i32, vp]
    L.hnsw_multi_search_batch.argtypes = [vp, vp, i64, i64, vp, vp, vp, vp, vp]
    L.hnsw_multi_search_batch_device.argtypes = [vp, vp, i64, i64, vp, vp, vp]
    L.hnsw_multi_copy_result.argtypes = [vp, i32, vp, vp]
    L.hnsw_multi_debug_counters.argtypes = [vp, vp]
    if hasattr(L, "hnsw_host_register"):      # (an older build of the library, loaded by tools/ab.py for comparison, lacks them)
        L.hnsw_host_register.argtypes = [vp, i64]
        L.hnsw_host_unregister.argtypes = [vp]
        L.hnsw_host_register.restype = L.hnsw_host_unregister.restype = i32
    if hasattr(L, "hnsw_host_alloc"):
        L.hnsw_host_alloc.argtypes = [vp, i64]
        L.hnsw_host_free.argtypes = [vp]
        L.hnsw_host_alloc.restype = L.hnsw_host_free.restype = i32
    for f in ("hnsw_search_layer_batch", "hnsw_search_one_batch", "hnsw_multi_create", "hnsw_multi_destroy",
              "hnsw_multi_num_replicas", "hnsw_multi_replica", "hnsw_multi_search_batch",
              "hnsw_multi_search_batch_device", "hnsw_multi_copy_result", "hnsw_multi_debug_counters"):
        getattr(L, f).restype = i32
    _lib = L
    return L


def _check(rc):
    if rc == OK:
        return
    msg = load().hnsw_last_error().decode()
    if rc in (ERR_BAD_ARG, ERR_EMPTY_INDEX, ERR_DEGREE_OVERFLOW):
        raise InvalidArgument(msg)
    raise Failure("[%d] %s" % (rc, msg))


def device_count():
    c = _C.c_int32(0)
    rc = load().hnsw_device_count(_C.byref(c))
    return c.value if rc == OK else 0


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_C.c_void_p)


def _rows(a):
    """[n][d] fp32 whose rows are contiguous but may be spaced (a Lacaml sub-matrix keeps its parent's
    leading dimension): returned as (array, row stride in floats) without copying when possible."""
    a = _np.asarray(a)
    if (a.dtype == _np.float32 and a.ndim == 2 and a.shape[0] > 0 and a.strides[1] == 4
            and a.strides[0] % 4 == 0 and a.strides[0] >= 4 * a.shape[1]):
        return a, a.strides[0] // 4
    a = _np.ascontiguousarray(a, dtype=_np.float32)
    return a, (a.shape[1] if a.ndim == 2 else 0)


class Hgraph:
    """The flattened form of Ohnsw.Hgraph.t (lib/ohnsw.ml:307-312) / Hnsw.Ba.Hgraph.t
    (lib/hnsw.ml:342-348) that crosses the C ABI: vectors + per-layer neighbour rows in the
    reference's iteration order, ids `id_base`-based (0 for Ohnsw, 1 for Hnsw.Ba).

    vectors [n][d] fp32; deg0 [n]; nbr0 [n][max_degree0]; upper = list of (nodes, deg, nbr) for
    layers 1..max_layer; entry_point id_base-based or None (empty hgraph)."""

    def __init__(self, vectors, deg0, nbr0, upper=(), entry_point=None, id_base=0,
                 max_degree=None, metric=METRIC_L2, expected_ef=0, expected_sem=SEM_OHNSW):
        self.vectors, self.row_stride = _rows(vectors)
        if self.vectors.ndim != 2:
            raise InvalidArgument("vectors must be [n][d]")
        self.n, self.d = self.vectors.shape
        self.deg0 = _np.ascontiguousarray(deg0, dtype=_np.int32)
        self.nbr0 = _np.ascontiguousarray(nbr0, dtype=_np.int32)
        if self.nbr0.ndim != 2 or self.nbr0.shape[0] != self.n or self.deg0.shape != (self.n,):
            raise InvalidArgument("deg0 / nbr0 shapes do not match n")
        self.max_degree0 = int(self.nbr0.shape[1])
        self.upper = []
        for (nodes, deg, nbr) in upper:
            nodes = _np.ascontiguousarray(nodes, dtype=_np.int64)
            deg = _np.ascontiguousarray(deg, dtype=_np.int32)
            nbr = _np.ascontiguousarray(nbr, dtype=_np.int32).reshape(len(nodes), -1)
            self.upper.append((nodes, deg, nbr))
        self.max_layer = len(self.upper)
        self.max_degree = int(max_degree if max_degree is not None else
                              (self.upper[0][2].shape[1] if self.upper else max(1, self.max_degree0 // 2)))
        self.id_base = int(id_base)
        self.entry_point = None if entry_point is None or entry_point < id_base else int(entry_point)
        self.metric = int(metric)
        # hnsw_index_desc.expected_ef / expected_semantics: the upload then also prepares searches with these parameters
        self.expected_ef, self.expected_sem = int(expected_ef), int(expected_sem)
        self._index = None
        self._device = None

    @classmethod
    def _from_handle(cls, handle, device, vectors, id_base, metric):
        """Wrap an index that already lives on the device (hnsw_build); the host copy of the
        graph is fetched on demand by export()."""
        self = cls.__new__(cls)
        self.vectors = vectors
        self.n, self.d = vectors.shape
        self._index, self._device = handle, device
        self.id_base, self.metric = int(id_base), int(metric)
        inf = IndexInfo()
        _check(load().hnsw_index_get_info(handle, _C.byref(inf)))
        self.max_degree0, self.max_degree, self.max_layer = inf.max_degree0, inf.max_degree, inf.max_layer
        self.entry_point = int(inf.entry_point) if inf.entry_point >= id_base else None
        self.deg0 = self.nbr0 = None
        self.upper = None
        return self

    def export(self):
        """Fetch the flattened graph from the device: fills deg0, nbr0, upper (ids id_base-based)."""
        L = load()
        self.deg0 = _np.empty(self.n, _np.int32)
        self.nbr0 = _np.empty((self.n, self.max_degree0), _np.int32)
        _check(L.hnsw_index_export_layer0(self.handle, _ptr(self.deg0), _ptr(self.nbr0)))
        self.upper = []
        for l in range(1, self.max_layer + 1):
            c = _C.c_int64(0)
            _check(L.hnsw_index_export_upper_count(self.handle, l, _C.byref(c)))
            nodes = _np.empty(c.value, _np.int64)
            deg = _np.empty(c.value, _np.int32)
            nbr = _np.empty((c.value, self.max_degree), _np.int32)
            _check(L.hnsw_index_export_upper(self.handle, l, _ptr(nodes), _ptr(deg), _ptr(nbr)))
            self.upper.append((nodes, deg, nbr))
        return self

    def locality_codes(self):
        """hnsw_index_locality_codes: the permutation of 0 .. n-1 that keys the visited set's bitmap blocks (introspection)."""
        out = _np.empty(self.n, _np.int32)
        _check(load().hnsw_index_locality_codes(self.handle, _ptr(out)))
        return out

    def visited_blocks(self, ef, sem=0):
        """hnsw_index_visited_blocks: 0 = searches at this ef use the tag cache, else log2 of the bitmap-block slots"""
        p = _SearchParams(ef, 1, FILL_OHNSW, sem)
        out = _C.c_int32(0)
        _check(load().hnsw_index_visited_blocks(self.handle, _C.byref(p), _C.byref(out)))
        return out.value

    def prepare(self, ef, sem=SEM_OHNSW):
        """hnsw_index_prepare: everything the first search with this ef and accept rule would do once (the visited-structure
        decision, the kernel's residency, its code object), now"""
        p = _SearchParams(ef, 1, FILL_OHNSW, sem)
        _check(load().hnsw_index_prepare(self.handle, _C.byref(p)))
        return self

    def stats(self):
        """Hgraph.Stats.compute (lib/hnsw.ml:353-375): {num_nodes, layer_sizes, layer_connectivity}; a layer's
        connectivity is the reference's mima record {min, max, mean, isolated}, `isolated` the list of node ids
        (descending, as the reference's fold conses it)."""
        out = {"num_nodes": self.n, "layer_sizes": {}, "layer_connectivity": {}}
        for l in range(self.max_layer + 1):
            st = LayerStats()
            _check(load().hnsw_index_layer_stats(self.handle, l, _C.byref(st)))
            iso = _np.empty(max(int(st.num_isolated), 1), _np.int64)
            c = _C.c_int64(0)
            _check(load().hnsw_index_layer_isolated(self.handle, l, _ptr(iso), int(st.num_isolated), _C.byref(c)))
            out["layer_sizes"][l] = int(st.num_nodes)
            out["layer_connectivity"][l] = {"min": st.min_degree, "max": st.max_degree,
                                            "mean": st.mean_degree, "isolated": iso[:min(c.value, int(st.num_isolated))].tolist()}
        return out

    def save(self, path):
        """Write the flattened index (vectors + graph) to `path` (hnsw_index_save)."""
        _check(load().hnsw_index_save(self.handle, str(path).encode()))

    @classmethod
    def load(cls, path, device=0):
        """Read a flattened index file straight into HBM (hnsw_index_load)."""
        h = _C.c_void_p()
        _check(load().hnsw_index_load(str(path).encode(), device, _C.byref(h)))
        inf = IndexInfo()
        _check(load().hnsw_index_get_info(h, _C.byref(inf)))
        self = cls.__new__(cls)
        self.vectors = None
        self.n, self.d = int(inf.n), int(inf.d)
        self._index, self._device = h, device
        self.id_base, self.metric = int(inf.id_base), int(inf.metric)
        self.max_degree0, self.max_degree, self.max_layer = inf.max_degree0, inf.max_degree, inf.max_layer
        self.entry_point = int(inf.entry_point) if inf.entry_point >= inf.id_base else None
        self.deg0 = self.nbr0 = self.upper = None
        return self

    def _desc(self):
        """(hnsw_index_desc, keep-alive) of the host copy of the flattened graph."""
        if self.vectors is None or self.deg0 is None:
            raise InvalidArgument("no host copy of the graph: call export() (and keep the vectors) first")
        nl = self.max_layer
        layers = (_LayerDesc * max(nl, 1))()
        for i, (nodes, deg, nbr) in enumerate(self.upper):
            if nbr.shape[1] != self.max_degree and len(nodes):
                raise InvalidArgument("upper rows must be max_degree wide")
            layers[i].n_nodes = len(nodes)
            layers[i].nodes, layers[i].deg, layers[i].nbr = nodes.ctypes.data, deg.ctypes.data, nbr.ctypes.data
        d = _IndexDesc()
        d.vectors = self.vectors.ctypes.data
        d.n, d.d, d.row_stride = self.n, self.d, getattr(self, "row_stride", self.vectors.strides[0] // 4)
        d.metric, d.id_base = self.metric, self.id_base
        d.max_degree0, d.max_degree, d.max_layer = self.max_degree0, self.max_degree, nl
        d.entry_point = self.id_base - 1 if self.entry_point is None else self.entry_point
        d.deg0, d.nbr0 = self.deg0.ctypes.data, self.nbr0.ctypes.data
        d.upper = _C.cast(layers, _C.c_void_p)
        d.expected_ef, d.expected_semantics = getattr(self, "expected_ef", 0), getattr(self, "expected_sem", 0)
        return d, layers

    def to_device(self, device=0):
        """Upload to HBM (hnsw_index_create).  Idempotent per device."""
        if self._index is not None and self._device == device:
            return self
        self.release()
        d, _keep = self._desc()
        h = _C.c_void_p()
        _check(load().hnsw_index_create(_C.byref(d), device, _C.byref(h)))
        self._index, self._device = h, device
        return self

    @property
    def handle(self):
        if self._index is None:
            self.to_device(0)
        return self._index

    def info(self):
        inf = IndexInfo()
        _check(load().hnsw_index_get_info(self.handle, _C.byref(inf)))
        return inf

    def set_option(self, name, value):
        _check(load().hnsw_index_set_option(self.handle, name.encode(), int(value)))

    def row_bytes(self):
        """Bytes of one vector as the knn searches read it (d: byte rows, 4 d: float32 rows)."""
        v = _C.c_int64(0)
        _check(load().hnsw_index_row_bytes(self.handle, _C.byref(v)))
        return v.value

    def kernel_times(self):
        """(search kernel ms, ordering pre-pass ms, calls) averaged over the device-entry calls since the
        last call (needs set_option("time_kernels", 1)); waits for them."""
        s_, p_, n_ = _C.c_double(0), _C.c_double(0), _C.c_int32(0)
        _check(load().hnsw_index_kernel_times(self.handle, _C.byref(s_), _C.byref(p_), _C.byref(n_)))
        return s_.value, p_.value, n_.value

    def release(self):
        if self._index is not None:
            load().hnsw_index_destroy(self._index)
            self._index = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class _HostBlock:
    """page-locked memory from hnsw_host_alloc, exposed through the array interface; freed with its last numpy view"""

    def __init__(self, shape, dtype):
        dt = _np.dtype(dtype)
        self.shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.nbytes = max(1, int(_np.prod(self.shape, dtype=_np.int64)) * dt.itemsize)
        p = _C.c_void_p()
        _check(load().hnsw_host_alloc(_C.byref(p), self.nbytes))
        self.ptr = p.value
        self.__array_interface__ = {"data": (self.ptr, False), "shape": self.shape, "typestr": dt.str, "version": 3}

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                load().hnsw_host_free(_C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def host_empty(shape, dtype=_np.float32):
    """A numpy array in page-locked memory the library allocates (hnsw_host_alloc): the host-buffer entry points read such
    query matrices and write such result matrices directly from the device, without copies.  Freed with the array."""
    return _np.asarray(_HostBlock(shape, dtype))


def pin(array):
    """hnsw_host_register: page-lock a host array the caller keeps alive (its query or result matrix of a benchmark
    loop): the host-buffer entry points then access it directly from the device.  Undo with unpin() BEFORE the array is
    freed (host_empty() gives memory the library owns instead)."""
    a = _np.asarray(array)
    if not a.flags["C_CONTIGUOUS"]:
        raise InvalidArgument("pin: array must be C-contiguous")
    _check(load().hnsw_host_register(_ptr(a), a.nbytes))
    return array


def unpin(array):
    """hnsw_host_unregister"""
    _check(load().hnsw_host_unregister(_ptr(_np.asarray(array))))


def _search(hgraph, batch, ef, k, fill, counters=False, sem=0, out=None):
    Q, qs = _rows(batch)
    if Q.ndim != 2 or (Q.shape[0] and Q.shape[1] != hgraph.d):
        raise InvalidArgument("batch must be [nq][d]")
    nq = Q.shape[0]
    if out is not None:      # result matrices of the caller (e.g. pinned ones it reuses batch after batch)
        ids, dist = out
        if ids.shape != (nq, k) or dist.shape != (nq, k) or ids.dtype != _np.int32 or dist.dtype != _np.float32 \
                or not ids.flags["C_CONTIGUOUS"] or not dist.flags["C_CONTIGUOUS"]:
            raise InvalidArgument("out must be (int32 [nq][k], float32 [nq][k]), C-contiguous")
    else:
        ids = _np.empty((nq, k), _np.int32)
        dist = _np.empty((nq, k), _np.float32)
    nd = _np.zeros(nq, _np.uint32) if counters else None
    nh = _np.zeros(nq, _np.uint32) if counters else None
    p = _SearchParams(ef, k, fill, sem)
    _check(load().hnsw_search_batch(hgraph.handle, _ptr(Q), nq, max(qs, hgraph.d), _C.byref(p), _ptr(ids),
                                    _ptr(dist), _ptr(nd), _ptr(nh)))
    return (ids, dist, nd, nh) if counters else (ids, dist)


class Request:
    """A batch in flight (hnsw_search_submit): `wait()` returns what the synchronous call would have.
    Lets a caller overlap consecutive batches (the next one fills the drain of the previous one):
        r1 = submit(hg, b1, ef, k); r2 = submit(hg, b2, ef, k); ids1, d1 = r1.wait(); ..."""

    def __init__(self, hgraph, handle, nq, k, keep):
        self._hg, self._h, self.nq, self.k, self._keep = hgraph, handle, nq, k, keep

    def wait(self, counters=False, out=None):
        if self._h is None:
            raise InvalidArgument("request already waited for")
        if out is not None:
            ids, dist = out
            if (ids.shape != (self.nq, self.k) or dist.shape != (self.nq, self.k) or ids.dtype != _np.int32 or dist.dtype != _np.float32
                    or not ids.flags["C_CONTIGUOUS"] or not dist.flags["C_CONTIGUOUS"]):   # the library writes nq * k contiguous words
                raise InvalidArgument("out must be (int32 [nq][k], float32 [nq][k]), C-contiguous")
        else:
            ids = _np.empty((self.nq, self.k), _np.int32)
            dist = _np.empty((self.nq, self.k), _np.float32)
        nd = _np.zeros(self.nq, _np.uint32) if counters else None
        nh = _np.zeros(self.nq, _np.uint32) if counters else None
        h, self._h, self._keep = self._h, None, None
        _check(load().hnsw_search_wait(h, _ptr(ids), _ptr(dist), _ptr(nd), _ptr(nh)))
        return (ids, dist, nd, nh) if counters else (ids, dist)


def submit(hgraph, batch, ef, k, fill=FILL_OHNSW, sem=SEM_OHNSW):
    """hnsw_search_submit: copy the batch in, start the search, return at once."""
    Q, qs = _rows(batch)
    if Q.ndim != 2 or Q.shape[0] < 1 or Q.shape[1] != hgraph.d:
        raise InvalidArgument("batch must be [nq][d], nq >= 1")
    p = _SearchParams(ef, k, fill, sem)
    h = _C.c_void_p()
    _check(load().hnsw_search_submit(hgraph.handle, _ptr(Q), Q.shape[0], max(qs, hgraph.d), _C.byref(p), _C.byref(h)))
    return Request(hgraph, h, Q.shape[0], k, Q)


def search_batch_device(hgraph, d_queries, nq, q_stride, ef, k, d_ids, d_dist, d_ndist=0, d_nhops=0,
                        d_status=0, stream=0, fill=FILL_OHNSW, sem=SEM_OHNSW):
    """Asynchronous search on device pointers (ints), on HIP stream `stream` (int handle)."""
    p = _SearchParams(ef, k, fill, sem)
    _check(load().hnsw_search_batch_device(hgraph.handle, d_queries, nq, q_stride, _C.byref(p), d_ids,
                                           d_dist, d_ndist or None, d_nhops or None,
                                           d_status or None, stream or None))


def search_batch_h2d(hgraph, batch, ef, k, d_ids, d_dist, d_ndist=0, d_nhops=0, d_status=0, stream=0,
                     fill=FILL_OHNSW, sem=SEM_OHNSW):
    """hnsw_search_batch_h2d: queries from a HOST matrix (read by the device directly when it was registered with pin()),
    results left in device buffers (pointers as ints), asynchronous on HIP stream `stream`.  The matrix must stay alive
    until the stream has passed the call."""
    Q, qs = _rows(batch)
    if Q.ndim != 2 or Q.shape[0] < 1 or Q.shape[1] != hgraph.d:
        raise InvalidArgument("batch must be [nq][d], nq >= 1")
    p = _SearchParams(ef, k, fill, sem)
    _check(load().hnsw_search_batch_h2d(hgraph.handle, _ptr(Q), Q.shape[0], max(qs, hgraph.d), _C.byref(p), d_ids, d_dist,
                                        d_ndist or None, d_nhops or None, d_status or None, stream or None))
    return Q          # the array the device reads: keep it alive until the stream is synchronised


class Ohnsw:
    """lib/ohnsw.ml -- the imperative API, 0-based ids, ef == k unless `ef` is given."""

    @staticmethod
    def knn(hgraph, k, target, ef=None):
        """Ohnsw.knn hgraph visited ~k target (lib/ohnsw.ml:859-875) -> [(node, distance)] ascending
        (the reference returns a MinQueue popped in that order, :886-893)."""
        ids, dist = _search(hgraph, _np.asarray(target, _np.float32)[None, :], k if ef is None else ef, k, FILL_OHNSW)
        return [(int(i), float(d)) for i, d in zip(ids[0], dist[0]) if i >= hgraph.id_base]

    @staticmethod
    def knn_batch_bigarray(hgraph, k, batch, ef=None, counters=False, out=None):
        """Ohnsw.knn_batch_bigarray hgraph ~k batch (lib/ohnsw.ml:877-897) -> (ids, distances):
        ids [nq][k] (-1 where fewer than k were found), distances [nq][k] fp32 (NaN there).
        out = (ids, distances): write into the caller's matrices instead of fresh ones."""
        return _search(hgraph, batch, k if ef is None else ef, k, FILL_OHNSW, counters, out=out)

    @staticmethod
    def search_k(hgraph, layer, start_nodes, targets, k, ef=None, sem=SEM_OHNSW, counters=False):
        """Batched Ohnsw.search_k layer distance value visited start_nodes target k ... (lib/ohnsw.ml:
        543-588) on one layer: start_nodes = one id list per target (the start MinQueue; distances
        are recomputed), W bounded by ef (= k, the reference's only shape, unless given).
        -> list per target of [(node, distance)] nearest first (result_minq order)."""
        T, ts = _rows(_np.atleast_2d(_np.asarray(targets, _np.float32)))
        nq = T.shape[0]
        if len(start_nodes) != nq:
            raise InvalidArgument("one start list per target")
        ef = k if ef is None else ef
        ns = max([len(c) for c in start_nodes] + [1])
        st = _np.full((nq, ns), hgraph.id_base - 1, _np.int64)
        for i, c in enumerate(start_nodes):
            st[i, :len(c)] = c
        ids = _np.empty((nq, k), _np.int32)
        dist = _np.empty((nq, k), _np.float32)
        cnt = _np.empty(nq, _np.int32)
        nd = _np.zeros(nq, _np.uint32)
        nh = _np.zeros(nq, _np.uint32)
        p = _SearchParams(ef, k, FILL_OHNSW, sem)
        _check(load().hnsw_search_layer_batch(hgraph.handle, layer, _ptr(T), nq, max(ts, hgraph.d), _ptr(st), ns,
                                              _C.byref(p), _ptr(ids), _ptr(dist), _ptr(cnt), _ptr(nd), _ptr(nh)))
        res = [[(int(ids[i, j]), float(dist[i, j])) for j in range(cnt[i])] for i in range(nq)]
        return (res, nd, nh) if counters else res

    @staticmethod
    def search_one(hgraph, layer, start, targets, with_distance=False):
        """Batched Ohnsw.search_one layer distance value visited start_node target (lib/ohnsw.ml:492-512):
        -> node per target (and its distance)."""
        T, ts = _rows(_np.atleast_2d(_np.asarray(targets, _np.float32)))
        nq = T.shape[0]
        st = _np.ascontiguousarray(_np.broadcast_to(_np.asarray(start, _np.int64), (nq,)))
        node = _np.empty(nq, _np.int64)
        dist = _np.empty(nq, _np.float32)
        _check(load().hnsw_search_one_batch(hgraph.handle, layer, _ptr(T), nq, max(ts, hgraph.d), _ptr(st),
                                            _ptr(node), _ptr(dist)))
        return (node, dist) if with_distance else node

    @staticmethod
    def build_batch_bigarray(batch, num_connections, num_nodes_search_construction, seed=0,
                             metric=METRIC_L2, device=0, max_batch=0, batch_div=0, expected_ef=0, expected_sem=SEM_OHNSW):
        """Ohnsw.build_batch_bigarray distance batch ~num_connections ~num_nodes_search_construction
        (lib/ohnsw.ml:840-857), batched on the device (the OCaml builder itself stays OCaml; this
        is for hosts without one).  -> Hgraph resident in HBM.  expected_ef (optional): the finished index
        is also prepared for searches with that ef (hnsw_build_params.expected_ef)."""
        X = _np.ascontiguousarray(batch, dtype=_np.float32)
        if X.ndim != 2 or X.shape[0] < 1:
            raise InvalidArgument("batch must be [n][d], n >= 1")
        p = _BuildParams(num_connections, num_nodes_search_construction, metric, 0, seed, max_batch, batch_div, expected_ef, expected_sem)
        h = _C.c_void_p()
        _check(load().hnsw_build(_ptr(X), X.shape[0], X.shape[1], X.shape[1], _C.byref(p), device, _C.byref(h)))
        return Hgraph._from_handle(h, device, X, 0, metric)

    @staticmethod
    def select_neighbours(hgraph, targets, candidates, num_neighbours, keep_all_if_few=False, degrees=None):
        """Batched Ohnsw.select_neighbours distance value queue num_neighbours (lib/ohnsw.ml:647-663):
        candidates = list of id lists (one per target); returns the kept ids per target in selection
        order.  keep_all_if_few=True gives Hnsw_algo.SelectNeighbours' shortcut (hnsw_algo.ml:596-599)."""
        T = _np.ascontiguousarray(targets, dtype=_np.float32)
        nb = T.shape[0]
        stride = max([len(c) for c in candidates] + [1])
        cand = _np.zeros((nb, stride), _np.int32)
        cnt = _np.zeros(nb, _np.int32)
        for i, c in enumerate(candidates):
            cand[i, :len(c)] = c
            cnt[i] = len(c)
        deg = None
        if degrees is not None:   # ~do_not_isolate:true (lib/hnsw_algo.ml:591-592)
            deg = _np.full((nb, stride), 2, _np.int32)
            for i, dg in enumerate(degrees):
                deg[i, :len(dg)] = dg
        out = _np.empty((nb, num_neighbours), _np.int32)
        ocnt = _np.empty(nb, _np.int32)
        _check(load().hnsw_select_neighbours_batch(hgraph.handle, _ptr(T), nb, T.shape[1], _ptr(cand), _ptr(cnt),
                                                   stride, num_neighbours, int(keep_all_if_few), _ptr(deg),
                                                   _ptr(out), _ptr(ocnt)))
        return [out[i, :ocnt[i]].tolist() for i in range(nb)]

    @staticmethod
    def distance_l2(hgraph, queries, ids):
        """Batched Ohnsw.distance_l2 (lib/ohnsw.ml:899): out[q][j] = distance(queries[q], value ids[q][j])."""
        Q = _np.ascontiguousarray(queries, dtype=_np.float32)
        I = _np.ascontiguousarray(ids, dtype=_np.int32)
        out = _np.empty(I.shape, _np.float32)
        _check(load().hnsw_distance_batch(hgraph.handle, _ptr(Q), Q.shape[0], Q.shape[1], _ptr(I),
                                          I.shape[1], _ptr(out)))
        return out


class Ba:
    """Hnsw.Ba = MakeBatch(EuclideanBa) (lib/hnsw.ml:729-778, 817-819): 1-based ids, separate
    ~num_neighbours_search (ef) and ~num_neighbours (k)."""

    @staticmethod
    def knn(hgraph, point, num_neighbours_search, num_neighbours, nearest_k_compat=False):
        """-> [{node; distance_to_target}] nearest first (lib/hnsw.ml:763-767).  nearest_k_compat=True
        reproduces Nearest.nearest_k (lib/hnsw.ml:522-525): the k FARTHEST of W when ef > k."""
        ids, dist = _search(hgraph, _np.asarray(point, _np.float32)[None, :], num_neighbours_search,
                            num_neighbours, FILL_BA, sem=SEM_FUNCTOR_NEAREST_K if nearest_k_compat else SEM_FUNCTOR)
        return [(int(i), float(d)) for i, d in zip(ids[0], dist[0]) if i >= hgraph.id_base]

    @staticmethod
    def knn_batch(hgraph, batch, num_neighbours_search, num_neighbours, nearest_k_compat=False):
        """-> distances [nq][k] fp32, +inf where fewer than k were found (lib/hnsw.ml:769-777)."""
        return _search(hgraph, batch, num_neighbours_search, num_neighbours, FILL_BA,
                       sem=SEM_FUNCTOR_NEAREST_K if nearest_k_compat else SEM_FUNCTOR)[1]

    @staticmethod
    def search(hgraph, layer, start_nodes, targets, size_nearest):
        """Batched Hnsw_algo.Search.search hgraph layer visited ~start_nodes target ~size_nearest
        (lib/hnsw_algo.ml:350-391) -> Nearest.t per target as [(node, distance)] nearest first."""
        return Ohnsw.search_k(hgraph, layer, start_nodes, targets, size_nearest, sem=SEM_FUNCTOR)

    @staticmethod
    def search_one(hgraph, layer, start, targets):
        """Batched Hnsw_algo.Search.search_one (lib/hnsw_algo.ml:393-437) -> (node, distance) arrays
        (the value_distance it returns)."""
        return Ohnsw.search_one(hgraph, layer, start, targets, with_distance=True)


class MultiHgraph:
    """One host process, several GPUs: the flattened graph replicated on every listed
    device; knn_batch_bigarray / knn_batch split the batch into contiguous shards, one device each,
    and return the same arrays as the single-device calls."""

    def __init__(self, hgraph, devices):
        self.hgraph = hgraph
        self.devices = [int(x) for x in devices]
        d, _keep = hgraph._desc()
        dev = _np.asarray(self.devices, _np.int32)
        h = _C.c_void_p()
        _check(load().hnsw_multi_create(_C.byref(d), _ptr(dev), len(dev), _C.byref(h)))
        self._h = h

    def num_replicas(self):
        c = _C.c_int32(0)
        _check(load().hnsw_multi_num_replicas(self._h, _C.byref(c)))
        return c.value

    def _search(self, batch, ef, k, fill, sem, counters=False):
        Q, qs = _rows(batch)
        if Q.ndim != 2 or (Q.shape[0] and Q.shape[1] != self.hgraph.d):
            raise InvalidArgument("batch must be [nq][d]")
        nq = Q.shape[0]
        ids = _np.empty((nq, k), _np.int32)
        dist = _np.empty((nq, k), _np.float32)
        nd = _np.zeros(nq, _np.uint32) if counters else None
        nh = _np.zeros(nq, _np.uint32) if counters else None
        p = _SearchParams(ef, k, fill, sem)
        _check(load().hnsw_multi_search_batch(self._h, _ptr(Q), nq, max(qs, self.hgraph.d), _C.byref(p), _ptr(ids),
                                              _ptr(dist), _ptr(nd), _ptr(nh)))
        return (ids, dist, nd, nh) if counters else (ids, dist)

    def knn_batch_bigarray(self, k, batch, ef=None, counters=False):
        """Ohnsw.knn_batch_bigarray over all replicas (lib/ohnsw.ml:877-897)."""
        return self._search(batch, k if ef is None else ef, k, FILL_OHNSW, SEM_OHNSW, counters)

    def search_device(self, batch, ef, k, fill=FILL_OHNSW, sem=SEM_OHNSW):
        """hnsw_multi_search_batch_device: sharded search + RCCL all-gather, results left on the devices.
        -> (d_ids, d_dist): per-device pointers (ints) to each device's full [nq][k] table."""
        Q, qs = _rows(batch)
        if Q.ndim != 2 or Q.shape[0] < 1 or Q.shape[1] != self.hgraph.d:
            raise InvalidArgument("batch must be [nq][d], nq >= 1")
        G = len(self.devices)
        pi = (_C.c_void_p * G)()
        pd = (_C.c_void_p * G)()
        p = _SearchParams(ef, k, fill, sem)
        _check(load().hnsw_multi_search_batch_device(self._h, _ptr(Q), Q.shape[0], max(qs, self.hgraph.d), _C.byref(p), pi, pd))
        self._last = (Q.shape[0], k)
        return [int(x or 0) for x in pi], [int(x or 0) for x in pd]

    def debug_counters(self):
        """hnsw_multi_debug_counters -> {allgather, broadcast, peer_copies, repaired_shards}"""
        out = (_C.c_int64 * 4)()
        _check(load().hnsw_multi_debug_counters(self._h, out))
        return dict(zip(("allgather", "broadcast", "peer_copies", "repaired_shards"), [int(x) for x in out]))

    def copy_result(self, g):
        """device g's copy of the last search_device result -> (ids, dist) host arrays"""
        nq, k = self._last
        ids = _np.empty((nq, k), _np.int32)
        dist = _np.empty((nq, k), _np.float32)
        _check(load().hnsw_multi_copy_result(self._h, int(g), _ptr(ids), _ptr(dist)))
        return ids, dist

    def knn_batch(self, batch, num_neighbours_search, num_neighbours):
        """Hnsw.Ba.knn_batch over all replicas (lib/hnsw.ml:769-777)."""
        return self._search(batch, num_neighbours_search, num_neighbours, FILL_BA, SEM_FUNCTOR)[1]

    def release(self):
        if self._h is not None:
            load().hnsw_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
