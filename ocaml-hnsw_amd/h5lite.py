"""Minimal HDF5 access over libhdf5's C API (ctypes) -- enough for the ann-benchmarks files that
benchmark/dataset.ml:76-102 reads (datasets `train`, `test`, `distances`, `neighbors`; string
attribute `distance`).  The image ships libhdf5 (conda) but no h5py; h5py is used when importable.

Host-side data loading only -- nothing here touches the search path.
"""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

_lib = None
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5S_SELECT_SET = 0
H5T_VARIABLE = C.c_size_t(-1).value


class H5Error(IOError):
    pass


def _find():
    cands = [os.environ.get("HNSW_LIBHDF5")]
    cands += [ctypes.util.find_library("hdf5")]
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*",
                "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        cands += sorted(glob.glob(pat), key=len)
    for c in cands:
        if not c:
            continue
        try:
            return C.CDLL(c)
        except OSError:
            continue
    raise H5Error("libhdf5 not found (set HNSW_LIBHDF5=/path/to/libhdf5.so)")


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = _find()
    hid, vp, i = C.c_int64, C.c_void_p, C.c_int
    sig = {
        "H5open": (i, []), "H5Fopen": (hid, [C.c_char_p, C.c_uint, hid]),
        "H5Fcreate": (hid, [C.c_char_p, C.c_uint, hid, hid]), "H5Fclose": (i, [hid]),
        "H5Dopen2": (hid, [hid, C.c_char_p, hid]), "H5Dclose": (i, [hid]),
        "H5Dget_space": (hid, [hid]), "H5Dget_type": (hid, [hid]),
        "H5Dcreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]),
        "H5Dread": (i, [hid, hid, hid, hid, hid, vp]), "H5Dwrite": (i, [hid, hid, hid, hid, hid, vp]),
        "H5Sget_simple_extent_ndims": (i, [hid]), "H5Sget_simple_extent_dims": (i, [hid, vp, vp]),
        "H5Sselect_hyperslab": (i, [hid, i, vp, vp, vp, vp]), "H5Screate_simple": (hid, [i, vp, vp]),
        "H5Screate": (hid, [i]), "H5Sclose": (i, [hid]),
        "H5Lexists": (i, [hid, C.c_char_p, hid]),
        "H5Aexists": (i, [hid, C.c_char_p]), "H5Aopen": (hid, [hid, C.c_char_p, hid]),
        "H5Aget_type": (hid, [hid]), "H5Aread": (i, [hid, hid, vp]), "H5Aclose": (i, [hid]),
        "H5Acreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid]), "H5Awrite": (i, [hid, hid, vp]),
        "H5Tis_variable_str": (i, [hid]), "H5Tget_size": (C.c_size_t, [hid]), "H5Tclose": (i, [hid]),
        "H5Tcopy": (hid, [hid]), "H5Tset_size": (i, [hid, C.c_size_t]), "H5Tget_class": (i, [hid]),
        "H5Eset_auto2": (i, [hid, vp, vp]), "H5free_memory": (i, [vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    if L.H5open() < 0:
        raise H5Error("H5open failed")
    L.H5Eset_auto2(0, None, None)   # errors are reported through return codes
    _lib = L
    return L


def _native(dtype):
    name = {np.dtype(np.float32): "H5T_NATIVE_FLOAT_g", np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g",
            np.dtype(np.int32): "H5T_NATIVE_INT_g", np.dtype(np.int64): "H5T_NATIVE_LLONG_g"}[np.dtype(dtype)]
    return C.c_int64.in_dll(lib(), name).value


def _ok(v, what):
    if v < 0:
        raise H5Error("HDF5: %s failed" % what)
    return v


class File:
    """with File(path) as f: f.read("train", np.float32, limit=1000); f.attr("distance")"""

    def __init__(self, path, mode="r"):
        L = lib()
        p = os.fsencode(path)
        if mode == "r":
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            self.fid = L.H5Fopen(p, H5F_ACC_RDONLY, 0)
        elif mode == "w":
            self.fid = L.H5Fcreate(p, H5F_ACC_TRUNC, 0, 0)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if self.fid < 0:
            raise H5Error("cannot open %s as HDF5" % path)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def close(self):
        if getattr(self, "fid", -1) >= 0:
            lib().H5Fclose(self.fid)
            self.fid = -1

    def __contains__(self, name):
        return lib().H5Lexists(self.fid, name.encode(), 0) > 0

    def shape(self, name):
        L = lib()
        ds = _ok(L.H5Dopen2(self.fid, name.encode(), 0), "H5Dopen2(%s)" % name)
        try:
            sp = _ok(L.H5Dget_space(ds), "H5Dget_space")
            nd = _ok(L.H5Sget_simple_extent_ndims(sp), "ndims")
            dims = (C.c_uint64 * max(nd, 1))()
            L.H5Sget_simple_extent_dims(sp, dims, None)
            L.H5Sclose(sp)
            return tuple(int(dims[i]) for i in range(nd))
        finally:
            L.H5Dclose(ds)

    def read(self, name, dtype=np.float32, limit=None):
        """Dataset `name` converted to `dtype`; the first `limit` rows only if given (a hyperslab:
        nothing beyond them is read, as Hdf5_caml's read_float_array2 + sub_right would)."""
        L = lib()
        ds = _ok(L.H5Dopen2(self.fid, name.encode(), 0), "H5Dopen2(%s)" % name)
        try:
            sp = _ok(L.H5Dget_space(ds), "H5Dget_space")
            nd = _ok(L.H5Sget_simple_extent_ndims(sp), "ndims")
            dims = (C.c_uint64 * max(nd, 1))()
            L.H5Sget_simple_extent_dims(sp, dims, None)
            shape = [int(dims[i]) for i in range(nd)]
            msp = 0
            if limit is not None and nd >= 1 and limit < shape[0]:
                shape[0] = max(int(limit), 0)
                start = (C.c_uint64 * nd)(*([0] * nd))
                count = (C.c_uint64 * nd)(*shape)
                _ok(L.H5Sselect_hyperslab(sp, H5S_SELECT_SET, start, None, count, None), "hyperslab")
                msp = _ok(L.H5Screate_simple(nd, count, None), "H5Screate_simple")
            out = np.empty(shape, dtype)
            if out.size:
                _ok(L.H5Dread(ds, _native(dtype), msp, sp if msp else 0, 0, out.ctypes.data), "H5Dread(%s)" % name)
            if msp:
                L.H5Sclose(msp)
            L.H5Sclose(sp)
            return out
        finally:
            L.H5Dclose(ds)

    def attr(self, name, default=None):
        """String attribute of the root group (ann-benchmarks: `distance`)."""
        L = lib()
        if L.H5Aexists(self.fid, name.encode()) <= 0:
            return default
        a = _ok(L.H5Aopen(self.fid, name.encode(), 0), "H5Aopen")
        try:
            t = _ok(L.H5Aget_type(a), "H5Aget_type")
            try:
                if L.H5Tis_variable_str(t) > 0:
                    p = C.c_char_p()
                    _ok(L.H5Aread(a, t, C.addressof(p)), "H5Aread")
                    s = p.value.decode() if p.value is not None else ""
                    L.H5free_memory(C.cast(p, C.c_void_p))
                    return s
                n = L.H5Tget_size(t)
                buf = C.create_string_buffer(n + 1)
                _ok(L.H5Aread(a, t, buf), "H5Aread")
                return buf.raw[:n].split(b"\0")[0].decode()
            finally:
                L.H5Tclose(t)
        finally:
            L.H5Aclose(a)

    # ---- writing (tests, exporting a synthetic dataset in the ann-benchmarks layout) ----
    def write(self, name, array):
        L = lib()
        a = np.ascontiguousarray(array)
        dims = (C.c_uint64 * max(a.ndim, 1))(*a.shape)
        sp = _ok(L.H5Screate_simple(a.ndim, dims, None), "H5Screate_simple")
        ds = _ok(L.H5Dcreate2(self.fid, name.encode(), _native(a.dtype), sp, 0, 0, 0), "H5Dcreate2(%s)" % name)
        try:
            if a.size:
                _ok(L.H5Dwrite(ds, _native(a.dtype), 0, 0, 0, a.ctypes.data), "H5Dwrite")
        finally:
            L.H5Dclose(ds)
            L.H5Sclose(sp)

    def set_attr(self, name, value, variable=True):
        L = lib()
        c_s1 = C.c_int64.in_dll(L, "H5T_C_S1_g").value
        t = _ok(L.H5Tcopy(c_s1), "H5Tcopy")
        raw = value.encode()
        _ok(L.H5Tset_size(t, H5T_VARIABLE if variable else len(raw) + 1), "H5Tset_size")
        sp = _ok(L.H5Screate(0), "H5Screate")   # H5S_SCALAR
        a = _ok(L.H5Acreate2(self.fid, name.encode(), t, sp, 0, 0), "H5Acreate2")
        try:
            if variable:
                p = C.c_char_p(raw)
                _ok(L.H5Awrite(a, t, C.addressof(p)), "H5Awrite")
            else:
                _ok(L.H5Awrite(a, t, C.create_string_buffer(raw, len(raw) + 1)), "H5Awrite")
        finally:
            L.H5Aclose(a)
            L.H5Sclose(sp)
            L.H5Tclose(t)
