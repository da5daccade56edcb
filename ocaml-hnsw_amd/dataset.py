"""Dataset loaders and the reference's recall definition (benchmark/dataset.ml), host side.

    Dataset.random   benchmark/dataset.ml:47-58   (Lacaml Mat.random range U[-1, 1))
    Dataset.read     benchmark/dataset.ml:76-102  (ann-benchmarks HDF5: train / test / distances + attr "distance")
    Recall.compute   benchmark/dataset.ml:105-127 (distance-threshold recall, epsilon 1e-8)
plus the TEXMEX .fvecs / .ivecs files of the Makefile's download-data target (Makefile:27-28).
Ground truth comes from an exact scan (brute_force_knn_l2, benchmark/dataset.ml:15-30).
"""

import numpy as np


def read_fvecs(path, limit=None):
    """TEXMEX .fvecs: per vector an int32 d followed by d float32."""
    a = np.fromfile(path, dtype=np.int32)
    if a.size == 0:
        return np.zeros((0, 0), np.float32)
    d = int(a[0])
    a = a.reshape(-1, d + 1)
    if not (a[:, 0] == d).all():
        raise ValueError("%s: inconsistent vector lengths" % path)
    out = a[:, 1:].view(np.float32)
    return np.ascontiguousarray(out[:limit] if limit else out)


def read_ivecs(path, limit=None):
    a = np.fromfile(path, dtype=np.int32)
    if a.size == 0:
        return np.zeros((0, 0), np.int32)
    d = int(a[0])
    a = a.reshape(-1, d + 1)
    return np.ascontiguousarray(a[:limit, 1:] if limit else a[:, 1:])


def write_fvecs(path, X):
    X = np.ascontiguousarray(X, np.float32)
    n, d = X.shape
    out = np.empty((n, d + 1), np.int32)
    out[:, 0] = d
    out[:, 1:] = X.view(np.int32)
    out.tofile(path)


class Dataset:
    """{train; test; test_distances; distance} of benchmark/dataset.ml:32-45, row-major [n][d]."""

    def __init__(self, train, test, test_distances, distance="euclidean"):
        self.train = np.ascontiguousarray(train, np.float32)
        self.test = np.ascontiguousarray(test, np.float32)
        self.test_distances = np.ascontiguousarray(test_distances, np.float32)
        self.distance = distance

    @classmethod
    def random(cls, dim, num_train, num_test, k, seed=0):
        rng = np.random.default_rng(seed)
        train = rng.uniform(-1, 1, size=(num_train, dim)).astype(np.float32)
        test = rng.uniform(-1, 1, size=(num_test, dim)).astype(np.float32)
        return cls(train, test, brute_force_knn_l2(train, test, k))

    @classmethod
    def read(cls, path, limit_train=None, limit_test=None):
        """ann-benchmarks HDF5 (benchmark/dataset.ml:76-102: datasets `train`, `test`, `distances`,
        attribute `distance`; ?limit_train / ?limit_test keep the first columns, :88-93).  Uses h5py
        when importable, else libhdf5 through ctypes (h5lite)."""
        try:
            import h5py
        except ImportError:
            h5py = None
        if h5py is not None:
            with h5py.File(path, "r") as f:
                distance = f.attrs.get("distance", "euclidean")
                if isinstance(distance, bytes):
                    distance = distance.decode()
                train = np.asarray(f["train"][:limit_train], np.float32)
                test = np.asarray(f["test"][:limit_test], np.float32)
                dist = np.asarray(f["distances"][:limit_test], np.float32)
            return cls(train, test, dist, distance)
        try:
            from . import h5lite
        except ImportError:       # imported as a plain module (tools/)
            import h5lite
        with h5lite.File(path) as f:
            distance = f.attr("distance", "euclidean")
            train = f.read("train", np.float32, limit_train)
            test = f.read("test", np.float32, limit_test)
            dist = f.read("distances", np.float32, limit_test)
        return cls(train, test, dist, distance)

    def write(self, path, neighbors=None):
        """The same layout back out (e.g. a synthetic set for another tool)."""
        try:
            from . import h5lite
        except ImportError:
            import h5lite
        with h5lite.File(path, "w") as f:
            f.write("train", self.train)
            f.write("test", self.test)
            f.write("distances", self.test_distances)
            if neighbors is not None:
                f.write("neighbors", np.ascontiguousarray(neighbors, np.int32))
            f.set_attr("distance", self.distance)

    @classmethod
    def read_texmex(cls, base, query, groundtruth=None, k=10, limit_train=None, limit_test=None):
        train = read_fvecs(base, limit_train)
        test = read_fvecs(query, limit_test)
        if groundtruth is not None and limit_train is None:
            gt = read_ivecs(groundtruth, limit_test)[:, :k]
            d = np.sqrt(((train[gt] - test[:, None, :]) ** 2).sum(-1)).astype(np.float32)
        else:
            d = brute_force_knn_l2(train, test, k)
        return cls(train, test, d)


def brute_force_knn_l2(train, test, k, block=256):
    """benchmark/dataset.ml:15-30: for each test vector the k smallest L2 distances, ascending."""
    train = np.ascontiguousarray(train, np.float32)
    test = np.ascontiguousarray(test, np.float32)
    tn = (train.astype(np.float64) ** 2).sum(1)
    out = np.empty((test.shape[0], k), np.float32)
    for s in range(0, test.shape[0], block):
        q = test[s:s + block].astype(np.float64)
        d2 = tn[None, :] - 2.0 * (q @ train.T.astype(np.float64)) + (q ** 2).sum(1)[:, None]
        part = np.partition(d2, min(k, d2.shape[1]) - 1, axis=1)[:, :k]
        out[s:s + block] = np.sqrt(np.maximum(np.sort(part, axis=1), 0)).astype(np.float32)
    return out


class Recall:
    @staticmethod
    def compute(expected, got, epsilon=1e-8):
        """benchmark/dataset.ml:107-126: mean over queries of the fraction of returned distances
        <= the true k-th distance + epsilon.  expected / got: [nq][k]."""
        e = np.asarray(expected, np.float64)
        g = np.asarray(got, np.float64)
        if e.shape != g.shape:
            raise ValueError("Recall.compute: arrrays have unequal shapes")   # sic, dataset.ml:113
        return float(((g <= e[:, -1:] + epsilon).sum(1) / e.shape[1]).mean())

    @staticmethod
    def ids(expected_ids, got_ids):
        """id-set recall@k (the >= 0.95 gate of BASELINE.json uses this one)."""
        return float(np.mean([len(set(a) & set(b)) / len(a) for a, b in zip(np.asarray(expected_ids).tolist(), np.asarray(got_ids).tolist())]))
