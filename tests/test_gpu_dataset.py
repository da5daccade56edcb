"""benchmark/dataset.ml end to end on the GPU: a toy ann-benchmarks HDF5 file (train / test / distances + attribute
`distance`, dataset.ml:76-102) and a TEXMEX fvecs / ivecs triple (Makefile:27-28) go through the loaders of
ocaml_hnsw_amd.dataset, the device builder and Ohnsw.knn_batch_bigarray; ids and distance bits must equal the
oracle's search of the SAME graph, and Recall.compute (dataset.ml:105-127) of the GPU result must equal the
oracle's recall of its own result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import ocaml_hnsw_amd as H
    H.load()
    assert H.device_count() >= 1
    return H


def _toy(n, nq, d, seed):
    """SIFT-like toy vectors: clustered integers 0..218 (byte-valued: the byte-row path is the one that runs)."""
    rng = np.random.default_rng(seed)
    centres = rng.integers(20, 200, size=(24, d))
    X = np.clip(np.rint(centres[rng.integers(0, 24, n)] + rng.normal(0, 25, size=(n, d))), 0, 218).astype(np.float32)
    Q = np.clip(np.rint(centres[rng.integers(0, 24, nq)] + rng.normal(0, 25, size=(nq, d))), 0, 218).astype(np.float32)
    return X, Q


def _check_against_oracle(H, oracle, ds, M, efc, ef, k):
    import ocaml_hnsw_amd.dataset as D
    hg = H.Ohnsw.build_batch_bigarray(ds.train, M, efc, seed=4).export()
    ids, dist = H.Ohnsw.knn_batch_bigarray(hg, k, ds.test, ef=ef)
    g = oracle.Graph(hg.n, hg.entry_point, hg.deg0, hg.nbr0, [(a, b, c) for a, b, c in hg.upper])
    sp = oracle.Space.l2(ds.train, arith=oracle.TREE16)
    oids, odist = oracle.Ohnsw.knn_batch_bigarray(g, sp, ds.test, k=k, ef=ef, ties=oracle.TIES_CANONICAL)
    np.testing.assert_array_equal(ids, oids)
    np.testing.assert_array_equal(dist.view(np.uint32), odist.view(np.uint32))
    # the reference's recall (distance threshold, epsilon 1e-8) of both results against the file's ground truth
    r_gpu = D.Recall.compute(ds.test_distances[:, :k], dist)
    r_oracle = oracle.recall_distance_threshold(ds.test_distances[:, :k], odist)
    assert r_gpu == pytest.approx(r_oracle, abs=0)
    assert r_gpu > 0.9        # the toy set is easy: a broken loader (transposed rows, wrong limits) would not get here
    return r_gpu


def test_hdf5_file_to_gpu_search_equals_oracle(H, oracle, tmp_path):
    import ocaml_hnsw_amd.dataset as D
    import ocaml_hnsw_amd.h5lite as h5
    try:
        h5.lib()
    except h5.H5Error as e:
        pytest.skip(str(e))
    X, Q = _toy(6000, 150, 128, 1)
    k = 10
    truth = D.brute_force_knn_l2(X, Q, k)
    D.Dataset(X, Q, truth, "euclidean").write(tmp_path / "toy-128-euclidean.hdf5")
    ds = D.Dataset.read(tmp_path / "toy-128-euclidean.hdf5")
    assert ds.distance == "euclidean" and ds.train.shape == (6000, 128) and ds.test.shape == (150, 128)
    _check_against_oracle(H, oracle, ds, M=16, efc=100, ef=128, k=k)
    # limits keep the FIRST vectors (dataset.ml:88-93); the ground truth of a truncated train set is recomputed
    lim = D.Dataset.read(tmp_path / "toy-128-euclidean.hdf5", limit_train=2500, limit_test=40)
    assert lim.train.shape == (2500, 128) and lim.test.shape == (40, 128)
    lim.test_distances = D.brute_force_knn_l2(lim.train, lim.test, k)
    _check_against_oracle(H, oracle, lim, M=16, efc=100, ef=100, k=k)


def test_texmex_files_to_gpu_search_equals_oracle(H, oracle, tmp_path):
    import ocaml_hnsw_amd.dataset as D
    X, Q = _toy(5000, 120, 96, 2)
    k = 10
    D.write_fvecs(tmp_path / "toy_base.fvecs", X)
    D.write_fvecs(tmp_path / "toy_query.fvecs", Q)
    # ground-truth ids as .ivecs (100 per query in the real files; 20 here)
    d2 = ((Q[:, None, :].astype(np.float64) - X[None, :, :].astype(np.float64)) ** 2).sum(-1)
    gt = np.argsort(d2, axis=1, kind="stable")[:, :20].astype(np.int32)
    np.concatenate([np.full((gt.shape[0], 1), gt.shape[1], np.int32), gt], axis=1).tofile(tmp_path / "toy_groundtruth.ivecs")
    ds = D.Dataset.read_texmex(tmp_path / "toy_base.fvecs", tmp_path / "toy_query.fvecs", tmp_path / "toy_groundtruth.ivecs", k=k)
    np.testing.assert_array_equal(ds.train, X)
    np.testing.assert_allclose(ds.test_distances, D.brute_force_knn_l2(X, Q, k), rtol=1e-6)
    _check_against_oracle(H, oracle, ds, M=12, efc=80, ef=96, k=k)
