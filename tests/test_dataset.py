"""Host-side harness pieces that mirror benchmark/dataset.ml: loaders, brute force, recall."""
import numpy as np
import pytest


def test_fvecs_ivecs_roundtrip(tmp_path):
    import ocaml_hnsw_amd.dataset as D
    X = np.random.default_rng(0).normal(size=(37, 12)).astype(np.float32)
    D.write_fvecs(tmp_path / "x.fvecs", X)
    np.testing.assert_array_equal(D.read_fvecs(tmp_path / "x.fvecs"), X)
    np.testing.assert_array_equal(D.read_fvecs(tmp_path / "x.fvecs", limit=5), X[:5])
    I = np.arange(20, dtype=np.int32).reshape(4, 5)
    raw = np.concatenate([np.full((4, 1), 5, np.int32), I], axis=1)
    raw.tofile(tmp_path / "g.ivecs")
    np.testing.assert_array_equal(D.read_ivecs(tmp_path / "g.ivecs"), I)


def test_recall_matches_reference_definition(oracle):
    import ocaml_hnsw_amd.dataset as D
    ds = D.Dataset.random(dim=8, num_train=400, num_test=30, k=7, seed=1)
    sp = oracle.Space.l2(ds.train, arith=oracle.F64)
    _, exact = oracle.brute_force_knn(sp, ds.test, 7)
    np.testing.assert_allclose(ds.test_distances, exact, rtol=1e-6)
    got = exact.copy()
    got[:, -2:] += 0.5          # two of seven too far
    assert D.Recall.compute(ds.test_distances, got) == pytest.approx(5 / 7)
    assert D.Recall.compute(ds.test_distances, got) == pytest.approx(oracle.recall_distance_threshold(ds.test_distances, got))
    with pytest.raises(ValueError, match="unequal shapes"):
        D.Recall.compute(exact, exact[:, :3])


def _h5():
    import ocaml_hnsw_amd.h5lite as h5
    try:
        h5.lib()
    except h5.H5Error as e:
        pytest.skip(str(e))
    return h5


def test_hdf5_ann_benchmarks_layout_roundtrip(tmp_path):
    """benchmark/dataset.ml:76-102: train / test / distances + attribute `distance`; the limits keep
    the first vectors (:88-93).  Read back through libhdf5 (ctypes) -- the image has no h5py."""
    _h5()
    import ocaml_hnsw_amd.dataset as D
    ds = D.Dataset.random(dim=6, num_train=50, num_test=9, k=4, seed=3)
    ds.distance = "angular"
    nb = np.arange(36, dtype=np.int32).reshape(9, 4)
    ds.write(tmp_path / "toy.hdf5", neighbors=nb)
    back = D.Dataset.read(tmp_path / "toy.hdf5")
    np.testing.assert_array_equal(back.train, ds.train)
    np.testing.assert_array_equal(back.test, ds.test)
    np.testing.assert_array_equal(back.test_distances, ds.test_distances)
    assert back.distance == "angular"
    lim = D.Dataset.read(tmp_path / "toy.hdf5", limit_train=7, limit_test=2)
    np.testing.assert_array_equal(lim.train, ds.train[:7])
    np.testing.assert_array_equal(lim.test, ds.test[:2])
    np.testing.assert_array_equal(lim.test_distances, ds.test_distances[:2])


def test_hdf5_low_level(tmp_path):
    h5 = _h5()
    p = tmp_path / "x.h5"
    with h5.File(p, "w") as f:
        f.write("d64", np.linspace(0, 1, 12).reshape(3, 4))          # float64 on disk, converted on read
        f.write("ids", np.arange(10, dtype=np.int64))
        f.write("empty", np.zeros((0, 5), np.float32))
        f.set_attr("distance", "euclidean", variable=False)           # fixed-length string flavour
    with h5.File(p) as f:
        assert "d64" in f and "nope" not in f
        assert f.shape("d64") == (3, 4)
        np.testing.assert_allclose(f.read("d64", np.float32), np.linspace(0, 1, 12).reshape(3, 4).astype(np.float32))
        np.testing.assert_array_equal(f.read("ids", np.int32, limit=4), np.arange(4))
        assert f.read("empty").shape == (0, 5)
        assert f.attr("distance") == "euclidean"
        assert f.attr("missing", "dflt") == "dflt"
        with pytest.raises(h5.H5Error):
            f.read("nope")
    with pytest.raises(FileNotFoundError):
        h5.File(tmp_path / "missing.hdf5")
    (tmp_path / "junk.hdf5").write_bytes(b"not an hdf5 file")
    with pytest.raises(h5.H5Error):
        h5.File(tmp_path / "junk.hdf5")