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


def test_hdf5_reader_needs_h5py(tmp_path):
    import ocaml_hnsw_amd.dataset as D
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            D.Dataset.read(tmp_path / "missing.hdf5")
