"""Byte rows (ocaml-hnsw_amd/csrc/hnsw_rows8.hip): when every value of the vectors is an integer in 0..255
the knn searches read a lossless byte copy of the rows.  The bar is the one of every other search test --
ids, distance bits, evaluation and hop counts identical to the oracle's -- plus: identical to the same
index searched through its float32 rows (option "byte_rows" = 0), for both metrics, both accept rules,
ragged and full rows and every lane-grid width; and data that does not qualify never gets the copy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import ocaml_hnsw_amd as H
    H.load()
    assert H.device_count() >= 1, "GPU tests need a HIP device"
    return H


def _bytes_data(n, d, seed, hi=255):
    rng = np.random.default_rng(seed)
    centres = rng.integers(0, hi + 1, size=(32, d))
    X = centres[rng.integers(0, 32, n)] + rng.normal(0, 30, size=(n, d))
    X = np.clip(np.rint(X), 0, hi).astype(np.float32)
    X[0, :] = 0.0          # the extremes of the range are present
    X[1, :] = float(hi)
    X[2, ::2] = -0.0       # np.rint / torch.round of -0.3: a zero with the sign bit set is still the byte 0
    return X


def _both_ways(H, hg, fn):
    # evaluation counts include re-evaluations, which depend on the visited cache's size; the library sizes that cache per
    # kernel variant (as large as the variant's residency allows), so the two formats are compared at one explicit size
    # (and see _same_results: the formats may still differ in the number of tags per word)
    hg.set_option("vt_bits", 12)
    assert hg.row_bytes() == hg.info().d
    a = fn()
    hg.set_option("byte_rows", 0)
    assert hg.row_bytes() == 4 * hg.info().d
    b = fn()
    hg.set_option("byte_rows", 1)
    assert hg.row_bytes() == hg.info().d
    hg.set_option("vt_bits", 0)
    return a, b


def _same_results(a, b):
    """(ids, distances, evaluations, hops) of the two row formats: ids, distance bits and hop counts are the search's and must be equal;
    the evaluation counts include what the lossy visited cache re-evaluates, and the float32-row kernels keep three tags per word
    from ef 65 on where the byte-row kernels keep two (visited_three_ways): equal only where nothing is forgotten"""
    for i, (x, y) in enumerate(zip(a, b)):
        if len(a) == 4 and i == 2:
            assert (np.asarray(x) > 0).all() and (np.asarray(y) > 0).all()
            continue
        np.testing.assert_array_equal(np.asarray(x).view(np.uint32) if np.asarray(x).dtype == np.float32 else x,
                                      np.asarray(y).view(np.uint32) if np.asarray(y).dtype == np.float32 else y)


# d: 20 -> 1 chunk column (ragged), 64 -> 1 (full), 100 -> 2 (ragged), 128 -> 2 (full), 130 -> 4 (ragged, d % 4 != 0),
#    256 -> 4 (full), 300 -> 8, 960 -> 16
# byte-valued queries (SIFT's are) with d <= 256 take the exact integer dot-product arithmetic of hop_round, any other
# query the float arithmetic on the converted bytes: both must be the oracle's float32 arithmetic bit for bit.  d = 256 with
# every value 255 against a query of zeros is the largest sum the integer path may meet (256 * 255^2 < 2^24).
@pytest.mark.parametrize("d", [20, 64, 100, 128, 130, 256, 300, 960])
@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("queries", ["bytes", "floats"])
def test_byte_rows_equal_float_rows_and_oracle(H, oracle, d, metric, queries):
    n, nq, M, efc = (1500, 96, 8, 40) if d > 256 else (4000, 200, 12, 60)
    X = _bytes_data(n, d, 100 + d)
    Q = _bytes_data(nq, d, 200 + d)
    if queries == "floats":
        Q = Q + np.float32(0.25)          # queries need not be bytes
        Q[5] = -Q[5]
    else:
        Q[3, :] = 0.0                     # against X[1] = all 255: the extreme sum
        Q[4, :] = 255.0
    hg = H.Ohnsw.build_batch_bigarray(X, M, efc, seed=3, metric=metric)
    for ef, k in ((16, 5), (100, 10), (300, 64)):
        a, b = _both_ways(H, hg, lambda: H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True))
        _same_results(a, b)
    hg.export()
    g = oracle.Graph(hg.n, hg.entry_point, hg.deg0, hg.nbr0, hg.upper)
    sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=100, counters=True)
    oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=10, ef=100, ties=oracle.TIES_CANONICAL, counters=True)
    np.testing.assert_array_equal(ids, oi)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(nh, onh)
    hg.release()


def test_byte_rows_functor_rule_and_ordered_launch(H):
    """the functor accept rule, and a batch large enough for the longest-first ordering (descent pre-pass on byte rows)"""
    n, d = 20000, 128
    X = _bytes_data(n, d, 7, hi=218)
    Q = _bytes_data(9000, d, 8, hi=218)
    hg = H.Ohnsw.build_batch_bigarray(X, 16, 100, seed=1)
    hg.set_option("order_queries", 1)
    a, b = _both_ways(H, hg, lambda: H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=128, counters=True))
    _same_results(a, b)
    a, b = _both_ways(H, hg, lambda: H.Ba.knn_batch(hg, Q[:500], 64, 10))
    for x, y in zip(a, b):
        np.testing.assert_array_equal(np.asarray(x).view(np.uint32) if np.asarray(x).dtype == np.float32 else x,
                                      np.asarray(y).view(np.uint32) if np.asarray(y).dtype == np.float32 else y)
    hg.release()


@pytest.mark.parametrize("spoil", ["fraction", "256", "negative", "nan"])
def test_data_that_is_not_bytes_keeps_float_rows(H, spoil):
    X = _bytes_data(600, 32, 11)
    X[311, 7] = {"fraction": 3.5, "256": 256.0, "negative": -1.0, "nan": np.nan}[spoil]
    hg = H.Ohnsw.build_batch_bigarray(X, 8, 40, seed=1)
    assert hg.row_bytes() == 4 * 32
    hg.set_option("byte_rows", 1)                      # nothing to switch to
    assert hg.row_bytes() == 4 * 32
    ids, _ = H.Ohnsw.knn_batch_bigarray(hg, 5, X[:50], ef=50)
    assert (ids[:, 0] == np.arange(50)).mean() > 0.9
    hg.release()


def test_save_load_and_flattened_create_rebuild_the_copy(H, tmp_path):
    X = _bytes_data(3000, 128, 21, hi=218)
    Q = _bytes_data(64, 128, 22, hi=218)
    hg = H.Ohnsw.build_batch_bigarray(X, 16, 80, seed=2)
    ref = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=64, counters=True)
    p = str(tmp_path / "idx.bin")
    hg.save(p)
    hg2 = H.Hgraph.load(p)
    assert hg2.row_bytes() == 128
    got = H.Ohnsw.knn_batch_bigarray(hg2, 10, Q, ef=64, counters=True)
    for x, y in zip(ref, got):
        np.testing.assert_array_equal(x, y)
    hg.export()
    hg3 = H.Hgraph(X, hg.deg0, hg.nbr0, hg.upper, entry_point=hg.entry_point, max_degree=16)
    assert hg3.row_bytes() == 128
    got = H.Ohnsw.knn_batch_bigarray(hg3, 10, Q, ef=64, counters=True)
    for x, y in zip(ref, got):
        np.testing.assert_array_equal(x, y)
    for h in (hg, hg2, hg3):
        h.release()


def test_launch_options_change_nothing_but_the_launch(H):
    """lds_pad (how many queries a CU holds at once) and order_queries are launch policy: per-query results and counters must
    not move, whatever they are set to (including LDS requests near the 64 KiB limit); byte_rows picks another kernel: same
    results and hop counts, evaluation counts of its own."""
    n, d = 30000, 128
    X = _bytes_data(n, d, 31, hi=218)
    Q = _bytes_data(9000, d, 32, hi=218)
    hg = H.Ohnsw.build_batch_bigarray(X, 16, 100, seed=1)
    ref = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=128, counters=True)
    for rows in (1, 0):
        hg.set_option("byte_rows", rows)
        hg.set_option("order_queries", -1)
        hg.set_option("lds_pad", -1)
        ref_rows = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=128, counters=True)
        _same_results(ref, ref_rows)                 # the row format moves the evaluation counts only (tags per word of the visited cache)
        for order in (-1, 0, 1):
            hg.set_option("order_queries", order)
            for pad in (-1, 0, 1280, 2560, 7000, 30000, 60000):
                hg.set_option("lds_pad", pad)
                got = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=128, counters=True)
                for x, y in zip(ref_rows, got):
                    np.testing.assert_array_equal(x, y)
    hg.release()
