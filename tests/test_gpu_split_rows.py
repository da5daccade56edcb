"""Split rows (ocaml-hnsw_amd/csrc/hnsw_rows_split.hip): when a float32 row ends 1..32 bytes past a 128-byte line
(d = 100: 400 bytes) the index keeps the whole lines of every row in one table and the remaining 16 / 32 bytes of node
nbr0[c][j] beside slot (c, j) of the layer-0 adjacency; the knn searches read those on layer 0.  Same lanes, same
operands, same order of arithmetic: the bar is ids, distance bits, evaluation and hop counts identical to the oracle
and to the same index searched through its plain rows (option "split_rows" = 0), for both metrics, both accept rules,
tails of one and two chunks, every lane-grid width that can have one; shapes that do not qualify never get the copy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROWS_F32, ROWS_BYTES, ROWS_SPLIT = 0, 2, 3


@pytest.fixture(scope="module")
def H():
    import ocaml_hnsw_amd as H
    H.load()
    assert H.device_count() >= 1, "GPU tests need a HIP device"
    return H


def _unit(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, d)).astype(np.float32)
    return X / np.linalg.norm(X, axis=1, keepdims=True).astype(np.float32)


def _same(a, b):
    for x, y in zip(a, b):
        x, y = np.asarray(x), np.asarray(y)
        np.testing.assert_array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                      y.view(np.uint32) if y.dtype == np.float32 else y)


def _both_ways(hg, fn):
    # evaluation counts include re-evaluations, which depend on the visited cache's size; the library sizes that cache per
    # kernel variant (as large as the variant's residency allows), so the two formats are compared at one explicit size
    hg.set_option("vt_bits", 12)
    assert hg.info().row_format == ROWS_SPLIT
    a = fn()
    hg.set_option("split_rows", 0)
    assert hg.info().row_format == ROWS_F32
    b = fn()
    hg.set_option("split_rows", 1)
    assert hg.info().row_format == ROWS_SPLIT
    hg.set_option("vt_bits", 0)
    return a, b


# d -> float4 chunks: 36 -> 9 = 8 + 1 (one lane-grid column... two: NCH = 1 holds 16), 40 -> 10 = 8 + 2, 100 -> 25 = 24 + 1 (C3),
# 97 -> 25 with a partly filled tail chunk, 104 -> 26 = 24 + 2, 132 -> 33 = 32 + 1 (NCH = 4: a whole dead lane-grid column behind
# the tail), 232 -> 58 = 56 + 2, 292 -> 73 = 72 + 1 (NCH = 8), 548 -> 137 = 136 + 1 (NCH = 16)
@pytest.mark.parametrize("d", [36, 40, 97, 100, 104, 132, 232, 292, 548])
@pytest.mark.parametrize("metric", [0, 1])
def test_split_rows_equal_plain_rows_and_oracle(H, oracle, d, metric):
    n, nq, M, efc = (1500, 96, 8, 40) if d > 256 else (4000, 200, 12, 60)
    X, Q = _unit(n, d, 100 + d), _unit(nq, d, 200 + d)
    hg = H.Ohnsw.build_batch_bigarray(X, M, efc, seed=3, metric=metric)
    for ef, k in ((16, 5), (100, 10), (300, 64), (600, 100)):
        _same(*_both_ways(hg, lambda: H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)))
    _same(*_both_ways(hg, lambda: H.Ba.knn_batch(hg, Q[:64], 64, 10)))          # the functor accept rule
    hg.export()
    g = oracle.Graph(hg.n, hg.entry_point, hg.deg0, hg.nbr0, hg.upper)
    sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=100, counters=True)
    oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=10, ef=100, ties=oracle.TIES_CANONICAL, counters=True)
    np.testing.assert_array_equal(ids, oi)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(nh, onh)
    hg.release()


def test_rows_of_64_neighbours_ordered_launch_and_a_created_index(H, oracle, tmp_path):
    """M = 32 (adjacency rows of 64: every lane of the wave holds a neighbour and its tail slot), a batch large enough for
    the longest-first ordering, and the copy rebuilt by hnsw_index_create / hnsw_index_load from a flattened graph"""
    n, d = 20000, 100
    X, Q = _unit(n, d, 5), _unit(9000, d, 6)
    hg = H.Ohnsw.build_batch_bigarray(X, 32, 100, seed=1, metric=1)
    hg.set_option("order_queries", 1)
    a, b = _both_ways(hg, lambda: H.Ohnsw.knn_batch_bigarray(hg, 100, Q, ef=256, counters=True))
    _same(a, b)
    hg.export()
    g = oracle.Graph(hg.n, hg.entry_point, hg.deg0, hg.nbr0, hg.upper)
    sp = oracle.Space.ip(X, arith=oracle.TREE16)
    oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q[:300], k=100, ef=256, ties=oracle.TIES_CANONICAL, counters=True)
    np.testing.assert_array_equal(a[0][:300], oi)
    np.testing.assert_array_equal(a[1][:300].view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(a[3][:300], onh)
    p = str(tmp_path / "idx.bin")
    hg.save(p)
    hg2 = H.Hgraph.load(p)
    hg3 = H.Hgraph(X, hg.deg0, hg.nbr0, hg.upper, entry_point=hg.entry_point, max_degree=32, metric=1)
    for h in (hg2, hg3):
        assert h.info().row_format == ROWS_SPLIT
        h.set_option("order_queries", 1)
        h.set_option("vt_bits", 12)
        _same(a, H.Ohnsw.knn_batch_bigarray(h, 100, Q, ef=256, counters=True))
    for h in (hg, hg2, hg3):
        h.release()


@pytest.mark.parametrize("d", [20, 64, 96, 108, 128, 204])
def test_shapes_that_do_not_qualify_keep_plain_rows(H, d):
    """fewer than one whole line, rows that end on a line (d = 64, 96, 128), tails of three chunks and more"""
    X = _unit(600, d, 11)
    hg = H.Ohnsw.build_batch_bigarray(X, 8, 40, seed=1)
    assert hg.info().row_format == ROWS_F32
    hg.set_option("split_rows", 1)                     # nothing to switch to
    assert hg.info().row_format == ROWS_F32
    ids, _ = H.Ohnsw.knn_batch_bigarray(hg, 5, X[:50], ef=50)
    assert (ids[:, 0] == np.arange(50)).mean() > 0.9
    hg.release()


def test_byte_rows_take_precedence(H):
    rng = np.random.default_rng(1)
    X = rng.integers(0, 256, size=(800, 100)).astype(np.float32)
    hg = H.Ohnsw.build_batch_bigarray(X, 8, 40, seed=1)
    assert hg.info().row_format == ROWS_BYTES            # one 128-byte line per vector already
    hg.set_option("byte_rows", 0)
    assert hg.info().row_format == ROWS_F32              # no split copy was built beside the byte rows
    hg.release()


def test_layer_operators_and_distances_are_untouched(H, oracle):
    """search_k / search_one as operators and hnsw_distance_batch read the plain rows of a split index"""
    n, d = 3000, 100
    X, Q = _unit(n, d, 21), _unit(40, d, 22)
    hg = H.Ohnsw.build_batch_bigarray(X, 12, 60, seed=2)
    assert hg.info().row_format == ROWS_SPLIT
    hg.export()
    g = oracle.Graph(hg.n, hg.entry_point, hg.deg0, hg.nbr0, hg.upper)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    got = H.Ohnsw.search_k(hg, 0, [[hg.entry_point]] * 40, Q, 50)
    for q in range(40):
        want = oracle.Ohnsw.search_k(g, sp, [hg.entry_point], Q[q], 50, layer=0, ties=oracle.TIES_CANONICAL)
        assert [i for i, _ in got[q]] == [i for i, _ in want]
    ids = np.arange(0, 3000, 7, dtype=np.int32)[None, :].repeat(4, 0)
    dd = H.Ohnsw.distance_l2(hg, Q[:4], ids)
    want = np.sqrt(((X[ids].astype(np.float64) - Q[:4, None, :]) ** 2).sum(-1))
    assert np.max(np.abs(dd - want) / want) < 1e-5           # the north star's tolerance
    hg.release()
