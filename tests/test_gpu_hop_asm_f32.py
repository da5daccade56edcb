"""The hand-scheduled layer-0 loops over FLOAT32 rows (csrc/hnsw_hop_asm.hip.h, "The same loops over FLOAT32 rows": 65..128 and
129..256 dimensions, L2 and inner product, both accept rules, ef <= 256; full rows d = 125..128, ragged rows and split rows) against the
oracle.  The round of these loops is new text -- row loads, the distance in hop_round's operation order, a reduction that
pairs lane l with lane l ^ 8, 4, 2, 1 while folding the round's candidates into one register -- so the data here is chosen
for the arithmetic (values whose sums depend on the order of the additions, squares that underflow, negative zeros,
negative inner-product distances) as well as for the control flow the byte-row tests drive (ties everywhere, the tie list
and its overflow, rounds of 1 / 2 / 4 batches, every rank slot).  Bit parity throughout: ids, distance bits, hop counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import ocaml_hnsw_amd as H
    H.load()
    assert H.device_count() >= 1
    return H


def _hgraph(H, X, g, M, metric=0, split=0):
    hg = H.Hgraph(X, g.deg0, g.nbr0, g.upper, entry_point=g.entry_point, id_base=0, max_degree=M, metric=metric)
    hg.set_option("byte_rows", 0)                 # integer-valued test data would otherwise take the byte rows
    hg.set_option("split_rows", split)            # 17, 18, 25, 26 chunks would otherwise take the split rows (C++ loop)
    assert hg.row_bytes() == 4 * X.shape[1]
    return hg


def _space(oracle, X, metric):
    return (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)


def _check(H, oracle, hg, g, sp, Q, ef, k, ctx=""):
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
    oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=k, ef=ef, ties=oracle.TIES_CANONICAL, counters=True)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32), err_msg=ctx)
    np.testing.assert_array_equal(ids, oi, err_msg=ctx)
    np.testing.assert_array_equal(nh, onh, err_msg=ctx)
    assert (nd > 0).all(), ctx


def _check_functor(H, oracle, hg, g, sp, Q, ef, k, ctx=""):
    """the functor accept rule (Hnsw.Ba.knn_batch) under the (d, id) order: the loops leave a hop when an entry would enter the tie set"""
    import ocaml_hnsw_amd as A
    gi, gd, gnd, gnh = A._search(hg, Q, ef, k, A.FILL_BA, True, sem=A.SEM_FUNCTOR)
    cd, ci = oracle.Functor.knn_batch(g, sp, Q, ef, k, ties=oracle.TIES_CANONICAL, with_ids=True)
    np.testing.assert_array_equal(gd.view(np.uint32), cd.view(np.uint32), err_msg=ctx)
    np.testing.assert_array_equal(gi, ci, err_msg=ctx)
    assert (gnd > 0).all() and (gnh > 0).all(), ctx


# every slot count (W in 1 / 2 / 3 / 4 / 6 / 8 registers: ef <= 64 / 128 / 192 / 256 / 384 / 512) at both ends of its range
EFS = ((1, 1), (17, 5), (64, 64), (65, 10), (100, 100), (128, 10), (129, 20), (192, 10), (193, 30), (256, 256), (257, 10), (384, 60), (385, 10), (400, 400), (512, 64))


# d: 68 = 17 chunks (one lane of the second chunk), 76 = 19, 100 = 25, 123 = 31 chunks with a partial last one, 125 = full rows
# whose last chunk is padded, 128 = full
@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d", [68, 76, 100, 123, 125, 128])
def test_order_dependent_sums_every_slot_count(H, oracle, d, metric):
    rng = np.random.default_rng(1000 * metric + d)
    n = 4000
    # magnitudes spread over twelve binary orders: a sum of these is only right bit for bit in the reference's order
    X = (rng.normal(size=(n, d)) * np.exp2(rng.integers(-6, 6, size=(n, d)))).astype(np.float32)
    Q = (rng.normal(size=(120, d)) * np.exp2(rng.integers(-6, 6, size=(120, d)))).astype(np.float32)
    X[5, ::3] = -0.0
    Q[3, ::2] = -0.0
    Q[4] = X[7]
    sp = _space(oracle, X, metric)
    g = oracle.build_ohnsw(sp, 12, 60, seed=3)
    hg = _hgraph(H, X, g, 12, metric)
    for ef, k in EFS:
        _check(H, oracle, hg, g, sp, Q, ef, k, "metric %d d %d ef %d" % (metric, d, ef))
        if ef in (17, 100, 192):
            _check_functor(H, oracle, hg, g, sp, Q, ef, k, "functor metric %d d %d ef %d" % (metric, d, ef))
    hg.release()


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("levels", [2, 3, 40])
@pytest.mark.parametrize("d", [96, 128])
def test_ties_everywhere_float_rows(H, oracle, levels, d, metric):
    rng = np.random.default_rng(100 * levels + d + metric)
    n = 5000
    X = rng.integers(0, levels, size=(n, d)).astype(np.float32)
    X[rng.integers(0, n, 300)] = X[rng.integers(0, n, 300)]               # exact duplicates
    Q = rng.integers(0, levels, size=(150, d)).astype(np.float32)
    Q[:10] = X[:10]
    sp = _space(oracle, X, metric)
    g = oracle.build_ohnsw(sp, 12, 60, seed=3)
    hg = _hgraph(H, X, g, 12, metric)
    for ef, k in EFS:
        _check(H, oracle, hg, g, sp, Q, ef, k, "levels %d d %d metric %d ef %d" % (levels, d, metric, ef))
        _check_functor(H, oracle, hg, g, sp, Q, ef, k, "functor levels %d d %d metric %d ef %d" % (levels, d, metric, ef))
    hg.release()


# split rows (csrc/hnsw_rows_split.hip): d = 66 -> 17 chunks (16 in the main row + 1 tail chunk), 70 -> 18 (16 + 2), 100 -> 25 (24 + 1),
# 104 -> 26 (24 + 2): the loops read the tail chunks from the expanded node's tail row at the candidate's slot; rows of four
# chunk columns: 130 -> 33 (32 + 1: the tail in column 2), 136 -> 34, 164 -> 41 (40 + 1: column 2 half main), 200 -> 50 (48 + 2: the tail
# in column 3), 228 -> 57 (56 + 1: column 3 half main), 232 -> 58
@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d", [66, 70, 100, 104, 130, 136, 164, 200, 228, 232])
@pytest.mark.parametrize("kind", ["ties", "spread"])
def test_split_rows_through_the_loops(H, oracle, d, metric, kind):
    rng = np.random.default_rng(7000 + 10 * d + metric)
    n = 4000
    if kind == "ties":
        X = rng.integers(0, 3, size=(n, d)).astype(np.float32) * 0.5
        Q = rng.integers(0, 3, size=(120, d)).astype(np.float32) * 0.5
        X[rng.integers(0, n, 200)] = X[rng.integers(0, n, 200)]
    else:
        X = (rng.normal(size=(n, d)) * np.exp2(rng.integers(-5, 5, size=(n, d)))).astype(np.float32)
        Q = (rng.normal(size=(120, d)) * np.exp2(rng.integers(-5, 5, size=(120, d)))).astype(np.float32)
    Q[:5] = X[:5]
    sp = _space(oracle, X, metric)
    g = oracle.build_ohnsw(sp, 12, 60, seed=5)
    hg = _hgraph(H, X, g, 12, metric, split=1)
    assert hg.info().row_format == 3                                       # HNSW_ROWS_SPLIT: the format under test is in use
    for ef, k in EFS:
        _check(H, oracle, hg, g, sp, Q, ef, k, "split %s metric %d d %d ef %d" % (kind, metric, d, ef))
        _check_functor(H, oracle, hg, g, sp, Q, ef, k, "split functor %s metric %d d %d ef %d" % (kind, metric, d, ef))
    hg.release()


# rows of 129..256 dimensions (four chunks per lane): d = 132 -> 33 chunks (one lane of the third chunk), 200 -> 50, 250 -> 63 (the
# last chunk partial), 253 -> full rows whose last chunk is padded, 256 -> full
@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d", [132, 200, 250, 253, 256])
def test_rows_of_129_to_256_dimensions_through_the_loops(H, oracle, d, metric):
    rng = np.random.default_rng(8000 + 10 * d + metric)
    n = 3000
    X = (rng.normal(size=(n, d)) * np.exp2(rng.integers(-5, 5, size=(n, d)))).astype(np.float32)
    Q = (rng.normal(size=(100, d)) * np.exp2(rng.integers(-5, 5, size=(100, d)))).astype(np.float32)
    X[5, ::3] = -0.0
    Q[4] = X[7]
    sp = _space(oracle, X, metric)
    g = oracle.build_ohnsw(sp, 12, 60, seed=3)
    hg = _hgraph(H, X, g, 12, metric)
    for ef, k in EFS:
        _check(H, oracle, hg, g, sp, Q, ef, k, "n4 metric %d d %d ef %d" % (metric, d, ef))
        if ef in (17, 100, 192, 400):
            _check_functor(H, oracle, hg, g, sp, Q, ef, k, "n4 functor metric %d d %d ef %d" % (metric, d, ef))
    hg.release()
    # ties everywhere on the same shapes
    Xt = rng.integers(0, 3, size=(n, d)).astype(np.float32) * 0.5
    Qt = rng.integers(0, 3, size=(60, d)).astype(np.float32) * 0.5
    sp = _space(oracle, Xt, metric)
    g = oracle.build_ohnsw(sp, 12, 60, seed=4)
    hg = _hgraph(H, Xt, g, 12, metric)
    for ef, k in ((30, 10), (128, 20), (256, 30), (512, 40)):
        _check(H, oracle, hg, g, sp, Qt, ef, k, "n4 ties metric %d d %d ef %d" % (metric, d, ef))
        _check_functor(H, oracle, hg, g, sp, Qt, ef, k, "n4 ties functor metric %d d %d ef %d" % (metric, d, ef))
    hg.release()


def test_underflowing_squares_and_huge_values(H, oracle):
    """squares below the normal range (the flush-or-keep behaviour of the hand-written v_mul / v_fmac must be the compiler's),
    and sums near the top of the range"""
    rng = np.random.default_rng(9)
    n, d = 3000, 128
    for scale in (1e-19, 1e-22, 3e18):
        X = (rng.normal(size=(n, d)) * scale).astype(np.float32)
        Q = (rng.normal(size=(64, d)) * scale).astype(np.float32)
        sp = _space(oracle, X, 0)
        g = oracle.build_ohnsw(sp, 8, 40, seed=1)
        hg = _hgraph(H, X, g, 8)
        for ef, k in ((32, 10), (100, 10), (200, 10)):
            _check(H, oracle, hg, g, sp, Q, ef, k, "scale %g ef %d" % (scale, ef))
        hg.release()


@pytest.mark.parametrize("metric", [0, 1])
def test_wide_rows_and_long_lists_float_rows(H, oracle, metric):
    """M = 32: layer-0 rows of 64 neighbours, fresh lists longer than one 16-row round, all three round shapes."""
    rng = np.random.default_rng(7 + metric)
    n, d = 6000, 96
    centres = rng.normal(size=(12, d)) * 4
    X = (centres[rng.integers(0, 12, n)] + rng.normal(size=(n, d))).astype(np.float32)
    Q = (centres[rng.integers(0, 12, 200)] + rng.normal(size=(200, d))).astype(np.float32)
    sp = _space(oracle, X, metric)
    g = oracle.build_ohnsw(sp, 32, 80, seed=1)
    hg = _hgraph(H, X, g, 32, metric)
    for ef, k in ((48, 10), (128, 10), (250, 50), (500, 100)):
        _check(H, oracle, hg, g, sp, Q, ef, k, "M 32 metric %d ef %d" % (metric, ef))
        _check_functor(H, oracle, hg, g, sp, Q, ef, k, "functor M 32 metric %d ef %d" % (metric, ef))
    hg.release()


@pytest.mark.parametrize("ef", [64, 128, 256, 512])
def test_tie_list_overflow_through_the_float_loops(H, oracle, ef):
    """test_gpu_hop_asm.py::test_tie_list_overflow_through_the_loops on float32 rows: shells tied at the maximum are evicted
    unexpanded one by one; beyond 64 of them the query is flagged and the host entry point searches it again."""
    shells = ef - 1
    chain = 100
    n = 1 + shells + chain + 1
    pos = np.zeros(n, np.float32)
    pos[0] = 250.0
    pos[1:1 + shells] = 150.0
    pos[1 + shells:1 + shells + chain] = 149.0 - np.arange(chain)
    z = n - 1
    pos[z] = 1.0
    rows = [[] for _ in range(n)]
    width = 64
    first = list(range(1, 1 + shells))
    rows[0] = first[:width]
    rest = first[width:]
    hub = 1
    while rest:
        rows[hub] = rest[:width - 1]
        rest = rest[width - 1:]
        hub += 1
    rows[1] = rows[1][:width - 1] + [1 + shells]
    for i in range(chain - 1):
        rows[1 + shells + i] = [2 + shells + i]
    late = 1 + (shells * 2) // 3
    rows[late] = (rows[late] + [z])[:width]
    deg0 = np.array([len(r) for r in rows], np.int32)
    nbr0 = np.full((n, width), -1, np.int32)
    for i, r in enumerate(rows):
        nbr0[i, :len(r)] = r
    X = np.zeros((n, 128), np.float32)
    X[:, 0] = pos
    X[:, 1] = 0.5                                                          # not byte-valued anyway
    g = oracle.Graph(n, 0, deg0, nbr0)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    hg = H.Hgraph(X, deg0, nbr0, entry_point=0, max_degree=32)
    assert hg.row_bytes() == 4 * 128
    Q = np.zeros((3, 128), np.float32)
    want = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=10, ef=ef, ties=oracle.TIES_CANONICAL, counters=True)
    got = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=ef, counters=True)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    np.testing.assert_array_equal(got[3], want[3])
    hg.release()


def test_random_configurations_of_the_float_loop_shapes(H, oracle):
    """Random small problems inside the loops' domain (d 65..128, either metric, Ohnsw rule, ef 1..256), split rows on and off."""
    rng = np.random.default_rng(78)
    for trial in range(48):
        n = int(rng.integers(2, 700))
        d = int(rng.integers(65, 129))
        M = int(rng.choice([2, 4, 8, 16, 32]))
        metric = int(rng.integers(0, 2))
        kind = int(rng.integers(0, 3))
        ef = int(rng.choice([1, 3, 30, 63, 64, 65, 90, 127, 128, 129, 191, 255, 256, 257, 300, 511, 512]))
        k = int(rng.integers(1, min(ef, 100) + 1))
        if kind == 0:
            X = rng.normal(size=(n, d)).astype(np.float32); Q = rng.normal(size=(20, d)).astype(np.float32)
        elif kind == 1:
            X = rng.integers(-3, 4, size=(n, d)).astype(np.float32); Q = rng.integers(-3, 4, size=(20, d)).astype(np.float32)
        else:
            X = (rng.integers(0, 4, size=(n, d)) * 0.25).astype(np.float32); Q = (rng.integers(0, 4, size=(20, d)) * 0.25).astype(np.float32)
        sp = _space(oracle, X, metric)
        g = oracle.build_ohnsw(sp, M, 40, seed=trial)
        hg = _hgraph(H, X, g, M, metric, split=trial & 1)
        _check(H, oracle, hg, g, sp, Q, ef, k, "trial %d: n %d d %d M %d metric %d kind %d ef %d k %d" % (trial, n, d, M, metric, kind, ef, k))
        _check_functor(H, oracle, hg, g, sp, Q, ef, k, "functor trial %d: n %d d %d M %d metric %d kind %d ef %d k %d" % (trial, n, d, M, metric, kind, ef, k))
        hg.release()


def test_ordered_launch_and_device_entry_float_rows(H, oracle):
    """a batch large enough for the longest-first ordering (descent in the pre-pass kernel, issue priorities): the same
    results per query as the plain launch"""
    rng = np.random.default_rng(5)
    n, d = 4000, 128
    X = rng.normal(size=(n, d)).astype(np.float32)
    Q = rng.normal(size=(3000, d)).astype(np.float32)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 16, 60, seed=9)
    hg = _hgraph(H, X, g, 16)
    for ef, k in ((40, 10), (128, 10), (200, 20)):
        hg.set_option("order_queries", 0)
        plain = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
        hg.set_option("order_queries", 1)
        got = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
        for a, b in zip(plain, got):
            np.testing.assert_array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32), err_msg="ef %d" % ef)
    _check(H, oracle, hg, g, sp, Q[:200], 128, 10, "ordered")
    hg.release()
