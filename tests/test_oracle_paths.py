"""Oracle self-consistency: the two reference paths agree where the reference says they must,
the structural fixtures of lib/ohnsw.ml hold, and the documented nearest_k defect is reproduced
only behind its bug-compat switch."""
import numpy as np
import pytest


def _data(n, d, seed, integer=False):
    rng = np.random.default_rng(seed)
    if integer:
        return rng.integers(0, 219, size=(n, d)).astype(np.float32)
    return rng.uniform(-1, 1, size=(n, d)).astype(np.float32)


@pytest.fixture(scope="module")
def small(oracle):
    X = _data(3000, 24, 0)
    sp = oracle.Space.l2(X)
    g = oracle.build_ohnsw(sp, 8, 60, seed=1)
    Q = _data(64, 24, 1)
    return X, sp, g, Q


def test_visited_epoch_overflow(oracle):
    # lib/ohnsw.ml:285-295
    v = oracle.Visited(3)
    v.set_epoch(4611686018427387903 - 10)
    for _ in range(16):
        v.add(1); v.clear()
        assert not v.mem(1) and v.card() == 0
        v.add(1); v.clear()
        assert not v.mem(1) and v.card() == 0
    with pytest.raises(IndexError):  # :276-279
        v.mem(3)


def test_builder_shape(small, oracle):
    X, sp, g, Q = small
    M = 8
    assert g.deg0.max() <= 2 * M and g.max_layer >= 1
    # symmetric links on layer 0 (Graph.Test.invariant, lib/ohnsw.ml:217-225)
    adj = [set(g.nbr0[i, :g.deg0[i]].tolist()) for i in range(g.n)]
    assert all(i in adj[j] for i in range(g.n) for j in adj[i])
    # level law round_nearest => layer 1 holds ~ n/sqrt(M) nodes
    n1 = len(g.upper[0][0])
    assert 0.5 * g.n / np.sqrt(M) < n1 < 1.6 * g.n / np.sqrt(M)


def test_paths_agree_without_ties(small, oracle):
    """On tie-free data the imperative path, the functor path and the canonical order all return
    the same W (same algorithm, different containers)."""
    X, sp, g, Q = small
    for q in Q[:32]:
        a = oracle.Ohnsw.knn(g, sp, q, k=10, ef=40, ties=0)
        b = oracle.Ohnsw.knn(g, sp, q, k=10, ef=40, ties=1)
        c = oracle.Functor.knn(g, sp, q, 40, 10, ties=0)
        d = oracle.Functor.knn(g, sp, q, 40, 10, ties=1)
        assert a == b == c == d
        assert [x[1] for x in a] == sorted(x[1] for x in a)


def test_nearest_k_bug_compat(small, oracle):
    """Hnsw.Nearest.nearest_k keeps the k FARTHEST of the ef results when ef > k
    (lib/hnsw.ml:519-525); with ef == k it is correct."""
    X, sp, g, Q = small
    q = Q[0]
    full = oracle.Functor.knn(g, sp, q, 40, 40)
    bug = oracle.Functor.knn(g, sp, q, 40, 10, bug_compat_farthest_k=True)
    assert bug == full[30:40]
    assert oracle.Functor.knn(g, sp, q, 10, 10, bug_compat_farthest_k=True) == \
        oracle.Functor.knn(g, sp, q, 10, 10)


def test_batch_layout_and_recall(small, oracle):
    X, sp, g, Q = small
    ids, dist, nd, nh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=10, ef=64, counters=True)
    assert ids.shape == (64, 10) and dist.dtype == np.float32
    gt_ids, gt_dist = oracle.brute_force_knn(sp, Q, 10)
    rec = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids.tolist(), gt_ids.tolist())])
    assert rec > 0.9
    assert oracle.recall_distance_threshold(gt_dist, dist) >= rec - 1e-9
    fd = oracle.Functor.knn_batch(g, sp, Q, 64, 10)
    np.testing.assert_array_equal(fd, dist)
    assert (nd > nh).all() and (nh > 0).all()


def test_fill_values(oracle):
    # fewer than k reachable nodes: -1 / NaN (lib/ohnsw.ml:880-881) vs +inf (lib/hnsw.ml:771)
    X = _data(3, 4, 5)
    sp = oracle.Space.l2(X)
    g = oracle.Graph.from_lists([[1], [0], []], entry_point=0)
    ids, dist = oracle.Ohnsw.knn_batch_bigarray(g, sp, X[:1], k=4)
    assert ids[0].tolist()[2:] == [-1, -1] and np.isnan(dist[0, 2:]).all()
    fd = oracle.Functor.knn_batch(g, sp, X[:1], 4, 4)
    assert np.isinf(fd[0, 2:]).all()


def test_empty_graph_raises(oracle):
    X = _data(3, 4, 5)
    sp = oracle.Space.l2(X)
    g = oracle.Graph.from_lists([[], [], []], entry_point=-1)
    with pytest.raises(ValueError, match="empty hgraph"):  # lib/ohnsw.ml:862
        oracle.Ohnsw.knn(g, sp, X[0], k=2)


def test_arithmetic_modes(oracle):
    rng = np.random.default_rng(3)
    for d in (3, 32, 96, 100, 128, 784):
        a = rng.uniform(-1, 1, d).astype(np.float32)
        b = rng.uniform(-1, 1, d).astype(np.float32)
        exact = float(np.sum((a.astype(np.float64) - b.astype(np.float64)) ** 2))
        t = oracle.l2sq_tree16(a, b)
        assert abs(t - exact) <= 1e-6 * exact
        dot = oracle.dot_tree16(a, b)
        assert abs(dot - float(a.astype(np.float64) @ b.astype(np.float64))) < 1e-5
    # integer-valued (SIFT-like) data: every summation order is exact (sums < 2^24)
    a = rng.integers(0, 219, 128).astype(np.float32)
    b = rng.integers(0, 219, 128).astype(np.float32)
    assert oracle.l2sq_tree16(a, b) == float(np.sum((a - b) ** 2, dtype=np.float64))


def test_canonical_vs_heap_on_integer_ties(oracle):
    """SIFT-like integer data has exact distance ties; the canonical (d,id) order and the
    heap-defined order must still return identical distance profiles (tie-class equality)."""
    X = _data(2000, 8, 7, integer=True) // 40  # few distinct values => many ties
    sp = oracle.Space.l2(X)
    g = oracle.build_ohnsw(sp, 6, 40, seed=3)
    Q = X[:40] + 0
    ia, da = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=8, ef=24, ties=0)
    ib, db = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=8, ef=24, ties=1)
    same = np.mean(np.all(da == db, axis=1))
    assert same > 0.8  # different tie orders explore slightly different sets; profiles mostly equal


def test_c1_configuration_regression_pin(oracle):
    """BASELINE.json configs[0] (10 k random fp32 vectors d = 32, M 8, ef 32, k 10: the reference's own
    CPU-runnable case) through the oracle: results pinned by tests/golden/c1_oracle_checksum.json
    (regenerated by tests/golden/make_c1_checksum.py) so that the checker cannot drift unnoticed."""
    import importlib.util
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_c1_checksum", os.path.join(here, "make_c1_checksum.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    got = m.run()
    with open(os.path.join(here, "c1_oracle_checksum.json")) as f:
        want = json.load(f)
    assert got == want
    assert got["seq_f32"]["ids_sha256"] == got["tree16"]["ids_sha256"]      # the summation order moves no neighbour here
    assert 0.6 < got["seq_f32"]["recall_at_10"] < 0.9                        # ef = 32 on structureless data


def test_functor_tie_with_max_w_is_inserted_but_w_keeps_the_incumbent(oracle):
    """lib/hnsw.ml:494-506 + lib/hnsw_algo.ml:25-32, 360-364: a neighbour tied with max(W) is answered
    Inserted (so it is expanded later) while W keeps the incumbent.  Both tie modes of the oracle's functor
    path follow that rule (TIES_CANONICAL only fixes the order among equal keys); the imperative path's
    strict `<` (lib/ohnsw.ml:574) drops the twin."""
    vals = np.array([[5.0], [2.0], [-2.0], [0.5]], np.float32)   # node 3 is reachable only through node 2
    g = oracle.Graph.from_lists([[1, 2], [0], [0, 3], [2]], entry_point=0)
    sp = oracle.Space.l2(vals, arith=oracle.SEQ_F32)
    Q = np.zeros((1, 1), np.float32)
    for ties in (oracle.TIES_HEAP, oracle.TIES_CANONICAL):
        od, oi = oracle.Functor.knn_batch(g, sp, Q, 1, 1, ties=ties, with_ids=True)
        assert oi[0, 0] == 3 and od[0, 0] == 0.5
        res = oracle.Functor.search(g, sp, [0], Q[0], 1, ties=ties)
        assert res == [(3, 0.5)]
    oi, od = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=1, ef=1, ties=oracle.TIES_CANONICAL)
    assert oi[0, 0] == 1 and od[0, 0] == 2.0


def test_stats_restatement_follows_the_reference_fold(oracle):
    """Hgraph.Stats.min_max_connectivity (lib/hnsw.ml:361-368), worked by hand: the fold starts from (1000000, -1, 0, 0., []),
    takes min / max of Neighbours.length, sums as floats, and conses a node without neighbours onto `isolated` as the
    ascending Map.fold meets it -- so the list is descending."""
    g = oracle.Graph.from_lists([[1], [0, 2], [1], [], []], 0)
    st = oracle.Stats.compute(g)
    assert st == {"num_nodes": 5, "layer_sizes": {0: 5},
                  "layer_connectivity": {0: {"min": 0, "max": 2, "mean": 4 / 5, "isolated": [4, 3]}}}
    ring = oracle.Stats.compute(oracle.Graph.ring(5))      # Graph.Test.create_loop (lib/ohnsw.ml:205-212)
    assert ring["layer_connectivity"][0] == {"min": 2, "max": 2, "mean": 2.0, "isolated": []}
    # upper layers: the keys of layer l are the nodes present there; a present node with no neighbour is isolated
    deg0 = np.array([1, 1, 0], np.int32)
    nbr0 = np.array([[1], [0], [-1]], np.int32)
    up1 = (np.array([0, 2], np.int64), np.array([0, 0], np.int32), np.full((2, 1), -1, np.int32))
    st = oracle.Stats.compute(oracle.Graph(3, 0, deg0, nbr0, [up1]))
    assert st["layer_sizes"] == {0: 3, 1: 2}
    assert st["layer_connectivity"][0] == {"min": 0, "max": 1, "mean": 2 / 3, "isolated": [2]}
    assert st["layer_connectivity"][1] == {"min": 0, "max": 0, "mean": 0.0, "isolated": [2, 0]}
