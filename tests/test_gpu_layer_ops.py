"""GPU parity of the layer-level operators (hnsw_search_layer_batch = Ohnsw.search_k /
Hnsw_algo.Search.search, hnsw_search_one_batch = Ohnsw.search_one / Search.search_one) and of the
single-process multi-device entry points (hnsw_multi_*), through the C ABI, against the oracle."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

G = load_golden("ohnsw_inline_tests.json")


@pytest.fixture(scope="module")
def H():
    import ocaml_hnsw_amd as H
    H.load()
    assert H.device_count() >= 1, "GPU tests need a HIP device"
    return H


def _hgraph(H, X, g, id_base=0, metric=0, M=None):
    up = [(nodes + id_base, deg, np.where(nbr >= 0, nbr + id_base, -1)) for nodes, deg, nbr in g.upper]
    nbr0 = np.where(g.nbr0 >= 0, g.nbr0 + id_base, -1)
    ep = None if g.entry_point < 0 else g.entry_point + id_base
    return H.Hgraph(X, g.deg0, nbr0, up, entry_point=ep, id_base=id_base, max_degree=M, metric=metric)


def _kat_graph(o, kind, n):
    return o.Graph.from_lists([[] for _ in range(n)]) if kind == "isolated" else o.Graph.ring(n)


# ---- the reference's inline tests, called the way the reference calls them -------------------------
@pytest.mark.parametrize("case", G["search_k"], ids=lambda c: c["ref"])
def test_reference_search_k_kats_layer_operator(H, oracle, case):
    """lib/ohnsw.ml:593-644: search_k layer distance value visited {start} target k.  d = 1 makes
    the L2 distance |a-b| (the tests' distance, :361)."""
    vals = np.array(case["values"], np.float32)[:, None]
    hg = _hgraph(H, vals, _kat_graph(oracle, case["graph"], len(vals)), M=2)
    got = H.Ohnsw.search_k(hg, 0, [[case["start"]]], [[case["target"]]], case["k"])[0]
    assert [n for n, _ in got] == [n for n, _ in case["expect"]]
    for (_, d), (_, e) in zip(got, case["expect"]):
        assert d == pytest.approx(e, rel=1e-6, abs=1e-6)
    # the functor path's Search.search gives the same W on these tie-free cases
    got_f = H.Ba.search(hg, 0, [[case["start"]]], [[case["target"]]], case["k"])[0]
    assert [n for n, _ in got_f] == [n for n, _ in case["expect"]]


@pytest.mark.parametrize("case", G["search_one"], ids=lambda c: c["ref"])
def test_reference_search_one_kats_layer_operator(H, oracle, case):
    """lib/ohnsw.ml:514-534: search_one layer distance value visited start target."""
    vals = np.array(case["values"], np.float32)[:, None]
    hg = _hgraph(H, vals, _kat_graph(oracle, case["graph"], len(vals)), M=2)
    got = H.Ohnsw.search_one(hg, 0, case["start"], [[case["target"]]])
    assert got.tolist() == [case["expect"]]


# ---- seeded parity on built graphs, every layer ------------------------------------------------------
@pytest.fixture(scope="module")
def built(H, oracle):
    rng = np.random.default_rng(77)
    X = rng.integers(0, 12, size=(5000, 24)).astype(np.float32)     # integer grid: exact ties
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 6, 50, seed=9)
    assert g.max_layer >= 2
    return X, sp, g


def _layer_nodes(g, layer):
    return np.arange(g.n) if layer == 0 else g.upper[layer - 1][0]


@pytest.mark.parametrize("id_base", [0, 1])
def test_search_k_every_layer_matches_oracle(H, oracle, built, id_base):
    X, sp, g = built
    hg = _hgraph(H, X, g, id_base=id_base, M=6)
    rng = np.random.default_rng(5 + id_base)
    for layer in range(g.max_layer + 1):
        nodes = _layer_nodes(g, layer)
        for ef, k in ((1, 1), (7, 7), (40, 12), (64, 64), (130, 100), (300, 5)):
            nq = 12
            T = (X[rng.integers(0, g.n, nq)] + rng.integers(0, 2, size=(nq, X.shape[1]))).astype(np.float32)
            starts = [rng.choice(nodes, size=int(rng.integers(1, min(ef, len(nodes), 70) + 1)), replace=False).tolist()
                      for _ in range(nq)]
            got, nd, nh = H.Ohnsw.search_k(hg, layer, [[s + id_base for s in st] for st in starts], T, k, ef=ef, counters=True)
            got_f = H.Ba.search(hg, layer, [[s + id_base for s in st] for st in starts], T, ef)
            for j in range(nq):
                want, c = oracle.Ohnsw.search_k(g, sp, starts[j], T[j], ef, layer=layer, ties=oracle.TIES_CANONICAL, counters=True)
                ctx = dict(layer=layer, ef=ef, k=k, j=j, n_start=len(starts[j]))
                assert [n - id_base for n, _ in got[j]] == [n for n, _ in want[:k]], ctx
                assert [np.float32(d).view(np.uint32) for _, d in got[j]] == \
                       [np.float32(d).view(np.uint32) for _, d in want[:k]], ctx
                assert int(nh[j]) == c.n_hops, ctx
                assert int(nd[j]) >= len(starts[j]), ctx
                want_f = oracle.Functor.search(g, sp, starts[j], T[j], ef, layer=layer, ties=oracle.TIES_CANONICAL)
                assert [n - id_base for n, _ in got_f[j]] == [n for n, _ in want_f], ctx


def test_search_one_every_layer_matches_both_reference_versions(H, oracle, built):
    X, sp, g = built
    hg = _hgraph(H, X, g, M=6)
    rng = np.random.default_rng(6)
    for layer in range(g.max_layer + 1):
        nodes = _layer_nodes(g, layer)
        nq = 60
        T = (X[rng.integers(0, g.n, nq)] + rng.integers(0, 3, size=(nq, X.shape[1]))).astype(np.float32)
        start = rng.choice(nodes, size=nq)
        node, dist = H.Ba.search_one(hg, layer, start, T)
        for j in range(nq):
            a = oracle.Ohnsw.search_one(g, sp, int(start[j]), T[j], layer=layer)
            b, bd = oracle.Functor.search_one(g, sp, int(start[j]), T[j], layer=layer, ties=oracle.TIES_CANONICAL)
            # both reference versions walk to a node at the same distance; on the simple version's
            # node the GPU agrees exactly
            assert int(node[j]) == a, (layer, j)
            assert np.float32(dist[j]).view(np.uint32) == np.float32(bd).view(np.uint32), (layer, j)


def test_start_node_missing_from_layer_and_ragged_lists(H, oracle, built):
    """A start node that is not on the layer has no neighbours there (MapGraph.adjacent,
    lib/hnsw.ml:146-149): W is just the seeds.  Lists of different lengths are padded with
    id_base - 1."""
    X, sp, g = built
    hg = _hgraph(H, X, g, M=6)
    top = g.max_layer
    on_top = set(g.upper[top - 1][0].tolist())
    absent = [i for i in range(g.n) if i not in on_top][:3]
    got = H.Ohnsw.search_k(hg, top, [absent, absent[:1]], X[:2], 8)
    assert sorted(n for n, _ in got[0]) == sorted(absent)
    assert [n for n, _ in got[1]] == absent[:1]
    want = oracle.Ohnsw.search_k(g, sp, absent, X[0], 8, layer=top, ties=oracle.TIES_CANONICAL)
    assert [n for n, _ in got[0]] == [n for n, _ in want]


def test_layer_operator_tie_overflow_is_exact(H, oracle):
    """The crafted graph of test_tie_overflow_beyond_lds_stack, through search_k on layer 0."""
    n = 229
    pos = np.zeros(n, np.float32)
    pos[0] = 20.0
    pos[1:128] = 10.0
    pos[128:228] = 9.0 - 0.01 * np.arange(100)
    pos[228] = 0.1
    rows = [[] for _ in range(n)]
    rows[0] = [1] + list(range(2, 65))
    rows[1] = list(range(65, 128)) + [128]
    for i in range(99):
        rows[128 + i] = [129 + i]
    rows[40] = [228]
    deg0 = np.array([len(r) for r in rows], np.int32)
    nbr0 = np.full((n, 64), -1, np.int32)
    for i, r in enumerate(rows):
        nbr0[i, :len(r)] = r
    X = pos[:, None]
    g = oracle.Graph(n, 0, deg0, nbr0)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    hg = H.Hgraph(X, deg0, nbr0, entry_point=0, max_degree=32)
    want = oracle.Ohnsw.search_k(g, sp, [0], np.zeros(1, np.float32), 128, ties=oracle.TIES_CANONICAL)
    got = H.Ohnsw.search_k(hg, 0, [[0]], np.zeros((1, 1), np.float32), 128)[0]
    assert 228 in [n for n, _ in want]
    assert [n for n, _ in got] == [n for n, _ in want]


def test_layer_operator_errors(H, oracle, built):
    X, sp, g = built
    hg = _hgraph(H, X, g, M=6)
    with pytest.raises(H.InvalidArgument):
        H.Ohnsw.search_k(hg, g.max_layer + 1, [[0]], X[:1], 4)          # Hgraph.layer out of range
    with pytest.raises(H.InvalidArgument):
        H.Ohnsw.search_k(hg, 0, [[0, 1, 2]], X[:1], 2)                  # n_start > ef
    with pytest.raises(H.InvalidArgument):
        H.Ohnsw.search_k(hg, 0, [[g.n]], X[:1], 2)                      # Vector.get out of range
    with pytest.raises(H.InvalidArgument):
        H.Ohnsw.search_one(hg, 0, -1, X[:1])
    with pytest.raises(H.InvalidArgument):
        H.Ohnsw.search_one(hg, -1, 0, X[:1])


# ---- one process, several devices ------------------------------------------------------------------
def test_multi_device_replicas_equal_single_device(H, oracle, built):
    """hnsw_multi_*: replicas (here three on device 0; the box has one GPU) + contiguous shards give
    the arrays of the single-device call bit for bit, for batch sizes around the shard count."""
    X, sp, g = built
    hg = _hgraph(H, X, g, M=6)
    hg1 = _hgraph(H, X, g, id_base=1, M=6)
    multi = H.MultiHgraph(hg, [0, 0, 0])
    multi1 = H.MultiHgraph(hg1, [0, 0])
    assert multi.num_replicas() == 3
    rng = np.random.default_rng(3)
    for nq in (0, 1, 2, 3, 4, 7, 301):
        Q = (X[rng.integers(0, g.n, nq)] + rng.integers(0, 2, size=(nq, X.shape[1]))).astype(np.float32).reshape(nq, X.shape[1])
        a = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=48, counters=True)
        b = multi.knn_batch_bigarray(10, Q, ef=48, counters=True)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32))
        np.testing.assert_array_equal(H.Ba.knn_batch(hg1, Q, 48, 10).view(np.uint32),
                                      multi1.knn_batch(Q, 48, 10).view(np.uint32))
    oi, od = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=10, ef=48, ties=oracle.TIES_CANONICAL)
    np.testing.assert_array_equal(b[0], oi)
    multi.release()
    multi1.release()


def test_multi_rccl_exchange_leaves_the_full_result_on_every_device(H, oracle, built):
    """hnsw_multi_search_batch_device: sharded search, then the exchange -- ncclAllGather on communicators
    from ncclCommInitAll -- leaves the full [nq][k] table on every device.  The box has one GPU, so the RCCL
    path runs at communicator size 1 (devices [0]: ncclCommInitAll, group, ncclAllGather / ncclBroadcast
    are all really called); three replicas on device 0 exercise the shard layout and the unequal-shard
    case through the same-device copy exchange."""
    X, sp, g = built
    hg = _hgraph(H, X, g, M=6)
    rng = np.random.default_rng(5)
    one = H.MultiHgraph(hg, [0])
    three = H.MultiHgraph(hg, [0, 0, 0])
    for nq in (1, 3, 64, 301):
        Q = (X[rng.integers(0, g.n, nq)] + rng.integers(0, 2, size=(nq, X.shape[1]))).astype(np.float32)
        want_i, want_d = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=48)
        for multi in (one, three):
            d_ids, d_dist = multi.search_device(Q, 48, 10)
            assert all(d_ids) and all(d_dist) and len(d_ids) == multi.num_replicas()
            for r in range(multi.num_replicas()):
                gi, gd = multi.copy_result(r)
                np.testing.assert_array_equal(gi, want_i)
                np.testing.assert_array_equal(gd.view(np.uint32), want_d.view(np.uint32))
    one.release()
    three.release()


def test_multi_device_call_reuses_buffers_checks_parameters_and_repairs_flagged_shards(H, oracle, built):
    """hnsw_multi_search_batch_device twice in a row hands back the same device tables (buffers are kept, nothing is
    re-allocated for an equal-sized batch) with the second batch's results; a bad k is a parameter error, not an
    allocation; and a shard whose queries overflow their tie lists is searched again AFTER the exchange and its slice
    re-sent, so the table every device ends with is the exact one."""
    X, sp, g = built
    hg = _hgraph(H, X, g, M=6)
    rng = np.random.default_rng(9)
    three = H.MultiHgraph(hg, [0, 0, 0])
    Qa = (X[rng.integers(0, g.n, 100)] + 1).astype(np.float32)
    Qb = (X[rng.integers(0, g.n, 100)] + 2).astype(np.float32)
    pa = three.search_device(Qa, 48, 10)
    pb = three.search_device(Qb, 48, 10)
    assert pa == pb                                                        # same tables, reused
    want_i, want_d = H.Ohnsw.knn_batch_bigarray(hg, 10, Qb, ef=48)
    for r in range(3):
        gi, gd = three.copy_result(r)
        np.testing.assert_array_equal(gi, want_i)
        np.testing.assert_array_equal(gd.view(np.uint32), want_d.view(np.uint32))
    with pytest.raises(H.InvalidArgument, match="k=10 > ef=5"):
        three.search_device(Qa, 5, 10)
    three.release()
    # the tie-overflow scenario of tests/test_gpu_parity.py::test_tie_overflow_beyond_lds_stack on two replicas
    n = 229
    pos = np.zeros(n, np.float32)
    pos[0] = 20.0
    pos[1:128] = 10.0
    pos[128:228] = 9.0 - 0.01 * np.arange(100)
    pos[228] = 0.1
    rows = [[] for _ in range(n)]
    rows[0] = [1] + list(range(2, 65))
    rows[1] = list(range(65, 128)) + [128]
    for i in range(99):
        rows[128 + i] = [129 + i]
    rows[40] = [228]
    deg0 = np.array([len(r) for r in rows], np.int32)
    nbr0 = np.full((n, 64), -1, np.int32)
    for i, r in enumerate(rows):
        nbr0[i, :len(r)] = r
    Xo = pos[:, None]
    go = oracle.Graph(n, 0, deg0, nbr0)
    spo = oracle.Space.l2(Xo, arith=oracle.TREE16)
    hgo = H.Hgraph(Xo, deg0, nbr0, entry_point=0, max_degree=32)
    Q = np.zeros((5, 1), np.float32)
    Q[3] = 30.0                                                            # one query of the second shard that does not overflow
    want = oracle.Ohnsw.knn_batch_bigarray(go, spo, Q, k=10, ef=128, ties=oracle.TIES_CANONICAL)
    two = H.MultiHgraph(hgo, [0, 0])
    got = two.knn_batch_bigarray(10, Q, ef=128)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    two.search_device(Q, 128, 10)
    for r in range(2):
        gi, gd = two.copy_result(r)                                        # repaired on EVERY device
        np.testing.assert_array_equal(gi, want[0])
        np.testing.assert_array_equal(gd.view(np.uint32), want[1].view(np.uint32))
    c2 = two.debug_counters()
    assert c2["repaired_shards"] >= 2 and c2["peer_copies"] > 0 and c2["allgather"] == 0 and c2["broadcast"] == 0, c2
    two.release()
    # the same batch through RCCL at communicator size 1 (devices [0]: every listed device distinct): the all-gather of the
    # whole table, then -- the shard is flagged -- the exactness fallback and the re-send of the repaired shard, which is
    # the ncclBroadcast branch of the exchange (the one unequal shards take on a multi-GPU node)
    one = H.MultiHgraph(hgo, [0])
    for nq_ in (5, 3):                                                     # 3: no flagged query left out, an odd count
        Qn = Q[:nq_]
        wn = oracle.Ohnsw.knn_batch_bigarray(go, spo, Qn, k=10, ef=128, ties=oracle.TIES_CANONICAL)
        before = one.debug_counters()
        one.search_device(Qn, 128, 10)
        after = one.debug_counters()
        gi, gd = one.copy_result(0)
        np.testing.assert_array_equal(gi, wn[0])
        np.testing.assert_array_equal(gd.view(np.uint32), wn[1].view(np.uint32))
        assert after["allgather"] == before["allgather"] + 2, (before, after)            # ids + distances
        assert after["broadcast"] == before["broadcast"] + 2, (before, after)            # the repaired shard, ids + distances
        assert after["repaired_shards"] == before["repaired_shards"] + 1 and after["peer_copies"] == 0
    one.release()


def test_multi_exchange_refused_part_way_aborts_instead_of_hanging(H, oracle, built, monkeypatch):
    """An enqueue refused inside the RCCL group (hook HNSW_MULTI_FAIL_ENQUEUE = device index) must not leave the other devices'
    collectives standing: the call returns Failure with the communicators aborted (it does not hang in its own stream
    synchronisation), and the next search on the handle re-creates them and is exact.  Communicator size 1 here (one GPU)."""
    X, sp, g = built
    hg = _hgraph(H, X, g, M=6)
    one = H.MultiHgraph(hg, [0])
    Q = (X[:40] + 1).astype(np.float32)
    want_i, want_d = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=48)
    one.search_device(Q, 48, 10)                       # communicators exist
    before = one.debug_counters()
    monkeypatch.setenv("HNSW_MULTI_FAIL_ENQUEUE", "0")
    with pytest.raises(H.Failure, match="communicators were aborted"):
        one.search_device(Q, 48, 10)
    monkeypatch.delenv("HNSW_MULTI_FAIL_ENQUEUE")
    assert one.debug_counters()["allgather"] == before["allgather"]           # nothing of the refused exchange was counted
    one.search_device(Q, 48, 10)                       # new communicators, same table
    gi, gd = one.copy_result(0)
    np.testing.assert_array_equal(gi, want_i)
    np.testing.assert_array_equal(gd.view(np.uint32), want_d.view(np.uint32))
    assert one.debug_counters()["allgather"] == before["allgather"] + 2
    one.release()


def test_multi_device_errors(H, oracle, built):
    X, sp, g = built
    hg = _hgraph(H, X, g, M=6)
    with pytest.raises(H.InvalidArgument):
        H.MultiHgraph(hg, [0, H.device_count() + 5])                   # no such device
    with pytest.raises(H.InvalidArgument):
        H.MultiHgraph(hg, [])
    empty = H.Hgraph(np.zeros((0, 4), np.float32), np.zeros(0, np.int32), np.zeros((0, 4), np.int32))
    m = H.MultiHgraph(empty, [0, 0])
    with pytest.raises(H.InvalidArgument, match="empty hgraph"):
        m.knn_batch_bigarray(1, np.zeros((5, 4), np.float32))
    with pytest.raises(H.InvalidArgument):
        H.MultiHgraph(hg, [0, 0]).knn_batch_bigarray(10, X[:4], ef=5)   # k > ef, message from a worker thread


# ---- submit / wait: batches in flight -------------------------------------------------------------------
def test_submit_wait_equals_synchronous_call(H, oracle, built):
    """hnsw_search_submit / hnsw_search_wait: several batches in flight on the handle's streams, waited
    for out of order, give exactly what hnsw_search_batch gives for each of them."""
    X, sp, g = built
    hg = _hgraph(H, X, g, M=6)
    rng = np.random.default_rng(8)
    batches = [(X[rng.integers(0, g.n, nq)] + rng.integers(0, 2, size=(nq, X.shape[1]))).astype(np.float32)
               for nq in (1, 700, 5000, 33, 2048, 9)]
    for rounds in range(2):                                     # the second round reuses pooled request buffers
        reqs = [H.submit(hg, b, 40, 7) for b in batches]
        for j in (3, 0, 5, 1, 4, 2):
            ids, dist, nd, nh = reqs[j].wait(counters=True)
            want = H.Ohnsw.knn_batch_bigarray(hg, 7, batches[j], ef=40, counters=True)
            for a, b in zip((ids, dist, nd, nh), want):
                np.testing.assert_array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
    with pytest.raises(H.InvalidArgument):
        reqs[0].wait()                                          # a request is waited for once
    with pytest.raises(H.InvalidArgument):
        H.submit(hg, batches[1], 5, 7)                          # k > ef
    r = H.submit(hg, batches[1], 40, 7)                         # never waited for: released with the index
    del r
    hg.release()


def test_submit_wait_overflow_fallback_is_exact(H, oracle):
    """The tie-overflow fallback (test_layer_operator_tie_overflow_is_exact's graph) through a request."""
    n = 229
    pos = np.zeros(n, np.float32)
    pos[0] = 20.0
    pos[1:128] = 10.0
    pos[128:228] = 9.0 - 0.01 * np.arange(100)
    pos[228] = 0.1
    rows = [[] for _ in range(n)]
    rows[0] = [1] + list(range(2, 65))
    rows[1] = list(range(65, 128)) + [128]
    for i in range(99):
        rows[128 + i] = [129 + i]
    rows[40] = [228]
    deg0 = np.array([len(r) for r in rows], np.int32)
    nbr0 = np.full((n, 64), -1, np.int32)
    for i, r in enumerate(rows):
        nbr0[i, :len(r)] = r
    X = pos[:, None]
    g = oracle.Graph(n, 0, deg0, nbr0)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    hg = H.Hgraph(X, deg0, nbr0, entry_point=0, max_degree=32)
    Q = np.zeros((3, 1), np.float32)
    want = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=10, ef=128, ties=oracle.TIES_CANONICAL)
    r1, r2 = H.submit(hg, Q, 128, 10), H.submit(hg, Q[:1], 128, 10)
    ids2, _ = r2.wait()
    ids1, dist1 = r1.wait()
    np.testing.assert_array_equal(ids1, want[0])
    np.testing.assert_array_equal(dist1.view(np.uint32), want[1].view(np.uint32))
    np.testing.assert_array_equal(ids2, want[0][:1])


# ---- longest-first ordering of large batches --------------------------------------------------------------
def test_longest_first_ordering_changes_nothing_but_the_order(H, oracle, built):
    """Option "order_queries": a descent pre-pass + a sort decide in which order the queries of a batch are
    launched (farthest layer-0 entry first).  Results AND counters per query must be exactly those of the
    unordered launch -- for every kernel variant touched: several W sizes, both accept rules, both
    metrics, a graph without upper layers, tiny batches, and through the host-buffer and request paths."""
    X, sp, g = built
    rng = np.random.default_rng(12)
    Q = (X[rng.integers(0, g.n, 3000)] + rng.integers(0, 2, size=(3000, X.shape[1]))).astype(np.float32)
    hg = _hgraph(H, X, g, M=6)
    hg1 = _hgraph(H, X, g, id_base=1, M=6)
    for ef, k in ((10, 10), (100, 20), (200, 7), (600, 50)):
        for nq in (1, 5, 3000):
            hg.set_option("order_queries", 0)
            plain = H.Ohnsw.knn_batch_bigarray(hg, k, Q[:nq], ef=ef, counters=True)
            hg.set_option("order_queries", 1)
            ordered = H.Ohnsw.knn_batch_bigarray(hg, k, Q[:nq], ef=ef, counters=True)
            for a, b in zip(plain, ordered):
                np.testing.assert_array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
    hg.set_option("order_queries", 1)
    oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q[:300], k=20, ef=100, ties=oracle.TIES_CANONICAL, counters=True)
    gi, gd, gnd, gnh = H.Ohnsw.knn_batch_bigarray(hg, 20, Q[:300], ef=100, counters=True)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(gnh, onh)
    # functor rule, 1-based ids, request path
    hg1.set_option("order_queries", 0)
    want = H.Ba.knn_batch(hg1, Q, 64, 9)
    hg1.set_option("order_queries", 1)
    np.testing.assert_array_equal(H.Ba.knn_batch(hg1, Q, 64, 9).view(np.uint32), want.view(np.uint32))
    r = H.submit(hg1, Q, 64, 9, fill=H.FILL_BA, sem=H.SEM_FUNCTOR)
    np.testing.assert_array_equal(r.wait()[1].view(np.uint32), want.view(np.uint32))
    # inner product, and a graph with no upper layer at all (the "descent" is the entry point's distance)
    Xn = X / np.maximum(np.linalg.norm(X, axis=1, keepdims=True), 1e-6)
    spi = oracle.Space.ip(Xn.astype(np.float32), arith=oracle.TREE16)
    gi_ = oracle.build_ohnsw(spi, 6, 40, seed=2)
    hgi = _hgraph(H, Xn.astype(np.float32), gi_, metric=1, M=6)
    flat = H.Hgraph(X, g.deg0, g.nbr0, entry_point=g.entry_point, max_degree=6)
    for h_, q_ in ((hgi, (Q / np.maximum(np.linalg.norm(Q, axis=1, keepdims=True), 1e-6)).astype(np.float32)), (flat, Q)):
        h_.set_option("order_queries", 0)
        a = H.Ohnsw.knn_batch_bigarray(h_, 10, q_, ef=50, counters=True)
        h_.set_option("order_queries", 1)
        b = H.Ohnsw.knn_batch_bigarray(h_, 10, q_, ef=50, counters=True)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32))


def test_kernel_times_diagnostic(H, oracle, built):
    import torch
    X, sp, g = built
    hg = _hgraph(H, X, g, M=6)
    dev = torch.device("cuda", 0)
    Qd = torch.from_numpy(X[:2000].copy()).to(dev)
    ids = torch.empty((2000, 5), dtype=torch.int32, device=dev)
    dist = torch.empty((2000, 5), dtype=torch.float32, device=dev)
    hg.set_option("time_kernels", 1)
    for mode, expect_prepass in ((0, False), (1, True)):
        hg.set_option("order_queries", mode)
        assert hg.kernel_times()[2] >= 0                      # reset
        for _ in range(3):
            H.search_batch_device(hg, Qd.data_ptr(), 2000, X.shape[1], 30, 5, ids.data_ptr(), dist.data_ptr())
        s_ms, p_ms, calls = hg.kernel_times()
        assert calls == 3 and s_ms > 0
        assert (p_ms > 0) == expect_prepass
    hg.set_option("time_kernels", 0)
    H.search_batch_device(hg, Qd.data_ptr(), 2000, X.shape[1], 30, 5, ids.data_ptr(), dist.data_ptr())
    torch.cuda.synchronize()
    assert hg.kernel_times()[2] == 0
