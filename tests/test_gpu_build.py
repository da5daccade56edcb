"""GPU graph builder (batched restatement of Ohnsw.build_batch_bigarray, lib/ohnsw.ml:766-857):
structural invariants of the reference's graphs, search parity on the built graph, determinism,
and quality next to the sequential CPU restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import ocaml_hnsw_amd as H
    H.load()
    assert H.device_count() >= 1
    return H


def _uniform(n, d, seed):
    return np.random.default_rng(seed).uniform(-1, 1, size=(n, d)).astype(np.float32)


def _oracle_graph(o, hg):
    up = [(nodes, deg, nbr) for nodes, deg, nbr in hg.upper]
    return o.Graph(hg.n, hg.entry_point, hg.deg0, hg.nbr0, up)


@pytest.fixture(scope="module")
def built(H, oracle):
    X = _uniform(20000, 32, 1)
    hg = H.Ohnsw.build_batch_bigarray(X, 8, 100, seed=7).export()
    return X, hg


def test_structure(H, oracle, built):
    X, hg = built
    M = 8
    n = hg.n
    assert hg.nbr0.shape == (n, 2 * M) and hg.deg0.max() <= 2 * M          # lib/ohnsw.ml:818
    rows = [hg.nbr0[i, :hg.deg0[i]] for i in range(n)]
    sets = [set(r.tolist()) for r in rows]
    assert all(len(s) == len(r) for s, r in zip(sets, rows))                # no duplicate links
    assert all(i not in s for i, s in enumerate(sets))                      # no self links
    assert all(i in sets[j] for i in range(n) for j in sets[i])             # symmetric (:217-225)
    assert (hg.deg0 > 0).mean() > 0.999
    for nodes, deg, nbr in hg.upper:
        assert deg.max() <= M
        slot = {int(v): s for s, v in enumerate(nodes)}
        for s, v in enumerate(nodes):
            for u in nbr[s, :deg[s]]:
                assert int(v) in nbr[slot[int(u)], :deg[slot[int(u)]]]
    # same level law and RNG as the CPU restatement: identical layer membership for one seed
    sp = oracle.Space.l2(X[:3000])
    g = oracle.build_ohnsw(sp, M, 20, seed=7)
    for l, (nodes, _, _) in enumerate(g.upper):
        mine = hg.upper[l][0]
        np.testing.assert_array_equal(nodes, mine[mine < 3000])


def test_search_parity_on_built_graph(H, oracle, built):
    X, hg = built
    Q = _uniform(300, 32, 2)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = _oracle_graph(oracle, hg)
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=64, counters=True)
    oids, odist, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=10, ef=64, ties=oracle.TIES_CANONICAL, counters=True)
    np.testing.assert_array_equal(ids, oids)
    np.testing.assert_array_equal(dist.view(np.uint32), odist.view(np.uint32))
    np.testing.assert_array_equal(nh, onh)


def test_quality_next_to_sequential_builder(H, oracle, built):
    X, hg = built
    Q = _uniform(300, 32, 3)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    gt, _ = oracle.brute_force_knn(sp, Q, 10)
    ids, _ = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=64)
    rec_gpu = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids.tolist(), gt.tolist())])
    g = oracle.build_ohnsw(sp, 8, 100, seed=7)
    oids, _ = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=10, ef=64)
    rec_cpu = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(oids.tolist(), gt.tolist())])
    print("recall@10 ef=64: gpu-built %.3f, sequential cpu-built %.3f; mean deg0 %.1f vs %.1f"
          % (rec_gpu, rec_cpu, hg.deg0.mean(), g.deg0.mean()))
    assert rec_gpu >= rec_cpu - 0.03


def test_deterministic(H, built):
    X, hg = built
    again = H.Ohnsw.build_batch_bigarray(X, 8, 100, seed=7).export()
    np.testing.assert_array_equal(hg.nbr0, again.nbr0)
    for a, b in zip(hg.upper, again.upper):
        np.testing.assert_array_equal(a[2], b[2])
    assert hg.entry_point == again.entry_point


@pytest.mark.parametrize("n,d,M,efc,metric", [(1, 8, 4, 10, 0), (2, 8, 4, 10, 0), (50, 5, 4, 16, 0),
                                             (3000, 100, 16, 64, 1), (1500, 200, 6, 40, 0)])
def test_small_and_odd_shapes(H, oracle, n, d, M, efc, metric):
    X = _uniform(n, d, 4)
    if metric:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    hg = H.Ohnsw.build_batch_bigarray(X, M, efc, seed=1, metric=metric).export()
    sets = [set(hg.nbr0[i, :hg.deg0[i]].tolist()) for i in range(n)]
    assert all(i in sets[j] for i in range(n) for j in sets[i])
    sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
    g = _oracle_graph(oracle, hg)
    k = min(5, n)
    ids, dist = H.Ohnsw.knn_batch_bigarray(hg, k, X[:20], ef=max(k, 20))
    oids, odist = oracle.Ohnsw.knn_batch_bigarray(g, sp, X[:20], k=k, ef=max(k, 20), ties=oracle.TIES_CANONICAL)
    np.testing.assert_array_equal(ids, oids)
    np.testing.assert_array_equal(dist.view(np.uint32), odist.view(np.uint32))


@pytest.mark.parametrize("metric,d,M,nc", [(0, 24, 8, 60), (0, 128, 32, 200), (1, 100, 16, 40), (0, 5, 4, 3)])
def test_select_neighbours_operator_matches_oracle(H, oracle, metric, d, M, nc):
    """hnsw_select_neighbours_batch vs Ohnsw.select_neighbours (lib/ohnsw.ml:647-663) of the oracle,
    canonical tie order, on random candidate sets; plus the functor path's keep-all shortcut."""
    rng = np.random.default_rng(d)
    X = rng.integers(0, 30, size=(3000, d)).astype(np.float32) if metric == 0 else _uniform(3000, d, 5)
    if metric:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    hg = H.Hgraph(X, np.zeros(3000, np.int32), np.full((3000, 2), -1, np.int32), entry_point=0, metric=metric)
    sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
    T = X[rng.integers(0, 3000, 40)] + (0 if metric else rng.integers(0, 2, size=(40, d))).astype(np.float32) if not metric else X[rng.integers(0, 3000, 40)]
    cands = [rng.choice(3000, size=rng.integers(1, nc + 1), replace=False).tolist() for _ in range(40)]
    got = H.Ohnsw.select_neighbours(hg, T, cands, M)
    for t, c, g in zip(T, cands, got):
        want = oracle.Ohnsw.select_neighbours(sp, c, t, M, ties=oracle.TIES_CANONICAL)
        assert g == want
    got2 = H.Ohnsw.select_neighbours(hg, T, [c[:M] for c in cands], M, keep_all_if_few=True)
    for c, g in zip(cands, got2):
        assert sorted(g) == sorted(c[:M])


def test_save_load_stats_roundtrip(H, oracle, built, tmp_path):
    X, hg = built
    p = tmp_path / "index.hnsw"
    hg.save(p)
    again = H.Hgraph.load(p)
    Q = _uniform(200, 32, 9)
    a = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=50)
    b = H.Ohnsw.knn_batch_bigarray(again, 10, Q, ef=50)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    st = again.stats()                                  # Hgraph.Stats (lib/hnsw.ml:353-375)
    assert st["num_nodes"] == hg.n and st["layer_sizes"][0] == hg.n
    assert st["layer_connectivity"][0]["max"] == int(hg.deg0.max())
    assert st["layer_connectivity"][0]["mean"] == pytest.approx(float(hg.deg0.mean()))
    assert st["layer_sizes"][1] == len(hg.upper[0][0])
    (tmp_path / "bad.hnsw").write_bytes(b"not an index")
    with pytest.raises(H.InvalidArgument, match="not a flattened hnsw index"):
        H.Hgraph.load(tmp_path / "bad.hnsw")


def _same_stats(got, want):
    assert got["num_nodes"] == want["num_nodes"]
    assert got["layer_sizes"] == want["layer_sizes"]
    assert sorted(got["layer_connectivity"]) == sorted(want["layer_connectivity"])
    for layer, w in want["layer_connectivity"].items():
        g = got["layer_connectivity"][layer]
        assert (g["min"], g["max"]) == (w["min"], w["max"]), layer
        assert g["mean"] == w["mean"] or (np.isnan(g["mean"]) and np.isnan(w["mean"])), layer    # same sum / count in double
        assert g["isolated"] == w["isolated"], layer                                             # ids, the reference's list order


def test_stats_equal_the_reference_fold_on_every_layer(H, oracle, built):
    """Hgraph.Stats.compute (lib/hnsw.ml:353-375): layer sizes, min / max / mean and the isolated node LIST of every layer,
    device (hnsw_index_layer_stats + hnsw_index_layer_isolated) against the oracle's restatement of the fold."""
    # (1) a hand-made three-layer graph with isolated nodes on every layer, 1-based ids as Hnsw.Ba has them
    n = 12
    X = _uniform(n, 8, 3)
    adj0 = {1: [2, 3], 2: [1], 3: [1], 4: [], 5: [6], 6: [5], 7: [], 8: [9], 9: [8], 10: [], 11: [12], 12: [11]}
    deg0 = np.array([len(adj0[i + 1]) for i in range(n)], np.int32)
    nbr0 = np.full((n, 4), 0, np.int32)                   # padding below id_base
    for i in range(n):
        nbr0[i, :deg0[i]] = adj0[i + 1]
    up1 = (np.array([2, 5, 7, 9, 12], np.int64), np.array([1, 0, 0, 1, 0], np.int32),
           np.array([[9, 0], [0, 0], [0, 0], [2, 0], [0, 0]], np.int32))
    up2 = (np.array([7, 9], np.int64), np.array([0, 0], np.int32), np.zeros((2, 2), np.int32))
    hg = H.Hgraph(X, deg0, nbr0, [up1, up2], entry_point=9, id_base=1, max_degree=2).to_device()
    got = hg.stats()
    assert got["layer_connectivity"][0] == {"min": 0, "max": 2, "mean": 10 / 12, "isolated": [10, 7, 4]}
    assert got["layer_connectivity"][1] == {"min": 0, "max": 1, "mean": 2 / 5, "isolated": [12, 7, 5]}
    assert got["layer_connectivity"][2] == {"min": 0, "max": 0, "mean": 0.0, "isolated": [9, 7]}
    assert got["layer_sizes"] == {0: 12, 1: 5, 2: 2}
    g = oracle.Graph(n, 8, deg0, np.where(nbr0 > 0, nbr0 - 1, -1),
                     [(u[0] - 1, u[1], np.where(u[2] > 0, u[2] - 1, -1)) for u in (up1, up2)])
    want = oracle.Stats.compute(g)
    for lc in want["layer_connectivity"].values():
        lc["isolated"] = [i + 1 for i in lc["isolated"]]  # the oracle is 0-based
    _same_stats(got, want)
    hg.release()
    # (2) a graph built on the device (uniform data: a few nodes lose every link when their neighbours shrink them away)
    X, hb = built
    hb.export()
    _same_stats(hb.stats(), oracle.Stats.compute(oracle.Graph(hb.n, hb.entry_point, hb.deg0, hb.nbr0, hb.upper)))
    # (3) tie-heavy integer data, M = 4: many isolated nodes on layer 0
    Xt = np.random.default_rng(4).integers(0, 3, size=(3000, 6)).astype(np.float32)
    ht = H.Ohnsw.build_batch_bigarray(Xt, 4, 20, seed=2)
    ht.export()
    want = oracle.Stats.compute(oracle.Graph(ht.n, ht.entry_point, ht.deg0, ht.nbr0, ht.upper))
    _same_stats(ht.stats(), want)
    print("isolated nodes per layer (tie-heavy set):", [len(v["isolated"]) for v in want["layer_connectivity"].values()])
    ht.release()


def test_select_neighbours_functor_variant(H, oracle):
    """Hnsw_algo.SelectNeighbours.select_neighbours (lib/hnsw_algo.ml:572-609): the keep-all shortcut
    and ~do_not_isolate:true, against the oracle's restatement (canonical order)."""
    rng = np.random.default_rng(5)
    X = rng.integers(0, 20, size=(2000, 12)).astype(np.float32)
    hg = H.Hgraph(X, np.zeros(2000, np.int32), np.full((2000, 2), -1, np.int32), entry_point=0)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    M = 8
    for trial in range(30):
        base = int(rng.integers(0, 2000))
        nc = int(rng.integers(1, 40))
        cand = rng.choice(2000, size=nc, replace=False).tolist()
        degs = rng.integers(0, 6, size=nc).tolist()
        if sum(d <= 1 for d in degs) >= M:
            continue
        cd = [float(np.sqrt(np.float64(np.float32(oracle.l2sq_tree16(X[c], X[base]))))) for c in cand]
        want = oracle.Functor.select_neighbours(sp, cand, cd, M, cand_degree=degs, do_not_isolate=True,
                                                ties=oracle.TIES_CANONICAL)
        got = H.Ohnsw.select_neighbours(hg, X[base][None, :], [cand], M, keep_all_if_few=True, degrees=[degs])[0]
        assert sorted(got) == sorted(want), (trial, got, want)
        if nc > M:
            assert got == want            # same selection order when the heuristic runs


def test_randomized_builds(H, oracle):
    """Random builder configurations: symmetric links, degree caps, no self links / duplicates, and
    search parity on the produced graph, every time."""
    rng = np.random.default_rng(77)
    for trial in range(24):
        n = int(rng.integers(1, 2500))
        d = int(rng.choice([1, 3, 8, 33, 64, 100, 130]))
        M = int(rng.choice([2, 4, 8, 16, 32]))
        efc = int(rng.choice([1, 5, 33, 64, 65, 200, 300]))
        metric = int(rng.integers(0, 2))
        X = rng.integers(0, int(rng.choice([3, 1000])), size=(n, d)).astype(np.float32)
        if metric:
            X = X + rng.uniform(0.01, 0.5, size=X.shape).astype(np.float32)
            X /= np.linalg.norm(X, axis=1, keepdims=True)
        hg = H.Ohnsw.build_batch_bigarray(X, M, efc, seed=trial, metric=metric, max_batch=int(rng.choice([0, 64, 1000])),
                                          batch_div=int(rng.choice([0, 2, 50]))).export()
        ctx = dict(trial=trial, n=n, d=d, M=M, efc=efc, metric=metric)
        assert hg.deg0.max(initial=0) <= 2 * M, ctx
        sets = []
        for i in range(n):
            row = hg.nbr0[i, :hg.deg0[i]].tolist()
            assert len(set(row)) == len(row) and i not in row, ctx
            sets.append(set(row))
        assert all(i in sets[j] for i in range(n) for j in sets[i]), ctx
        for nodes, deg, nbr in hg.upper:
            assert deg.max(initial=0) <= M, ctx
        sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
        g = _oracle_graph(oracle, hg)
        k = min(4, n)
        Q = X[rng.integers(0, n, 8)]
        ids, dist = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=max(k, 30))
        oi, od = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=k, ef=max(k, 30), ties=oracle.TIES_CANONICAL)
        assert np.array_equal(ids, oi) and np.array_equal(dist.view(np.uint32), od.view(np.uint32)), ctx


@pytest.mark.parametrize("kind,metric,n,d,M,efc", [
    ("uniform", 0, 3000, 16, 6, 40),        # generic position: no exact ties
    ("levels", 0, 2500, 6, 8, 60),          # few distinct coordinates: exact distance ties everywhere
    ("unit", 1, 3000, 24, 8, 50),           # inner product (distance 1 - <a,b>, negative values possible)
    ("sift", 0, 2000, 128, 16, 100),        # the C2 shape: integer-valued, d = 128, M = 16
    ("sift", 0, 2500, 32, 32, 80),          # M = 32: layer-0 rows of 64 entries, a dropped node's list is [q] + a full row = 65
    ("levels", 0, 2500, 6, 32, 80),         # the same with exact ties everywhere
])
def test_sequential_build_equals_the_reference_insert_link_for_link(H, oracle, kind, metric, n, d, M, efc):
    """max_batch = 1: every node is inserted on its own, and the link step runs neighbour by neighbour as
    Ohnsw.insert does (lib/ohnsw.ml:766-837: set_connections_for_new_node, then the shrink of every
    over-full neighbour in list order, Neighbours.remove reversing what it keeps).  The device-built graph
    must then equal the oracle's restatement of that function -- entry point, max layer, layer membership,
    and every adjacency list in ITERATION ORDER -- which pins the construction search, select_neighbours,
    the back-link and the shrink kernels against the reference's own control flow."""
    rng = np.random.default_rng(11)
    if kind == "uniform":
        X = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    elif kind == "levels":
        X = rng.integers(0, 4, size=(n, d)).astype(np.float32)
    elif kind == "unit":
        X = rng.normal(size=(n, d))
        X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
    else:
        centres = rng.integers(20, 200, size=(32, d))
        X = np.clip(np.rint(centres[rng.integers(0, 32, n)] + rng.normal(0, 25, size=(n, d))), 0, 218).astype(np.float32)
    sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
    want = oracle.build_ohnsw(sp, M, efc, seed=5, ties=oracle.TIES_CANONICAL)
    hg = H.Ohnsw.build_batch_bigarray(X, M, efc, seed=5, metric=metric, max_batch=1).export()
    assert hg.entry_point == want.entry_point
    assert hg.max_layer == want.max_layer
    np.testing.assert_array_equal(hg.deg0, want.deg0)
    np.testing.assert_array_equal(hg.nbr0, want.nbr0)                      # rows in iteration order, -1 padded
    assert len(hg.upper) == len(want.upper)
    for (nodes, deg, nbr), (wn, wd, wb) in zip(hg.upper, want.upper):
        np.testing.assert_array_equal(nodes, wn)
        np.testing.assert_array_equal(deg, wd)
        np.testing.assert_array_equal(nbr, wb)


def test_default_build_with_64_entry_rows_is_symmetric(H, oracle):
    """M = 32 (C3 / C5's graph shape: layer-0 rows are 64 entries wide, the widest the tables allow), default batching:
    single-node batches (a node that raises max_layer, a one-node tail) go through the sequential link kernel, whose
    survivor lists can hold 65 entries.  Graph.Test.invariant (lib/ohnsw.ml:217-225): every link has its back-link."""
    rng = np.random.default_rng(17)
    n, d, M = 5001, 24, 32
    centres = rng.integers(20, 200, size=(16, d))
    X = np.clip(np.rint(centres[rng.integers(0, 16, n)] + rng.normal(0, 20, size=(n, d))), 0, 218).astype(np.float32)
    hg = H.Ohnsw.build_batch_bigarray(X, M, 80, seed=2).export()
    links = set()
    for i in range(n):
        for v in hg.nbr0[i, :hg.deg0[i]]:
            links.add((i, int(v)))
    missing = [(a, b) for (a, b) in links if (b, a) not in links]
    assert not missing, "asymmetric layer-0 links: %s" % missing[:5]
    assert hg.deg0.max() <= 2 * M
    for nodes, deg, nbr in hg.upper:
        pos = {int(v): j for j, v in enumerate(nodes)}
        for j, v in enumerate(nodes):
            for u in nbr[j, :deg[j]]:
                ju = pos[int(u)]
                assert int(v) in nbr[ju, :deg[ju]].tolist(), "asymmetric upper-layer link %d -> %d" % (v, u)


def test_hub_with_more_than_64_new_links_in_one_batch_stays_symmetric(H, oracle):
    """One node close to everything collects far more than 64 back-links in a single batch: they are merged
    64 at a time (nothing is refused or dropped), and the graph keeps the reference's invariants
    (Graph.Test.invariant, lib/ohnsw.ml:217-225: symmetric links; degree caps)."""
    rng = np.random.default_rng(3)
    n, d, M = 6000, 32, 8
    X = rng.normal(size=(n, d)).astype(np.float32) * 10
    X[0] = 0.0
    X[1:] /= np.linalg.norm(X[1:], axis=1, keepdims=True)                  # a shell around the hub at the origin: in 32
    # dimensions two shell points are about sqrt(2) apart, so the hub (distance 1) is every node's nearest
    # neighbour and every insertion of a batch links to it
    hg = H.Ohnsw.build_batch_bigarray(X, M, 60, seed=2, max_batch=4096, batch_div=1).export()
    sets = [set(hg.nbr0[i, :hg.deg0[i]].tolist()) for i in range(n)]
    assert all(len(s) == hg.deg0[i] for i, s in enumerate(sets))            # no duplicates
    assert all(i in sets[j] for i in range(n) for j in sets[i])             # symmetric
    assert hg.deg0.max() <= 2 * M
    assert hg.deg0[0] == 2 * M                                              # the hub's row is full, and was re-selected many times
