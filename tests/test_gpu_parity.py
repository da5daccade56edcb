"""GPU parity: the HIP search path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bar: ids bit-exact, distances bit-exact under the oracle's OG_TREE16 arithmetic
(the kernel's summation order) and within 1e-5 relative of the reference-style sequential fp32
arithmetic (north star tolerance)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # north-star tolerance on L2 distances


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


def _dataset(kind, n, d, seed):
    rng = np.random.default_rng(seed)
    if kind == "sift":      # clustered integers 0..218 stored as fp32
        centres = rng.integers(20, 200, size=(64, d))
        X = centres[rng.integers(0, 64, n)] + rng.normal(0, 25, size=(n, d))
        return np.clip(np.rint(X), 0, 218).astype(np.float32)
    if kind == "uniform":   # Lacaml Mat.random range (benchmark/dataset.ml:48), C1
        return rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
    if kind == "unit":      # GloVe/DEEP-like: N(0,1) normalised
        X = rng.normal(size=(n, d))
        return (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
    raise ValueError(kind)


# ---- the reference's own known-answer tests, through the GPU ------------------------------------
G = load_golden("ohnsw_inline_tests.json")


def _kat_graph(o, kind, n):
    return o.Graph.from_lists([[] for _ in range(n)]) if kind == "isolated" else o.Graph.ring(n)


@pytest.mark.parametrize("case", G["search_k"], ids=lambda c: c["ref"])
def test_reference_search_k_kats(H, oracle, case):
    """lib/ohnsw.ml:593-644: with d = 1 the L2 distance IS |a-b|; max_layer = 0 and
    entry_point = start make knn == search_k from that start node."""
    vals = np.array(case["values"], np.float32)[:, None]
    g = _kat_graph(oracle, case["graph"], len(vals))
    g.entry_point = case["start"]
    hg = _hgraph(H, vals, g, M=2)
    k = min(case["k"], 64)
    got = H.Ohnsw.knn(hg, k, np.array([case["target"]], np.float32))
    assert [n for n, _ in got] == [n for n, _ in case["expect"]]
    for (_, d), (_, e) in zip(got, case["expect"]):
        assert d == pytest.approx(e, rel=1e-6, abs=1e-6)


@pytest.mark.parametrize("case", G["search_one"], ids=lambda c: c["ref"])
def test_reference_search_one_kats(H, oracle, case):
    """lib/ohnsw.ml:514-534: put the test graph on layer 1 over an isolated layer 0; the descent
    (search_one) result is then the single layer-0 result."""
    vals = np.array(case["values"], np.float32)[:, None]
    n = len(vals)
    g1 = _kat_graph(oracle, case["graph"], n)
    nbr = np.full((n, 2), -1, np.int32)
    nbr[:, :g1.nbr0.shape[1]] = g1.nbr0[:, :2]
    hg = H.Hgraph(vals, np.zeros(n, np.int32), np.full((n, 4), -1, np.int32),
                  [(np.arange(n), g1.deg0, nbr)], entry_point=case["start"], max_degree=2)
    got = H.Ohnsw.knn(hg, 1, np.array([case["target"]], np.float32))
    assert [n_ for n_, _ in got] == [case["expect"]]


# ---- seeded parity against the oracle -------------------------------------------------------------
CASES = [
    # name, data, n, d, metric, M, efC, ef, k, nq
    ("sift_like_c2_shape", "sift", 12000, 128, 0, 16, 100, 128, 10, 300),
    ("uniform_c1", "uniform", 10000, 32, 0, 8, 100, 32, 10, 300),
    ("glove_like_ip_c3_shape", "unit", 6000, 100, 1, 32, 80, 256, 100, 100),
    ("deep_like_c5_shape", "unit", 6000, 96, 0, 32, 80, 512, 10, 100),
    ("mnist_like_d784", "uniform", 1500, 784, 0, 15, 60, 10, 10, 40),   # benchmark.ml:118,126 shape
    ("ragged_d3", "uniform", 2000, 3, 0, 6, 40, 20, 5, 100),
    ("ef1", "uniform", 3000, 16, 0, 8, 40, 1, 1, 100),
    ("ef_1000", "uniform", 3000, 20, 0, 8, 40, 1000, 50, 20),
]


@pytest.fixture(scope="module", params=CASES, ids=lambda c: c[0])
def case(request, oracle, H):
    name, kind, n, d, metric, M, efc, ef, k, nq = request.param
    X = _dataset(kind, n, d, 11)
    Q = _dataset(kind, nq, d, 12)
    mk = oracle.Space.l2 if metric == 0 else oracle.Space.ip
    sp = mk(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, M, efc, seed=5)
    hg = _hgraph(H, X, g, metric=metric, M=M)
    return dict(name=name, X=X, Q=Q, sp=sp, g=g, hg=hg, ef=ef, k=k, metric=metric, mk=mk)


def test_bit_parity_with_oracle(H, oracle, case):
    c = case
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(c["hg"], c["k"], c["Q"], ef=c["ef"], counters=True)
    oids, odist, ond, onh = oracle.Ohnsw.knn_batch_bigarray(
        c["g"], c["sp"], c["Q"], k=c["k"], ef=c["ef"], ties=oracle.TIES_CANONICAL, counters=True)
    np.testing.assert_array_equal(ids, oids)
    np.testing.assert_array_equal(dist.view(np.uint32), odist.view(np.uint32))
    # the same candidates were expanded; the lossy visited cache may only ADD evaluations
    np.testing.assert_array_equal(nh, onh)
    assert (nd.astype(np.int64) >= 1).all()


def test_within_tolerance_of_reference_arithmetic(H, oracle, case):
    """Reference-style arithmetic (sequential fp32 sum, sqrt in double; lib/ohnsw.ml:899): near-ties
    may swap, so compare rank-wise distances (1e-5 relative) and demand id equality on the queries
    whose result has no near-tie."""
    c = case
    ids, dist = H.Ohnsw.knn_batch_bigarray(c["hg"], c["k"], c["Q"], ef=c["ef"])
    sp = c["mk"](c["X"], arith=oracle.SEQ_F32)
    oids, odist = oracle.Ohnsw.knn_batch_bigarray(c["g"], sp, c["Q"], k=c["k"], ef=c["ef"], ties=oracle.TIES_HEAP)
    ok = np.isfinite(odist)
    assert np.array_equal(ok, np.isfinite(dist))
    scale = np.maximum(np.abs(odist), 1e-3 if c["metric"] else 1e-30)
    rel = np.where(ok, np.abs(dist - odist) / scale, 0.0)
    # the contract itself (north star: "L2 distances within 1e-5 relative"): wherever both sides report the SAME node
    # at a rank, its two distances differ by at most 1e-5 relative -- every such entry, no slack
    same = ok & (ids == oids)
    worst_same = float(rel[same].max()) if same.any() else 0.0
    print("\n[tolerance] %s: max relative distance difference on %d same-node entries: %.3g (bound %.0e); "
          "rank-wise over all %d entries: max %.3g, share above the bound %.2g"
          % (c.get("name", "?"), int(same.sum()), worst_same, REL_TOL, int(ok.sum()), float(rel.max()),
             float((rel[ok] > REL_TOL).mean()) if ok.any() else 0.0))
    assert worst_same <= REL_TOL
    # rank-wise (different nodes may sit at a rank after a near-tie swap or a diverged walk): the same bound on 99.9 %
    close = ~ok | (rel <= REL_TOL)
    assert close[ok].mean() > 0.999
    # ids: a differing position is legitimate only as a swap inside the tolerance band
    differs = ids != oids
    unexplained = differs & ~close
    assert unexplained.any(axis=1).mean() <= 0.02
    near_tie = np.zeros_like(differs)
    near_tie[:, 1:] |= np.abs(odist[:, 1:] - odist[:, :-1]) <= 8 * REL_TOL * scale[:, 1:]
    near_tie[:, :-1] |= near_tie[:, 1:].copy()
    near_tie[:, -1] = True   # the k-th boundary can trade with rank k+1
    assert (differs & ~near_tie).any(axis=1).mean() <= 0.02


def test_visited_cache_size_never_changes_results(H, oracle, case):
    c = case
    base = H.Ohnsw.knn_batch_bigarray(c["hg"], c["k"], c["Q"], ef=c["ef"], counters=True)
    try:
        for bits in (4, 9, 13):
            c["hg"].set_option("vt_bits", bits)
            got = H.Ohnsw.knn_batch_bigarray(c["hg"], c["k"], c["Q"], ef=c["ef"], counters=True)
            np.testing.assert_array_equal(got[0], base[0])
            np.testing.assert_array_equal(got[1].view(np.uint32), base[1].view(np.uint32))
            np.testing.assert_array_equal(got[3], base[3])
    finally:
        c["hg"].set_option("vt_bits", 0)


def test_recall_property(H, oracle, case):
    """Domain property: ascending distances, no duplicate ids, sane recall vs brute force."""
    c = case
    ids, dist = H.Ohnsw.knn_batch_bigarray(c["hg"], c["k"], c["Q"], ef=c["ef"])
    fin = np.where(np.isfinite(dist), dist, np.inf)
    assert (np.diff(fin, axis=1) >= 0).all()
    for row in ids[:50]:
        r = row[row >= 0]
        assert len(set(r.tolist())) == len(r)
    gt, _ = oracle.brute_force_knn(c["sp"], c["Q"][:50], c["k"])
    rec = np.mean([len(set(a) & set(b)) / c["k"] for a, b in zip(ids[:50].tolist(), gt.tolist())])
    assert rec > 0.25  # random high-d data is hard (the reference's own note: benchmark.ml:145)


# ---- ties, duplicates: the overflow stack of entries evicted while tied with max(W) ----------------
@pytest.mark.parametrize("levels", [3, 8, 40])
def test_exact_ties_and_duplicates(H, oracle, levels):
    rng = np.random.default_rng(levels)
    X = rng.integers(0, levels, size=(4000, 6)).astype(np.float32)   # few distinct points, many duplicates
    Q = rng.integers(0, levels, size=(200, 6)).astype(np.float32)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 8, 60, seed=2)
    hg = _hgraph(H, X, g, M=8)
    for ef, k in ((16, 16), (64, 10), (200, 100)):
        ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
        oids, odist, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=k, ef=ef, ties=oracle.TIES_CANONICAL, counters=True)
        np.testing.assert_array_equal(dist.view(np.uint32), odist.view(np.uint32))
        np.testing.assert_array_equal(ids, oids)
        np.testing.assert_array_equal(nh, onh)


# ---- API surface and error behaviour -------------------------------------------------------------
@pytest.fixture(scope="module")
def tiny(H, oracle):
    X = _dataset("uniform", 500, 10, 3)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 6, 30, seed=9)
    return X, sp, g


def test_ba_api_is_one_based_and_inf_filled(H, oracle, tiny):
    X, sp, g = tiny
    hg1 = _hgraph(H, X, g, id_base=1, M=6)
    Q = X[:20] + 0.01
    d = H.Ba.knn_batch(hg1, Q, num_neighbours_search=40, num_neighbours=7)
    od = oracle.Functor.knn_batch(g, sp, Q, 40, 7, ties=oracle.TIES_CANONICAL)
    np.testing.assert_array_equal(d.view(np.uint32), od.view(np.uint32))
    one = H.Ba.knn(hg1, Q[0], 40, 7)
    ref = oracle.Functor.knn(g, sp, Q[0], 40, 7, ties=oracle.TIES_CANONICAL)
    assert [n for n, _ in one] == [n + 1 for n, _ in ref]       # node ids are matrix columns (lib/hnsw.ml:325)


def test_fewer_than_k_reachable(H, oracle):
    X = _dataset("uniform", 3, 4, 5)
    hg = H.Hgraph(X, [1, 1, 0], [[1, -1], [0, -1], [-1, -1]], entry_point=0, max_degree=1)
    ids, dist = H.Ohnsw.knn_batch_bigarray(hg, 4, X[:1])
    assert ids[0, 2:].tolist() == [-1, -1] and np.isnan(dist[0, 2:]).all()   # lib/ohnsw.ml:880-881
    hg1 = H.Hgraph(X, [1, 1, 0], [[2, 0], [1, 0], [0, 0]], entry_point=1, id_base=1, max_degree=1)
    d = H.Ba.knn_batch(hg1, X[:1], 4, 4)
    assert np.isinf(d[0, 2:]).all()                                        # lib/hnsw.ml:771


def test_empty_hgraph_raises_invalid_argument(H):
    X = _dataset("uniform", 3, 4, 5)
    hg = H.Hgraph(X, [0, 0, 0], np.full((3, 2), -1), entry_point=None)
    with pytest.raises(H.InvalidArgument, match="knn: empty hgraph"):       # lib/ohnsw.ml:862
        H.Ohnsw.knn(hg, 2, X[0])
    hg0 = H.Hgraph(np.zeros((0, 4), np.float32), np.zeros(0, np.int32), np.zeros((0, 2), np.int32))
    with pytest.raises(H.InvalidArgument, match="knn: empty hgraph"):
        H.Ohnsw.knn_batch_bigarray(hg0, 2, X)


def test_bad_arguments(H, tiny):
    X, sp, g = tiny
    hg = _hgraph(H, X, g, M=6)
    with pytest.raises(H.InvalidArgument, match="k=5 > ef=3"):
        H.Ohnsw.knn_batch_bigarray(hg, 5, X[:2], ef=3)
    with pytest.raises(H.InvalidArgument):
        H.Ohnsw.knn_batch_bigarray(hg, 0, X[:2])
    ids, dist = H.Ohnsw.knn_batch_bigarray(hg, 3, X[:0])                    # empty batch
    assert ids.shape == (0, 3)
    bad = H.Hgraph(X[:3], [3, 0, 0], [[1, 2, 0], [-1, -1, -1], [-1, -1, -1]], entry_point=0)
    bad.deg0[0] = 4
    with pytest.raises(H.InvalidArgument, match="max_degree0"):            # never truncate
        bad.to_device()
    oob = H.Hgraph(X[:3], [1, 0, 0], [[7], [-1], [-1]], entry_point=0)
    with pytest.raises(H.InvalidArgument, match="out of range"):
        oob.to_device()


def test_single_node(H):
    X = np.array([[1.0, 2.0, 2.0]], np.float32)
    hg = H.Hgraph(X, [0], [[-1, -1]], entry_point=0)
    assert H.Ohnsw.knn(hg, 1, np.zeros(3, np.float32)) == [(0, 3.0)]


def test_distance_batch_matches_oracle(H, oracle, tiny):
    X, sp, g = tiny
    hg = _hgraph(H, X, g, M=6)
    rng = np.random.default_rng(1)
    Q = _dataset("uniform", 30, 10, 8)
    ids = rng.integers(0, 500, size=(30, 37)).astype(np.int32)
    got = H.Ohnsw.distance_l2(hg, Q, ids)
    want = np.array([[np.float32(np.sqrt(np.float64(np.float32(oracle.l2sq_tree16(X[j], q))))) for j in row]
                     for q, row in zip(Q, ids)], np.float32)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    exact = np.sqrt(((X[ids].astype(np.float64) - Q[:, None, :]) ** 2).sum(-1))
    assert np.max(np.abs(got - exact) / exact) < REL_TOL


def test_large_batch_property(H, oracle, tiny):
    """Size-independent property at a batch far larger than the oracle is run on: every query equal
    to a database vector must return that vector (or a duplicate) at distance 0 first, and the
    result for a query is independent of the batch it travels in."""
    X, sp, g = tiny
    hg = _hgraph(H, X, g, M=6)
    reps = np.tile(X, (40, 1))                      # 20 000 queries
    ids, dist = H.Ohnsw.knn_batch_bigarray(hg, 5, reps, ef=40)
    assert (dist[:, 0] == 0).mean() > 0.99
    np.testing.assert_array_equal(ids[:500], ids[500:1000])
    np.testing.assert_array_equal(ids[:500], ids[-500:])


def test_cpp_front_end_mirror(H):
    """host/hnsw_front.hpp (C++ mirror of Ohnsw / Hnsw.Ba over the C ABI) against the reference's
    search_k known answers; built by __graft_entry__.build()."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "cpp", "test_front")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "front-end ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("case", G["select_neighbours"], ids=lambda c: c["ref"])
def test_reference_select_neighbours_kats(H, case):
    """lib/ohnsw.ml:665-764 through the device operator (d = 1: L2 is |a-b|)."""
    vals = np.array(case["values"], np.float32)[:, None]
    n = len(vals)
    hg = H.Hgraph(vals, np.zeros(n, np.int32), np.full((n, 2), -1, np.int32), entry_point=0)
    M = min(case["M"], 64)
    got = H.Ohnsw.select_neighbours(hg, np.array([[case["target"]]], np.float32), [case["candidates"]], M)
    assert sorted(got[0]) == case["expect"]


def test_tie_overflow_beyond_lds_stack(H, oracle):
    """More than 64 entries evicted while tied with max(W) and still unexpanded (lib/ohnsw.ml:568 keeps
    them poppable): the asynchronous entry point flags the query, the host entry point re-runs it with a
    global slab and must equal the oracle -- including a node reachable only through such an entry."""
    import torch
    # 1-D positions, query at 0: E far, 127 identical "shell" points at 10, a chain approaching, Z close
    n = 229
    pos = np.zeros(n, np.float32)
    pos[0] = 20.0
    pos[1:128] = 10.0
    pos[128:228] = 9.0 - 0.01 * np.arange(100)
    pos[228] = 0.1
    rows = [[] for _ in range(n)]
    rows[0] = [1] + list(range(2, 65))            # E -> H + 63 shells
    rows[1] = list(range(65, 128)) + [128]        # H -> 63 shells + head of the chain
    for i in range(99):
        rows[128 + i] = [129 + i]
    rows[40] = [228]                              # a shell evicted late is the only way to Z
    deg0 = np.array([len(r) for r in rows], np.int32)
    nbr0 = np.full((n, 64), -1, np.int32)
    for i, r in enumerate(rows):
        nbr0[i, :len(r)] = r
    X = pos[:, None]
    g = oracle.Graph(n, 0, deg0, nbr0)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    hg = H.Hgraph(X, deg0, nbr0, entry_point=0, max_degree=32)
    Q = np.zeros((3, 1), np.float32)
    want = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=10, ef=128, ties=oracle.TIES_CANONICAL, counters=True)
    assert 228 in want[0][0]                       # the oracle does reach Z
    got = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=128, counters=True)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    np.testing.assert_array_equal(got[3], want[3])
    # the same through page-locked matrices: the fallback's re-run writes the repaired rows into the caller's memory in place
    Qp = H.host_empty(Q.shape, np.float32)
    Qp[:] = Q
    oi, od = H.host_empty((3, 10), np.int32), H.host_empty((3, 10), np.float32)
    got = H.Ohnsw.knn_batch_bigarray(hg, 10, Qp, ef=128, counters=True, out=(oi, od))
    np.testing.assert_array_equal(oi, want[0])
    np.testing.assert_array_equal(od.view(np.uint32), want[1].view(np.uint32))
    np.testing.assert_array_equal(got[3], want[3])
    dev = torch.device("cuda", 0)
    Qd = torch.from_numpy(Q).to(dev)
    ids = torch.empty((3, 10), dtype=torch.int32, device=dev)
    dd = torch.empty((3, 10), dtype=torch.float32, device=dev)
    st = torch.zeros(3, dtype=torch.int32, device=dev)
    H.search_batch_device(hg, Qd.data_ptr(), 3, 1, 128, 10, ids.data_ptr(), dd.data_ptr(), 0, 0, st.data_ptr(), 0)
    torch.cuda.synchronize()
    assert ((st.cpu().numpy() & 1) == 1).all()     # flagged: more than 64 tied evicted entries
    assert not np.array_equal(ids.cpu().numpy(), want[0])      # ... and incomplete: Z is missing
    # opt-in exact mode of the device entry point: the flagged queries are listed and searched again ON THE DEVICE, on the
    # caller's stream (option device_fallback_slab_bytes; a flagged query needs 4 n bytes of slab)
    with pytest.raises(H.InvalidArgument, match="holds no query"):
        hg.set_option("device_fallback_slab_bytes", 4 * n - 1)
    nd_t = torch.zeros(3, dtype=torch.int32, device=dev)
    nh_t = torch.zeros(3, dtype=torch.int32, device=dev)
    for room, repaired in ((4 * n * 8, 3), (4 * n * 2, 2)):              # room for all three; for two of them
        hg.set_option("device_fallback_slab_bytes", room)
        st.zero_()
        H.search_batch_device(hg, Qd.data_ptr(), 3, 1, 128, 10, ids.data_ptr(), dd.data_ptr(), nd_t.data_ptr(), nh_t.data_ptr(), st.data_ptr(), 0)
        torch.cuda.synchronize()
        ok = (st.cpu().numpy() & 1) == 0
        assert ok.sum() == repaired                                       # the ones the slab had no room for keep their flag
        np.testing.assert_array_equal(ids.cpu().numpy()[ok], want[0][ok])
        np.testing.assert_array_equal(dd.cpu().numpy()[ok].view(np.uint32), want[1][ok].view(np.uint32))
        np.testing.assert_array_equal(nh_t.cpu().numpy()[ok], want[3][ok])
    hg.set_option("device_fallback_slab_bytes", 0)                        # freed: flags only again
    st.zero_()
    H.search_batch_device(hg, Qd.data_ptr(), 3, 1, 128, 10, ids.data_ptr(), dd.data_ptr(), 0, 0, st.data_ptr(), 0)
    torch.cuda.synchronize()
    assert ((st.cpu().numpy() & 1) == 1).all()


def test_host_queries_device_results(H, oracle):
    """hnsw_search_batch_h2d: queries from a host matrix (registered: read by the device directly; pageable: staged), results
    in device buffers on the caller's stream -- the same bits as the all-host and the all-device call, for a batch large
    enough to be ordered (the pre-pass then leaves the device copy of the queries) and for a small one."""
    import torch
    dev = torch.device("cuda", 0)
    X = _dataset("sift", 20000, 128, 3)
    hg = H.Ohnsw.build_batch_bigarray(X, 16, 80, seed=2)
    for nq in (9000, 300):
        Q = _mmap_array((nq, 128), np.float32, _dataset("sift", nq, 128, 4 + nq))
        want_i, want_d, want_nd, want_nh = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=128, counters=True)
        ids = torch.empty((nq, 10), dtype=torch.int32, device=dev)
        dd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
        nd = torch.zeros(nq, dtype=torch.int32, device=dev)
        nh = torch.zeros(nq, dtype=torch.int32, device=dev)
        st = torch.zeros(nq, dtype=torch.int32, device=dev)
        for registered in (False, True):
            if registered:
                H.pin(Q)
            try:
                ids.fill_(-9)
                keep = H.search_batch_h2d(hg, Q, 128, 10, ids.data_ptr(), dd.data_ptr(), nd.data_ptr(), nh.data_ptr(), st.data_ptr(), 0)
                torch.cuda.synchronize()
                del keep
            finally:
                if registered:
                    H.unpin(Q)
            np.testing.assert_array_equal(ids.cpu().numpy(), want_i)
            np.testing.assert_array_equal(dd.cpu().numpy().view(np.uint32), want_d.view(np.uint32))
            np.testing.assert_array_equal(nd.cpu().numpy().astype(np.uint32), want_nd)
            np.testing.assert_array_equal(nh.cpu().numpy().astype(np.uint32), want_nh)
            assert not (st.cpu().numpy() & 1).any()
    with pytest.raises(H.InvalidArgument):
        H.search_batch_h2d(hg, X[:4, :64], 16, 4, ids.data_ptr(), dd.data_ptr())          # wrong dimension
    with pytest.raises(H.InvalidArgument, match="k=20 > ef=5"):
        H.search_batch_h2d(hg, X[:4], 5, 20, ids.data_ptr(), dd.data_ptr())
    L = H.load()
    import ctypes
    p = H._SearchParams(16, 4, 0, 0)
    assert L.hnsw_search_batch_h2d(hg.handle, None, 4, 128, ctypes.byref(p), ids.data_ptr(), dd.data_ptr(), None, None, None, None) == -1   # HNSW_ERR_BAD_ARG
    assert L.hnsw_search_batch_h2d(hg.handle, X.ctypes.data, 0, 128, ctypes.byref(p), ids.data_ptr(), dd.data_ptr(), None, None, None, None) == 0   # empty batch
    out = ctypes.c_void_p()
    assert L.hnsw_host_alloc(ctypes.byref(out), 0) == -1 and L.hnsw_host_alloc(None, 64) == -1
    assert L.hnsw_host_free(None) == 0
    hg.release()


@pytest.mark.parametrize("levels", [3, 8])
def test_functor_accept_rule_on_ties(H, oracle, levels):
    """Hnsw.Ba: Nearest.insert_distance (lib/hnsw.ml:494-506) accepts an element that is not farther
    than max(W); under the canonical (distance, id) order the GPU's HNSW_SEM_FUNCTOR must equal the
    oracle's functor path on tie-heavy data -- and differ from the Ohnsw rule somewhere."""
    rng = np.random.default_rng(levels + 100)
    X = rng.integers(0, levels, size=(4000, 6)).astype(np.float32)
    Q = rng.integers(0, levels, size=(200, 6)).astype(np.float32)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 8, 60, seed=2)
    hg1 = _hgraph(H, X, g, id_base=1, M=8)
    differs = False
    for ef, k in ((16, 16), (64, 10)):
        d = H.Ba.knn_batch(hg1, Q, num_neighbours_search=ef, num_neighbours=k)
        od, oi = oracle.Functor.knn_batch(g, sp, Q, ef, k, ties=oracle.TIES_CANONICAL, with_ids=True)
        np.testing.assert_array_equal(d.view(np.uint32), od.view(np.uint32))
        one = [H.Ba.knn(hg1, Q[j], ef, k) for j in range(10)]
        for j in range(10):
            assert [n for n, _ in one[j]] == [int(x) + 1 for x in oi[j] if x >= 0]
        ids_ohnsw, _ = H.Ohnsw.knn_batch_bigarray(_hgraph(H, X, g, M=8), k, Q, ef=ef)
        differs = differs or not np.array_equal(ids_ohnsw, oi)
    assert differs


def test_randomized_small_configurations(H, oracle):
    """Many small random configurations across every template boundary (ef around 64/128/256, d around
    4/64/128/256, row widths up to 64, 0..5 upper layers, both metrics, both id bases, duplicate
    points): ids, distances and hop counts must equal the oracle's bit for bit in all of them."""
    rng = np.random.default_rng(2024)
    efs = [1, 2, 7, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300]
    ds = [1, 2, 3, 4, 5, 16, 31, 63, 64, 65, 100, 127, 128, 129, 200, 256, 257, 300]
    checked = 0
    for trial in range(70):
        n = int(rng.integers(1, 400))
        d = int(rng.choice(ds))
        M = int(rng.choice([2, 3, 5, 8, 16, 32]))
        ef = int(rng.choice(efs))
        k = int(rng.integers(1, ef + 1))
        metric = int(rng.integers(0, 2))
        id_base = int(rng.integers(0, 2))
        levels = int(rng.choice([2, 4, 50]))         # few distinct coordinate values => exact ties
        X = rng.integers(0, levels, size=(n, d)).astype(np.float32)
        if metric == 1:
            X = X + rng.uniform(0, 0.5, size=X.shape).astype(np.float32)
            X /= np.maximum(np.linalg.norm(X, axis=1, keepdims=True), 1e-6)
        sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
        g = oracle.build_ohnsw(sp, M, int(rng.integers(4, 60)), seed=int(rng.integers(0, 1000)))
        hg = _hgraph(H, X, g, id_base=id_base, metric=metric, M=M)
        nq = 6
        Q = X[rng.integers(0, n, nq)] + (rng.integers(0, 2, size=(nq, d)) if metric == 0 else 0)
        Q = Q.astype(np.float32)
        ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
        oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=k, ef=ef, ties=oracle.TIES_CANONICAL, counters=True)
        oi = np.where(oi >= 0, oi + id_base, -1)
        ctx = dict(trial=trial, n=n, d=d, M=M, ef=ef, k=k, metric=metric, id_base=id_base, levels=levels)
        assert np.array_equal(ids, oi), ctx
        assert np.array_equal(dist.view(np.uint32), od.view(np.uint32)), ctx
        assert np.array_equal(nh, onh), ctx
        fd = H.Ba.knn_batch(hg, Q, num_neighbours_search=ef, num_neighbours=k) if id_base == 1 else None
        if fd is not None:
            ofd = oracle.Functor.knn_batch(g, sp, Q, ef, k, ties=oracle.TIES_CANONICAL)
            assert np.array_equal(fd.view(np.uint32), ofd.view(np.uint32)), ctx
        checked += 1
    assert checked == 70


def test_strided_vectors_and_queries(H, oracle, tiny):
    """A Lacaml sub-matrix keeps its parent's leading dimension (benchmark/dataset.ml:91-93): rows that
    are contiguous but spaced must work for the vector table (row_stride) and the batch (q_stride)."""
    X, sp, g = tiny
    big = np.full((X.shape[0], X.shape[1] + 7), 9e9, np.float32)
    big[:, :X.shape[1]] = X
    Xv = big[:, :X.shape[1]]
    hg = H.Hgraph(Xv, g.deg0, g.nbr0, g.upper, entry_point=g.entry_point, max_degree=6)
    assert hg.row_stride == X.shape[1] + 7
    Qbig = np.full((40, X.shape[1] + 3), -7e9, np.float32)
    Qbig[:, :X.shape[1]] = X[:40] + 0.01
    ids, dist = H.Ohnsw.knn_batch_bigarray(hg, 5, Qbig[:, :X.shape[1]], ef=30)
    oi, od = oracle.Ohnsw.knn_batch_bigarray(g, sp, X[:40] + 0.01, k=5, ef=30, ties=oracle.TIES_CANONICAL)
    np.testing.assert_array_equal(ids, oi)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))


def test_inner_product_with_negative_distances(H, oracle):
    """1 - <a,b> is negative for long vectors: the ordered-key transform must keep the order across the
    sign change, and the reported distances must be the oracle's, bit for bit."""
    rng = np.random.default_rng(31)
    X = (rng.normal(size=(3000, 20)) * 2.0).astype(np.float32)          # dots from about -60 to +60
    Q = (rng.normal(size=(80, 20)) * 2.0).astype(np.float32)
    sp = oracle.Space.ip(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 8, 60, seed=4)
    hg = _hgraph(H, X, g, metric=1, M=8)
    for ef, k in ((10, 10), (100, 30)):
        ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
        oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=k, ef=ef, ties=oracle.TIES_CANONICAL, counters=True)
        np.testing.assert_array_equal(ids, oi)
        np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))
        np.testing.assert_array_equal(nh, onh)
        assert (dist < 0).any()
        assert (np.diff(dist, axis=1) >= 0).all()
    # both signs, through the gathered-distance entry point
    pick = rng.integers(0, 3000, size=(80, 50)).astype(np.int32)
    got = H.Ohnsw.distance_l2(hg, Q, pick)
    want = np.array([[np.float32(1.0) - np.float32(oracle.dot_tree16(X[j], q)) for j in row] for q, row in zip(Q, pick)], np.float32)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    assert (got < 0).any() and (got > 0).any()


@pytest.mark.parametrize("n,d,metric,M,ef,k", [(3000, 20, 0, 8, 1024, 1024),     # the largest W (16 slots), k = ef
                                                 (600, 1024, 0, 6, 64, 10),        # the widest rows (16 chunks per lane)
                                                 (600, 1000, 1, 6, 70, 70),        # ragged last chunk at that width, inner product
                                                 (2500, 7, 0, 32, 500, 3)])        # 64-wide adjacency rows, ef far above k
def test_extremes_of_the_supported_range(H, oracle, n, d, metric, M, ef, k):
    rng = np.random.default_rng(n + d)
    X = rng.normal(size=(n, d)).astype(np.float32)
    Q = rng.normal(size=(12, d)).astype(np.float32)
    sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, M, 40, seed=3)
    hg = _hgraph(H, X, g, metric=metric, M=M)
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
    oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=k, ef=ef, ties=oracle.TIES_CANONICAL, counters=True)
    np.testing.assert_array_equal(ids, oi)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(nh, onh)
    with pytest.raises(H.InvalidArgument):
        H.Ohnsw.knn_batch_bigarray(hg, k + 1, Q, ef=k)                  # k > ef is rejected, never clamped


def test_beyond_the_supported_range_is_refused(H, oracle, tiny):
    X, sp, g = tiny
    hg = _hgraph(H, X, g, M=6)
    with pytest.raises(H.Failure, match="ef=1025"):
        H.Ohnsw.knn_batch_bigarray(hg, 5, X[:2], ef=1025)
    with pytest.raises(H.Failure, match="d=1025"):
        H.Hgraph(np.zeros((4, 1025), np.float32), np.zeros(4, np.int32), np.full((4, 2), -1, np.int32), entry_point=0).to_device(0)
    with pytest.raises(H.Failure, match="max_degree0=65"):
        H.Hgraph(np.zeros((4, 8), np.float32), np.zeros(4, np.int32), np.full((4, 65), -1, np.int32), entry_point=0).to_device(0)


def test_prepare_and_expected_ef_change_nothing_but_the_first_call(H, oracle, tiny):
    """hnsw_index_prepare / hnsw_index_desc.expected_ef (ABI 3): one-time work moved in front of the first search; parameter
    errors are those of a search (an explicit prepare reports them, construction with an unusable expected_ef ignores it);
    results are the same bits with and without."""
    X, sp, g = tiny
    plain = _hgraph(H, X, g, M=6)
    want = H.Ohnsw.knn_batch_bigarray(plain, 5, X[:40], ef=20, counters=True)
    fw = H.Ba.knn_batch(plain, X[:40], 20, 5)
    for kw in ({"expected_ef": 20}, {"expected_ef": 20, "expected_sem": H.SEM_FUNCTOR}, {"expected_ef": 5000}, {"expected_ef": -3}):
        up = [(nodes, deg, nbr) for nodes, deg, nbr in g.upper]
        hg = H.Hgraph(X, g.deg0, g.nbr0, up, entry_point=g.entry_point, id_base=0, max_degree=6, **kw).to_device(0)
        got = H.Ohnsw.knn_batch_bigarray(hg, 5, X[:40], ef=20, counters=True)
        for a, b in zip(want, got):
            np.testing.assert_array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
        np.testing.assert_array_equal(H.Ba.knn_batch(hg, X[:40], 20, 5).view(np.uint32), fw.view(np.uint32))
        hg.release()
    plain.prepare(20).prepare(300, sem=H.SEM_FUNCTOR).prepare(20)          # any number of shapes, twice the same
    got = H.Ohnsw.knn_batch_bigarray(plain, 5, X[:40], ef=20, counters=True)
    for a, b in zip(want, got):
        np.testing.assert_array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
    with pytest.raises(H.Failure, match="ef=1025"):
        plain.prepare(1025)
    with pytest.raises(H.InvalidArgument):
        plain.prepare(0)
    empty = H.Hgraph(np.zeros((0, 4), np.float32), np.zeros(0, np.int32), np.zeros((0, 4), np.int32), expected_ef=16).to_device(0)
    with pytest.raises(H.InvalidArgument, match="empty hgraph"):
        empty.prepare(16)
    built = H.Ohnsw.build_batch_bigarray(X, 6, 40, seed=2, expected_ef=20)  # hnsw_build_params.expected_ef
    unbuilt = H.Ohnsw.build_batch_bigarray(X, 6, 40, seed=2)
    for a, b in zip(H.Ohnsw.knn_batch_bigarray(built, 5, X[:40], ef=20), H.Ohnsw.knn_batch_bigarray(unbuilt, 5, X[:40], ef=20)):
        np.testing.assert_array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def test_nearest_k_compat_reproduces_the_reference_output(H, oracle):
    """Hnsw.Ba.knn* with ~num_neighbours_search > ~num_neighbours returns, in the reference, the k FARTHEST
    members of W (Nearest.nearest_k, lib/hnsw.ml:522-525).  HNSW_SEM_FUNCTOR_NEAREST_K reproduces that
    output; the default does not."""
    rng = np.random.default_rng(17)
    X = rng.integers(0, 9, size=(3000, 10)).astype(np.float32)
    Q = rng.integers(0, 9, size=(60, 10)).astype(np.float32)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 8, 50, seed=6)
    hg1 = _hgraph(H, X, g, id_base=1, M=8)
    for ef, k in ((40, 7), (130, 20), (10, 10), (5000 // 50, 64)):
        want_d, want_i = oracle.Functor.knn_batch(g, sp, Q, ef, k, ties=oracle.TIES_CANONICAL,
                                                  bug_compat_farthest_k=True, with_ids=True)
        got = H.Ba.knn_batch(hg1, Q, ef, k, nearest_k_compat=True)
        np.testing.assert_array_equal(got.view(np.uint32), want_d.view(np.uint32))
        one = H.Ba.knn(hg1, Q[3], ef, k, nearest_k_compat=True)
        assert [n for n, _ in one] == [int(x) + 1 for x in want_i[3] if x >= 0]
        plain = H.Ba.knn_batch(hg1, Q, ef, k)
        if ef > k:
            assert (plain[:, 0] <= got[:, 0]).all() and (plain[:, 0] < got[:, 0]).any()
        else:
            np.testing.assert_array_equal(plain.view(np.uint32), got.view(np.uint32))


def _mmap_array(shape, dtype, fill=None):
    """a numpy array on an anonymous mmap region of its own (page aligned, nothing else in its pages)"""
    import mmap
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    a = np.frombuffer(mmap.mmap(-1, max(nbytes, 1)), dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    if fill is not None:
        a[...] = fill
    return a


def test_registered_host_arrays_and_caller_owned_results(H, oracle):
    """Page-locked matrices of the caller -- registered (hnsw_host_register / hnsw_host_unregister: the caller owns the
    lifetime) or allocated by the library (hnsw_host_alloc) -- and results written into them: the device accesses them
    directly, same bits as the plain call, for the synchronous call and for submit / wait; registering twice is fine;
    arrays registered only in part are refused."""
    X = _dataset("sift", 5000, 128, 3)
    Q0 = _dataset("sift", 3000, 128, 4)
    hg = H.Ohnsw.build_batch_bigarray(X, 16, 80, seed=2)
    k, ef = 10, 128
    want_i, want_d = H.Ohnsw.knn_batch_bigarray(hg, k, Q0, ef=ef)
    for how in ("registered", "allocated"):
        if how == "registered":
            Q = _mmap_array(Q0.shape, np.float32, Q0)
            ids, dist = _mmap_array((Q.shape[0], k), np.int32, -7), _mmap_array((Q.shape[0], k), np.float32, 0)
            for a in (Q, ids, dist):
                H.pin(a)
            H.pin(Q)                                 # twice: not an error
        else:
            Q = H.host_empty(Q0.shape, np.float32)
            Q[:] = Q0
            ids, dist = H.host_empty((Q.shape[0], k), np.int32), H.host_empty((Q.shape[0], k), np.float32)
            ids[:] = -7
        try:
            got = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, out=(ids, dist))
            assert got[0] is ids and got[1] is dist
            np.testing.assert_array_equal(ids, want_i)
            np.testing.assert_array_equal(dist.view(np.uint32), want_d.view(np.uint32))
            ids[:] = -7
            got = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, out=(ids, dist), counters=True)     # counters staged, results direct
            np.testing.assert_array_equal(ids, want_i)
            assert (got[2] > 0).all()
            ids[:] = -7
            r = H.submit(hg, Q, ef, k)
            r.wait(out=(ids, dist))
            np.testing.assert_array_equal(ids, want_i)
            np.testing.assert_array_equal(dist.view(np.uint32), want_d.view(np.uint32))
            # page-locked queries, pageable results and the other way round
            pi, pd = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef)
            np.testing.assert_array_equal(pi, want_i)
            ids[:] = -7
            H.Ohnsw.knn_batch_bigarray(hg, k, Q0, ef=ef, out=(ids, dist))
            np.testing.assert_array_equal(ids, want_i)
            # a sub-matrix of a page-locked matrix (rows 100..): still inside the registered range
            sub_i, sub_d = H.Ohnsw.knn_batch_bigarray(hg, k, Q[100:], ef=ef)
            np.testing.assert_array_equal(sub_i, want_i[100:])
        finally:
            if how == "registered":
                for a in (Q, ids, dist):
                    H.unpin(a)
        if how == "registered":
            with pytest.raises(H.InvalidArgument, match="no range registered"):
                H.unpin(ids)                         # not registered any more
        else:
            with pytest.raises(H.InvalidArgument, match="hnsw_host_alloc"):
                H.unpin(ids)                         # the library's own block: freed with the array, never unregistered
    ids, dist = np.full((Q0.shape[0], k), -7, np.int32), np.zeros((Q0.shape[0], k), np.float32)
    with pytest.raises(H.InvalidArgument):
        H.Ohnsw.knn_batch_bigarray(hg, k, Q0, ef=ef, out=(ids[:, :5], dist))
    # a strided view or a Fortran-ordered matrix has the right shape and dtype but is not nq * k contiguous words
    big_i, big_d = np.zeros((Q0.shape[0], 2 * k), np.int32), np.zeros((Q0.shape[0], 2 * k), np.float32)
    for bad in ((big_i[:, :k], big_d[:, :k]), (np.asfortranarray(ids), dist)):
        r = H.submit(hg, Q0, ef, k)
        with pytest.raises(H.InvalidArgument, match="contiguous"):
            r.wait(out=bad)
        r.wait(out=(ids, dist))                  # the request is still there to be waited for
        np.testing.assert_array_equal(ids, want_i)
    assert not big_i.any() and not big_d.any()
    # an existing registration that covers only the head of the array does not make the whole array registered
    whole = _mmap_array((1 << 22,), np.uint8, 0)
    H.pin(whole[:1 << 20])
    try:
        with pytest.raises(H.InvalidArgument, match="registered already"):
            H.pin(whole)
    finally:
        H.unpin(whole[:1 << 20])
    # a small array inside the malloc heap registers like any other (its pages stay mapped while the block is alive)
    small = np.arange(64, dtype=np.float32)
    H.pin(small)
    H.unpin(small)
    # an array somebody else has page-locked (here: torch) is accepted, never unregistered by the library, and searched through copies
    import torch
    theirs = torch.from_numpy(Q0.copy()).pin_memory()
    Qt = theirs.numpy()
    H.pin(Qt)
    ti, td = H.Ohnsw.knn_batch_bigarray(hg, k, Qt, ef=ef)
    np.testing.assert_array_equal(ti, want_i)
    H.unpin(Qt)
    assert theirs.is_pinned()
    ti, td = H.Ohnsw.knn_batch_bigarray(hg, k, Qt, ef=ef)           # still theirs, still valid
    np.testing.assert_array_equal(ti, want_i)


def test_unregistering_a_matrix_the_device_still_reads_waits(H, oracle):
    """VERDICT r04 item 2.  hnsw_search_batch_h2d returns while the device still reads the caller's registered query matrix in
    place (the ordering pre-pass of a large batch; the search kernel itself for a small one).  Round 4's second suite abort was a
    test that unregistered at that moment: hipHostUnregister pulled the pages from under the kernel -- a GPU page fault, the
    process gone.  The defined behaviour now: hnsw_host_unregister / hnsw_host_free WAIT for the readers the library knows of
    (an event behind the launch, per range and stream); results are the plain call's."""
    import torch
    dev = torch.device("cuda", 0)
    X = _dataset("sift", 20000, 128, 3)
    hg = H.Ohnsw.build_batch_bigarray(X, 16, 80, seed=2)
    stream = torch.cuda.Stream()
    for nq in (9000, 300):
        Q0 = _dataset("sift", nq, 128, 40 + nq)
        want_i, want_d = H.Ohnsw.knn_batch_bigarray(hg, 10, Q0, ef=128)
        ids = torch.full((nq, 10), -9, dtype=torch.int32, device=dev)
        dd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        for how in ("registered", "allocated"):
            if how == "registered":
                Q = _mmap_array((nq, 128), np.float32, Q0)
                H.pin(Q)
            else:
                Q = H.host_empty((nq, 128), np.float32)
                Q[:] = Q0
            ids.fill_(-9)
            torch.cuda.synchronize()
            keep = H.search_batch_h2d(hg, Q, 128, 10, ids.data_ptr(), dd.data_ptr(), stream=stream.cuda_stream)
            # no synchronisation: the launch is in flight (or still queued) when the range is taken away
            if how == "registered":
                H.unpin(Q)
            else:
                del keep, Q                              # the last references: _HostBlock.__del__ -> hnsw_host_free
            stream.synchronize()
            np.testing.assert_array_equal(ids.cpu().numpy(), want_i)
            np.testing.assert_array_equal(dd.cpu().numpy().view(np.uint32), want_d.view(np.uint32))
    hg.release()
