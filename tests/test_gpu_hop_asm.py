"""The hand-scheduled layer-0 loops (csrc/hnsw_hop_asm.hip.h: byte rows of 65..128 and of 129..256 dimensions, byte-valued
queries, L2 and inner product, both accept rules; W in one / two / three / four / six / eight key registers per lane = ef <= 64 /
65..128 / 129..192 / 193..256 / 257..384 / 385..512, every boundary among the ef values below) against the oracle on
data chosen to drive their seldom-taken paths: exact distance ties everywhere (the general rank with id comparison, the
"node already in W" check), entries evicted while tied with the new maximum (the tie list in LDS, its pop when W has no
unexpanded member left, its overflow and the host's exactness fallback), rounds of 1 / 2 / 4 batches and lists longer
than one round, every rank slot of the cascading shift.  Bit parity throughout: ids, distance bits, hop counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import ocaml_hnsw_amd as H
    H.load()
    assert H.device_count() >= 1
    return H


def _hgraph(H, X, g, M):
    return H.Hgraph(X, g.deg0, g.nbr0, g.upper, entry_point=g.entry_point, id_base=0, max_degree=M)


def _check(H, oracle, hg, g, sp, Q, ef, k, ctx=""):
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
    oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=k, ef=ef, ties=oracle.TIES_CANONICAL, counters=True)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32), err_msg=ctx)
    np.testing.assert_array_equal(ids, oi, err_msg=ctx)
    np.testing.assert_array_equal(nh, onh, err_msg=ctx)
    # (evaluation counts are not compared: the same hops expand the same rows, but the lossy visited cache re-evaluates
    # a data-dependent share of them -- a third on these 5000-node tie-heavy sets at ef 256 -- which never changes W)
    assert (nd > 0).all(), ctx


@pytest.mark.parametrize("levels", [2, 3, 6, 40])
@pytest.mark.parametrize("d", [65, 100, 128])
def test_ties_everywhere_every_slot_count(H, oracle, levels, d):
    rng = np.random.default_rng(100 * levels + d)
    n = 5000
    X = rng.integers(0, levels, size=(n, d)).astype(np.float32)
    X[rng.integers(0, n, 300)] = X[rng.integers(0, n, 300)]               # exact duplicates
    Q = rng.integers(0, levels, size=(150, d)).astype(np.float32)
    Q[:10] = X[:10]                                                        # queries that ARE data points
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 12, 60, seed=3)
    hg = _hgraph(H, X, g, 12)
    assert hg.to_device(0).row_bytes() == d                                # byte rows: the loops under test run
    for ef, k in ((1, 1), (17, 5), (64, 64), (65, 10), (100, 100), (128, 10), (129, 20), (160, 160), (192, 10), (193, 10), (256, 256), (257, 10), (384, 384), (385, 10), (400, 400), (512, 64)):
        _check(H, oracle, hg, g, sp, Q, ef, k, "levels %d d %d ef %d" % (levels, d, ef))


def _check_functor(H, oracle, hg, g, sp, Q, ef, k, ctx=""):
    """Hnsw_algo.Search / Hnsw.Ba.knn_batch (the functor accept rule) under the (d, id) order: ids, distance bits, counters"""
    import ocaml_hnsw_amd as A
    gi, gd, gnd, gnh = A._search(hg, Q, ef, k, A.FILL_BA, True, sem=A.SEM_FUNCTOR)
    cd, ci = oracle.Functor.knn_batch(g, sp, Q, ef, k, ties=oracle.TIES_CANONICAL, with_ids=True)
    np.testing.assert_array_equal(gd.view(np.uint32), cd.view(np.uint32), err_msg=ctx)
    np.testing.assert_array_equal(gi, ci, err_msg=ctx)
    assert (gnd > 0).all() and (gnh > 0).all(), ctx


@pytest.mark.parametrize("levels", [2, 3, 6, 40])
@pytest.mark.parametrize("d", [65, 128])
def test_functor_rule_through_the_loops(H, oracle, levels, d):
    """The loops instantiated for the functor rule leave a hop whenever an entry would enter the tie set and come back when
    the set is empty again (search_layer): tie-heavy data makes them leave all the time, in every slot count."""
    rng = np.random.default_rng(300 * levels + d)
    n = 5000
    X = rng.integers(0, levels, size=(n, d)).astype(np.float32)
    X[rng.integers(0, n, 300)] = X[rng.integers(0, n, 300)]
    Q = rng.integers(0, levels, size=(150, d)).astype(np.float32)
    Q[:10] = X[:10]
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 12, 60, seed=3)
    hg = _hgraph(H, X, g, 12)
    assert hg.to_device(0).row_bytes() == d
    for ef, k in ((1, 1), (17, 5), (64, 64), (65, 10), (100, 100), (128, 10), (129, 20), (160, 160), (192, 10), (193, 10), (256, 256), (257, 10), (384, 384), (385, 10), (400, 400), (512, 64)):
        _check_functor(H, oracle, hg, g, sp, Q, ef, k, "functor levels %d d %d ef %d" % (levels, d, ef))


@pytest.mark.parametrize("levels", [2, 6, 40])
@pytest.mark.parametrize("d", [70, 128])
def test_inner_product_on_byte_rows_through_the_loops(H, oracle, levels, d):
    """byte rows under the inner product (distance 1 - <x, q>: large negative integers here): the dot-product block without the
    x.x term and the sign-flipped key, both accept rules, every slot count"""
    rng = np.random.default_rng(500 * levels + d)
    n = 5000
    X = rng.integers(0, levels, size=(n, d)).astype(np.float32)
    X[rng.integers(0, n, 300)] = X[rng.integers(0, n, 300)]
    Q = rng.integers(0, levels, size=(150, d)).astype(np.float32)
    Q[:10] = X[:10]
    Q[10] = 0.0                                                            # every distance 1.0: one huge tie
    sp = oracle.Space.ip(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 12, 60, seed=3)
    hg = H.Hgraph(X, g.deg0, g.nbr0, g.upper, entry_point=g.entry_point, id_base=0, max_degree=12, metric=1)
    assert hg.to_device(0).row_bytes() == d
    for ef, k in ((1, 1), (17, 5), (64, 64), (65, 10), (128, 10), (129, 20), (192, 100), (256, 256), (300, 300), (384, 10), (512, 10)):
        _check(H, oracle, hg, g, sp, Q, ef, k, "ip levels %d d %d ef %d" % (levels, d, ef))
        _check_functor(H, oracle, hg, g, sp, Q, ef, k, "ip functor levels %d d %d ef %d" % (levels, d, ef))


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("levels", [2, 6, 256])
@pytest.mark.parametrize("d", [129, 200, 256])
def test_byte_rows_of_129_to_256_dimensions_through_the_loops(H, oracle, levels, d, metric):
    """four dwords per lane and row (NCH = 4): twice the loads and dot products per batch, the same loop otherwise; d = 256 with
    every value 255 against a query of zeros is the largest sum the integer arithmetic may meet (256 * 255^2 < 2^24)"""
    rng = np.random.default_rng(900 * levels + d + metric)
    n = 4000
    X = rng.integers(0, levels, size=(n, d)).astype(np.float32)
    X[rng.integers(0, n, 200)] = X[rng.integers(0, n, 200)]
    X[1, :] = float(levels - 1)
    Q = rng.integers(0, levels, size=(120, d)).astype(np.float32)
    Q[:8] = X[:8]
    Q[8] = 0.0
    Q[9] = float(levels - 1)
    sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 12, 60, seed=3)
    hg = H.Hgraph(X, g.deg0, g.nbr0, g.upper, entry_point=g.entry_point, id_base=0, max_degree=12, metric=metric)
    assert hg.to_device(0).row_bytes() == d
    for ef, k in ((1, 1), (17, 5), (64, 64), (100, 10), (128, 128), (192, 10), (256, 50), (400, 20), (512, 512)):
        _check(H, oracle, hg, g, sp, Q, ef, k, "n4 metric %d levels %d d %d ef %d" % (metric, levels, d, ef))
        if ef in (17, 100, 192, 400):
            _check_functor(H, oracle, hg, g, sp, Q, ef, k, "n4 functor metric %d levels %d d %d ef %d" % (metric, levels, d, ef))


def test_wide_rows_and_long_lists(H, oracle):
    """M = 32: layer-0 rows of 64 neighbours, fresh lists longer than one 16-row round, all four batch shapes."""
    rng = np.random.default_rng(7)
    n, d = 6000, 96
    centres = rng.integers(20, 200, size=(12, d))
    X = np.clip(np.rint(centres[rng.integers(0, 12, n)] + rng.normal(0, 30, size=(n, d))), 0, 255).astype(np.float32)
    Q = np.clip(np.rint(centres[rng.integers(0, 12, 200)] + rng.normal(0, 30, size=(200, d))), 0, 255).astype(np.float32)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 32, 80, seed=1)
    hg = _hgraph(H, X, g, 32)
    for ef, k in ((48, 10), (128, 10), (250, 50), (500, 100)):
        _check(H, oracle, hg, g, sp, Q, ef, k, "M 32 ef %d" % ef)
        _check_functor(H, oracle, hg, g, sp, Q, ef, k, "functor M 32 ef %d" % ef)


@pytest.mark.parametrize("ef", [64, 128, 256, 512])
def test_tie_list_overflow_through_the_loops(H, oracle, ef):
    """The scenario of test_tie_overflow_beyond_lds_stack with byte-valued 128-dimensional vectors, sized per slot
    count: ef - 1 identical "shell" points fill W behind the far entry node; a chain of ever closer points then evicts
    them one by one while they are still unexpanded and tied with the new maximum -- more than 64 of them, so the LDS
    list overflows (ef >= 128), the query is flagged and the host entry point searches it again; a point reachable only
    through a late shell must be found, as the oracle finds it."""
    import torch
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
    # entry E -> the first shell H and as many shells as fit; H and further shells fan out to the rest and to the chain head
    first = list(range(1, 1 + shells))
    rows[0] = first[:width]
    rest = first[width:]
    hub = 1
    while rest:
        rows[hub] = rest[:width - 1]
        rest = rest[width - 1:]
        hub += 1
    rows[1] = rows[1][:width - 1] + [1 + shells]                           # ... and the head of the chain
    for i in range(chain - 1):
        rows[1 + shells + i] = [2 + shells + i]
    late = 1 + (shells * 2) // 3                                           # a shell that is evicted late: the only way to Z
    rows[late] = (rows[late] + [z])[:width]
    deg0 = np.array([len(r) for r in rows], np.int32)
    nbr0 = np.full((n, width), -1, np.int32)
    for i, r in enumerate(rows):
        nbr0[i, :len(r)] = r
    X = np.zeros((n, 128), np.float32)
    X[:, 0] = pos
    g = oracle.Graph(n, 0, deg0, nbr0)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    hg = H.Hgraph(X, deg0, nbr0, entry_point=0, max_degree=32)
    Q = np.zeros((3, 128), np.float32)
    want = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=10, ef=ef, ties=oracle.TIES_CANONICAL, counters=True)
    got = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=ef, counters=True)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    np.testing.assert_array_equal(got[3], want[3])
    # the device-pointer entry point has no fallback: it flags exactly the queries whose list outgrew its 64 LDS slots
    dev = torch.device("cuda", 0)
    Qd = torch.from_numpy(Q).to(dev)
    ids = torch.empty((3, 10), dtype=torch.int32, device=dev)
    dd = torch.empty((3, 10), dtype=torch.float32, device=dev)
    st = torch.zeros(3, dtype=torch.int32, device=dev)
    H.search_batch_device(hg.to_device(0), Qd.data_ptr(), 3, 128, ef, 10, ids.data_ptr(), dd.data_ptr(), 0, 0, st.data_ptr(), 0)
    torch.cuda.synchronize()
    flagged = (st & 1).cpu().numpy().astype(bool)
    if ef == 128:
        assert flagged.all()         # 100 evictions of unexpanded shells tied with the new maximum: more than the 64 LDS slots
    if ef == 64:
        assert not flagged.any()     # W cannot hold that many
    if not flagged.any():
        np.testing.assert_array_equal(ids.cpu().numpy(), want[0])


def test_random_configurations_of_the_loop_shapes(H, oracle):
    """Random small problems inside the loops' domain (d 65..128, byte values, L2, Ohnsw rule, ef 1..256)."""
    rng = np.random.default_rng(77)
    for trial in range(60):
        n = int(rng.integers(2, 700))
        d = int(rng.integers(65, 129))
        M = int(rng.choice([2, 4, 8, 16, 32]))
        levels = int(rng.choice([2, 4, 16, 219]))
        ef = int(rng.choice([1, 3, 30, 63, 64, 65, 90, 127, 128, 129, 160, 191, 192, 193, 255, 256, 257, 300, 384, 385, 511, 512]))
        k = int(rng.integers(1, min(ef, 100) + 1))
        X = rng.integers(0, levels, size=(n, d)).astype(np.float32)
        Q = rng.integers(0, levels, size=(20, d)).astype(np.float32)
        sp = oracle.Space.l2(X, arith=oracle.TREE16)
        g = oracle.build_ohnsw(sp, M, 40, seed=trial)
        hg = _hgraph(H, X, g, M)
        _check(H, oracle, hg, g, sp, Q, ef, k, "trial %d: n %d d %d M %d levels %d ef %d k %d" % (trial, n, d, M, levels, ef, k))
        _check_functor(H, oracle, hg, g, sp, Q, ef, k, "functor trial %d: n %d d %d M %d levels %d ef %d k %d" % (trial, n, d, M, levels, ef, k))


def test_issue_priorities_and_ordering_change_nothing(H, oracle, monkeypatch):
    """An ordered launch runs its first blocks and its late starters at raised issue priority (launch_priorities,
    csrc/hnsw_capi.hip; HNSW_PRIO="head,tail" overrides the bounds), and its descent runs in the pre-pass kernel: ids,
    distance bits, hop and evaluation counts per query must be those of the plain launch, whatever the bounds."""
    rng = np.random.default_rng(5)
    n, d = 4000, 128
    X = rng.integers(0, 200, size=(n, d)).astype(np.float32)
    Q = rng.integers(0, 200, size=(3000, d)).astype(np.float32)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 16, 60, seed=9)
    hg = _hgraph(H, X, g, 16)
    for ef, k in ((40, 10), (128, 10), (200, 20)):
        hg.set_option("order_queries", 0)
        monkeypatch.delenv("HNSW_PRIO", raising=False)
        plain = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
        hg.set_option("order_queries", 1)
        for prio in (None, "0,2000000000", "3000,0", "100,2900", "1,1"):
            if prio is None:
                monkeypatch.delenv("HNSW_PRIO", raising=False)
            else:
                monkeypatch.setenv("HNSW_PRIO", prio)
            got = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
            for a, b in zip(plain, got):
                np.testing.assert_array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32), err_msg="ef %d HNSW_PRIO %s" % (ef, prio))
    monkeypatch.delenv("HNSW_PRIO", raising=False)
    oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q[:200], k=10, ef=128, ties=oracle.TIES_CANONICAL, counters=True)
    np.testing.assert_array_equal(H.Ohnsw.knn_batch_bigarray(hg, 10, Q[:200], ef=128)[0], oi)
