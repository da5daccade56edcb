"""Visited as bitmap blocks over locality codes (option "visited_blocks"; visited_blocks_mem_add in hnsw_device.hip.h, its
hand-scheduled twin HNSW_HOP_FILTER_*_BLK, the codes of hnsw_locality.hip) against the reference's exact Visited
(lib/ohnsw.ml:256-268, used at :571-572), restated in the oracle.

A visited structure that forgets can only ADD evaluations; one that invents a visit would lose neighbours.  So for every
mode (0 = the tag cache, 1 = the blocks, -1 = the handle's own measurement): ids, distance bits and hop counts equal the
oracle's -- on small graphs built by the oracle's restatement of Ohnsw.insert (every kernel shape the blocks run in: the
hand-scheduled loops over float32 rows with W in three / four / six / eight registers, both accept rules, and the C++ loop of other shapes)
and on clustered unit vectors at moderate and at BASELINE configuration 5's full size, where the tag cache re-evaluated a
third of the rows in round 4 -- plus the bound on the evaluations the device adds.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import ocaml_hnsw_amd as H
    H.load()
    assert H.device_count() >= 1, "GPU tests need a HIP device"
    return H


def _unit(rng, n, d):
    X = rng.normal(size=(n, d))
    return (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)


def _clustered_np(n, d, seed, centres, spread=1.5):
    rng = np.random.default_rng(4321)
    cen = _unit(rng, centres, d)
    rng = np.random.default_rng(seed)
    X = cen[rng.integers(0, centres, n)] + spread * rng.normal(size=(n, d)).astype(np.float32) / np.sqrt(d)
    return (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)


def _clustered_gpu(n, d, seed, centres=256, spread=1.5):
    """bench.py's clustered_unit_vectors (others.C3_clustered / C5_clustered): unit vectors around `centres` random directions"""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    cen = torch.randn((centres, d), generator=g, device=dev)
    cen = cen / cen.norm(dim=1, keepdim=True)
    g.manual_seed(seed)
    out = np.empty((n, d), np.float32)
    for s in range(0, n, 1 << 20):
        m = min(1 << 20, n - s)
        idx = torch.randint(0, centres, (m,), generator=g, device=dev)
        x = cen[idx] + spread * torch.randn((m, d), generator=g, device=dev) / (d ** 0.5)
        out[s:s + m] = (x / x.norm(dim=1, keepdim=True)).cpu().numpy()
    return out


def _hgraph(H, X, g, metric, M):
    return H.Hgraph(X, g.deg0, g.nbr0, g.upper, entry_point=g.entry_point, id_base=0, max_degree=M, metric=metric)


SMALL = [
    # name, d, metric, M, efC, ef, k: float32 rows of 65..128 dimensions -> the hand-scheduled block filter (W in 3 / 4 / 6 / 8 registers)
    ("ragged_l2_8slots", 96, 0, 16, 60, 512, 10),
    ("split_ip_4slots", 100, 1, 16, 60, 256, 50),
    ("full_l2_4slots", 128, 0, 12, 60, 200, 10),
    ("ragged_ip_6slots", 72, 1, 12, 60, 300, 10),
    ("ragged_l2_8slots_ef400", 80, 0, 12, 60, 400, 10),
    ("full_ip_3slots", 128, 1, 12, 60, 129, 10),
    ("ragged_l2_6slots", 96, 0, 16, 60, 384, 384),
    # byte-valued data of 65..128 dimensions (searched through the byte rows): the byte-row loops' block filter
    ("bytes_l2_4slots", 128, 0, 16, 60, 200, 10),
    ("bytes_l2_8slots", 100, 0, 12, 60, 400, 10),
    ("bytes_ip_4slots", 96, 1, 12, 60, 256, 20),
    ("bytes_l2_3slots", 128, 0, 16, 60, 180, 10),
    ("bytes_ip_6slots", 100, 1, 12, 60, 320, 10),
    # float32 rows of 129..256 dimensions: the four-chunk loops' block filter
    ("d200_ragged_8slots", 200, 0, 8, 40, 400, 10),
    ("d256_full_ip_4slots", 256, 1, 8, 40, 200, 10),
    ("d132_split_3slots", 132, 0, 8, 40, 160, 10),
    ("d132_split_4slots", 132, 0, 8, 40, 250, 10),
    # other shapes: the C++ loop's block filter
    ("d32_4slots", 32, 0, 8, 40, 200, 10),
    ("d300_8slots", 300, 0, 8, 40, 400, 10),
    ("ef1000_16slots", 96, 0, 8, 40, 1000, 20),
]


@pytest.mark.parametrize("case", SMALL, ids=lambda c: c[0])
def test_blocks_equal_the_exact_visited_set_on_small_graphs(H, oracle, case):
    name, d, metric, M, efc, ef, k = case
    if name.startswith("bytes"):
        rng = np.random.default_rng(11)
        cen = rng.integers(20, 200, size=(12, d))
        X = np.clip(np.rint(cen[rng.integers(0, 12, 5000)] + rng.normal(0, 25, size=(5000, d))), 0, 218).astype(np.float32)
        Q = np.clip(np.rint(cen[rng.integers(0, 12, 120)] + rng.normal(0, 25, size=(120, d))), 0, 218).astype(np.float32)
    else:
        X = _clustered_np(5000, d, 11, 12)
        Q = _clustered_np(120, d, 12, 12)
    sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, M, efc, seed=5)
    assert len(g.upper) >= 1, "the codes need an upper layer"
    hg = _hgraph(H, X, g, metric, M)
    assert (hg.row_bytes() == d) == name.startswith("bytes")
    oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=k, ef=ef, ties=oracle.TIES_CANONICAL, counters=True)
    fd, fi = oracle.Functor.knn_batch(g, sp, Q, ef, k, ties=oracle.TIES_CANONICAL, with_ids=True)
    for mode in (1, 0):
        hg.set_option("visited_blocks", mode)
        ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
        np.testing.assert_array_equal(ids, oi)
        np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))
        np.testing.assert_array_equal(nh, onh)
        # (the oracle counts the reference's distance CALLS: search_one evaluates its start node again on every upper layer,
        # lib/ohnsw.ml:496; the kernel carries the key down -- one evaluation fewer per upper layer, never more)
        extra = nd.astype(np.int64) - ond.astype(np.int64) + len(g.upper)
        assert (extra >= 0).all()
        if mode == 1:      # 5000 codes are twenty blocks: they all stay (a neighbour is only lost when two new blocks of one set meet in one hop)
            assert extra.sum() <= 0.02 * ond.sum()
        gi, gd = H._search(hg, Q, ef, k, H.FILL_BA, sem=H.SEM_FUNCTOR)        # Hnsw.Ba: the functor rule on the same loops
        np.testing.assert_array_equal(gi, fi)
        np.testing.assert_array_equal(gd.view(np.uint32), fd.view(np.uint32))
    hg.release()


def test_locality_codes_are_a_permutation_and_need_an_upper_layer(H, oracle):
    X = _clustered_np(4000, 96, 3, 8)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, 8, 40, seed=2)
    hg = _hgraph(H, X, g, 0, 8)
    L = hg.locality_codes()
    np.testing.assert_array_equal(np.sort(L), np.arange(len(X)))
    # neighbours in the graph are neighbours in the numbering far more often than under the ids
    near = lambda code: np.mean([np.mean(np.abs(code[g.nbr0[v, :g.deg0[v]]] - code[v]) < 256) for v in range(0, len(X), 7) if g.deg0[v]])
    assert near(L.astype(np.int64)) > max(0.4, 2.5 * near(np.arange(len(X))))
    hg.release()
    flat = H.Hgraph(X[:50], np.zeros(50, np.int32), np.full((50, 4), -1, np.int32), [], entry_point=0, max_degree=2)
    with pytest.raises(H.Failure):
        flat.locality_codes()
    flat.set_option("visited_blocks", 1)                                   # nothing to build the codes from: the tag cache, quietly
    ids, dist = H.Ohnsw.knn_batch_bigarray(flat, 1, X[:3], ef=200)
    assert ids[:, 0].tolist() == [0, 0, 0]
    flat.release()


@pytest.mark.parametrize("bits", [3, 4, 5])
@pytest.mark.parametrize("shape", [("ragged_l2", 96, 0, 300), ("split_ip", 100, 1, 200), ("bytes_l2", 128, 0, 400)], ids=lambda s: s[0])
def test_tiny_directories_evict_and_contest_without_inventing_visits(H, oracle, monkeypatch, shape, bits):
    """HNSW_BLK_BITS forces a directory of 8 / 16 / 32 slots on a graph of 80 blocks: nearly every hop claims slots that other
    lanes of the same hop claim too, and blocks are evicted while their nodes are still around -- the regime in which a wrong
    ownership test would set a bit under another block's directory word.  Results must still be the oracle's, bit for bit
    (only the evaluations may grow)."""
    name, d, metric, ef = shape
    monkeypatch.setenv("HNSW_BLK_BITS", str(bits))
    if name == "bytes_l2":
        rng = np.random.default_rng(5)
        cen = rng.integers(20, 200, size=(16, d))
        X = np.clip(np.rint(cen[rng.integers(0, 16, 20000)] + rng.normal(0, 25, size=(20000, d))), 0, 218).astype(np.float32)
        Q = np.clip(np.rint(cen[rng.integers(0, 16, 60)] + rng.normal(0, 25, size=(60, d))), 0, 218).astype(np.float32)
    else:
        X = _clustered_np(20000, d, 21, 8)
        Q = _clustered_np(60, d, 22, 8)
    hg = H.Ohnsw.build_batch_bigarray(X, 16, 60, seed=3, metric=metric)
    hg.export()
    g = oracle.Graph(hg.n, hg.entry_point, hg.deg0, hg.nbr0, hg.upper)
    sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
    oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=10, ef=ef, ties=oracle.TIES_CANONICAL, counters=True)
    fd, fi = oracle.Functor.knn_batch(g, sp, Q, ef, 10, ties=oracle.TIES_CANONICAL, with_ids=True)
    hg.set_option("visited_blocks", 1)
    assert hg.visited_blocks(ef) == bits
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=ef, counters=True)
    np.testing.assert_array_equal(ids, oi)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(nh, onh)
    assert (nd.astype(np.int64) + len(g.upper) >= ond).all()
    gi, gd = H._search(hg, Q, ef, 10, H.FILL_BA, sem=H.SEM_FUNCTOR)
    np.testing.assert_array_equal(gi, fi)
    np.testing.assert_array_equal(gd.view(np.uint32), fd.view(np.uint32))
    hg.release()


def _sample_against_oracle(oracle, c, mode, n_sample, bound):
    H, hg = c["H"], c["hg"]
    hg.set_option("visited_blocks", mode)
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, c["k"], c["Q"], ef=c["ef"], counters=True)
    sel = np.random.default_rng(0).choice(len(c["Q"]), n_sample, replace=False)
    if "oracle" not in c:
        c["oracle"] = oracle.Ohnsw.knn_batch_bigarray(c["g"], c["sp"], c["Q"][sel], k=c["k"], ef=c["ef"], ties=oracle.TIES_CANONICAL, counters=True)
    oi, od, ond, onh = c["oracle"]
    np.testing.assert_array_equal(ids[sel], oi)
    np.testing.assert_array_equal(dist[sel].view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(nh[sel], onh)
    extra = nd[sel].astype(np.int64) - ond.astype(np.int64) + len(c["g"].upper)      # (the oracle counts search_one's start node once per upper layer)
    assert (extra >= 0).all()
    over = extra.sum() / ond.sum()
    print("  visited_blocks %2d: %.0f evaluations per query in the oracle, +%.1f %% on the device" % (mode, ond.mean(), 100 * over))
    if bound is not None:
        assert over <= bound, "re-evaluations %.3f above the bound %.3f" % (over, bound)
    return over, nd


def _clustered_case(oracle, H, n, d, metric, ef, k, centres, nq=2000, expected_ef=0):
    X = _clustered_gpu(n, d, 12, centres)
    Q = _clustered_gpu(nq, d, 112, centres)
    hg = H.Ohnsw.build_batch_bigarray(X, 32, 200, seed=1, metric=metric, expected_ef=expected_ef)
    hg.export()
    g = oracle.Graph(hg.n, hg.entry_point, hg.deg0, hg.nbr0, hg.upper)
    sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
    return dict(H=H, X=X, Q=Q, hg=hg, g=g, sp=sp, ef=ef, k=k)


@pytest.mark.parametrize("shape", [("c5_shape", 96, 0, 512, 10, 0.10), ("c3_shape", 100, 1, 256, 100, 0.06)], ids=lambda s: s[0])
def test_clustered_200k_parity_and_reevaluation_bound(H, oracle, shape):
    """the regime of round 4's weak point (clustered unit vectors, large ef) at moderate size: n = 200 000
    around 16 directions (12 500 nodes each, as many as a walk at ef 512 visits), M 32.  Both visited structures must return the
    oracle's bits; the blocks must stay within the bound of VERDICT item 1, and the handle's own measurement must pick them."""
    name, d, metric, ef, k, bound = shape
    c = _clustered_case(oracle, H, 200_000, d, metric, ef, k, centres=16)
    over_tags, _ = _sample_against_oracle(oracle, c, 0, 60, None)
    over_blk, nd_blk = _sample_against_oracle(oracle, c, 1, 60, bound)
    assert over_blk < over_tags
    _, nd_auto = _sample_against_oracle(oracle, c, -1, 60, bound)
    np.testing.assert_array_equal(nd_auto, nd_blk)                         # measured on 256 of its own vectors: the blocks
    c["hg"].release()


def test_structureless_vectors_keep_the_tag_cache(H, oracle):
    """N(0,1) unit vectors (BASELINE.md's prescription for C3 / C5) have no neighbourhoods a numbering could keep together: the
    blocks would forget more than the tags do, and the handle's measurement must say so."""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    x = torch.randn((200_000 + 1000, 96), generator=g, device=dev)
    x = (x / x.norm(dim=1, keepdim=True)).cpu().numpy()
    X, Q = x[:200_000], x[200_000:]
    hg = H.Ohnsw.build_batch_bigarray(X, 32, 200, seed=1)
    out = {}
    for mode in (0, 1, -1):
        hg.set_option("visited_blocks", mode)
        out[mode] = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=512, counters=True)
    for mode in (1, -1):
        for a, b in zip(out[0][:2], out[mode][:2]):
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
        np.testing.assert_array_equal(out[0][3], out[mode][3])             # hops
    assert out[1][2].sum() > out[0][2].sum()                               # the blocks re-evaluate more here ...
    np.testing.assert_array_equal(out[-1][2], out[0][2])                   # ... so left to itself the handle keeps the tags
    # ... and then does not keep the per-slot code table (n x max_degree0 x 4 bytes: 2.56 GB at 10 M nodes) it built
    # for the measurement: only the per-node codes (n x 4 bytes) stay, and hnsw_index_info.device_bytes says so
    n, S0 = 200_000, 64
    without_table = int(hg.info().device_bytes)
    hg.set_option("visited_blocks", 1)
    H.Ohnsw.knn_batch_bigarray(hg, 10, Q[:8], ef=512)
    assert int(hg.info().device_bytes) == without_table + n * S0 * 4          # an explicit 1 re-makes it (one fill kernel) ...
    hg.set_option("visited_blocks", -1)
    H.Ohnsw.knn_batch_bigarray(hg, 10, Q[:8], ef=512)                       # ... and the next measurement that chooses the tags drops it again
    assert int(hg.info().device_bytes) == without_table
    L = hg.locality_codes()                                                # introspection needs the per-node codes only
    np.testing.assert_array_equal(np.sort(L), np.arange(n))
    assert int(hg.info().device_bytes) == without_table
    hg.release()


def _timed(fn, reps):
    import time
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return ts


def test_prepared_shapes_are_saved_and_a_loaded_index_starts_at_full_speed(H, oracle, tmp_path, capfd):
    """the visited-structure decision (codes + measurement: 0.1-1.3 s) belongs to construction.
    hnsw_build with expected_ef makes it there; hnsw_index_save writes codes and decisions down (format 2); hnsw_index_load
    adopts them -- nothing is built or measured again (HNSW_DEBUG_VISITED prints a line per measurement: none after a load) --
    and prepares the saved shapes, so the first search of a loaded index is as fast as the steady state.  A format 1 file
    (no trailer) and a file whose codes are not a permutation still load and still give the same bits."""
    import os
    import struct
    os.environ["HNSW_DEBUG_VISITED"] = "1"
    try:
        n, d, ef, k = 200_000, 96, 512, 10
        c = _clustered_case(oracle, H, n, d, 0, ef, k, centres=16, nq=4000, expected_ef=ef)
        hg, Q = c["hg"], c["Q"]
        err = capfd.readouterr().err
        assert err.count("visited set for ef 512") == 1, err               # measured once, inside hnsw_build
        bits = hg.visited_blocks(ef)
        assert bits > 0
        base = int(hg.info().device_bytes)
        first = _timed(lambda: H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef), 1)[0]
        steady = sorted(_timed(lambda: H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef), 7))
        assert capfd.readouterr().err.count("visited set for") == 0        # ... and never again
        want = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
        path = str(tmp_path / "c.idx")
        hg.save(path)
        hg.release()
        lg = H.Hgraph.load(path)
        assert capfd.readouterr().err.count("visited set for") == 0        # adopted, not measured
        assert int(lg.info().device_bytes) == base                         # codes and per-slot table in place after the load
        lfirst = _timed(lambda: H.Ohnsw.knn_batch_bigarray(lg, k, Q, ef=ef), 1)[0]
        lsteady = sorted(_timed(lambda: H.Ohnsw.knn_batch_bigarray(lg, k, Q, ef=ef), 7))
        assert lg.visited_blocks(ef) == bits
        got = H.Ohnsw.knn_batch_bigarray(lg, k, Q, ef=ef, counters=True)
        for a, b in zip(want, got):
            np.testing.assert_array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
        print("  4000 queries at ef 512: built with expected_ef: first call %.2f ms, then %.2f ms (median of 7); loaded: first %.2f ms, then %.2f ms"
              % (1e3 * first, 1e3 * steady[3], 1e3 * lfirst, 1e3 * lsteady[3]))
        # the first call still allocates the handle's scratch for this batch size (a few hipMalloc): well under the 100+ ms of
        # a measurement, and bounded here at half a steady call
        assert first <= 1.5 * steady[3] and lfirst <= 1.5 * lsteady[3], (first, steady, lfirst, lsteady)
        lg.release()
        raw = open(path, "rb").read()
        at = raw.rindex(b"PREP")
        # (a) format 1: the file as rounds 1-5 wrote it
        p1 = str(tmp_path / "v1.idx")
        open(p1, "wb").write(raw[:8] + struct.pack("<I", 1) + raw[12:at])
        # (b) codes that are not a permutation (two equal entries)
        n_dec = struct.unpack("<I", raw[at + 4:at + 8])[0]
        codes_at = at + 8 + 12 * n_dec + 4
        assert struct.unpack("<I", raw[codes_at - 4:codes_at])[0] == 1 and len(raw) == codes_at + 4 * n
        p2 = str(tmp_path / "bad.idx")
        open(p2, "wb").write(raw[:codes_at] + raw[codes_at + 4:codes_at + 8] + raw[codes_at + 4:])
        for pth in (p1, p2):
            g2 = H.Hgraph.load(pth)
            got = H.Ohnsw.knn_batch_bigarray(g2, k, Q[:500], ef=ef)
            np.testing.assert_array_equal(got[0], want[0][:500])
            np.testing.assert_array_equal(got[1].view(np.uint32), want[1][:500].view(np.uint32))
            assert g2.visited_blocks(ef) == bits                           # measured again on demand, same answer
            g2.release()
        assert capfd.readouterr().err.count("visited set for ef 512") == 2
    finally:
        os.environ.pop("HNSW_DEBUG_VISITED", None)


def test_c5_clustered_full_size(H, oracle):
    """BASELINE configuration 5's shape (n = 10 M, d 96, M 32, ef 512) on bench.py's clustered generator
    (others.C5_clustered): a sample against the oracle -- ids, distance bits, hops -- and the device's re-evaluations at most
    10 % of the oracle's count (round 4: 40 %).  About 60 s, 12 GB of host memory."""
    try:
        import psutil
        free = psutil.virtual_memory().available
    except Exception:
        free = None
    if free is not None and free < 24 << 30:
        pytest.skip("needs about 12 GB of host memory; %.1f GB free" % (free / 2 ** 30))
    c = _clustered_case(oracle, H, 10_000_000, 96, 0, 512, 10, centres=256, nq=10_000, expected_ef=512)
    # built with expected_ef = 512, the index has its codes, its decision and its code objects: the first search
    # call is a steady-state call (round 5: + 1.2 s inside it), and device_bytes states the 2.6 GB of tables
    hg, Q = c["hg"], c["Q"]
    assert hg.visited_blocks(512) > 0
    assert int(hg.info().device_bytes) >= 10_000_000 * (96 * 4 + 64 * 4 + (1 + 64) * 4)
    first = _timed(lambda: H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=512), 1)[0]
    steady = sorted(_timed(lambda: H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=512), 5))
    print("  C5 clustered, 10 000 queries at ef 512: first call %.2f ms, then %.2f ms (median of 5)" % (1e3 * first, 1e3 * steady[2]))
    assert first <= 1.10 * steady[2], (first, steady)
    over, _ = _sample_against_oracle(oracle, c, -1, 50, 0.10)
    c["hg"].release()
