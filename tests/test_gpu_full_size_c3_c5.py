"""GPU parity at BASELINE.json's full C3 and C5 sizes (the graph is built on the GPU, exported, and a
sample of the 10 000-query batch is checked bit for bit against the oracle on the same graph: ids,
distance bits, hop counts; the whole batch through size-independent properties).

  C3  "GloVe-like": n = 1 183 514, d = 100, N(0,1) L2-normalised, inner product (distance 1 - <a,b>),
      M 32, efConstruction 200, ef 256, k 100                                   (seed 2)
  C5  "DEEP-like":  n = 10 000 000, d = 96, N(0,1) normalised, L2, M 32, efConstruction 200,
      ef 512, k 10 -- 6.4 GB of index in HBM, the HBM-bandwidth configuration   (seed 3)

Both shapes have ragged rows (d not a multiple of 64: the masked-row kernel variants) and W in 4 / 8
key registers per lane.  Budget on the one-GPU box: C3 about 15 s, C5 about 90 s (45 s of it the build,
3.8 GB of host vectors + 2.6 GB of exported adjacency); C5 is skipped, with the reason, when the host
has less than 24 GB of free memory.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit_vectors(n, d, seed):
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = np.empty((n, d), np.float32)
    step = 1 << 20
    for s in range(0, n, step):
        m = min(step, n - s)
        x = torch.randn((m, d), generator=g, device=dev)
        out[s:s + m] = (x / x.norm(dim=1, keepdim=True)).cpu().numpy()
    return out


def _build(oracle, n, d, metric, M, efc, ef, k, seed, nq=10_000):
    import ocaml_hnsw_amd as H
    H.load()
    assert H.device_count() >= 1
    t0 = time.time()
    X = _unit_vectors(n, d, seed)
    Q = _unit_vectors(nq, d, seed + 100)
    t1 = time.time()
    hg = H.Ohnsw.build_batch_bigarray(X, M, efc, seed=1, metric=metric)
    t2 = time.time()
    hg.export()
    g = oracle.Graph(hg.n, hg.entry_point, hg.deg0, hg.nbr0, hg.upper)
    sp = (oracle.Space.ip if metric else oracle.Space.l2)(X, arith=oracle.TREE16)
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
    print("full size n=%d d=%d metric=%d: data %.1fs, GPU build %.1fs, export + first search %.1fs" %
          (n, d, metric, t1 - t0, t2 - t1, time.time() - t2))
    return dict(H=H, X=X, Q=Q, hg=hg, g=g, sp=sp, ids=ids, dist=dist, nd=nd, nh=nh, ef=ef, k=k, n=n, metric=metric)


def _sample_parity(oracle, c, n_sample):
    sel = np.random.default_rng(0).choice(len(c["Q"]), n_sample, replace=False)
    oi, od, ond, onh = oracle.Ohnsw.knn_batch_bigarray(c["g"], c["sp"], c["Q"][sel], k=c["k"], ef=c["ef"],
                                                       ties=oracle.TIES_CANONICAL, counters=True)
    np.testing.assert_array_equal(c["ids"][sel], oi)
    np.testing.assert_array_equal(c["dist"][sel].view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(c["nh"][sel], onh)                  # the same candidates were expanded
    extra = c["nd"][sel].astype(np.int64) - ond.astype(np.int64)      # the lossy visited cache only ADDS evaluations
    assert (extra >= 0).all() and extra.sum() < 0.08 * ond.sum()
    print("  sample of %d: ids, distance bits and hop counts equal; %.0f evaluations/query (oracle), GPU re-evaluations +%.2f%%"
          % (n_sample, ond.mean(), 100.0 * extra.sum() / ond.sum()))


def _whole_batch_properties(c):
    ids, dist, k = c["ids"], c["dist"], c["k"]
    assert (ids < c["n"]).all()
    short = np.where((ids < 0).any(1))[0]
    assert len(short) <= 10                                            # isolated nodes, as in the C2 test
    full = np.setdiff1d(np.arange(len(ids)), short)
    assert (np.diff(dist[full], axis=1) >= 0).all()                    # ascending (lib/ohnsw.ml:886-893)
    assert all(len(set(r)) == k for r in ids[full[:1000]].tolist())    # no node twice
    # distances are what they claim to be: 1e-5 relative (north star) against fp64
    j = full[::197]
    V = c["X"][ids[j]].astype(np.float64)
    q = c["Q"][j][:, None, :].astype(np.float64)
    want = (1.0 - (V * q).sum(-1)) if c["metric"] else np.sqrt(((V - q) ** 2).sum(-1))
    np.testing.assert_allclose(dist[j], want, rtol=1e-5, atol=1e-6)
    # batch independence (lib/ohnsw.ml:883-895 is a pure map over the query columns)
    H, hg = c["H"], c["hg"]
    sub = np.arange(0, len(ids), 7)[::-1].copy()
    ri, rd = H.Ohnsw.knn_batch_bigarray(hg, k, c["Q"][sub], ef=c["ef"])
    np.testing.assert_array_equal(ri, ids[sub])
    np.testing.assert_array_equal(rd.view(np.uint32), dist[sub].view(np.uint32))


@pytest.fixture(scope="module")
def c3(oracle):
    return _build(oracle, 1_183_514, 100, 1, 32, 200, 256, 100, seed=2)


def test_c3_sample_bit_parity_with_oracle(oracle, c3):
    _sample_parity(oracle, c3, 200)


def test_c3_whole_batch_properties(c3):
    _whole_batch_properties(c3)


def test_c3_functor_path_equals_oracle_on_a_sample(oracle, c3):
    """Hnsw.Ba.knn_batch on the C3 shape (inner product: negative distances, no exact ties): the functor
    rule and the Ohnsw rule coincide without ties, so both oracles must agree with the GPU."""
    import ocaml_hnsw_amd as A
    sel = np.arange(0, 10_000, 211)
    gi, gd = A._search(c3["hg"], c3["Q"][sel], c3["ef"], c3["k"], A.FILL_BA, sem=A.SEM_FUNCTOR)
    od, oi = oracle.Functor.knn_batch(c3["g"], c3["sp"], c3["Q"][sel], c3["ef"], c3["k"], ties=oracle.TIES_CANONICAL, with_ids=True)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))


@pytest.fixture(scope="module")
def c5(oracle):
    try:
        import psutil
        free = psutil.virtual_memory().available
    except Exception:
        free = None
    if free is not None and free < 24 << 30:
        pytest.skip("C5 needs about 12 GB of host memory for the vectors, the exported graph and the oracle's "
                    "copies; only %.1f GB are free on this host" % (free / 2 ** 30))
    return _build(oracle, 10_000_000, 96, 0, 32, 200, 512, 10, seed=3)


def test_c5_sample_bit_parity_with_oracle(oracle, c5):
    _sample_parity(oracle, c5, 100)


def test_c5_whole_batch_properties(c5):
    _whole_batch_properties(c5)


def test_c5_self_queries(c5):
    H, hg, X = c5["H"], c5["hg"], c5["X"]
    j = np.arange(0, c5["n"], 4999)
    ids, dist = H.Ohnsw.knn_batch_bigarray(hg, 1, X[j], ef=c5["ef"])
    hit = ids[:, 0] == j
    assert hit.mean() > 0.3                                             # structureless data (recall@10 is 0.33 here): the walk need not reach every point
    assert (dist[hit, 0] == 0).all()
