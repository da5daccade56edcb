#!/usr/bin/env python3
"""One-off soak (not collected by pytest): many more random configurations than tests/test_gpu_parity.py runs,
search and ordering options included, GPU against the oracle bit for bit.
    python tests/soak_random_parity.py [trials] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    import torch  # noqa: F401  (one HIP runtime per process)
except ImportError:
    pass
import ocaml_hnsw_amd as H  # noqa: E402
from oracle import oracle as o  # noqa: E402

o.build(); o.lib(); H.load()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
# every boundary between slot counts (W in 1 / 2 / 3 / 4 / 6 / 8 / 16 registers: ef 64 | 128 | 192 | 256 | 384 | 512 | 1024) from both sides
efs = [1, 2, 3, 7, 31, 63, 64, 65, 100, 127, 128, 129, 160, 191, 192, 193, 200, 255, 256, 257, 300, 320, 383, 384, 385, 511, 512, 513, 700, 1024]
ds = [1, 2, 3, 4, 5, 7, 16, 31, 33, 63, 64, 65, 96, 100, 127, 128, 129, 200, 255, 256, 257, 300, 511, 513, 784]
if os.environ.get("SOAK_DS"):       # e.g. SOAK_DS=65-128 SOAK_EFS=1-256: one kernel family (here: the hand-scheduled loops' domain)
    lo, hi = (int(x) for x in os.environ["SOAK_DS"].split("-"))
    ds = list(range(lo, hi + 1))
if os.environ.get("SOAK_EFS"):
    lo, hi = (int(x) for x in os.environ["SOAK_EFS"].split("-"))
    efs = [e for e in efs if lo <= e <= hi] + [int(x) for x in np.random.default_rng(7).integers(lo, hi + 1, 8)]
bad = 0
for trial in range(trials):
    n = int(rng.integers(1, 1500))
    d = int(rng.choice(ds))
    M = int(rng.choice([2, 3, 5, 8, 12, 16, 24, 32]))
    ef = int(rng.choice(efs))
    k = int(rng.integers(1, ef + 1))
    metric = int(rng.integers(0, 2))
    id_base = int(rng.integers(0, 2))
    levels = int(rng.choice([2, 3, 4, 50, 1000]))
    X = rng.integers(0, levels, size=(n, d)).astype(np.float32)
    # byte-valued data (every value an integer 0..255) is searched through the library's byte rows; a third of the
    # inner-product trials keep such data as it is (no normalisation) so that path meets both metrics
    ip_bytes = metric == 1 and levels <= 50 and rng.integers(0, 3) == 0
    if (metric == 1 and not ip_bytes) or levels == 1000:
        X = X + rng.uniform(0, 0.5, size=X.shape).astype(np.float32)
    if metric == 1 and not ip_bytes:
        X /= np.maximum(np.linalg.norm(X, axis=1, keepdims=True), 1e-6)
    sp = (o.Space.ip if metric else o.Space.l2)(X, arith=o.TREE16)
    g = o.build_ohnsw(sp, M, int(rng.integers(4, 80)), seed=int(rng.integers(0, 10000)))
    up = [(nodes + id_base, deg, np.where(nbr >= 0, nbr + id_base, -1)) for nodes, deg, nbr in g.upper]
    hg = H.Hgraph(X, g.deg0, np.where(g.nbr0 >= 0, g.nbr0 + id_base, -1), up, entry_point=g.entry_point + id_base,
                  id_base=id_base, max_degree=M, metric=metric)
    hg.set_option("order_queries", int(rng.integers(-1, 2)))
    hg.set_option("vt_bits", int(rng.choice([0, 4, 6, 9, 11, 13])))
    hg.set_option("byte_rows", int(rng.integers(0, 4) != 0))        # (no effect where the data has no byte copy)
    hg.set_option("split_rows", int(rng.integers(0, 4) != 0))       # (no effect where the row shape has no split copy)
    os.environ["HNSW_BLK_BITS"] = str(int(rng.choice([0, 3, 4, 6])))  # Visited as bitmap blocks (ef > 128): the directory that fits, or a tiny one
    hg.set_option("visited_blocks", int(rng.choice([0, 1, 1])))
    nq = int(rng.choice([1, 3, 17, 64, 200]))
    Q = (X[rng.integers(0, n, nq)] + (rng.integers(0, 2, size=(nq, d)) if metric == 0 else 0)).astype(np.float32)
    if rng.integers(0, 3) == 0:
        Q[rng.integers(0, nq)] += np.float32(0.5)                      # a query that is not byte-valued among byte-valued ones
    out = None
    if rng.integers(0, 3) == 0:                                        # page-locked matrices: read and written by the device in place
        Qp = H.host_empty(Q.shape, np.float32)
        Qp[:] = Q
        Q = Qp
        if rng.integers(0, 2) == 0:
            out = (H.host_empty((nq, k), np.int32), H.host_empty((nq, k), np.float32))
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True, out=out)
    oi, od, ond, onh = o.Ohnsw.knn_batch_bigarray(g, sp, Q, k=k, ef=ef, ties=o.TIES_CANONICAL, counters=True)
    oi = np.where(oi >= 0, oi + id_base, -1)
    ok = np.array_equal(ids, oi) and np.array_equal(dist.view(np.uint32), od.view(np.uint32)) and np.array_equal(nh, onh)
    fd = H.Ba.knn_batch(hg, Q, ef, k)
    ofd = o.Functor.knn_batch(g, sp, Q, ef, k, ties=o.TIES_CANONICAL)
    ok = ok and np.array_equal(fd.view(np.uint32), ofd.view(np.uint32))
    layer = int(rng.integers(0, g.max_layer + 1))
    nodes = np.arange(n) if layer == 0 else g.upper[layer - 1][0]
    st = rng.choice(nodes, size=int(rng.integers(1, min(ef, len(nodes), 70) + 1)), replace=False).tolist()
    got = H.Ohnsw.search_k(hg, layer, [[s_ + id_base for s_ in st]], Q[:1], min(k, ef), ef=ef)[0]
    want = o.Ohnsw.search_k(g, sp, st, Q[0], ef, layer=layer, ties=o.TIES_CANONICAL)[:min(k, ef)]
    ok = ok and [a - id_base for a, _ in got] == [a for a, _ in want]
    if not ok:
        bad += 1
        print("MISMATCH", dict(trial=trial, n=n, d=d, M=M, ef=ef, k=k, metric=metric, id_base=id_base, levels=levels, nq=nq, layer=layer), flush=True)
    hg.release()
    if trial % 50 == 49:
        print("trial %d: %d mismatches so far" % (trial + 1, bad), flush=True)
print("soak done: %d trials, %d mismatches" % (trials, bad))
sys.exit(1 if bad else 0)
