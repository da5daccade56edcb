#!/usr/bin/env python3
"""Debugging aid for hnsw_hop_asm.hip.h (library built with -DHNSW_ASM_DEBUG): run the hand-written loop for the first
N hops of every query and hipcc's loop for the rest; the first N at which results differ from N = 0 is the faulty hop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ocaml_hnsw_amd as H
rng = np.random.default_rng(5)
n, d, nq, ef, k = int(os.environ.get("N", 3000)), 128, int(os.environ.get("NQ", 8)), int(os.environ.get("EF", 100)), 10
if os.environ.get("KIND") == "sift":
    import torch, bench
    dev = torch.device("cuda", 0)
    X = bench.make_sift_like(n, d, 1, dev, int(os.environ.get("CENTRES", 64)), 25.0).cpu().numpy()
    Q = bench.make_sift_like(nq, d, 2, dev, int(os.environ.get("CENTRES", 64)), 25.0).cpu().numpy()
else:
    X = rng.integers(0, 219, size=(n, d)).astype(np.float32)
    Q = rng.integers(0, 219, size=(nq, d)).astype(np.float32)
hg = H.Ohnsw.build_batch_bigarray(X, 16, 100, seed=1)
hg.set_option("order_queries", 1)
ref = None
for N in [0] + list(range(int(os.environ.get("MINN", 1)), int(os.environ.get("MAXN", 40)), int(os.environ.get("STEP", 1)))):
    hg.set_option("lds_pad", N)
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
    cur = (ids.copy(), dist.copy(), nd.copy(), nh.copy())
    if ref is None:
        ref = cur
        print("N=0: n_dist", nd.tolist(), "hops", nh.tolist())
        continue
    same = all(np.array_equal(a, b) for a, b in zip(cur, ref))
    bad = [q for q in range(nq) if not (np.array_equal(ids[q], ref[0][q]) and nd[q] == ref[2][q] and nh[q] == ref[3][q])]
    if not same or os.environ.get("VERBOSE"):
        print("N=%d: %s  n_dist %s hops %s  bad queries %s" % (N, "same" if same else "DIFFERENT", nd.tolist(), nh.tolist(), bad), flush=True)
