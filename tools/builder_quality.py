#!/usr/bin/env python3
"""Does the batched device builder cost recall where it matters?  The harder SIFT-like set (256 blobs, sigma 40) at
n = 100 k: recall@10 at ef 128 / 192 on the graph built with the default batching and on the graph built one node at
a time (max_batch = 1 = Ohnsw.insert, pinned link for link against the reference, lib/ohnsw.ml:766-837)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ocaml_hnsw_amd as H
import bench
dev = torch.device("cuda", 0)
n, d, M, efc, k, nq = int(os.environ.get("N", 100000)), 128, 16, 200, 10, 2000
cen, sig = int(os.environ.get("CENTRES", 256)), float(os.environ.get("SIGMA", 40))
Xd = bench.make_sift_like(n, d, 1, dev, cen, sig); Qd = bench.make_sift_like(nq, d, 2, dev, cen, sig)
X, Q = Xd.cpu().numpy(), Qd.cpu().numpy()
gt = bench.brute_force_topk(Xd, Qd, k)
CONFIGS = (("batched (default)", {}), ("batch_div 64", {"batch_div": 64}), ("max_batch 256", {"max_batch": 256}), ("max_batch 16", {"max_batch": 16}), ("sequential (max_batch 1)", {"max_batch": 1}))
for name, kw in CONFIGS[:int(os.environ.get("NCONF", 5))]:
    if name.startswith("sequential") and os.environ.get("NO_SEQ"):
        continue
    t = time.time()
    hg = H.Ohnsw.build_batch_bigarray(X, M, efc, seed=1, **kw)
    bt = time.time() - t
    out = []
    for ef in (64, 128, 192, 256):
        ids, _, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
        out.append("ef %d: recall %.4f (%.0f evals)" % (ef, bench.recall_ids(ids, gt), nd.mean()))
    st = hg.layer_stats(0) if hasattr(hg, "layer_stats") else None
    print("%-26s build %.1fs  %s  %s" % (name, bt, "  ".join(out), ("deg mean %.1f isolated %d" % (st.mean_degree, st.num_isolated)) if st else ""), flush=True)
    hg.release()
