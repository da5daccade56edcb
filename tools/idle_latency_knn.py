#!/usr/bin/env python3
"""Single-query latency of the host-buffer call (Ohnsw.knn: one query per call, test/test.ml:122) on the C2 index: median of
200 calls at ef = k = 10 and at ef 128, with the small-call path (a page-locked block of the handle's: HNSW_SMALL_CALLS=1, the
default) and without.  (GPU box)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    import ocaml_hnsw_amd as H
    import bench
    dev = torch.device("cuda", 0)
    X = bench.make_sift_like(1000000, 128, 1, dev).cpu().numpy()
    hg = H.Ohnsw.build_batch_bigarray(X, 16, 200, seed=1)
    Q = bench.make_sift_like(256, 128, 2, dev).cpu().numpy()
    for ef, k in ((10, 10), (128, 10)):
        for nq in (1, 64):
            ts = []
            for i in range(220):
                q = Q[i % 190:i % 190 + nq]
                t = time.perf_counter()
                H.Ohnsw.knn_batch_bigarray(hg, k, q, ef=ef)
                ts.append(time.perf_counter() - t)
            ts = sorted(ts[20:])
            print("  HNSW_SMALL_CALLS=%s ef %3d, %2d queries per call: median %.1f us (min %.1f)" % (os.environ.get("HNSW_SMALL_CALLS", "1"), ef, nq, 1e6 * ts[len(ts) // 2], 1e6 * ts[0]), flush=True)
    sys.exit(0)
for v in ("0", "1"):
    subprocess.call([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, HNSW_SMALL_CALLS=v))
