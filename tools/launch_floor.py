#!/usr/bin/env python3
"""How fast can the chip start one-wave workgroups of the search kernel?  100 k queries at ef = 1, 8, 32 on the C2
index (short walks): the time per query is then mostly launch + prologue, i.e. the floor under a large batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ocaml_hnsw_amd as H
import bench
dev = torch.device("cuda", 0)
n, d = 1000000, 128
cache = "/tmp/ab_c2_%d.idx" % n
if os.path.exists(cache):
    hg = H.Hgraph.load(cache)
else:
    hg = H.Ohnsw.build_batch_bigarray(bench.make_sift_like(n, d, 1, dev).cpu().numpy(), 16, 200, seed=1); hg.save(cache)
stream = torch.cuda.current_stream()
for order in (0, -1):
    hg.set_option("order_queries", order)
    for ef in (1, 8, 32, 64, 128):
        for nq in (10000, 100000):
            k = 1
            Qd = bench.make_sift_like(nq, d, 2, dev)
            ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
            nh = torch.zeros(nq, dtype=torch.int32, device=dev)
            def go(c=False):
                H.search_batch_device(hg, Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), 0, nh.data_ptr() if c else 0, 0, stream.cuda_stream)
            go(True); torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream); go(); b.record(stream); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
            ms = float(np.median(ts))
            print("order=%2d ef=%3d nq=%6d: %.4f ms  %.1f Mq/s  hops %.1f  -> %.1f ns per workgroup chip-wide" % (order, ef, nq, ms, nq / ms / 1e3, float(nh.float().mean()), ms * 1e6 / nq), flush=True)
