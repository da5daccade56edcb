#!/usr/bin/env python3
"""Ordering pre-pass (descent kernel + sort) and search kernel times of the library's own HIP events against the batch
size, C2 index (run on the GPU box): is the pre-pass one pass or two?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ocaml_hnsw_amd as H
import bench

dev = torch.device("cuda", 0)
n, d, M, efc, k, ef = 1000000, 128, 16, 200, 10, 128
Xd = bench.make_sift_like(n, d, 1, dev, 4096, 25.0)
hg = H.Ohnsw.build_batch_bigarray(Xd.cpu().numpy(), M, efc, seed=1)
stream = torch.cuda.current_stream()
for nq in [int(x) for x in os.environ.get("NQS", "2048,4096,5000,6144,8192,9000,10000,12000,16384,20000").split(",")]:
    Qd = bench.make_sift_like(nq, d, 2, dev, 4096, 25.0)
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    hg.set_option("order_queries", 1)
    hg.set_option("time_kernels", 1)
    hg.kernel_times()
    for _ in range(12):
        H.search_batch_device(hg.to_device(0), Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), 0, 0, 0, stream.cuda_stream)
    torch.cuda.synchronize()
    sm, pm, calls = hg.kernel_times()
    hg.set_option("time_kernels", 0)
    print("nq=%6d: pre-pass %.1f us, search kernel %.1f us (%d calls)" % (nq, 1e3 * pm, 1e3 * sm, calls), flush=True)
