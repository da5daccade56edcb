#!/usr/bin/env python3
"""A lone 64-query batch on the C2 index against ef (host wall time around one device call, synchronised): the fixed cost
(launch + descent) at ef = 1 and the layer-0 walk growing with ef.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocaml_hnsw_amd as H
import bench
dev = torch.device("cuda", 0)
n, d, M, efc = 1000000, 128, 16, 200
Xd = bench.make_sift_like(n, d, 1, dev, 4096, 25.0)
hg = H.Ohnsw.build_batch_bigarray(Xd.cpu().numpy(), M, efc, seed=1)
stream = torch.cuda.current_stream()
nq = 64
Qd = bench.make_sift_like(nq, d, 2, dev, 4096, 25.0)
for ef, k in ((1, 1), (2, 1), (4, 1), (10, 10), (16, 10), (32, 10), (64, 10), (128, 10)):
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    nd = torch.zeros(nq, dtype=torch.int32, device=dev); nh = torch.zeros(nq, dtype=torch.int32, device=dev)
    ts = []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        H.search_batch_device(hg.to_device(0), Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), nd.data_ptr(), nh.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("ef %3d: %.1f us median (min %.1f), hops %.1f, evals %.1f" % (ef, 1e6 * ts[len(ts) // 2], 1e6 * ts[0], nh.float().mean().item(), nd.float().mean().item()), flush=True)
