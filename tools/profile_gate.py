#!/usr/bin/env python3
"""The harder SIFT-like set of bench.py's `secondary` (256 blobs, sigma 40; n = 1 M, d = 128, M 16, efConstruction 200) at the
ef of its recall gate, alone in a process: what tools/profile_cmd.sh wraps to give the gate's kernels -- byte rows
hnsw_search_kernel<2,4,3,0,0,2,0> at ef 176 (W in three registers), float32 rows <2,4,3,0,0,1,*> -- a rocprofv3 summary of
their own (the gate kernel had none).  20 launches of the 10 000-query batch per row format, queries
resident in HBM; the evaluation / hop means of the batch are printed for the per-hop figures.

    tools/profile_cmd.sh r06_gate trace,inst,wait,fetch python3 $PWD/tools/profile_gate.py [ef]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import ocaml_hnsw_amd as H  # noqa: E402

ef = int(sys.argv[1]) if len(sys.argv) > 1 else 176
n, d, nq, k = 1_000_000, 128, 10_000, 10
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
X = bench.make_sift_like(n, d, seed=1, device=dev, n_centres=256, sigma=40.0)
Q = bench.make_sift_like(nq, d, seed=2, device=dev, n_centres=256, sigma=40.0)
hg = H.Ohnsw.build_batch_bigarray(X.cpu().numpy(), 16, 200, seed=1, device=0, expected_ef=ef)
ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
nd = torch.zeros(nq, dtype=torch.int32, device=dev)
nh = torch.zeros(nq, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream()
for rows in (1, 0):
    hg.set_option("byte_rows", rows)
    H.search_batch_device(hg, Q.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), nd.data_ptr(), nh.data_ptr(), 0, st.cuda_stream)
    torch.cuda.synchronize()
    print("ef %d, %s rows: %.1f evaluations and %.1f hops per query (device counters), kernel %s" %
          (ef, "byte" if rows else "float32", nd.float().mean().item(), nh.float().mean().item(),
           bench.search_kernel_name(d, ef, 0, 0, 2 if rows else -1, hg.visited_blocks(ef))), flush=True)
    for _ in range(20):
        H.search_batch_device(hg, Q.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), 0, 0, 0, st.cuda_stream)
    torch.cuda.synchronize()
