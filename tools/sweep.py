#!/usr/bin/env python3
"""Build the C2 index once, then time the search kernel over batch sizes / visited-cache sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ocaml_hnsw_amd as H
import bench

dev = torch.device("cuda", 0)
n, d, M, efc = int(os.environ.get("N", 1000000)), 128, 16, 200
sigma = float(os.environ.get("SIGMA", 25)); centres = int(os.environ.get("CENTRES", 4096))
Xd = bench.make_sift_like(n, d, 1, dev, centres, sigma)
X = Xd.cpu().numpy()
t = time.time(); hg = H.Ohnsw.build_batch_bigarray(X, M, efc, seed=1); print("build %.2fs" % (time.time() - t), flush=True)
stream = torch.cuda.current_stream()

def run(nq, ef, k=10, vt=0, reps=5):
    Qd = bench.make_sift_like(nq, d, 2, dev, centres, sigma)
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    nd = torch.zeros(nq, dtype=torch.int32, device=dev); nh = torch.zeros(nq, dtype=torch.int32, device=dev)
    hg.set_option("vt_bits", vt)
    def go(c=False):
        H.search_batch_device(hg, Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), nd.data_ptr() if c else 0, nh.data_ptr() if c else 0, 0, stream.cuda_stream)
    go(True); torch.cuda.synchronize()
    ndm, nhm = nd.float().mean().item(), nh.float().mean().item()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); go(); b.record(stream); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    bq = ndm * (4 * d + 4) + nhm * 4 * 2 * M + 4 * d + 8 * k   # uses GPU n_dist (incl. re-evals)
    ns = min(500, nq)
    gt = bench.brute_force_topk(Xd, Qd[:ns], k)
    rec = bench.recall_ids(ids.cpu().numpy()[:ns], gt)
    print("nq=%7d ef=%4d vt=%2d: %8.3f ms  %10.0f q/s  n_dist(gpu)=%.0f hops=%.0f  gpu-bytes %.2f TB/s  recall %.3f" %
          (nq, ef, vt, ms, nq / ms * 1e3, ndm, nhm, bq * nq / ms / 1e9, rec), flush=True)

for spec in sys.argv[1:]:
    nq, ef, vt = (int(x) for x in spec.split(","))
    run(nq, ef, vt=vt)

if os.environ.get("PREF_STATS"):
    nq, ef, k = 10000, 128, 10
    Qd = bench.make_sift_like(nq, d, 2, dev, centres, sigma)
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    nh = torch.zeros(nq, dtype=torch.int32, device=dev); st = torch.zeros(nq, dtype=torch.int32, device=dev)
    H.search_batch_device(hg, Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), 0, nh.data_ptr(), st.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    hits = (st >> 8).float().sum().item(); hops = nh.float().sum().item()
    print("prefetch hit rate: %.3f (%d hits / %d hops)" % (hits / hops, hits, hops))
    # host-buffer (PCIe-inclusive) entry point
    Qh = Qd.cpu().numpy()
    import time
    H.Ohnsw.knn_batch_bigarray(hg, k, Qh, ef=ef)
    t = time.perf_counter()
    for _ in range(5):
        H.Ohnsw.knn_batch_bigarray(hg, k, Qh, ef=ef)
    dt = (time.perf_counter() - t) / 5
    print("host-buffer hnsw_search_batch (H2D + kernel + D2H + sync): %.3f ms/batch = %.0f q/s" % (dt * 1e3, nq / dt))
