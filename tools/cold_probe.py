#!/usr/bin/env python3
"""Where does a process's FIRST search go?  (GPU box)  Times, after an index build, the first 1-query device call (the search
kernels' code object), the first ordered call (the ordering pre-pass's), the first 10 k device call (its scratch) and the first
host-buffer call (the handle's stream, flag word, scratch).  With HNSW_WARM_UP=0 (round 4's behaviour): 1.4 + 5.2 + 0.4 + 0.9 ms;
by default index construction has paid the first two and the handle's state already (warm_up, hnsw_capi.hip).
    python tools/cold_probe.py [steps|host]"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import ocaml_hnsw_amd as H, bench
dev = torch.device("cuda", 0)
n, d, nq, k, ef = 1000000, 128, 10000, 10, 128
X = bench.make_sift_like(n, d, 1, dev).cpu().numpy()
hg = H.Ohnsw.build_batch_bigarray(X, 16, 200, seed=1)
Qd = bench.make_sift_like(nq, d, 2, dev)
ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream()
def t(f, name):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); print("%-50s %.3f ms" % (name, 1e3 * (time.perf_counter() - t0)), flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "steps"
if mode == "steps":
    t(lambda: H.search_batch_device(hg, Qd.data_ptr(), 1, d, 1, 1, ids.data_ptr(), dd.data_ptr(), 0, 0, 0, st.cuda_stream), "device call, 1 query ef 1 (variant TU load)")
    t(lambda: H.search_batch_device(hg, Qd.data_ptr(), 1, d, 128, 10, ids.data_ptr(), dd.data_ptr(), 0, 0, 0, st.cuda_stream), "device call, 1 query ef 128")
    hg.set_option("order_queries", 1)
    t(lambda: H.search_batch_device(hg, Qd.data_ptr(), 1, d, 128, 10, ids.data_ptr(), dd.data_ptr(), 0, 0, 0, st.cuda_stream), "ordered device call, 1 query (order TU load)")
    hg.set_option("order_queries", -1)
    t(lambda: H.search_batch_device(hg, Qd.data_ptr(), nq, d, 128, 10, ids.data_ptr(), dd.data_ptr(), 0, 0, 0, st.cuda_stream), "device call, 10 k (scratch for the ordering)")
    t(lambda: H.search_batch_device(hg, Qd.data_ptr(), nq, d, 128, 10, ids.data_ptr(), dd.data_ptr(), 0, 0, 0, st.cuda_stream), "device call, 10 k again")
Qh = H.host_empty((nq, d), np.float32); Qh[:] = Qd.cpu().numpy()
hi = H.host_empty((nq, k), np.int32); hd = H.host_empty((nq, k), np.float32)
t(lambda: H.Ohnsw.knn_batch_bigarray(hg, k, Qh, ef=ef, out=(hi, hd)), "host call, 10 k (stream, flag, scratch)")
t(lambda: H.Ohnsw.knn_batch_bigarray(hg, k, Qh, ef=ef, out=(hi, hd)), "host call, 10 k again")
