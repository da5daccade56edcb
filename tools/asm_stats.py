import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import ocaml_hnsw_amd as H, bench
dev = torch.device("cuda", 0)
n, d, nq, ef, k = 1000000, 128, 2000, 128, 10
hg = H.Hgraph.load("/tmp/ab_c2_1000000.idx") if os.path.exists("/tmp/ab_c2_1000000.idx") else H.Ohnsw.build_batch_bigarray(bench.make_sift_like(n, d, 1, dev).cpu().numpy(), 16, 200, seed=1)
Qd = bench.make_sift_like(nq, d, 2, dev)
ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
nd = torch.zeros(nq, dtype=torch.int32, device=dev); nh = torch.zeros(nq, dtype=torch.int32, device=dev); st = torch.zeros(nq, dtype=torch.int32, device=dev)
H.search_batch_device(hg, Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), nd.data_ptr(), nh.data_ptr(), st.data_ptr(), 0)
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.uint32)
print("hops %.1f  prefetch hits %.1f  unsafe pops %.1f per query" % (nh.float().mean().item(), ((s >> 8) & 0xFFF).mean(), (s >> 20).mean()))
