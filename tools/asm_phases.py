import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ocaml_hnsw_amd as H, bench
dev = torch.device("cuda", 0)
n, d, ef, k = 1000000, 128, int(os.environ.get("EF", 128)), 10
kw = dict(n_centres=256, sigma=40.0) if int(os.environ.get("HARD", 0)) else {}     # HARD=1: the harder SIFT-like set of bench.py's `secondary`
hg = H.Ohnsw.build_batch_bigarray(bench.make_sift_like(n, d, 1, dev, **kw).cpu().numpy(), 16, 200, seed=1)   # built by the library under test (1 s): no stale cache
names = ["pop + adjacency row", "visited filter + speculative fetch + compaction", "round: ids, row loads, dot products, reduction, accept", "insertions"]
for nq in (64, 10000):
    hg.set_option("order_queries", 0)
    Qd = bench.make_sift_like(nq, d, 2, dev, **kw)
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    nd = torch.zeros(nq, dtype=torch.int32, device=dev); nh = torch.zeros(nq, dtype=torch.int32, device=dev)
    for _ in range(3):
        H.search_batch_device(hg, Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), nd.data_ptr(), nh.data_ptr(), 0, 0)
        torch.cuda.synchronize()
    c = nd.cpu().numpy().astype(np.uint32).astype(np.float64); h = nh.cpu().numpy().astype(np.float64)
    print("ef %d%s%s phase %s (%s), nq=%d: %.0f cycles per query (median), %.0f per hop, %.1f hops" % (ef, " hard set" if kw else "", " W in powers of two" if os.environ.get("HNSW_NSLOT_POW2") else "", os.environ.get("PHASE"), names[int(os.environ.get("PHASE", 0))], nq, np.median(c), np.median(c / h), np.median(h)), flush=True)
