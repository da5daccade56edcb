#!/usr/bin/env python3
"""Tag cache against bitmap blocks (option "visited_blocks") on one index: same results, evaluations, time.  (GPU box)

    N=10000000 D=96 M=32 EF=512 METRIC=0 KIND=clustered python tools/visited_blocks_ab.py      # C5_clustered's shape
    KIND=clustered python tools/visited_blocks_ab.py                                            # C3_clustered's shape
    KIND=unit ...                                                                                # the structureless twin

Prints, per mode (0 tags, 1 blocks, -1 the handle's own measurement), evaluations and hops per query, the median time of
REPS device-resident 10 k batches, and whether ids, distance bits and hop counts equal mode 0's (they must).  (The comparison with
the exact Visited of lib/ohnsw.ml:256-268 is tests/test_gpu_visited_blocks.py's and bench.py's: gpu_reevaluation_overhead.)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import ocaml_hnsw_amd as H

N = int(os.environ.get("N", 1183514)); D = int(os.environ.get("D", 100)); M = int(os.environ.get("M", 32))
EF = int(os.environ.get("EF", 256)); METRIC = int(os.environ.get("METRIC", 1)); K = int(os.environ.get("K", 10))
NQ = int(os.environ.get("NQ", 10000)); REPS = int(os.environ.get("REPS", 5)); KIND = os.environ.get("KIND", "clustered")
SEM = int(os.environ.get("SEM", 0))
dev = torch.device("cuda", 0)


def vectors(n, seed, centres=256, spread=1.5):
    g = torch.Generator(device=dev)
    if KIND == "clustered":
        g.manual_seed(4321)
        cen = torch.randn((centres, D), generator=g, device=dev); cen = cen / cen.norm(dim=1, keepdim=True)
    g.manual_seed(seed)
    out = np.empty((n, D), np.float32)
    for s in range(0, n, 1 << 20):
        m = min(1 << 20, n - s)
        if KIND == "clustered":
            idx = torch.randint(0, centres, (m,), generator=g, device=dev)
            x = cen[idx] + spread * torch.randn((m, D), generator=g, device=dev) / (D ** 0.5)
        else:
            x = torch.randn((m, D), generator=g, device=dev)
        out[s:s + m] = (x / x.norm(dim=1, keepdim=True)).cpu().numpy()
    return out


t0 = time.time()
X = vectors(N, 12)
Q = vectors(NQ, 112)
hg = H.Ohnsw.build_batch_bigarray(X, M, 200, seed=1, metric=METRIC)
print("n %d d %d M %d ef %d metric %d %s: built in %.0f s" % (N, D, M, EF, METRIC, KIND, time.time() - t0), flush=True)
Qd = torch.from_numpy(Q).to(dev)
ids = torch.empty((NQ, K), dtype=torch.int32, device=dev)
dist = torch.empty((NQ, K), dtype=torch.float32, device=dev)
nd = torch.zeros(NQ, dtype=torch.int32, device=dev)
nh = torch.zeros(NQ, dtype=torch.int32, device=dev)
st = torch.zeros(NQ, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream()
ref = None
for mode in [int(x) for x in os.environ.get("MODES", "0,1,-1").split(",")]:
    hg.set_option("visited_blocks", mode)
    t1 = time.time()

    def go(c=False):
        H.search_batch_device(hg, Qd.data_ptr(), NQ, D, EF, K, ids.data_ptr(), dist.data_ptr(), nd.data_ptr() if c else 0,
                              nh.data_ptr() if c else 0, st.data_ptr(), stream.cuda_stream, sem=SEM)
    go(True)
    torch.cuda.synchronize()
    first = time.time() - t1
    got = (ids.cpu().numpy().copy(), dist.cpu().numpy().view(np.uint32).copy(), nh.cpu().numpy().copy())
    ndm, nhm = float(nd.float().mean().item()), float(nh.float().mean().item())
    ts = []
    for _ in range(REPS):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); go(); b.record(stream); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    same = "-" if ref is None else str(all(np.array_equal(x, y) for x, y in zip(ref, got)))
    if ref is None:
        ref = got
    print("visited_blocks %2d: %.0f evaluations, %.1f hops per query; %.3f ms per %d-query batch (min %.3f); first call %.2f s; "
          "same ids / distance bits / hops as mode 0: %s; flagged %d" % (mode, ndm, nhm, ts[len(ts) // 2], NQ, ts[0], first, same,
                                                                        int((st & 1).sum().item())), flush=True)
