#!/usr/bin/env python3
"""Would starting the walks of a host batch in PHASES (as their queries arrive over PCIe) end the step earlier?  (GPU box)

Builds C2's shape, takes the hop counts of eight 10 k batches from the kernel's own counters, and evaluates a simple model of the
host step: phase i of P (queries in matrix order) can start its walks at  a_i = (i + 1) / P x T_pcie + T_tail ; a walk of h hops takes
h x T_hop; the step ends with the last walk.  P = 1 is today's step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import ocaml_hnsw_amd as H
from bench import make_sift_like

dev = torch.device("cuda", 0)
n, d, nq, k, ef = 1_000_000, 128, 10_000, 10, 128
X = make_sift_like(n, d, 1, dev)
hg = H.Ohnsw.build_batch_bigarray(X.cpu().numpy(), 16, 200, seed=1, metric=0)
T_PCIE, T_TAIL, T_HOP = 94.0, 35.0, 1.28       # us: transfer of the whole batch; last descents + sort + launch; a hop under load
res = {}
for b in range(8):
    Q = make_sift_like(nq, d, 100 + b, dev).cpu().numpy()
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
    nh = nh.astype(np.float64)
    line = []
    for P in (1, 2, 3, 4, 6, 8):
        size = (nq + P - 1) // P
        end = 0.0
        for i in range(P):
            part = nh[i * size:(i + 1) * size]
            if len(part):
                end = max(end, (i + 1) / P * T_PCIE + T_TAIL + part.max() * T_HOP)
        res.setdefault(P, []).append(end)
        line.append("P=%d %.0f" % (P, end))
    print("batch %d: hops mean %.1f p99 %.0f p99.9 %.0f max %.0f | step end (us): %s" %
          (b, nh.mean(), np.percentile(nh, 99), np.percentile(nh, 99.9), nh.max(), "  ".join(line)), flush=True)
for P, v in res.items():
    print("P=%d: mean step end %.1f us (%.3f of P=1)" % (P, np.mean(v), np.mean(v) / np.mean(res[1])))
