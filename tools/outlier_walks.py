#!/usr/bin/env python3
"""Does the ordering key (distance to the entry node of layer 0) single out the few walks that end a launch?  (GPU box)

C2's shape, eight 10 k batches: hop counts from the kernel, the key from the reference's own descent (search_one per layer).
For f = 1..10 % of the queries taken by key: is the longest walk among them, what is the longest walk left, and a model of the
host step if those f % start walking as soon as their own query has arrived and the rest behind the pre-pass (today: all behind it)."""
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
kw = dict(n_centres=256, sigma=40.0) if int(os.environ.get("HARD", 0)) else {}
X = make_sift_like(n, d, 1, dev, **kw)
hg = H.Ohnsw.build_batch_bigarray(X.cpu().numpy(), 16, 200, seed=1, metric=0)
T_PCIE, T_PRE, T_DESC, T_HOP = 94.0, 135.0, 25.0, 1.28
rng = np.random.default_rng(5)
tot = {}
for b in range(8):
    Q = make_sift_like(nq, d, 100 + b, dev, **kw).cpu().numpy()
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
    node = np.full(nq, hg.entry_point, np.int64)
    key = None
    for layer in range(hg.max_layer, 0, -1):
        node, key = H.Ohnsw.search_one(hg, layer, node, Q, with_distance=True)
    nh = nh.astype(np.float64)
    order = np.argsort(-key, kind="stable")            # farthest entry first: the library's launch order
    rank_of_longest = int(np.where(order == int(np.argmax(nh)))[0][0])
    arrive = rng.uniform(0, T_PCIE, nq)                # rows arrive in no particular order (tools/pcie_read.hip)
    line = []
    for f in (0, 1, 2, 3, 5, 10, 20):
        m = nq * f // 100
        early, rest = order[:m], order[m:]
        end_rest = T_PRE + nh[rest].max() * T_HOP
        end_early = (arrive[early] + T_DESC + nh[early] * T_HOP).max() if m else 0.0
        end = max(end_rest, end_early)
        tot.setdefault(f, []).append(end)
        line.append("%d%%: %.0f (rest %.0f hops)" % (f, end, nh[rest].max()))
    print("batch %d: longest walk %d hops at rank %d by key; corr(key, hops) %.2f | step end us: %s" %
          (b, nh.max(), rank_of_longest, np.corrcoef(key, nh)[0, 1], "  ".join(line)), flush=True)
for f, v in tot.items():
    print("early fraction %2d %%: mean step end %.1f us (%.3f of today's)" % (f, np.mean(v), np.mean(v) / np.mean(tot[0])))
