#!/usr/bin/env python3
"""What would another visited-cache policy save?  (run on the GPU box)

The knn kernels keep Visited (lib/ohnsw.ml:259-262) as a lossy cache of node tags in LDS: 2 ways of 16-bit tags per word, the
new tag enters way 0, way 0 moves to way 1.  A forgotten node that comes back is evaluated again (its row is fetched again);
W ignores it, so results never change -- but on clustered data with a large ef a third of the evaluations are such
re-evaluations (others.C3_clustered, others.C5_clustered).  This script replays the exact search of a few queries on the
GPU-built graph (numpy; the order of expansions does not depend on the cache) and feeds every hop's neighbour row to
simulated caches: sets x ways x replacement policy -> evaluations per query.

    python tools/visited_cache_sim.py [queries]        # C3_clustered's shape (env N, D, EF, M as in tools/sweep.py)
"""
import heapq
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import ocaml_hnsw_amd as H

N = int(os.environ.get("N", 1183514)); D = int(os.environ.get("D", 100)); M = int(os.environ.get("M", 32))
EF = int(os.environ.get("EF", 256)); METRIC = int(os.environ.get("METRIC", 1)); NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)


def clustered(n, seed, centres=256, spread=1.5):
    g = torch.Generator(device=dev); g.manual_seed(4321)
    cen = torch.randn((centres, D), generator=g, device=dev); cen = cen / cen.norm(dim=1, keepdim=True)
    g.manual_seed(seed)
    out = np.empty((n, D), np.float32)
    for s in range(0, n, 1 << 20):
        m = min(1 << 20, n - s)
        idx = torch.randint(0, centres, (m,), generator=g, device=dev)
        x = cen[idx] + spread * torch.randn((m, D), generator=g, device=dev) / (D ** 0.5)
        out[s:s + m] = (x / x.norm(dim=1, keepdim=True)).cpu().numpy()
    return out


X = clustered(N, 12)
Q = clustered(NQ, 112)
hg = H.Ohnsw.build_batch_bigarray(X, M, 200, seed=1, metric=METRIC)
hg.export()
deg0, nbr0, upper, ep = hg.deg0, hg.nbr0, hg.upper, hg.entry_point
ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=EF, counters=True)
print("device: %.0f evaluations, %.0f hops per query (ef %d)" % (nd.mean(), nh.mean(), EF), flush=True)


def dist_to(q, rows):
    v = X[rows] @ q
    return (1.0 - v) if METRIC else ((X[rows] - q) ** 2).sum(1)


class Cache:
    def __init__(self, set_bits, ways, policy):
        self.sb, self.ways, self.policy = set_bits, ways, policy
        self.sets = {}

    def lookup_insert(self, node):
        s = node & ((1 << self.sb) - 1)
        tag = node >> self.sb
        ws = self.sets.setdefault(s, [])
        if tag in ws:
            if self.policy == "lru":
                ws.remove(tag); ws.insert(0, tag)
            return True
        ws.insert(0, tag)
        del ws[self.ways:]
        return False


CONFIGS = [(11, 2, "fifo"), (11, 2, "lru"), (11, 3, "fifo"), (11, 3, "lru"), (11, 4, "fifo"), (12, 2, "fifo"), (12, 2, "lru"), (13, 2, "fifo")]
tot = {c: 0 for c in CONFIGS}
exact = 0
for qi in range(NQ):
    q = Q[qi]
    start = int(ids[qi, 0])                                    # near the query: the walk from here is the converged part of the search
    d0 = float(dist_to(q, np.array([start]))[0])
    visited = {start}
    cand = [(d0, start)]
    W = [(-d0, start)]
    caches = {c: Cache(*c) for c in CONFIGS}
    for c in caches.values():
        c.lookup_insert(start)
    n_exact = 1
    while cand:
        dc, c_ = heapq.heappop(cand)
        if len(W) >= EF and dc > -W[0][0]:
            break
        row = nbr0[c_, :deg0[c_]]
        row = row[row >= 0]
        for cfg, ca in caches.items():
            tot[cfg] += sum(0 if ca.lookup_insert(int(nb)) else 1 for nb in row)
        fresh = [int(nb) for nb in row if int(nb) not in visited]
        if not fresh:
            continue
        visited.update(fresh)
        n_exact += len(fresh)
        dd = dist_to(q, np.array(fresh))
        for nb, dn in zip(fresh, dd):
            dn = float(dn)
            if len(W) < EF or dn < -W[0][0]:
                heapq.heappush(cand, (dn, nb))
                heapq.heappush(W, (-dn, nb))
                if len(W) > EF:
                    heapq.heappop(W)
    exact += n_exact
print("exact visited set: %.0f evaluations per query (layer-0 walk from the nearest neighbour; the device's walk starts further out)" % (exact / NQ))
for cfg in CONFIGS:
    print("2^%d sets x %d ways, %s: %.0f evaluations per query (+%.1f %%)" % (cfg[0], cfg[1], cfg[2], tot[cfg] / NQ, 100.0 * (tot[cfg] / max(exact, 1) - 1)))
