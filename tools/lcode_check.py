#!/usr/bin/env python3
"""How good are the locality codes (hnsw_locality.hip)?  (GPU box)  On clustered unit vectors: the share of layer-0 links
whose two ends lie within 4096 codes of each other -- under the ids, under the device's codes and under the generator's own
cluster order --, how often consecutive codes share a cluster, and how far a node's code is from its greedy / exact nearest
layer-2 node's (the greedy descent alone ends in the wrong cluster for half of the vectors: why the codes use a small search)."""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
os.environ.setdefault("BLOCKS", "0")
N = int(os.environ.get("N", 2000000)); D = 96; M = 32
import ocaml_hnsw_amd as H
dev = torch.device("cuda", 0)
def clustered(n, seed, centres=256, spread=1.5):
    g = torch.Generator(device=dev); g.manual_seed(4321)
    cen = torch.randn((centres, D), generator=g, device=dev); cen = cen / cen.norm(dim=1, keepdim=True)
    g.manual_seed(seed)
    out = np.empty((n, D), np.float32); cl = np.empty(n, np.int64)
    for s in range(0, n, 1 << 20):
        m = min(1 << 20, n - s)
        idx = torch.randint(0, centres, (m,), generator=g, device=dev)
        x = cen[idx] + spread * torch.randn((m, D), generator=g, device=dev) / (D ** 0.5)
        out[s:s + m] = (x / x.norm(dim=1, keepdim=True)).cpu().numpy(); cl[s:s+m] = idx.cpu().numpy()
    return out, cl
X, cl = clustered(N, 12)
hg = H.Ohnsw.build_batch_bigarray(X, M, 200, seed=1, metric=0)
hg.export()
L = hg.locality_codes().astype(np.int64)
print("permutation:", np.array_equal(np.sort(L), np.arange(N)), "max_layer", hg.max_layer, [len(u[0]) for u in hg.upper])
rng = np.random.default_rng(0)
samp = rng.choice(N, 2000, replace=False)
def quality(code, name):
    near = tot = 0
    for v in samp:
        nb = hg.nbr0[v, :hg.deg0[v]]
        near += int((np.abs(code[nb] - code[v]) < 4096).sum()); tot += len(nb)
    print("%-8s fraction of layer-0 links within 4096 codes: %.3f" % (name, near / tot))
quality(np.arange(N), "ident")
quality(L, "device")
order = np.lexsort((np.arange(N), cl)); Lc = np.empty(N, np.int64); Lc[order] = np.arange(N)
quality(Lc, "cluster")
# how do codes relate to clusters: for consecutive codes, how often same cluster
inv = np.empty(N, np.int64); inv[L] = np.arange(N)
print("same cluster for consecutive codes:", float((cl[inv[1:]] == cl[inv[:-1]]).mean()))
# emulate descent in python through the operator
T = hg.max_layer
Qs = X[samp[:500]]
cur = np.full(len(Qs), hg.entry_point, np.int64)
for l in range(T, 1, -1):
    cur = H.Ohnsw.search_one(hg, l, cur, Qs)
hub_codes = L[cur]
print("python descent to layer 2: |L[v]-L[hub]| median", float(np.median(np.abs(L[samp[:500]] - hub_codes))), "same cluster as hub:", float((cl[cur] == cl[samp[:500]]).mean()))
# exact nearest layer-2 hub
hubs = np.asarray(hg.upper[1][0], np.int64)
Hm = torch.from_numpy(X[hubs]).to(dev); xs = torch.from_numpy(Qs).to(dev)
nn = hubs[(xs @ Hm.T).argmax(1).cpu().numpy()]
print("greedy hub == exact nearest hub:", float((nn == cur).mean()), " |L[v]-L[exact hub]| median", float(np.median(np.abs(L[samp[:500]] - L[nn]))))
