#!/usr/bin/env python3
"""The host call (hnsw_search_batch, page-locked matrices) in phases: HNSW_HOST_PHASES x HNSW_PHASE_ORDER against the plain call.  (GPU box)

C2's shape, eight rotating 10 k batches as bench.py's headline loop; per setting the mean and median time of a call and
whether ids / distance bits equal the plain call's."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import ocaml_hnsw_amd as H
from bench import make_sift_like

dev = torch.device("cuda", 0)
n, d, nq, k = 1_000_000, 128, int(os.environ.get("NQ", 10_000)), 10
ef = int(os.environ.get("EF", 128))
hard = int(os.environ.get("HARD", 0))
kw = dict(n_centres=256, sigma=40.0) if hard else {}
X = make_sift_like(n, d, 1, dev, **kw)
hg = H.Ohnsw.build_batch_bigarray(X.cpu().numpy(), 16, 200, seed=1, metric=0)
if os.environ.get("BYTE_ROWS"):
    hg.set_option("byte_rows", int(os.environ["BYTE_ROWS"]))
NB = 8
Qs = []
for b in range(NB):
    q = H.host_empty((nq, d), np.float32)
    q[:] = make_sift_like(nq, d, 100 + b, dev, **kw).cpu().numpy()
    Qs.append(q)
oi = H.host_empty((nq, k), np.int32); od = H.host_empty((nq, k), np.float32)
settings = [(0, 0)]
for spec in os.environ.get("SETTINGS", "2:0,2:2,2:3,3:4,3:6,3:7,4:8,4:12,4:14,4:15,6:48,6:56").split(","):
    a, b = spec.split(":")
    settings.append((int(a), int(b)))
ref = None
for P, mask in settings + [(0, 0)]:
    os.environ["HNSW_HOST_PHASES"] = str(P); os.environ["HNSW_PHASE_ORDER"] = str(mask)
    got = []
    for b in range(NB):
        H.Ohnsw.knn_batch_bigarray(hg, k, Qs[b], ef=ef, out=(oi, od))
        got.append((oi.copy(), od.view(np.uint32).copy()))
    if ref is None:
        ref = got
    same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(ref, got))
    ts = []
    for rep in range(5):
        for b in range(NB):
            t0 = time.perf_counter()
            H.Ohnsw.knn_batch_bigarray(hg, k, Qs[b], ef=ef, out=(oi, od))
            ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print("phases %d order mask %2d: mean %.4f ms  median %.4f  min %.4f  max %.4f  (%.2f M q/s)  same results: %s" %
          (P, mask, ts.mean(), np.median(ts), ts.min(), ts.max(), nq / ts.mean() / 1e3, same), flush=True)
