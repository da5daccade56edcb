#!/usr/bin/env python3
"""A/B of environment knobs the library reads per call, on C2's shape: host call and device-resident call, eight rotating batches.  (GPU box)

    SETTINGS="HNSW_ORDER_STOP_LAYER=2;HNSW_ORDER_STOP_LAYER=3" python tools/env_ab.py
Each setting is an &-separated list of NAME=VALUE (values may hold commas: HNSW_PRIO=1024,8192); the plain library is measured first and last."""
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
kw = dict(n_centres=256, sigma=40.0) if int(os.environ.get("HARD", 0)) else {}
X = make_sift_like(n, d, 1, dev, **kw)
hg = H.Ohnsw.build_batch_bigarray(X.cpu().numpy(), 16, 200, seed=1, metric=0)
if os.environ.get("BYTE_ROWS"):
    hg.set_option("byte_rows", int(os.environ["BYTE_ROWS"]))
NB = 8
Qh, Qd = [], []
for b in range(NB):
    qd = make_sift_like(nq, d, 100 + b, dev, **kw)
    q = H.host_empty((nq, d), np.float32)
    q[:] = qd.cpu().numpy()
    Qh.append(q); Qd.append(qd)
oi = H.host_empty((nq, k), np.int32); od = H.host_empty((nq, k), np.float32)
ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream()
settings = [""] + [s for s in os.environ.get("SETTINGS", "").split(";") if s] + [""]
ref = None
for s in settings:
    names = []
    for kv in [x for x in s.split("&") if x]:
        a, b = kv.split("=", 1)
        os.environ[a] = b; names.append(a)
    got = []
    for b in range(NB):
        H.Ohnsw.knn_batch_bigarray(hg, k, Qh[b], ef=ef, out=(oi, od))
        got.append((oi.copy(), od.view(np.uint32).copy()))
    if ref is None:
        ref = got
    same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(ref, got))
    th = []
    for rep in range(5):
        for b in range(NB):
            t0 = time.perf_counter()
            H.Ohnsw.knn_batch_bigarray(hg, k, Qh[b], ef=ef, out=(oi, od))
            th.append(time.perf_counter() - t0)
    for b in range(NB):
        H.search_batch_device(hg, Qd[b].data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), 0, 0, 0, stream.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(5):
        for b in range(NB):
            H.search_batch_device(hg, Qd[b].data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), 0, 0, 0, stream.cuda_stream)
    torch.cuda.synchronize()
    td = (time.perf_counter() - t0) / (5 * NB) * 1e3
    th = np.array(th) * 1e3
    print("%-40s host call mean %.4f ms median %.4f | device-resident %.4f ms per step | same results: %s" %
          (s or "(plain)", th.mean(), np.median(th), td, same), flush=True)
    for a in names:
        del os.environ[a]
