#!/usr/bin/env python3
"""Distance micro-benchmark: the counterpart of bench_dist/bench_dist.ml (1 M calls of distance_l2 at
d = 784, one fixed `a`, fresh `b` each call, prints checksum, s/call, calls/s) on the device:
batched gathered distances over random rows, bytes = pairs * 4 * d."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ocaml_hnsw_amd as H

dev = torch.device("cuda", 0)
L = H.load()
SHAPES = ((784, 1_000_000), (128, 4_000_000), (100, 4_000_000), (96, 4_000_000))
if os.environ.get("DIST_ONLY"):          # one shape only (tools/dist_ab.sh, profiling)
    SHAPES = tuple(s_ for s_ in SHAPES if s_[0] == int(os.environ["DIST_ONLY"]))
for d, n in SHAPES:
    g = torch.Generator(device=dev); g.manual_seed(d)
    X = torch.rand((n, d), generator=g, device=dev).cpu().numpy()
    hg = H.Hgraph(X, np.zeros(n, np.int32), np.full((n, 2), -1, np.int32), entry_point=0).to_device(0)
    nq, m = 1024, 1024                       # 1 M pairs, like bench_dist.ml's 1 M calls
    Q = torch.rand((nq, d), generator=g, device=dev)
    ids = torch.randint(0, n, (nq, m), generator=g, device=dev, dtype=torch.int32)
    out = torch.empty((nq, m), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream()
    def go():
        rc = L.hnsw_distance_batch_device(hg.handle, Q.data_ptr(), nq, d, ids.data_ptr(), m, out.data_ptr(), st.cuda_stream)
        assert rc == 0, L.hnsw_last_error()
    go(); torch.cuda.synchronize()
    ts = []
    for _ in range(int(os.environ.get('DIST_REPS', 9))):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); go(); b.record(st); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    pairs = nq * m
    ref = torch.sqrt(((X_t := torch.from_numpy(X[:1]).to(dev)) * 0).sum()) if False else None
    chk = float(out.double().sum())
    print("d=%4d n=%8d: %d pairs in %.3f ms = %.3g s/call, %.3g calls/s, %.2f TB/s gathered (checksum %.6g)" %
          (d, n, pairs, ms, ms * 1e-3 / pairs, pairs / (ms * 1e-3), pairs * 4 * d / (ms * 1e-3) / 1e12, chk), flush=True)
    del hg, X
