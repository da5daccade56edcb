#!/usr/bin/env python3
"""Build the C2 index once, then time the search kernel over batch sizes / visited-cache sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ocaml_hnsw_amd as H
import bench

dev = torch.device("cuda", 0)
n, d, M, efc = int(os.environ.get("N", 1000000)), int(os.environ.get("D", 128)), int(os.environ.get("M", 16)), int(os.environ.get("EFC", 200))
K = int(os.environ.get("K", 10)); METRIC = int(os.environ.get("METRIC", 0)); KIND = os.environ.get("KIND", "sift")
sigma = float(os.environ.get("SIGMA", 25)); centres = int(os.environ.get("CENTRES", 4096))

def make(nn, seed):
    if KIND == "sift":
        return bench.make_sift_like(nn, d, seed, dev, centres, sigma)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    if KIND == "uniform":
        return torch.rand((nn, d), generator=g, device=dev) * 2 - 1
    if KIND == "cunit":      # unit vectors clustered around CENTRES random directions (bench.py: others.C3_clustered)
        gc = torch.Generator(device=dev); gc.manual_seed(4321)
        cen = torch.randn((centres, d), generator=gc, device=dev); cen = cen / cen.norm(dim=1, keepdim=True)
        out = torch.empty((nn, d), device=dev)
        for s0 in range(0, nn, 1 << 20):
            m = min(1 << 20, nn - s0)
            idx = torch.randint(0, centres, (m,), generator=g, device=dev)
            x = cen[idx] + sigma * torch.randn((m, d), generator=g, device=dev) / (d ** 0.5)
            out[s0:s0 + m] = x / x.norm(dim=1, keepdim=True)
        return out
    x = torch.randn((nn, d), generator=g, device=dev)      # "unit": N(0,1) normalised (GloVe/DEEP-like)
    return x / x.norm(dim=1, keepdim=True)

def truth(Xd_, Q, k):
    if METRIC == 0:
        return bench.brute_force_topk(Xd_, Q, k)
    ids = []
    for s0 in range(0, Q.shape[0], 128):
        ids.append(torch.topk(Q[s0:s0 + 128] @ Xd_.T, k, dim=1, largest=True).indices)
    return torch.cat(ids).cpu().numpy()

Xd = make(n, 1)
X = Xd.cpu().numpy()
# INDEX_CACHE=<path>: build once, save the flattened index (hnsw_index_save), load it on later runs
# (the PMC passes of tools/profile_cmd.sh run this script once per counter group)
cache = os.environ.get("INDEX_CACHE")
t = time.time()
if cache and os.path.exists(cache):
    hg = H.Hgraph.load(cache)
    print("index loaded from %s in %.2fs (n=%d d=%d M=%d metric=%d %s)" % (cache, time.time() - t, n, d, M, METRIC, KIND), flush=True)
else:
    hg = H.Ohnsw.build_batch_bigarray(X, M, efc, seed=1, metric=METRIC)
    print("build %.2fs (n=%d d=%d M=%d efC=%d metric=%d %s)" % (time.time() - t, n, d, M, efc, METRIC, KIND), flush=True)
    if cache:
        t = time.time(); hg.save(cache); print("index saved to %s in %.2fs" % (cache, time.time() - t), flush=True)
stream = torch.cuda.current_stream()
for kv in filter(None, os.environ.get("OPTS", "").split(",")):      # OPTS="split_rows=0,order_queries=1": hnsw_index_set_option
    hg.set_option(kv.split("=")[0], int(kv.split("=")[1]))
print("row format %d, %d B of index" % (hg.info().row_format, hg.info().device_bytes), flush=True)

# hooks called after every run(); registered by a wrapper that executes this file (the oracle is test
# infrastructure and is only touched from tests/)
POST_RUN = globals().get("POST_RUN", [])

def run(nq, ef, k=K, vt=0, reps=int(os.environ.get("REPS", 5)), pad=-1):
    Qd = make(nq, 2)
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    nd = torch.zeros(nq, dtype=torch.int32, device=dev); nh = torch.zeros(nq, dtype=torch.int32, device=dev)
    hg.set_option("vt_bits", vt); hg.set_option("lds_pad", pad)
    def go(c=False):
        H.search_batch_device(hg, Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), nd.data_ptr() if c else 0, nh.data_ptr() if c else 0, 0, stream.cuda_stream)
    go(True); torch.cuda.synchronize()
    ndm, nhm = nd.float().mean().item(), nh.float().mean().item()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); go(); b.record(stream); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    bq = ndm * (hg.row_bytes() + 4) + nhm * 4 * 2 * M + 4 * d + 8 * k   # uses GPU n_dist (incl. re-evals); a row is d bytes when the index serves byte rows
    ns = min(200, nq)
    gt = truth(Xd, Qd[:ns], k)
    rec = bench.recall_ids(ids.cpu().numpy()[:ns], gt)
    print("nq=%7d ef=%4d vt=%2d pad=%5d: %8.3f ms  %10.0f q/s  n_dist(gpu)=%.0f hops=%.0f  gpu-bytes %.2f TB/s  recall %.3f" %
          (nq, ef, vt, pad, ms, nq / ms * 1e3, ndm, nhm, bq * nq / ms / 1e9, rec), flush=True)
    for hook in POST_RUN:   # e.g. tests/sweep_with_oracle.py compares a sample with the CPU oracle
        hook(dict(nq=nq, ef=ef, k=k, Qd=Qd, ids=ids, dist=dist, nd=nd, nh=nh, go=go, hg=hg, X=X, metric=METRIC, make=make))

for spec in sys.argv[1:]:
    f = [int(x) for x in spec.split(",")]     # nq,ef,vt[,lds_pad]  (lds_pad -1 = the library's choice)
    run(f[0], f[1], vt=f[2], pad=f[3] if len(f) > 3 else -1)

if os.environ.get("PREF_STATS"):
    nq, ef, k = 10000, 128, 10
    Qd = make(nq, 2)
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    nh = torch.zeros(nq, dtype=torch.int32, device=dev); st = torch.zeros(nq, dtype=torch.int32, device=dev)
    H.search_batch_device(hg, Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), 0, nh.data_ptr(), st.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    hits = (st >> 8).float().sum().item(); hops = nh.float().sum().item()
    print("prefetch hit rate: %.3f (%d hits / %d hops)" % (hits / hops, hits, hops))
    # host-buffer (PCIe-inclusive) entry point
    Qh = Qd.cpu().numpy()
    import time
    H.Ohnsw.knn_batch_bigarray(hg, k, Qh, ef=ef)
    t = time.perf_counter()
    for _ in range(5):
        H.Ohnsw.knn_batch_bigarray(hg, k, Qh, ef=ef)
    dt = (time.perf_counter() - t) / 5
    print("host-buffer hnsw_search_batch (H2D + kernel + D2H + sync): %.3f ms/batch = %.0f q/s" % (dt * 1e3, nq / dt))

if os.environ.get("ORDER_EXPERIMENT"):
    # does co-scheduling spatially close queries on one XCD pay?  (host-side emulation)
    nq, ef, k = 10000, 128, 10
    Qd = make(nq, 2)
    g = torch.Generator(device=dev); g.manual_seed(1234)
    cen = torch.randint(20, 200, (centres, d), generator=g, device=dev).float()
    cid = torch.cdist(Qd, cen).argmin(1)
    order = torch.argsort(cid)
    per = (nq + 7) // 8
    b = torch.arange(nq, device=dev)
    spos = (b % 8) * per + b // 8
    spos = torch.clamp(spos, max=nq - 1)
    variants = {"unsorted": Qd, "sorted (block-contiguous)": Qd[order], "sorted + xcd-aware": Qd[order][spos]}
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    for name, Qv in variants.items():
        Qv = Qv.contiguous()
        ts = []
        for _ in range(7):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            H.search_batch_device(hg, Qv.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), 0, 0, 0, stream.cuda_stream)
            e.record(stream); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
        print("%-28s %.3f ms  %.0f q/s" % (name, np.median(ts), nq / np.median(ts) * 1e3), flush=True)

if os.environ.get("OVERLAP_EXPERIMENT"):
    # Is the 10 k batch paying pipeline fill/drain (time = latency + nq/throughput) or is the larger
    # batch simply hitting in cache?  Back-to-back 10 k batches on ONE stream vs alternated over
    # TWO streams (the tail of batch i overlaps the head of batch i+1; different queries each).
    nq, ef, k, nb = 10000, 128, 10, 16
    Qs = [make(nq, 100 + i) for i in range(nb)]
    outs = [(torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev)) for _ in range(nb)]
    for nstreams in (1, 2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        ts = []
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(nb):
                s_ = streams[i % nstreams]
                H.search_batch_device(hg, Qs[i].data_ptr(), nq, d, ef, k, outs[i][0].data_ptr(), outs[i][1].data_ptr(), 0, 0, 0, s_.cuda_stream)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / nb)
        ms = float(np.median(ts)) * 1e3
        print("%d stream(s): %.3f ms per 10k batch  %.0f q/s" % (nstreams, ms, nq / ms * 1e3), flush=True)

if os.environ.get("HOP_STATS"):
    nq, ef, k = 10000, 128, 10
    Qd = make(nq, 2)
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    nd = torch.zeros(nq, dtype=torch.int32, device=dev); nh = torch.zeros(nq, dtype=torch.int32, device=dev)
    H.search_batch_device(hg, Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), nd.data_ptr(), nh.data_ptr(), 0, stream.cuda_stream)
    torch.cuda.synchronize()
    for name, v in (("hops", nh.cpu().numpy()), ("evaluations", nd.cpu().numpy())):
        print("%s per query: min %d p10 %d p50 %d p90 %d p99 %d max %d mean %.1f" %
              (name, v.min(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(), v.mean()), flush=True)

if os.environ.get("TIMELINE"):
    # needs a library built with -DHNSW_TIMING: out_ndist / out_nhops carry each query's start / end (10 ns ticks)
    for nq in (7168, 10000, 20000):
        ef, k = 128, 10
        Qd = make(nq, 2)
        ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
        t0 = torch.zeros(nq, dtype=torch.int32, device=dev); t1 = torch.zeros(nq, dtype=torch.int32, device=dev)
        for _ in range(3):
            H.search_batch_device(hg, Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), t0.data_ptr(), t1.data_ptr(), 0, stream.cuda_stream)
            torch.cuda.synchronize()
        a = t0.cpu().numpy().astype(np.uint32).astype(np.int64); b = t1.cpu().numpy().astype(np.uint32).astype(np.int64)
        base = a.min(); a = (a - base) / 100.0; b = (b - base) / 100.0      # microseconds
        dur = b - a
        print("nq=%d: launch span %.0f us; start p50 %.0f p90 %.0f max %.0f; duration p10 %.0f p50 %.0f p90 %.0f max %.0f us" %
              (nq, b.max(), np.percentile(a, 50), np.percentile(a, 90), a.max(), np.percentile(dur, 10), np.percentile(dur, 50), np.percentile(dur, 90), dur.max()), flush=True)
        first = a < 5.0
        print("   first wave: %d queries, duration p50 %.0f us, end p10 %.0f p50 %.0f p90 %.0f; later starters: %d, duration p50 %.0f us" %
              (first.sum(), np.median(dur[first]), np.percentile(b[first], 10), np.percentile(b[first], 50), np.percentile(b[first], 90),
               (~first).sum(), np.median(dur[~first]) if (~first).any() else 0), flush=True)
        edges = np.arange(0, b.max() + 50, 50)
        act = [int(((a <= t) & (b > t)).sum()) for t in edges]
        print("   queries in flight every 50 us:", act, flush=True)

if os.environ.get("PHASES"):
    # needs a library built with -DHNSW_PHASE_TIMING: the counters carry shader-clock cycles per phase
    for nq in (64, 10000):
        ef, k = 128, 10
        Qd = make(nq, 2)
        ids = torch.empty((nq, k), dtype=torch.int32, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
        a = torch.zeros(nq, dtype=torch.int32, device=dev); b = torch.zeros(nq, dtype=torch.int32, device=dev); c = torch.zeros(nq, dtype=torch.int32, device=dev)
        hg.set_option("order_queries", 0)
        for _ in range(3):
            H.search_batch_device(hg, Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(), a.data_ptr(), b.data_ptr(), c.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
        st = c.cpu().numpy().astype(np.uint32)
        if os.environ["PHASES"] == "2":    # library built with -DHNSW_PHASE_TIMING=2
            fill = a.cpu().numpy().astype(np.uint32).astype(np.float64); steady = b.cpu().numpy().astype(np.uint32).astype(np.float64)
            print("nq=%d: insertions per query %.1f; insert cycles while W fills %.0f, once full %.0f (medians); cycles per insertion %.0f" %
                  (nq, st.mean(), np.median(fill), np.median(steady), (fill.sum() + steady.sum()) / st.sum()), flush=True)
            continue
        p0 = a.cpu().numpy().astype(np.uint32).astype(np.float64); p2 = b.cpu().numpy().astype(np.uint32).astype(np.float64)
        p3 = (st & 0xFFFFF).astype(np.float64); p1 = ((st >> 20) * 64).astype(np.float64)
        tot = p0 + p1 + p2 + p3
        print("nq=%d: cycles per query (median): pop/adjacency/filter %.0f (%.0f%%)  prefetch+compaction %.0f (%.0f%%)  ids/rows/arith/keys %.0f (%.0f%%)  insert %.0f (%.0f%%)  total %.0f" %
              (nq, np.median(p0), 100 * p0.sum() / tot.sum(), np.median(p1), 100 * p1.sum() / tot.sum(), np.median(p2), 100 * p2.sum() / tot.sum(),
               np.median(p3), 100 * p3.sum() / tot.sum(), np.median(tot)), flush=True)

if os.environ.get("SUBMIT_WAIT"):
    # host-buffer batches in flight (hnsw_search_submit / hnsw_search_wait) against the synchronous call
    nq, ef, k, nb = 10000, 128, 10, 16
    Qs = [make(nq, 200 + i).cpu().numpy() for i in range(nb)]
    H.Ohnsw.knn_batch_bigarray(hg, k, Qs[0], ef=ef)
    t = time.perf_counter()
    for i in range(nb):
        H.Ohnsw.knn_batch_bigarray(hg, k, Qs[i], ef=ef)
    sync_ms = (time.perf_counter() - t) / nb * 1e3
    for depth in (1, 2, 3):
        for rep in range(2):
            t = time.perf_counter()
            inflight = []
            for i in range(nb):
                inflight.append(H.submit(hg, Qs[i], ef, k))
                if len(inflight) > depth:
                    inflight.pop(0).wait()
            while inflight:
                inflight.pop(0).wait()
            dt = (time.perf_counter() - t) / nb * 1e3
        print("host buffers, %d batch(es) in flight beyond the one being waited for: %.3f ms per 10k batch = %.0f q/s (synchronous call %.3f ms)" %
              (depth, dt, nq / dt * 1e3, sync_ms), flush=True)
