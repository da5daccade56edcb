#!/usr/bin/env python3
"""A/B timing of library variants on the C2 workload (run on the GPU box).

    python tools/ab.py exp/base.so exp/new.so ...        # one subprocess per library
    python tools/ab.py --one <lib>                        # (internal) measure one library

Per library: 64 queries on an idle chip, the 10 k batch (the headline), 100 k queries, the harder set's 10 k
batch when AB_HARD=1; byte rows and (AB_F32=1) float32 rows; AB_SEM=1: the functor accept rule (Hnsw.Ba) instead of Ohnsw's.  Prints the median of REPS device calls and an
md5 over ids, distance bits, evaluation and hop counts of the 10 k batch: variants that are meant to be exact
must print the same digest.  The index is built by the first library and saved; the others load the file.
"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(lib):
    os.environ["HNSW_LIB_PATH"] = os.path.abspath(lib)
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import ocaml_hnsw_amd as H
    import bench
    dev = torch.device("cuda", 0)
    n, d, M, efc, k, ef = int(os.environ.get("N", 1000000)), 128, 16, 200, 10, int(os.environ.get("EF", 128))
    reps = int(os.environ.get("REPS", 9))
    sets = [("c2", 4096, 25.0)] + ([("hard", 256, 40.0)] if os.environ.get("AB_HARD") else [])
    stream = torch.cuda.current_stream()
    for name, centres, sigma in sets:
        # the graph depends on the builder that made it: key the cached file on everything that decides it, the FIRST
        # library's content included (variants meant to be exact search the same file and must print the same digest)
        key = hashlib.md5(open(os.environ.get("AB_BUILDER_LIB", lib), "rb").read()).hexdigest()[:10]
        cache = "/tmp/ab_%s_n%d_M%d_efc%d_seed1_%s.idx" % (name, n, M, efc, key)
        Xd = bench.make_sift_like(n, d, 1, dev, centres, sigma)
        if os.path.exists(cache):
            hg = H.Hgraph.load(cache)
            print("  [%s] index loaded from %s" % (name, cache), flush=True)
        else:
            t = time.time()
            hg = H.Ohnsw.build_batch_bigarray(Xd.cpu().numpy(), M, efc, seed=1)
            print("  [%s] index built in %.1fs" % (name, time.time() - t), flush=True)
            hg.save(cache)
        del Xd
        for rows in ([1, 0] if os.environ.get("AB_F32") else [1]):
            hg.set_option("byte_rows", rows)
            out = []
            digest = None
            for nq in [int(x) for x in os.environ.get("AB_NQ", "64,10000,100000").split(",")]:
                Qd = bench.make_sift_like(nq, d, 2, dev, centres, sigma)
                ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
                dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
                nd = torch.zeros(nq, dtype=torch.int32, device=dev)
                nh = torch.zeros(nq, dtype=torch.int32, device=dev)
                st = torch.zeros(nq, dtype=torch.int32, device=dev)

                def go(c=False):
                    H.search_batch_device(hg, Qd.data_ptr(), nq, d, ef, k, ids.data_ptr(), dist.data_ptr(),
                                          nd.data_ptr() if c else 0, nh.data_ptr() if c else 0, st.data_ptr() if c else 0, stream.cuda_stream,
                                          sem=int(os.environ.get("AB_SEM", 0)))
                go(True)
                torch.cuda.synchronize()
                if nq == 10000 or digest is None:
                    h = hashlib.md5()
                    for a in (ids, dist.view(torch.int32), nd, nh, st & 1):
                        h.update(a.cpu().numpy().tobytes())
                    digest = h.hexdigest()[:12]
                    ndm, nhm = float(nd.float().mean()), float(nh.float().mean())
                ts = []
                for _ in range(reps):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream); go(); b.record(stream); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
                out.append((nq, float(np.median(ts)), float(np.min(ts))))
            print("  [%s %s] " % (name, "byte" if rows else "f32 ") +
                  "  ".join("nq=%d: %.4f ms (min %.4f) %.2f Mq/s" % (nq, ms, mn, nq / ms / 1e3) for nq, ms, mn in out) +
                  "  | n_dist %.1f hops %.1f md5 %s" % (ndm, nhm, digest), flush=True)
        hg.release()


def main():
    if sys.argv[1] == "--one":
        one(sys.argv[2])
        return
    os.environ.setdefault("AB_BUILDER_LIB", os.path.abspath(sys.argv[1]))     # every variant searches the first library's graph
    for lib in sys.argv[1:]:
        print("== %s" % lib, flush=True)
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--one", lib])
        if rc != 0:
            print("   FAILED rc=%d" % rc, flush=True)
            sys.exit(rc)     # a faulting variant: start no further GPU work


if __name__ == "__main__":
    main()
