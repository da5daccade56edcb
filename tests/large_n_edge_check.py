#!/usr/bin/env python3
"""One-off check (not collected by pytest; it uses the oracle, so it lives under tests/) at the edge of the hand-scheduled loops' 32-bit row offsets: the layer-0 loop forms the byte
offset (id + 1) * S0 * 4 + lane * 4 in 32 bits and is taken only while ((n + 1) * S0 < 2^30).  Two indexes of byte-valued
65-dimensional vectors, M = 16 (S0 = 32): n just BELOW the limit (the loop runs with offsets up to 2^32 - 128) and n just
ABOVE it (hipcc's loop with 64-bit addresses takes over); both must equal the oracle bit for bit.
    python tests/large_n_edge_check.py            (about 45 GB of host memory, a few minutes on the GPU)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import ocaml_hnsw_amd as H  # noqa: E402
from oracle import oracle as o  # noqa: E402

o.build(); o.lib(); H.load()
dev = torch.device("cuda", 0)
d, M, efc, ef, k, nq = 65, 16, 40, 128, 10, 256
limit = (1 << 30) // (2 * M) - 1            # largest n with (n + 1) * S0 < 2^30 is limit - 1


def data(n, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    gc = torch.Generator(device=dev); gc.manual_seed(99)
    cen = torch.randint(20, 200, (4096, d), generator=gc, device=dev).float()
    out = np.empty((n, d), np.float32)
    for s in range(0, n, 1 << 21):
        m = min(1 << 21, n - s)
        idx = torch.randint(0, 4096, (m,), generator=g, device=dev)
        out[s:s + m] = torch.clamp(torch.round(cen[idx] + 25.0 * torch.randn((m, d), generator=g, device=dev)), 0, 218).cpu().numpy()
    return out


for n, what in ((limit - 1, "just below the limit: hand-scheduled loop, 32-bit offsets up to 2^32 - 128"),
                (limit + 1000, "just above the limit: the compiler's loop, 64-bit addresses")):
    t = time.time()
    X = data(n, 1)
    Q = data(nq, 2)
    print("n = %d (%s): data %.0fs" % (n, what, time.time() - t), flush=True)
    t = time.time()
    hg = H.Ohnsw.build_batch_bigarray(X, M, efc, seed=1)
    print("  built in %.0fs, max_layer %d, row bytes %d" % (time.time() - t, hg.max_layer, hg.row_bytes()), flush=True)
    assert hg.row_bytes() == d
    # queries whose walks end in the LAST rows of the tables: the highest node ids (copies of them as queries)
    Q[:64] = X[n - 64:]
    ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
    hg.set_option("order_queries", 1)
    ids2, dist2, _, nh2 = H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, counters=True)
    assert np.array_equal(ids, ids2) and np.array_equal(nh, nh2)
    t = time.time()
    hg.export()
    g = o.Graph(hg.n, hg.entry_point, hg.deg0, hg.nbr0, hg.upper)
    sp = o.Space.l2(X, arith=o.TREE16)
    oi, od, ond, onh = o.Ohnsw.knn_batch_bigarray(g, sp, Q, k=k, ef=ef, ties=o.TIES_CANONICAL, counters=True)
    ok = np.array_equal(ids, oi) and np.array_equal(dist.view(np.uint32), od.view(np.uint32)) and np.array_equal(nh, onh)
    print("  oracle on %d queries (%.0fs): ids / distance bits / hop counts %s; %d of the first 64 queries find their own node (ids %d..%d) first"
          % (nq, time.time() - t, "EQUAL" if ok else "DIFFER", int((ids[:64, 0] == np.arange(n - 64, n)).sum()), n - 64, n - 1), flush=True)
    assert ok
    hg.release()
    del X, g, sp, hg
print("large-n edge check ok")
