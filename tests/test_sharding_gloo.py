"""The N > 1 path on CPU: world_size-2 (and 3, unequal shards) gloo processes shard a query batch,
each produces its shard's results, and the all-gather reproduces the single-process result
bit for bit.  The per-shard search runs on the CPU oracle here (no GPU in this tier); the
sharding, graph replication and gather code is the code bench.py runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ocaml_hnsw_amd.sharding as sh
        from oracle import oracle as o
        rng = np.random.default_rng(0)
        X = rng.uniform(-1, 1, size=(1500, 12)).astype(np.float32)
        Q = rng.uniform(-1, 1, size=(nq, 12)).astype(np.float32)
        sp = o.Space.l2(X)
        M = 6
        g0 = o.build_ohnsw(sp, M, 30, seed=3) if rank == 0 else None

        class G:  # what Hgraph.export() provides
            pass
        hg = None
        if rank == 0:
            hg = G()
            hg.max_layer, hg.entry_point, hg.n = g0.max_layer, g0.entry_point, g0.n
            hg.deg0, hg.nbr0, hg.upper = g0.deg0, g0.nbr0, g0.upper
        deg0, nbr0, upper, entry = sh.replicate_graph(dist, torch.device("cpu"), hg, M)
        g = o.Graph(len(deg0), entry, deg0, nbr0, upper)
        sh.assert_same_on_all_ranks(dist, torch.device("cpu"), {"X": torch.from_numpy(X), "Q": torch.from_numpy(Q)})

        def search_shard(lo, hi):   # the per-shard search: the CPU oracle here, the HIP kernel in bench.py
            ids, dd = o.Ohnsw.knn_batch_bigarray(g, sp, Q[lo:hi], k=5, ef=20, ties=o.TIES_CANONICAL)
            return torch.from_numpy(ids), torch.from_numpy(dd)
        ai, ad = sh.sharded_search(dist, search_shard, nq, 5)    # the function bench.py's strong-scaling leg runs
        if rank == world - 1:      # a rank with different data must be caught
            bad = X.copy(); bad[0, 0] += 1.0
        else:
            bad = X
        caught = False
        try:
            sh.assert_same_on_all_ranks(dist, torch.device("cpu"), {"X": torch.from_numpy(bad)})
        except RuntimeError:
            caught = True
        assert caught, "differing replicas were not detected"
        if rank == 0:
            fi, fd = o.Ohnsw.knn_batch_bigarray(g0, sp, Q, k=5, ef=20, ties=o.TIES_CANONICAL)
            ok = np.array_equal(ai.numpy(), fi) and np.array_equal(ad.numpy().view(np.uint32), fd.view(np.uint32))
            open(os.path.join(tmp, "ok"), "w").write("1" if ok else "0")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nq", [(2, 64), (3, 50)])
def test_sharded_equals_single_process(tmp_path, world, nq):
    mp.spawn(_worker, args=(world, _free_port(), nq, str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / "ok").read() == "1"


def test_shard_bounds_cover_the_batch():
    import ocaml_hnsw_amd.sharding as sh
    for nq in (0, 1, 7, 10000):
        for world in (1, 2, 3, 8):
            b = [sh.shard_bounds(nq, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == nq
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
