"""Query-batch data parallelism: one process per GPU, replicated index, contiguous query shards,
one all-gather of the per-shard results (RCCL over xGMI when the backend is "nccl"; the same code
runs on "gloo" for CPU tests).

The reference's batch loop is a pure map over query columns (lib/ohnsw.ml:883-895), so the batch
shards with no data-path collective other than collecting the results.
"""
import numpy as _np


def shard_bounds(nq, world, rank):
    """Contiguous shards [g*nq/G, (g+1)*nq/G), remainder to the last ranks."""
    lo = (nq * rank) // world
    hi = (nq * (rank + 1)) // world
    return lo, hi


def replicate_graph(dist, dev, hg, M, src=0):
    """Broadcast a flattened graph (deg0, nbr0, upper layers, entry point) from rank `src`.
    `hg` is the exported Hgraph on `src` and ignored elsewhere.  Returns host numpy arrays
    (deg0, nbr0, upper, entry_point) on every rank."""
    import torch
    rank = dist.get_rank()
    meta = torch.zeros(3, dtype=torch.int64, device=dev)
    if rank == src:
        meta[0], meta[1], meta[2] = hg.max_layer, hg.entry_point, hg.n
    dist.broadcast(meta, src)
    max_layer, entry, n = int(meta[0]), int(meta[1]), int(meta[2])

    def bc(arr, shape, dtype):
        t = torch.from_numpy(_np.ascontiguousarray(arr)).to(dev) if rank == src else torch.empty(shape, dtype=dtype, device=dev)
        dist.broadcast(t, src)
        return t.cpu().numpy()

    deg0 = bc(hg.deg0 if rank == src else None, (n,), torch.int32)
    nbr0 = bc(hg.nbr0 if rank == src else None, (n, 2 * M), torch.int32)
    upper = []
    for l in range(max_layer):
        cnt = torch.tensor([len(hg.upper[l][0]) if rank == src else 0], dtype=torch.int64, device=dev)
        dist.broadcast(cnt, src)
        c = int(cnt[0])
        nodes = bc(hg.upper[l][0] if rank == src else None, (c,), torch.int64)
        dg = bc(hg.upper[l][1] if rank == src else None, (c,), torch.int32)
        nb = bc(hg.upper[l][2] if rank == src else None, (c, M), torch.int32)
        upper.append((nodes, dg, nb))
    return deg0, nbr0, upper, entry


def all_gather_results(dist, ids, dists, counts=None):
    """ids [nq_g][k] int32, dists [nq_g][k] fp32 (torch tensors on the backend's device) ->
    concatenation over ranks in rank order.  Equal shards use all_gather_into_tensor (one
    collective per array); unequal shards are padded to the largest."""
    import torch
    world = dist.get_world_size()
    if counts is None:
        counts = [ids.shape[0]] * world
    mx = max(counts)
    k = ids.shape[1]
    if all(c == mx for c in counts) and hasattr(dist, "all_gather_into_tensor"):
        out_i = torch.empty((world * mx, k), dtype=ids.dtype, device=ids.device)
        out_d = torch.empty((world * mx, k), dtype=dists.dtype, device=dists.device)
        dist.all_gather_into_tensor(out_i, ids.contiguous())
        dist.all_gather_into_tensor(out_d, dists.contiguous())
        return out_i, out_d
    pad_i = torch.full((mx, k), -1, dtype=ids.dtype, device=ids.device)
    pad_d = torch.full((mx, k), float("nan"), dtype=dists.dtype, device=dists.device)
    pad_i[:ids.shape[0]] = ids
    pad_d[:dists.shape[0]] = dists
    li = [torch.empty_like(pad_i) for _ in range(world)]
    ld = [torch.empty_like(pad_d) for _ in range(world)]
    dist.all_gather(li, pad_i)
    dist.all_gather(ld, pad_d)
    return (torch.cat([t[:c] for t, c in zip(li, counts)]), torch.cat([t[:c] for t, c in zip(ld, counts)]))


def sharded_search(dist, search_shard, nq, k):
    """One sharded pass over a global batch of `nq` queries: this rank searches its contiguous shard
    (search_shard(lo, hi) -> (ids [hi-lo][k] int32, dists [hi-lo][k] fp32) as torch tensors on the
    backend's device), then one all-gather leaves the full [nq][k] result on every rank.
    bench.py's strong-scaling leg and the gloo tests run exactly this function."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(nq, world, rank)
    ids, dd = search_shard(lo, hi)
    counts = [shard_bounds(nq, world, r)[1] - shard_bounds(nq, world, r)[0] for r in range(world)]
    return all_gather_results(dist, ids, dd, counts)


def assert_same_on_all_ranks(dist, dev, tensors):
    """Every rank generated its own copy of the synthetic data: compare a checksum of each tensor
    (float64 sum and a strided sample's sum) across ranks and fail loudly on any difference."""
    import torch
    world = dist.get_world_size()
    for name, t in tensors.items():
        flat = t.reshape(-1)
        sig = torch.stack([flat.double().sum(), flat[::max(1, flat.numel() // 65536)].double().mul(1.000001).sum(),
                           torch.tensor(float(flat.numel()), dtype=torch.float64, device=flat.device)]).to(dev)
        gathered = [torch.empty_like(sig) for _ in range(world)]
        dist.all_gather(gathered, sig)
        for r, g in enumerate(gathered):
            if not torch.equal(g, gathered[0]):
                raise RuntimeError("rank %d holds different %s than rank 0 (checksums %s vs %s)" % (r, name, g.tolist(), gathered[0].tolist()))
