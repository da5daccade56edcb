#!/usr/bin/env python3
"""tools/sweep.py with the CPU oracle beside it (the oracle is test infrastructure: only code under
tests/ touches it).  Same command line and environment as tools/sweep.py, plus

    ORACLE=1   print the oracle's exact (true visited set) evaluation and hop counts per (ef, k)
    PARITY=n   compare the first n queries of every run with the oracle on the same graph: ids,
               distance bits, hop counts

    PARITY=100 N=10000000 D=96 M=32 EFC=200 K=10 KIND=unit python tests/sweep_with_oracle.py 10000,512,0
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_state = {}


def _oracle_side(ctx):
    from oracle import oracle as o
    if "g" not in _state:
        hg = ctx["hg"]
        hg.export()
        _state["sp"] = (o.Space.ip if ctx["metric"] else o.Space.l2)(ctx["X"], arith=o.TREE16)
        _state["g"] = o.Graph(hg.n, hg.entry_point, hg.deg0, hg.nbr0, hg.upper)
    return o, _state["g"], _state["sp"]


def post_run(ctx):
    import torch
    k, ef = ctx["k"], ctx["ef"]
    if os.environ.get("ORACLE"):
        o, g, sp = _oracle_side(ctx)
        Qs = ctx["Qd"][:200].cpu().numpy()
        r = o.Ohnsw.knn_batch_bigarray(g, sp, Qs, k=k, ef=ef, ties=o.TIES_CANONICAL, counters=True)
        print("   oracle (exact visited set) ef=%d: n_dist=%.0f n_hops=%.0f" % (ef, r[2].mean(), r[3].mean()), flush=True)
    if os.environ.get("PARITY"):
        o, g, sp = _oracle_side(ctx)
        n = min(int(os.environ["PARITY"]), ctx["nq"])
        ctx["go"](True)
        torch.cuda.synchronize()
        oi, od, ond, onh = o.Ohnsw.knn_batch_bigarray(g, sp, ctx["Qd"][:n].cpu().numpy(), k=k, ef=ef, ties=o.TIES_CANONICAL, counters=True)
        gi, gd = ctx["ids"][:n].cpu().numpy(), ctx["dist"][:n].cpu().numpy()
        print("   parity on %d queries: ids %s, distance bits %s, hop counts %s; oracle n_dist %.0f (GPU re-evaluations +%.1f%%)" %
              (n, np.array_equal(gi, oi), np.array_equal(gd.view(np.uint32), od.view(np.uint32)),
               np.array_equal(ctx["nh"][:n].cpu().numpy(), onh), ond.mean(),
               100 * (ctx["nd"][:n].float().mean().item() / ond.mean() - 1)), flush=True)


if __name__ == "__main__":
    spec = importlib.util.spec_from_file_location("sweep", os.path.join(ROOT, "tools", "sweep.py"))
    mod = importlib.util.module_from_spec(spec)
    mod.POST_RUN = [post_run]
    spec.loader.exec_module(mod)
