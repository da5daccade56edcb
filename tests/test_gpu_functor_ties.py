"""The functor path (Hnsw.Ba / Hnsw_algo.Search.search) on tie-heavy data, GPU against the FAITHFUL
oracle (TIES_HEAP = the in-tree pairing heap of lib/hnsw_algo.ml:17-66, restated verbatim-in-behaviour).

What the reference pins in-tree (lib/hnsw.ml:494-506 with lib/hnsw_algo.ml:25-32, 360-364): a neighbour
whose distance EQUALS max(W).d is answered `Inserted`, W keeps the incumbent, and the neighbour is still
pushed to VisitMe and expanded later.  HNSW_SEM_FUNCTOR implements exactly that rule; what it does not
reproduce is the pairing heap's SHAPE-dependent order among equal keys (which of several equally near
candidates VisitMe pops first, which of several equally far members of W remove_max takes): there the
kernel and the oracle's TIES_CANONICAL mode use (distance, id).

So against TIES_HEAP the comparison is tie-class aware:
  * the distance arrays of the full ef-sized W must be bit-equal;
  * ids must be equal as SETS inside every class of equal distance strictly below max(W).d;
  * inside the farthest class (the one max(W) cuts) the ids may be a different subset of the same class.
The test asserts the per-query agreement rate (>= 0.99; measured: 2699 of 2700 query runs, see
profiles/r02_functor_tie_agreement.txt) and prints it; bit-exact agreement with TIES_CANONICAL (the same
rule, (d, id) order) is asserted in full.
"""
import numpy as np
import pytest

from test_gpu_parity import H, _hgraph, _dataset  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _tie_class_compare(gi, gd, oi, od):
    """-> (distance arrays equal, ids equal up to the farthest class) for one query; -1 ids = fill"""
    gv, ov = gi >= 0, oi >= 0
    if gv.sum() != ov.sum():
        return False, False
    gi, gd, oi, od = gi[gv], gd[gv], oi[ov], od[ov]
    if not np.array_equal(gd.view(np.uint32), od.view(np.uint32)):
        return False, False
    if len(gd) == 0:
        return True, True
    inner = gd.view(np.uint32) != gd.view(np.uint32)[-1]          # classes strictly below max(W).d
    ok = True
    for dv in np.unique(gd[inner].view(np.uint32)):
        sel = gd.view(np.uint32) == dv
        ok = ok and set(gi[sel].tolist()) == set(oi[sel].tolist())
    return True, ok


SUITES = [
    ("levels3", dict(kind="levels", levels=3, n=4000, d=6, M=8, efc=60, efs=(16, 64))),
    ("levels8", dict(kind="levels", levels=8, n=4000, d=6, M=8, efc=60, efs=(16, 64))),
    ("levels40", dict(kind="levels", levels=40, n=4000, d=6, M=8, efc=60, efs=(16, 64, 128))),
    ("sift12k", dict(kind="sift", n=12000, d=128, M=16, efc=100, efs=(64, 128))),
]


@pytest.mark.parametrize("name,cfg", SUITES, ids=[s[0] for s in SUITES])
def test_functor_semantics_against_the_pairing_heap_oracle(H, oracle, name, cfg):
    rng = np.random.default_rng(700 + [s[0] for s in SUITES].index(name))
    if cfg["kind"] == "levels":
        X = rng.integers(0, cfg["levels"], size=(cfg["n"], cfg["d"])).astype(np.float32)
        Q = rng.integers(0, cfg["levels"], size=(300, cfg["d"])).astype(np.float32)
    else:
        X = _dataset("sift", cfg["n"], cfg["d"], 5)
        Q = _dataset("sift", 300, cfg["d"], 6)
    sp = oracle.Space.l2(X, arith=oracle.TREE16)
    g = oracle.build_ohnsw(sp, cfg["M"], cfg["efc"], seed=3)
    hg = _hgraph(H, X, g, id_base=0, M=cfg["M"])
    import ocaml_hnsw_amd as A
    for ef in cfg["efs"]:
        gi, gd = A._search(hg, Q, ef, ef, A.FILL_BA, sem=A.SEM_FUNCTOR)
        # (1) the same rule under the (d, id) order: bit-exact, ids included
        cd, ci = oracle.Functor.knn_batch(g, sp, Q, ef, ef, ties=oracle.TIES_CANONICAL, with_ids=True)
        np.testing.assert_array_equal(gi, ci)
        np.testing.assert_array_equal(gd.view(np.uint32), cd.view(np.uint32))
        # (2) the faithful pairing-heap oracle: tie-class-aware agreement
        hd, hi = oracle.Functor.knn_batch(g, sp, Q, ef, ef, ties=oracle.TIES_HEAP, with_ids=True)
        nd = ni = 0
        first_bad = None
        for j in range(len(Q)):
            dok, iok = _tie_class_compare(gi[j], gd[j], hi[j], hd[j])
            nd += dok
            ni += dok and iok
            if not dok and first_bad is None:
                first_bad = j
        rate_d, rate_i = nd / len(Q), ni / len(Q)
        print("functor ties %-8s ef=%3d: distance profile equal %.3f, tie-class ids equal %.3f of %d queries%s"
              % (name, ef, rate_d, rate_i, len(Q), "" if first_bad is None else "  (first differing query %d)" % first_bad))
        # the rule itself is reproduced; what remains is the heap-shape-dependent order among equal keys.
        # A regression to the old accept rule (ties broken by id, tied neighbours never expanded) drops
        # these rates far below the bounds.
        assert rate_d >= 0.99, (name, ef, rate_d)
        assert rate_i >= 0.99, (name, ef, rate_i)


def test_tied_neighbour_is_expanded_although_it_never_enters_w(H, oracle):
    """A hand-made reproducer of the pinned behaviour: target 0 on a line of equidistant twins.
    Node 1 and node 2 sit at the same distance; with ef = 1 the second one evaluated ties with max(W),
    is answered Inserted, W keeps the incumbent -- and the twin is still expanded, which is the only way
    to reach the true nearest neighbour 3 behind it."""
    #   values: node0 = 5 (entry), node1 = +2, node2 = -2 (twin of 1), node3 = 0.5 reachable only via 2
    vals = np.array([[5.0], [2.0], [-2.0], [0.5]], np.float32)
    adj = [[1, 2], [0], [0, 3], [2]]
    g = oracle.Graph.from_lists(adj, entry_point=0)
    sp = oracle.Space.l2(vals, arith=oracle.TREE16)
    hg = _hgraph(H, vals, g, id_base=0, M=2)
    import ocaml_hnsw_amd as A
    Q = np.zeros((1, 1), np.float32)
    for ties in (oracle.TIES_HEAP, oracle.TIES_CANONICAL):
        od, oi = oracle.Functor.knn_batch(g, sp, Q, 1, 1, ties=ties, with_ids=True)
        assert oi[0, 0] == 3 and od[0, 0] == 0.5
    gi, gd = A._search(hg, Q, 1, 1, A.FILL_BA, sem=A.SEM_FUNCTOR)
    assert gi[0, 0] == 3 and gd[0, 0] == 0.5
    # the imperative path (strict <, lib/ohnsw.ml:574) never expands the twin and stops at node 1
    oi2, od2 = oracle.Ohnsw.knn_batch_bigarray(g, sp, Q, k=1, ef=1, ties=oracle.TIES_CANONICAL)
    gi2, gd2 = H.Ohnsw.knn_batch_bigarray(hg, 1, Q, ef=1)
    assert gi2[0, 0] == oi2[0, 0] and gd2[0, 0] == od2[0, 0]
