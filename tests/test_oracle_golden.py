"""The oracle against the reference's own 34 inline known-answer tests
(lib/ohnsw.ml:514-534, 593-644, 665-764), transcribed in tests/golden/ohnsw_inline_tests.json."""
import pytest

from conftest import load_golden

G = load_golden("ohnsw_inline_tests.json")


def _graph(o, kind, n):
    if kind == "isolated":  # Graph.create n
        return o.Graph.from_lists([[] for _ in range(n)])
    return o.Graph.ring(n)  # Graph.Test.create_loop


def test_golden_count():
    assert len(G["search_one"]) == 7 and len(G["search_k"]) == 7 and len(G["select_neighbours"]) == 20


@pytest.mark.parametrize("case", G["search_one"], ids=lambda c: c["ref"])
@pytest.mark.parametrize("paper", [False, True])
def test_search_one(oracle, case, paper):
    o = oracle
    g = _graph(o, case["graph"], len(case["values"]))
    sp = o.Space.scalar(case["values"])
    assert o.Ohnsw.search_one(g, sp, case["start"], case["target"], paper=paper) == case["expect"]
    # the functor path's search_one (lib/hnsw_algo.ml:393-437) finds the same node
    node, _ = o.Functor.search_one(g, sp, case["start"], case["target"])
    assert node == case["expect"]


@pytest.mark.parametrize("case", G["search_k"], ids=lambda c: c["ref"])
@pytest.mark.parametrize("ties", [0, 1])
def test_search_k(oracle, case, ties):
    o = oracle
    g = _graph(o, case["graph"], len(case["values"]))
    sp = o.Space.scalar(case["values"])
    got = o.Ohnsw.search_k(g, sp, [case["start"]], case["target"], case["k"], ties=ties)
    assert [n for n, _ in got] == [n for n, _ in case["expect"]]
    for (_, d), (_, e) in zip(got, case["expect"]):
        assert d == pytest.approx(e, abs=1e-12)
    # Hnsw_algo.Search.search (lib/hnsw_algo.ml:350-391) returns the same W on these inputs
    got_f = o.Functor.search(g, sp, [case["start"]], case["target"], case["k"], ties=ties)
    assert [n for n, _ in got_f] == [n for n, _ in case["expect"]]


@pytest.mark.parametrize("case", G["select_neighbours"], ids=lambda c: c["ref"])
@pytest.mark.parametrize("ties", [0, 1])
def test_select_neighbours(oracle, case, ties):
    o = oracle
    sp = o.Space.scalar(case["values"])
    got = o.Ohnsw.select_neighbours(sp, case["candidates"], case["target"], case["M"], ties=ties)
    assert sorted(got) == case["expect"]  # Neighbours.Test.to_sorted_list, lib/ohnsw.ml:141
