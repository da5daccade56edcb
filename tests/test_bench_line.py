"""bench.py's ONE stdout line (round 5's 25 KB line fell off the driver's record: BENCH_r05.json "parsed": null), the aggregation
of the live rocprofv3 counter rows, and the self-launch of `python3 bench.py --gpus N` typed without a launcher -- all on CPU.

The canned result is a real one: profiles/r05_bench_n1.json, the full dictionary round 5's bench.py printed on an MI355X."""
import copy
import json
import os
import sys
import textwrap

import pytest

from conftest import ROOT

import bench

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def canned():
    with open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")) as f:
        d = json.load(f)
    d["roofline"]["latency_floor"] = {"longest_walk_hops": 263, "longest_walk_hops_mean_over_batches": 241.5, "longest_walk_hops_max_over_batches": 263,
                                      "mean_hops": 133.2, "lone_launch_ms": 0.2712, "lone_hop_us": 1.0312, "floor_ms": 0.249,
                                      "lone_launch_results_equal": True, "kernel_ms": 0.312, "kernel_over_floor": 1.253, "wave_occupancy": 0.7528}
    d["one_process"] = {"value": 4.1e7, "unit": "queries/s", "n_gpus": 2, "global_batch": 20000, "ms_per_step": 0.48, "scaling": "weak",
                        "devices": [0, 1], "exchange": "rccl", "equals_single_device": True, "what": "prose " * 40}
    return d


def strict(line):
    def no_constant(name):
        raise AssertionError("non-strict JSON token %s" % name)
    return json.loads(line, parse_constant=no_constant)


def test_driver_line_is_short_strict_and_complete():
    line = bench.driver_line(canned())
    assert "\n" not in line and len(line.encode()) < bench.LINE_LIMIT == 4096
    out = strict(line)
    for key in REQUIRED:
        assert key in out, key
    assert out["value"] == 21935123.1 and out["n_gpus"] == 1 and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert len(out["config"]["workload"]) <= 300
    r = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "bytes_per_query", "latency_floor"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["latency_floor"]["kernel_over_floor"] == 1.253
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "queries/s" and c["sample"]
    assert out["checks"]["failed"] == [] and out["checks"]["passed"] == 10 and out["checks"]["parity_queries"] == 4000 and out["recall_at_10"] == 1.0
    assert out["harder_set_at_recall_gate"]["ef"] == 176 and out["harder_set_at_recall_gate"]["parity"] is True
    assert set(out["others"]) == {"C1", "C3", "C3_clustered", "C5", "C5_clustered"}
    assert out["others"]["C5_clustered"]["parity"] is True and out["others"]["C5"]["frac"] == 0.7838
    assert out["one_process"]["equals_single_device"] is True and "what" not in out["one_process"]
    # numbers only: no prose field of the detail survives (the longest strings are the workload and the sample)
    def longest(o):
        if isinstance(o, dict):
            return max([longest(v) for v in o.values()] + [0])
        return len(o) if isinstance(o, str) else 0
    assert longest({k: v for k, v in out.items() if k != "config"}) <= 160


def test_a_failed_check_is_named_on_the_line():
    d = canned()
    d["checks"]["parity_dist_bits_equal"] = False
    out = strict(bench.driver_line(d))
    assert out["checks"]["failed"] == ["parity_dist_bits_equal"] and out["checks"]["passed"] == 9


def test_driver_line_has_no_nan_or_infinity():
    d = canned()
    d["roofline"]["frac"] = float("nan")
    d["others"]["C3"]["value"] = float("inf")
    d["bench_dist"]["gathered_TBps"] = float("-inf")
    out = strict(bench.driver_line(d))
    assert out["roofline"]["frac"] is None and out["others"]["C3"]["value"] is None and out["bench_dist"]["gathered_TBps"] is None


def test_driver_line_drops_optional_objects_before_it_outgrows_the_limit():
    d = canned()
    d["checks"] = dict(d["checks"], **{"count_%03d" % i: i for i in range(400)})        # a future leg that adds 6 KB of counters
    line = bench.driver_line(d)
    assert len(line) < bench.LINE_LIMIT
    out = strict(line)
    assert out["truncated"] is True and out["checks"] is None
    for key in REQUIRED:                                   # the contract's keys never go
        assert out[key] is not None or key == "vs_baseline", key


def test_driver_line_of_a_minimal_multi_gpu_result():
    d = copy.deepcopy(canned())
    for k in ("float32_rows", "harder_set_at_recall_gate", "others", "bench_dist", "functor_api", "cold", "cpu_baseline"):
        d[k] = None
    d["n_gpus"] = 8
    d["strong"] = {"value": 1.0e8, "unit": "queries/s", "global_batch": 10000, "n_gpus": 8, "ms_per_step": 0.1, "scaling": "strong",
                   "equals_single_device": True, "what": "..."}
    out = strict(bench.driver_line(d))
    assert out["n_gpus"] == 8 and out["cpu_baseline"] is None and out["strong"]["equals_single_device"] is True


def _row(kernel, grid, counter, value):
    return {"Kernel_Name": kernel, "Grid_Size": str(grid), "Counter_Name": counter, "Counter_Value": str(value)}


def test_counter_means_keys_by_kernel_instance_and_launch_size():
    k2 = "void hnsw_dev::hnsw_search_kernel<2, 4, 2, 0, 0, 2, 0>(hnsw_dev::IndexView, hnsw_dev::SearchArgs)"
    k1 = "void hnsw_dev::hnsw_search_kernel<2, 4, 2, 0, 0, 1, 0>(hnsw_dev::IndexView, hnsw_dev::SearchArgs)"
    rows = []
    # an index construction's warm_up: one query through the same instance, twice per handle (plain + ordered launch)
    rows += [_row(k2, 64, "FETCH_SIZE", 3.0) for _ in range(4)]
    # the handle's visited-structure measurement: 256 queries
    rows += [_row(k1, 64 * 256, "FETCH_SIZE", 9000.0) for _ in range(2)]
    # the launches that count: 10 000 queries, five per row format
    rows += [_row(k2, 64 * 10000, "FETCH_SIZE", 600000.0 + i) for i in range(5)]
    rows += [_row(k2, 64 * 10000, "SQ_INSTS_SALU", 7.0e7) for _ in range(5)]
    rows += [_row(k1, 64 * 10000, "FETCH_SIZE", 2100000.0) for _ in range(5)]
    rows += [_row("void hnsw_dev::hnsw_descent_kernel<2>(...)", 64 * 10000, "FETCH_SIZE", 1.0)]      # not a search kernel
    rows += [_row(k2, "", "FETCH_SIZE", 1.0)]                                                       # a row without a grid
    m = bench.counter_means(rows, 10000)
    assert set(m) == {("hnsw_search_kernel<2,4,2,0,0,2,0>", 10000), ("hnsw_search_kernel<2,4,2,0,0,1,0>", 10000)}
    assert m[("hnsw_search_kernel<2,4,2,0,0,2,0>", 10000)] == {"FETCH_SIZE": 600002.0, "SQ_INSTS_SALU": 7.0e7}
    assert m[("hnsw_search_kernel<2,4,2,0,0,1,0>", 10000)] == {"FETCH_SIZE": 2100000.0}
    # round 5's mistake, for the record: all nine dispatches of the instance in one mean is 0.56 of the truth
    diluted = (4 * 3.0 + 5 * 600002.0) / 9
    assert diluted / m[("hnsw_search_kernel<2,4,2,0,0,2,0>", 10000)]["FETCH_SIZE"] < 0.6
    # several launch sizes on request
    m2 = bench.counter_means(rows, (1, 256))
    assert m2[("hnsw_search_kernel<2,4,2,0,0,2,0>", 1)]["FETCH_SIZE"] == 3.0
    assert m2[("hnsw_search_kernel<2,4,2,0,0,1,0>", 256)]["FETCH_SIZE"] == 9000.0
    # the name bench.py derives for a shape is the name the rows are keyed by
    assert bench.search_kernel_name(128, 128, 0, 0, 2) == "hnsw_search_kernel<2,4,2,0,0,2,0>"
    assert bench.search_kernel_name(128, 128, 0, 0, -1) == "hnsw_search_kernel<2,4,2,0,0,1,0>"


@pytest.mark.timeout(300)
def test_self_launch_starts_the_ranks_and_relays_rank_zeros_line(tmp_path):
    """`python3 bench.py --gpus 2` without a launcher: bench.self_launch starts the ranks through torch.distributed.run on
    127.0.0.1 as fresh children, hands back rank 0's one line and the children's status (a stand-in script on gloo here;
    bench.py itself needs a GPU)."""
    script = tmp_path / "ranks.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        print("noise on stdout from rank %d" % dist.get_rank(), flush=True)
        if dist.get_rank() == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "sum": t.item(), "argv": sys.argv[1:],
                              "master": os.environ["MASTER_ADDR"]}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(3 if "--fail" in sys.argv and dist.get_rank() == 1 else 0)
    """))
    rc, line = bench.self_launch(["--gpus", "2", "--backend", "gloo"], 2, script=str(script))
    assert rc == 0
    out = json.loads(line)
    assert out == {"n_gpus": 2, "sum": 3.0, "argv": ["--gpus", "2", "--backend", "gloo"], "master": "127.0.0.1"}
    rc, _ = bench.self_launch(["--fail"], 2, script=str(script))
    assert rc != 0                       # a failing rank fails the launch
