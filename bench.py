#!/usr/bin/env python3
"""bench.py -- queries/s of the MI355X HNSW search path on BASELINE.json's headline config.

Workload (configs[1], "C2"): SIFT1M-shaped synthetic data -- n = 1,000,000 x d = 128 clustered
integers 0..218 stored as fp32, M = 16, efConstruction = 200; search ef = 128, k = 10, a batch of
10,000 queries per GPU.  One step = one pass of the hot path (Ohnsw.knn_batch_bigarray,
lib/ohnsw.ml:877-897) over the batch, queries already resident in HBM; with N > 1 every rank
holds a replica of the index, searches its own 10,000-query shard and the per-shard results are
all-gathered over RCCL (weak scaling: per-GPU work fixed).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`value` is measured the way the reference's benchmark times knn_batch (benchmark/benchmark.ml:89-96) and SURVEY 8d
prescribes: host matrices in, host results out, one synchronous call per batch -- hnsw_search_batch, the body of
Ohnsw.knn_batch_bigarray (H2D of the queries + ordering pre-pass + search kernel + D2H of the results; the caller's
matrices in page-locked memory -- hnsw_host_alloc, or registered once with hnsw_host_register -- as a benchmark loop
that reuses its Bigarrays would have them: the device then reads and writes them directly).  Named beside it:
`device_resident` (hnsw_search_batch_device: queries already in HBM, results left there), `float32_rows` (the same
batch through the general-format kernel), `harder_set_at_recall_gate`, `functor_api` (Hnsw.Ba.knn_batch: the functor
module's accept rule through the same protocol), `drop_in` (pageable matrices; two requests in flight), `secondary` (a harder SIFT-like set), `others` (C3, C5), and for N > 1 `strong` (C4 as BASELINE.json words it:
ONE 10 k batch split over the N GPUs).  `roofline` is the search kernel's: its duration comes from HIP events the
library records around its own launches inside the timed region.

Prints ONE JSON line on rank 0.  The oracle (oracle/) is used only as the checker and as the
`cpu_baseline` leg (a single-thread C restatement of the reference's OCaml CPU path).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured streaming)
# the efs tried, in order, when a set misses recall@10 >= 0.95 at BASELINE's ef 128 (steps of 16 up to 192, coarser above)
EF_LADDER = (144, 160, 176, 192, 224, 256, 320, 384, 512, 768, 1024)


def host_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, min(n, 64))


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def make_sift_like(n, d, seed, device, n_centres=4096, sigma=25.0):
    """Clustered integers 0..218 as fp32: Gaussian blobs, clipped, rounded."""
    g = torch.Generator(device=device)
    g.manual_seed(1234)   # centres shared by base and query sets
    centres = torch.randint(20, 200, (n_centres, d), generator=g, device=device).float()
    g.manual_seed(seed)
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    step = 1 << 18
    for s in range(0, n, step):
        m = min(step, n - s)
        idx = torch.randint(0, n_centres, (m,), generator=g, device=device)
        noise = torch.randn((m, d), generator=g, device=device) * sigma
        out[s:s + m] = torch.clamp(torch.round(centres[idx] + noise), 0, 218)
    return out


def brute_force_topk(X, Q, k):
    """exact ground truth on the GPU (ids), integer data => exact fp32 arithmetic"""
    xn = (X * X).sum(1)
    ids = []
    for s in range(0, Q.shape[0], 256):
        q = Q[s:s + 256]
        d2 = xn[None, :] - 2.0 * (q @ X.T) + (q * q).sum(1)[:, None]
        ids.append(torch.topk(d2, k, dim=1, largest=False).indices)
    return torch.cat(ids).cpu().numpy()


def recall_ids(got, gt):
    return float(np.mean([len(set(a) & set(b)) / len(b) for a, b in zip(got.tolist(), gt.tolist())]))


def search_kernel_name(d, ef, metric, semf, rows=-1, blk=0):
    """The hnsw_search_kernel<NCH, RB, NSLOT, METRIC, SEMF, ROWS, BLK> instance the library launches for this shape
    (csrc/hnsw_internal.h: pick_nch / pick_nslot; csrc/hnsw_search_variants.hip: RB per NCH).  rows: 2 = byte rows,
    3 = split rows, -1 = plain float32 rows (1 when every chunk of the lane grid lies inside the row, else 0); blk: 1 = Visited as
    bitmap blocks (Hgraph.visited_blocks(ef) != 0)."""
    nchunks = (d + 3) // 4
    per_lane = (nchunks + 15) // 16
    nch = next(c for c in (1, 2, 4, 8, 16) if per_lane <= c)
    nslot = search_nslot(ef, nch)
    if rows < 0:
        rows = 1 if nchunks == 16 * nch else 0
    rb = {1: 8, 2: 4, 4: 4, 8: 2, 16: 1}[nch] if rows == 2 else {1: 8, 2: 4, 4: 2, 8: 1, 16: 1}[nch]
    return "hnsw_search_kernel<%d,%d,%d,%d,%d,%d,%d>" % (nch, rb, nslot, metric, semf, rows, 1 if blk else 0)


def search_nslot(ef, nch=2):
    """key registers per lane that hold W for this ef (pick_nslot_knn of the library: rows of 65..256 dimensions -- NCH 2 and 4 --
    also have three and six)"""
    for s_ in ((1, 2, 3, 4, 6, 8, 16) if nch in (2, 4) else (1, 2, 4, 8, 16)):
        if ef <= 64 * s_:
            return s_
    return 16


LINE_LIMIT = 4096          # bytes of the ONE stdout line (round 5's 25 KB line fell off the driver's record)
DETAIL_FILE = "bench_detail.json"


def _finite(o):
    """the same structure with every non-finite float replaced by None (strict JSON has no NaN / Infinity)"""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {str(k_): _finite(v_) for k_, v_ in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v_) for v_ in o]
    if isinstance(o, (np.floating, np.integer, np.bool_)):
        return _finite(o.item())
    return o


def _pick(src, *keys, **renamed):
    """{key: src[key]} for the keys src has (None when src is no dict); renamed: out_key = 'src_key'"""
    if not isinstance(src, dict):
        return None
    out = {k_: src[k_] for k_ in keys if k_ in src}
    out.update({k_: src[v_] for k_, v_ in renamed.items() if v_ in src})
    return out


def driver_line(detail):
    """The ONE stdout line: what the driver parses and a judge checks, numbers only, at most LINE_LIMIT bytes of strict JSON.
    `detail` is the full result dictionary (written to DETAIL_FILE and stderr by the caller); every prose field stays there.
    Shape follows the reference's own report -- a rate, a time per call, a recall (benchmark/benchmark.ml:95-98)."""
    d_ = _finite(detail)
    cfg = d_.get("config") or {}
    roof = d_.get("roofline") or {}
    out = {k_: d_.get(k_) for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                      "scaling", "vs_baseline", "dtype", "data")}
    c_ = _pick(cfg, "workload", "n", "d", "M", "ef_construction", "ef", "k", "queries_per_gpu", "global_batch", "parallelism") or {}
    c_["workload"] = str(c_.get("workload", ""))[:300]
    c_["rows"] = str(cfg.get("rows", ""))[:8].split(" ")[0]
    out["config"] = c_
    r_ = _pick(roof, "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "bytes_per_query", "row_bytes",
               "n_dist_per_query", "n_hops_per_query")
    if r_ is not None:
        r_["latency_floor"] = _pick(roof.get("latency_floor"), "longest_walk_hops", "lone_hop_us", "floor_ms", "kernel_over_floor",
                                    "wave_occupancy", "full_load_ms_per_batch")
        alg = (roof.get("bytes_per_query") or 0) * (cfg.get("queries_per_gpu") or 0)
        r_["traffic_over_algorithmic"] = round(roof["traffic"] / alg, 3) if roof.get("traffic") and alg else None
        iss = roof.get("issue") or {}
        r_["issue"] = _pick(iss, "valu", "salu", "clock_GHz", "instructions_per_hop")
    out["roofline"] = r_
    cb = d_.get("cpu_baseline")
    if isinstance(cb, dict):
        ac = cb.get("all_cores") or {}
        out["cpu_baseline"] = dict(_pick(cb, "value", "unit", "cores", "kind", "value_min", "value_max"),
                                   sample=str(cb.get("sample", ""))[:110], all_cores_value=ac.get("value"), all_cores=ac.get("cores"))
    else:
        out["cpu_baseline"] = None
    # the run's self-checks: every boolean one by name only if it FAILED (the names of the passed ones are in the detail), numbers kept
    ck = d_.get("checks")
    if isinstance(ck, dict):
        flags = {k_: v_ for k_, v_ in ck.items() if isinstance(v_, bool)}
        out["checks"] = dict({k_: v_ for k_, v_ in ck.items() if not isinstance(v_, bool)},
                             passed=sum(1 for v_ in flags.values() if v_), failed=sorted(k_ for k_, v_ in flags.items() if not v_))
    else:
        out["checks"] = None
    out["recall_at_10"] = (d_.get("checks") or {}).get("recall_at_10")
    fl, rfl = d_.get("float32_rows"), roof.get("float32_rows") or {}
    out["float32_rows"] = None if not isinstance(fl, dict) else dict(
        _pick(fl, "value", "ms_per_step", "device_resident_value", "frac", "kernel", "kernel_ms"), traffic=rfl.get("traffic"))
    out["device_resident"] = _pick(d_.get("device_resident"), "value", "ms_per_step", "kernel_ms", "prepass_ms")
    out["bandwidth_point"] = _pick(d_.get("bandwidth_point"), "value", "queries_per_gpu", "ms_per_step", "frac")
    hg_ = d_.get("harder_set_at_recall_gate")
    if isinstance(hg_, dict):
        hr, hf, hc = hg_.get("roofline") or {}, hg_.get("float32_rows") or {}, hg_.get("checks") or {}
        out["harder_set_at_recall_gate"] = dict(
            _pick(hg_, "ef", "recall_at_10", "value", "ms_per_step", "device_resident_value"),
            frac=hr.get("frac"), kernel=hr.get("kernel"), kernel_ms=hr.get("kernel_ms"), bytes_per_query=hr.get("bytes_per_query"),
            n_dist_per_query=hr.get("n_dist_per_query"), traffic=hr.get("traffic"),
            instructions_per_hop=(hr.get("issue") or {}).get("instructions_per_hop"),
            parity=(hc.get("parity_ids_equal") and hc.get("parity_dist_bits_equal")) if hc else None,
            float32_rows=_pick(hf, "value", "device_resident_value", "frac", "kernel_ms", "traffic"))
    else:
        out["harder_set_at_recall_gate"] = None
    oth = d_.get("others")
    if isinstance(oth, dict):
        o2 = {}
        for tag, v_ in oth.items():
            if not isinstance(v_, dict):
                continue
            if "skipped" in v_:
                o2[tag] = {"skipped": str(v_["skipped"])[:80]}
                continue
            if tag == "C1":
                ck = v_.get("checks") or {}
                o2[tag] = {"cpu_value": (v_.get("cpu_restatement") or {}).get("value"), "gpu_value": (v_.get("gpu_same_graph") or {}).get("value"),
                           "recall_at_10": v_.get("recall_at_10"), "parity": ck.get("gpu_bits_equal_oracle_kernel_order")}
                continue
            ro, ck = v_.get("roofline") or {}, v_.get("checks") or {}
            o2[tag] = {"value": v_.get("value"), "frac": ro.get("frac"), "kernel_ms": ro.get("kernel_ms"),
                       "recall_at_k": ck.get("recall_at_k"),
                       "parity": (ck.get("parity_ids_equal") and ck.get("parity_dist_bits_equal")) if "parity_ids_equal" in ck else None}
        out["others"] = o2
    else:
        out["others"] = None
    out["bench_dist"] = _pick(d_.get("bench_dist"), "d", "pairs", "ms", "calls_per_s", "gathered_TBps", "frac_of_hbm_peak")
    out["functor_api"] = _pick(d_.get("functor_api"), "value", "ms_per_step")
    out["cold"] = _pick(d_.get("cold"), "first_call_ms", "cold_cache_call_ms")
    for k_ in ("one_process", "strong"):
        out[k_] = _pick(d_.get(k_), "value", "ms_per_step", "n_gpus", "global_batch", "equals_single_device", "exchange", "scaling")
    out["detail"] = DETAIL_FILE
    # never above the limit: whatever a future leg adds, the optional objects go first, the contract's keys never
    line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    for k_ in ("cold", "functor_api", "bench_dist", "bandwidth_point", "others", "device_resident", "strong", "one_process",
               "float32_rows", "harder_set_at_recall_gate", "checks"):
        if len(line) < LINE_LIMIT:
            break
        out[k_] = None
        out["truncated"] = True
        line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:
        raise ValueError("driver line is %d bytes with every optional object dropped" % len(line))
    return line


def counter_means(rows, want_queries, wave=64):
    """Means of rocprofv3 --pmc rows (dicts of a *counter_collection.csv) per search-kernel INSTANCE and LAUNCH SIZE:
    {(kernel instance, e.g. 'hnsw_search_kernel<2,4,2,0,0,2,0>', queries per launch): {counter: mean over the dispatches}}
    for the launches of `want_queries` queries (an int or a collection of ints) only -- an index construction's warm_up
    dispatches one query through the same instance and the handle's visited-structure measurement 256, and neither belongs
    in the mean of the 10 000-query launches (round 5's roofline.traffic was 0.54x for exactly that).  Grid_Size = 64 x queries."""
    import re
    want = {int(want_queries)} if isinstance(want_queries, (int, np.integer)) else {int(w_) for w_ in want_queries}
    acc = {}
    for row in rows:
        m = re.search(r"hnsw_search_kernel<([^>]*)>", (row.get("Kernel_Name") or "").replace(" ", ""))
        if not m:
            continue
        try:
            nq_ = int(row.get("Grid_Size") or row.get("Grid_Size_X") or -1) // wave
        except ValueError:
            continue
        if nq_ not in want:
            continue
        a_ = acc.setdefault(("hnsw_search_kernel<%s>" % m.group(1), nq_), {}).setdefault(row["Counter_Name"], [0.0, 0])
        a_[0] += float(row["Counter_Value"])
        a_[1] += 1
    return {key_: {c_: v_[0] / v_[1] for c_, v_ in cs.items()} for key_, cs in acc.items()}


def self_launch(argv, gpus, extra_env=None, script=None):
    """`python3 bench.py --gpus N` typed without a launcher: start the N ranks as FRESH child processes (this process has
    made no GPU call yet), one per GPU, through torch.distributed.run on 127.0.0.1; relay rank 0's single stdout line and
    return the children's exit status.  Never an exec of a process that has touched the GPU."""
    import socket
    import subprocess
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ, **(extra_env or {}))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    log("starting %d ranks: %s" % (gpus, " ".join(cmd)))
    pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    out, _ = pr.communicate()
    lines = [l_ for l_ in out.decode(errors="replace").splitlines() if l_.startswith("{")]
    return pr.returncode, (lines[-1] if lines else None)


def _stdout_to_stderr():
    """Everything libraries write to fd 1 while the bench runs (RCCL prints a version banner there at
    communicator creation) goes to stderr: stdout carries the ONE JSON line only."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return saved


def _restore_stdout(saved):
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)     # C stdio buffers of native libraries, while fd 1 still is stderr
    except Exception:
        pass
    os.dup2(saved, 1)
    os.close(saved)


def main():
    saved_stdout = _stdout_to_stderr()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--nq", type=int, default=10_000, help="queries per GPU per step")
    ap.add_argument("--M", type=int, default=16)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--cpu-sample", type=int, default=4000, help="queries timed on the CPU restatement")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the harder SIFT-like data point (object `secondary`)")
    ap.add_argument("--no-builder-check", action="store_true", help="skip the batched-vs-sequential builder comparison inside `secondary` (about 75 s)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 counter passes (roofline.traffic / roofline.issue are then null)")
    ap.add_argument("--pmc-child", default=None, help="(internal) load this index file and run a few steps: the process rocprofv3 wraps")
    ap.add_argument("--pmc-phase", default="head", choices=("head", "hard"), help="(internal) which set's query batch the child generates")
    ap.add_argument("--no-bench-dist", action="store_true", help="skip the bench_dist counterpart (object `bench_dist`; about 8 s)")
    ap.add_argument("--no-others", action="store_true", help="skip the C3 / C5 configurations (object `others`; about 100 s)")
    ap.add_argument("--no-clustered", action="store_true", help="skip the clustered variant of C3 inside `others` (its dispatches carry C3's kernel name)")
    ap.add_argument("--pipelined", action="store_true",
                    help="also time the same steps alternated over two HIP streams (extra object `pipelined`, never `value`); "
                         "off by default so that the default run's kernel trace holds serialized launches only")
    ap.add_argument("--dataset", default=None,
                    help="optional real data instead of the synthetic C2 set: an ann-benchmarks .hdf5 file, or a "
                         "directory holding *base.fvecs and *query.fvecs (TEXMEX); n and d come from the file")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; the measured path) or gloo (functional rehearsal of N > 1 on fewer GPUs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "LOCAL_RANK" not in os.environ:
        # typed as the driver types the N = 1 form: start the ranks ourselves, as fresh children, before any GPU call here
        rc_, line_ = self_launch(sys.argv[1:], args.gpus)
        _restore_stdout(saved_stdout)
        if line_:
            print(line_, flush=True)
        raise SystemExit(rc_ if rc_ else (0 if line_ else 1))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d inside a torch.distributed.run of %d ranks" % (args.gpus, world))
    import ocaml_hnsw_amd as H
    H.load()
    if H.device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the search path has no CPU fallback")
    if args.pmc_child:
        # the process a rocprofv3 --pmc pass wraps: ONE index file the parent saved (--pmc-phase head: the headline's, the
        # parent's query batch by its seed; hard: the harder set's at the ef of its recall gate), five launches through the
        # byte rows and five through the float32 rows.  One phase per process: the rows of a pass then belong to one index.
        torch.cuda.set_device(0)
        dev_ = torch.device("cuda", 0)
        hgc = H.Hgraph.load(args.pmc_child)
        hard_ = args.pmc_phase == "hard"
        Qc = make_sift_like(args.nq, args.d, seed=2, device=dev_, **({"n_centres": 256, "sigma": 40.0} if hard_ else {}))
        ic = torch.empty((args.nq, args.k), dtype=torch.int32, device=dev_)
        dc = torch.empty((args.nq, args.k), dtype=torch.float32, device=dev_)
        stc = torch.cuda.current_stream()
        for rows_ in (1, 0):
            hgc.set_option("byte_rows", rows_)
            for _ in range(5):
                H.search_batch_device(hgc, Qc.data_ptr(), args.nq, args.d, args.ef, args.k, ic.data_ptr(), dc.data_ptr(), 0, 0, 0, stc.cuda_stream)
            torch.cuda.synchronize()
        _restore_stdout(saved_stdout)
        return
    gpu = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where collectives run
    dist = None
    # BENCH_FORCE_DIST=1 runs the N > 1 code path (RCCL init, broadcast, async all-gather, barrier) with
    # world_size 1 -- the only way to rehearse the nccl backend on a one-GPU box
    multi = world > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        # a host-side group: ranks that wait while rank 0 runs a one-process leg must not spin a kernel on their GPU
        host_pg = dist.new_group(backend="gloo") if args.backend == "nccl" else None

    n, d, nq, k, ef = args.n, args.d, args.nq, args.k, args.ef
    t0 = time.time()
    if args.dataset:
        # benchmark/dataset.ml:76-102 (HDF5) / Makefile:27-28 (TEXMEX): real vectors when the box has them
        import glob
        import ocaml_hnsw_amd.dataset as D
        if os.path.isdir(args.dataset):
            base = sorted(glob.glob(os.path.join(args.dataset, "*base.fvecs")))
            query = sorted(glob.glob(os.path.join(args.dataset, "*query.fvecs")))
            if not base or not query:
                raise SystemExit("--dataset %s: no *base.fvecs / *query.fvecs" % args.dataset)
            Xh, Qh = D.read_fvecs(base[0]), D.read_fvecs(query[0])
        else:
            ds = D.Dataset.read(args.dataset)
            Xh, Qh = ds.train, ds.test
        n, d = Xh.shape
        Xd = torch.from_numpy(Xh).to(dev)
        reps = -(-(world * nq) // Qh.shape[0])
        Qall = torch.from_numpy(np.tile(Qh, (reps, 1))[:world * nq]).to(dev)   # the file's queries, repeated to fill the batch
    else:
        Xd = make_sift_like(n, d, seed=1, device=dev)
        # every rank searches its own shard of the global batch of world * nq queries
        Qall = make_sift_like(world * nq, d, seed=2, device=dev)
    Qd = Qall[rank * nq:(rank + 1) * nq].contiguous()
    # The K timed steps (and the warm-up) ROTATE through NB distinct query batches (seeds 2 .. 9): the reference times one
    # call on queries the caches have never seen (benchmark/benchmark.ml:86-98); K steps on ONE batch would find the rows of
    # the walks in L2 / Infinity Cache from the second step on.  Batch 0 is the batch every check is made on.
    NB = 1 if args.dataset else max(1, int(os.environ.get("BENCH_BATCHES", "8")))
    Qd_b = [Qd] + [make_sift_like(world * nq, d, seed=2 + j, device=dev)[rank * nq:(rank + 1) * nq].contiguous() for j in range(1, NB)]
    if multi:   # every rank generated its own copy of the data: the copies must be identical
        import ocaml_hnsw_amd.sharding as sharding
        sharding.assert_same_on_all_ranks(dist, cdev, {"X": Xd, "Q": Qall})
    X = Xd.cpu().numpy()
    log("data: n=%d d=%d nq/gpu=%d (%.1fs)" % (n, d, nq, time.time() - t0))

    # ---- index: built on the GPU by rank 0, replicated to every rank ----
    t0 = time.time()
    if rank == 0:
        hg = H.Ohnsw.build_batch_bigarray(X, args.M, args.efc, seed=1, device=gpu)
        build_s = time.time() - t0
        log("graph built on the GPU in %.1fs (max_layer %d)" % (build_s, hg.max_layer))
        if multi or not args.no_cpu:
            hg.export()
    if multi:
        deg0, nbr0, upper, entry = sharding.replicate_graph(dist, cdev, hg if rank == 0 else None, args.M)
        if rank != 0:
            hg = H.Hgraph(X, deg0, nbr0, upper, entry_point=entry, id_base=0, max_degree=args.M).to_device(gpu)
        del deg0, nbr0, upper

    # ---- device buffers; the kernel is launched on torch's current stream ----
    # Results of one step live in ONE buffer ([ids | distance bits], int32) so the exchange is a
    # single all-gather; two such buffers alternate so that step i+1's search overlaps step i's
    # all-gather (the collective runs on RCCL's own stream).
    nres = nq * k
    res = [torch.empty(2 * nres, dtype=torch.int32, device=dev) for _ in range(2)]
    ids_v = [r[:nres].view(nq, k) for r in res]
    dist_v = [r[nres:].view(torch.float32).view(nq, k) for r in res]
    ids_d, dist_d = ids_v[0], dist_v[0]
    nd_d = torch.zeros(nq, dtype=torch.int32, device=dev)
    nh_d = torch.zeros(nq, dtype=torch.int32, device=dev)
    st_d = torch.zeros(nq, dtype=torch.int32, device=dev)   # per-query status: bit 0 = tie-overflow list outgrew its LDS slots
    if multi:
        all_res = [torch.empty(world * 2 * nres, dtype=torch.int32, device=cdev) for _ in range(2)]
    stream = torch.cuda.current_stream()
    # row format the knn kernel reads: byte rows (lossless, hnsw_rows8.hip) when every value is an integer in 0..255
    row_bytes = hg.row_bytes()
    byte_rows = row_bytes == d

    def kernel_name(bytes_):
        return search_kernel_name(d, ef, 0, 0, 2 if bytes_ else -1)

    def search(ef_, counters=False, slot=0, j=0):
        H.search_batch_device(hg, Qd_b[j % NB].data_ptr(), nq, d, ef_, k, ids_v[slot].data_ptr(), dist_v[slot].data_ptr(),
                              nd_d.data_ptr() if counters else 0, nh_d.data_ptr() if counters else 0,
                              st_d.data_ptr(), stream.cuda_stream)

    def gather(slot):
        """the exchange step: per-shard results -> every rank (one RCCL all-gather over xGMI), async"""
        src = res[slot] if args.backend == "nccl" else res[slot].cpu()
        return dist.all_gather_into_tensor(all_res[slot], src, async_op=True)

    def run_steps(ef_, steps, ev=None, j0=0):
        works = []
        for i in range(steps):
            slot = i & 1
            if multi and i >= 2:
                works[i - 2].wait()          # the gather that read this slot two steps ago is done
            if ev:
                ev[i][0].record(stream)
            search(ef_, slot=slot, j=j0 + i)
            if ev:
                ev[i][1].record(stream)
            if multi:
                works.append(gather(slot))
        for w in works[-2:]:
            w.wait()

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(ef_, steps, warmup, instrument=True):
        """device-resident steps (hnsw_search_batch_device: queries already in HBM, results left there).  instrument: HIP
        events around every step and (library option time_kernels) around the library's own launches -- five event records
        per step, which cost the stream about 0.02 ms per step; the pass that `device_resident.value` is quoted on runs
        without them, the instrumented pass of the same steps gives the kernel durations."""
        run_steps(ef_, warmup)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] if instrument else None
        sync()
        hg.set_option("time_kernels", 1 if instrument else 0)     # HIP events around the library's own launches, on the launch stream
        hg.kernel_times()
        t = time.perf_counter()
        run_steps(ef_, steps, ev, j0=warmup)
        sync()
        wall = time.perf_counter() - t
        if instrument:
            lib_times.update(zip(("search_ms", "prepass_ms", "calls"), hg.kernel_times()))
        hg.set_option("time_kernels", 0)
        per_step = sorted(a.elapsed_time(b) for a, b in ev) if instrument else [1e3 * wall / steps]
        if instrument:
            step_stats.update({"median": round(per_step[len(per_step) // 2], 4), "min": round(per_step[0], 4), "max": round(per_step[-1], 4)})
        kern_ms = float(np.mean(per_step))
        if multi:
            w = torch.tensor([wall], dtype=torch.float64, device=cdev)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            wall = float(w[0])
        return wall, kern_ms

    # ---- the headline protocol (benchmark/benchmark.ml:89-96): host matrices in, host results out, one
    #      synchronous call per batch.  N = 1: hnsw_search_batch itself (= the body of Ohnsw.knn_batch_bigarray) on the
    #      caller's registered matrices.  N > 1 (one process per GPU): the same sequence per rank with the exchange in it
    #      -- H2D of the rank's query shard, search, all-gather of the per-shard results over RCCL (the full table ends
    #      up resident on every GPU), D2H of the rank's own shard, stream synchronisation. ----
    # the caller's matrices live in page-locked memory (hnsw_host_alloc; hnsw_host_register on an mmap-backed array is the
    # same to the library): the device reads the queries and writes the results directly, no copies
    Qh_b = []
    for j in range(NB):
        q_ = H.host_empty((nq, d), np.float32)
        q_[:] = Qd_b[j].cpu().numpy()
        Qh_b.append(q_)
    Qh = Qh_b[0]
    if not multi:
        host_i = H.host_empty((nq, k), np.int32)
        host_d = H.host_empty((nq, k), np.float32)
    else:
        host_res = torch.empty(2 * nres, dtype=torch.int32).pin_memory()

    def host_step(ef_, slot=0, j=0, q_=None):
        q_ = Qh_b[j % NB] if q_ is None else q_
        if not multi:
            H.Ohnsw.knn_batch_bigarray(hg, k, q_, ef=ef_, out=(host_i, host_d))
            return
        # hnsw_search_batch_h2d: the rank's registered query matrix is read by the device directly (as in hnsw_search_batch),
        # the shard's results stay in HBM for the exchange
        H.search_batch_h2d(hg, q_, ef_, k, ids_v[slot].data_ptr(), dist_v[slot].data_ptr(), 0, 0, st_d.data_ptr(), stream.cuda_stream)
        with torch.cuda.stream(stream):
            gather(slot).wait()                           # the current stream waits for the collective
            host_res.copy_(res[slot], non_blocking=True)
        stream.synchronize()

    def timed_host(ef_, steps, warmup, instrument=False):
        """K synchronous host-protocol steps between two synchronisation points.  instrument: the library brackets its own
        launches with HIP events (option time_kernels: three event records per call, about 0.009 ms); the pass `value` is
        quoted on runs without them, an instrumented pass of the same steps gives the kernel durations of the roofline."""
        for i in range(warmup):
            host_step(ef_, i & 1, i)
        sync()
        hg.set_option("time_kernels", 1 if instrument else 0)
        hg.kernel_times()
        ts = []
        t = time.perf_counter()
        for i in range(steps):
            t1 = time.perf_counter()
            host_step(ef_, i & 1, warmup + i)
            ts.append(1e3 * (time.perf_counter() - t1))
        last_host_batch[0] = (warmup + steps - 1) % NB
        sync()
        wall = time.perf_counter() - t
        lt = dict(zip(("search_ms", "prepass_ms", "calls"), hg.kernel_times()))
        hg.set_option("time_kernels", 0)
        ts.sort()
        if multi:
            w = torch.tensor([wall], dtype=torch.float64, device=cdev)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            wall = float(w[0])
        return wall, {"median": round(ts[len(ts) // 2], 4), "min": round(ts[0], 4), "max": round(ts[-1], 4)}, lt

    lib_times = {}
    step_stats = {}
    last_host_batch = [0]
    # ---- the very first call after the index upload (nothing of this index in any cache; its scratch buffers allocated in it --
    #      the kernels' code objects, the handle's stream and flag word are paid at index construction), then calls with the
    #      caches flushed in front (2 GiB written) on batches the device has never seen: what ONE call of the reference's benchmark
    #      (benchmark/benchmark.ml:89-96) pays when it is not the twentieth of its kind ----
    cold = None
    if not args.dataset:
        sync()
        t1 = time.perf_counter()
        host_step(ef, 0, 0)
        sync()
        first_ms = 1e3 * (time.perf_counter() - t1)
        flush = torch.empty(1 << 31, dtype=torch.uint8, device=dev)
        qc = H.host_empty((nq, d), np.float32)
        cts = []
        for j in range(3):
            qc[:] = make_sift_like(world * nq, d, seed=40 + j, device=dev)[rank * nq:(rank + 1) * nq].cpu().numpy()
            flush.fill_(j + 1)
            sync()
            t1 = time.perf_counter()
            host_step(ef, 0, 0, q_=qc)
            if multi:
                sync()
            cts.append(1e3 * (time.perf_counter() - t1))
        del flush
        cts.sort()
        cold = {"first_call_ms": round(first_ms, 4), "cold_cache_call_ms": round(cts[1], 4), "cold_cache_call_ms_min": round(cts[0], 4),
                "cold_cache_call_ms_max": round(cts[2], 4),
                "what": "first_call_ms: the first search call after the index upload (its scratch buffers are allocated in it; the code objects of "
                        "the kernels, the handle's stream and flag word are paid at index construction since round 5: warm_up in hnsw_capi.hip -- "
                        "7 ms before); cold_cache_call_ms: median of 3 later calls of the headline protocol, each on a batch never searched "
                        "before and behind a 2 GiB write that empties L2 and the Infinity Cache"}
        log("first call after the index upload %.3f ms; calls on unseen batches behind a cache flush %.3f ms (median of 3)" % (first_ms, cts[1]))
    # the K timed steps of `value`, with the library's kernel events in them (roofline.kernel_ms is measured over THIS region;
    # three event records per call cost 0.003-0.015 ms of a 0.44 ms step), and once more without, for the record
    wall, host_stats, host_lib = timed_host(ef, args.steps, args.warmup, instrument=True)
    wall_hi = wall
    wall_plain, _, _ = timed_host(ef, args.steps, 1)
    qps = world * nq * args.steps / wall
    search_ms, prepass_ms = host_lib["search_ms"], host_lib["prepass_ms"]
    log("ef=%d, host matrices in and out: %.0f q/s, %.3f ms/step (median %.3f; %.3f without the kernel events); per step: ordering pre-pass %.3f ms + search kernel %.3f ms%s" %
        (ef, qps, 1e3 * wall / args.steps, host_stats["median"], 1e3 * wall_plain / args.steps, prepass_ms, search_ms, " [byte rows]" if byte_rows else ""))
    # ---- the same steps with the queries already resident in HBM and the results left there (hnsw_search_batch_device) ----
    wall_dev, _ = timed(ef, args.steps, args.warmup, instrument=False)
    wall_dev_i, kern_ms = timed(ef, args.steps, 1)
    headline_steps = dict(step_stats)
    dev_lib = dict(lib_times)
    qps_dev = world * nq * args.steps / wall_dev
    log("ef=%d, device-resident: %.0f q/s, %.3f ms/step; pre-pass %.3f ms + search kernel %.3f ms (device call %.3f ms)" %
        (ef, qps_dev, 1e3 * wall_dev / args.steps, dev_lib["prepass_ms"], dev_lib["search_ms"], kern_ms))
    # ---- the bandwidth point: 100 000 queries per GPU per step, resident (SURVEY 8e "scaling caveat": 10 000 queries are
    #      fewer than the 8192 wave slots plus their refill -- a launch that ends with its longest walk; ten times the batch
    #      shows what the memory system sustains).  N > 1: with the all-gather of the per-shard results in the step. ----
    big = None
    if not args.dataset:
        nqb = int(os.environ.get("BENCH_BIG_NQ", "100000"))
        Qbig = make_sift_like(world * nqb, d, seed=30, device=dev)[rank * nqb:(rank + 1) * nqb].contiguous()
        big_res = torch.empty(2 * nqb * k, dtype=torch.int32, device=dev)
        big_all = torch.empty(world * 2 * nqb * k, dtype=torch.int32, device=cdev) if multi else None

        def big_step():
            H.search_batch_device(hg, Qbig.data_ptr(), nqb, d, ef, k, big_res[:nqb * k].data_ptr(), big_res[nqb * k:].data_ptr(), 0, 0, 0, stream.cuda_stream)
            if multi:
                dist.all_gather_into_tensor(big_all, big_res if args.backend == "nccl" else big_res.cpu())
        big_step()
        sync()
        bsteps = max(3, args.steps // 4)
        t = time.perf_counter()
        for _ in range(bsteps):
            big_step()
        sync()
        bw_ = time.perf_counter() - t
        if multi:
            w_ = torch.tensor([bw_], dtype=torch.float64, device=cdev)
            dist.all_reduce(w_, op=dist.ReduceOp.MAX)
            bw_ = float(w_[0])
        big = {"value": round(world * nqb * bsteps / bw_, 1), "unit": "queries/s", "queries_per_gpu": nqb, "n_gpus": world, "steps": bsteps,
               "ms_per_step": round(1e3 * bw_ / bsteps, 4), "scaling": "weak",
               "what": "hnsw_search_batch_device on %d resident queries per GPU per step%s: the batch size at which a GPU is bound by its "
                       "memory system, not by the length of one walk" % (nqb, ", all-gather of the per-shard results in the step" if multi else "")}
        log("%d queries per GPU per step, resident: %.0f q/s, %.3f ms/step" % (nqb, big["value"], big["ms_per_step"]))
        del Qbig, big_res, big_all

    # ---- the same steps through the float32 rows (never `value`): the general-format kernel, the one the HBM
    #      roofline bounds (a byte row is a quarter of the bytes and leaves that regime) ----
    fp32_leg = None
    if byte_rows and world == 1:
        hg.set_option("byte_rows", 0)
        wall_fh, stats_fh, _ = timed_host(ef, args.steps, args.warmup)
        wall_f, _ = timed(ef, args.steps, args.warmup, instrument=False)
        _, kern_f = timed(ef, args.steps, 1)
        fp32_leg = {"wall": wall_f, "kern_ms": kern_f, "search_ms": lib_times["search_ms"], "prepass_ms": lib_times["prepass_ms"],
                    "host_wall": wall_fh, "host_stats": stats_fh}
        lib_times.clear(); lib_times.update(dev_lib)
        search(ef, counters=True)
        torch.cuda.synchronize()
        fp32_leg["ids"], fp32_leg["dist"] = ids_d.cpu().numpy().copy(), dist_d.cpu().numpy().copy()
        hg.set_option("byte_rows", 1)
        log("float32 rows: host protocol %.0f q/s; device-resident %.0f q/s, %.3f ms/step; pre-pass %.3f ms + search kernel %.3f ms" %
            (nq * args.steps / wall_fh, nq * args.steps / wall_f, 1e3 * wall_f / args.steps, fp32_leg["prepass_ms"], fp32_leg["search_ms"]))
    if not multi:
        host_last = (host_i.copy(), host_d.copy())
        host_last_j = last_host_batch[0]

    # ---- extra (--pipelined, N = 1, not `value`): the same steps alternated over two HIP streams ----
    # A single 10 k-query launch ends with a drain phase (the last queries to start run on a nearly
    # empty chip at their serial hop latency); a caller that keeps batches coming can start batch
    # i+1 while batch i drains.  The OCaml drop-in call is synchronous, so `value` stays the
    # serialized number; this is the sustained rate of the device-pointer entry point.
    pipelined = None
    if world == 1 and args.pipelined:
        side = [torch.cuda.Stream(device=dev) for _ in range(2)]
        def run_pipelined(steps):
            for i in range(steps):
                H.search_batch_device(hg, Qd.data_ptr(), nq, d, ef, k, ids_v[i & 1].data_ptr(), dist_v[i & 1].data_ptr(),
                                      0, 0, 0, side[i & 1].cuda_stream)
        run_pipelined(2)
        torch.cuda.synchronize()
        t = time.perf_counter()
        run_pipelined(args.steps)
        torch.cuda.synchronize()
        pw = time.perf_counter() - t
        pipelined = {"streams": 2, "value": round(nq * args.steps / pw, 1), "unit": "queries/s",
                     "ms_per_step": round(1e3 * pw / args.steps, 4),
                     "note": "consecutive 10k batches alternated over two HIP streams (batch i+1 fills batch i's drain); not `value`"}
        log("two streams: %.0f q/s, %.3f ms/step" % (pipelined["value"], pipelined["ms_per_step"]))

    # ---- recall@10 on rank 0 (exact ground truth on the GPU) ----
    checks = {}
    strong = None
    if multi:   # the gathered table holds every rank's [ids | distances] block at its place
        # (one more step, untimed, on batch 0 -- the timed steps rotate their batches, and the checks below search batch 0)
        last = 0
        search(ef, slot=last, j=0)
        gather(last).wait()
        torch.cuda.synchronize()
        blocks = all_res[last].view(world, 2 * nres)
        checks["gathered_shard_matches"] = bool(torch.equal(blocks[rank].to(dev), res[last]))
        checks["gathered_all_shards_nonempty"] = bool((blocks[:, :nres] >= 0).any(dim=1).all())
        # the gathered result must equal ONE device's search of the whole global batch (rank 0 does it)
        if rank == 0:
            gq = world * nq
            g_ids = torch.empty((gq, k), dtype=torch.int32, device=dev)
            g_dd = torch.empty((gq, k), dtype=torch.float32, device=dev)
            H.search_batch_device(hg, Qall.data_ptr(), gq, d, ef, k, g_ids.data_ptr(), g_dd.data_ptr(), 0, 0, 0, stream.cuda_stream)
            torch.cuda.synchronize()
            b_ids = blocks[:, :nres].reshape(gq, k).to(dev)
            b_dd = blocks[:, nres:].reshape(gq, k).to(dev)
            checks["gathered_equals_single_device"] = bool(torch.equal(b_ids, g_ids) and torch.equal(b_dd, g_dd.view(torch.int32)))
            del g_ids, g_dd, b_ids, b_dd

        # ---- strong scaling, as BASELINE.json words C4: ONE nq-query batch split over the N GPUs, results
        #      all-gathered so that every GPU holds the full [nq][k] table.  Never `value` (the driver computes
        #      scaling from the weak line); nq / N queries per GPU is far below the 7168 resident: a latency point.
        lo_s, hi_s = sharding.shard_bounds(nq, world, rank)
        s_ids = torch.empty((max(hi_s - lo_s, 1), k), dtype=torch.int32, device=dev)
        s_dd = torch.empty((max(hi_s - lo_s, 1), k), dtype=torch.float32, device=dev)
        Qg = Qall[:nq]

        def search_shard(lo, hi):
            if hi > lo:
                H.search_batch_device(hg, Qg[lo:hi].data_ptr(), hi - lo, d, ef, k, s_ids.data_ptr(), s_dd.data_ptr(), 0, 0, 0, stream.cuda_stream)
            i_, d_ = s_ids[:hi - lo], s_dd[:hi - lo]
            if args.backend != "nccl":
                torch.cuda.synchronize()
                i_, d_ = i_.cpu(), d_.cpu()
            return i_, d_

        for _ in range(max(1, args.warmup)):
            full_i, full_d = sharding.sharded_search(dist, search_shard, nq, k)
        sync()
        t = time.perf_counter()
        for _ in range(args.steps):
            full_i, full_d = sharding.sharded_search(dist, search_shard, nq, k)
        sync()
        sw = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device=cdev)
        dist.all_reduce(sw, op=dist.ReduceOp.MAX)
        swall = float(sw[0])
        strong = {"value": round(nq * args.steps / swall, 1), "unit": "queries/s", "global_batch": nq, "n_gpus": world,
                  "ms_per_step": round(1e3 * swall / args.steps, 4), "scaling": "strong",
                  "what": "one %d-query batch split into %d contiguous shards, per-shard search + all-gather of [ids | distances] per step" % (nq, world)}
        if rank == 0:
            one_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
            one_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
            H.search_batch_device(hg, Qg.data_ptr(), nq, d, ef, k, one_i.data_ptr(), one_d.data_ptr(), 0, 0, 0, stream.cuda_stream)
            torch.cuda.synchronize()
            strong["equals_single_device"] = bool(torch.equal(full_i.to(dev), one_i) and torch.equal(full_d.to(dev).view(torch.int32), one_d.view(torch.int32)))
            log("strong scaling (one %d-query batch over %d GPUs): %.0f q/s, %.3f ms/step" % (nq, world, strong["value"], strong["ms_per_step"]))
    search(ef, counters=True)
    torch.cuda.synchronize()
    got = ids_d.cpu().numpy()
    got_dist = dist_d.cpu().numpy()
    gpu_nd = nd_d.cpu().numpy().astype(np.int64)
    gpu_nh = nh_d.cpu().numpy().astype(np.int64)
    # the device-pointer entry point has no exactness fallback: a query whose list of tied, evicted, still
    # expandable entries outgrew its 64 LDS slots is only FLAGGED there (it may then miss neighbours); the
    # host-buffer entry point searches such queries again with a global slab.  Count them.
    flagged = int((st_d & 1).sum().item())
    checks["tie_overflow_flagged"] = flagged

    # ---- what bounds a 10 k launch of the byte-row kernel: it lasts as long as its longest walk.  The hops of the longest walk
    #      of every rotated batch (the kernel counts hops), and a launch that holds ONLY batch 0's 64 longest walks -- each wave
    #      with a CU to itself, the descent in the pre-pass as in the timed launches: its search-kernel duration is the floor no
    #      10 k launch containing those walks can go below, whatever the memory system sustains. ----
    latency_floor = None
    if world == 1 and rank == 0 and nq >= 64:
        longest = []
        for j in range(NB):
            search(ef, counters=True, slot=1, j=j)
            torch.cuda.synchronize()
            longest.append(int(nh_d.max().item()))
        top = torch.argsort(torch.from_numpy(gpu_nh).to(dev), descending=True)[:64]
        Q64 = Qd[top].contiguous()
        i64 = torch.empty((64, k), dtype=torch.int32, device=dev)
        d64 = torch.empty((64, k), dtype=torch.float32, device=dev)
        hg.set_option("order_queries", 1)
        hg.set_option("time_kernels", 1)
        for phase_ in range(2):
            for _ in range(3 if phase_ == 0 else 20):
                H.search_batch_device(hg, Q64.data_ptr(), 64, d, ef, k, i64.data_ptr(), d64.data_ptr(), 0, 0, 0, stream.cuda_stream)
            torch.cuda.synchronize()
            lone_ms, lone_pre, _ = hg.kernel_times()
        hg.set_option("time_kernels", 0)
        hg.set_option("order_queries", -1)
        same64 = bool(torch.equal(i64, ids_d[top]))
        hops64 = int(gpu_nh[top[0].item()])
        latency_floor = {"longest_walk_hops": longest[0], "longest_walk_hops_mean_over_batches": round(float(np.mean(longest)), 1),
                         "longest_walk_hops_max_over_batches": max(longest), "mean_hops": round(float(gpu_nh.mean()), 1),
                         "lone_launch_ms": round(lone_ms, 4), "lone_hop_us": round(1e3 * lone_ms / max(hops64, 1), 4),
                         "floor_ms": round(lone_ms * float(np.mean(longest)) / max(hops64, 1), 4),
                         "lone_launch_results_equal": same64}
        log("latency floor: longest walk %d hops (mean over batches %.1f); 64 longest walks alone %.4f ms = %.3f us per hop" %
            (longest[0], float(np.mean(longest)), lone_ms, 1e3 * lone_ms / max(hops64, 1)))
        del Q64, i64, d64

    # ---- N > 1: the ONE-PROCESS form of C4, the one an OCaml program (a single process) can reach:
    #      hnsw_multi_create on devices 0 .. N-1 + hnsw_multi_search_batch -- host matrices in and out, the batch split into N
    #      contiguous shards, one RCCL all-gather inside the library (lib/ohnsw.ml:883-895 is the map being sharded).  Rank 0
    #      runs it while the other ranks wait on the host; same global batch of N x nq queries as the weak line.  Never `value`. ----
    one_process = None
    if multi:
        sync()
        if rank == 0:
            try:
                ndev = torch.cuda.device_count()
                devs = list(range(world)) if ndev >= world else [g_ % ndev for g_ in range(world)]
                mh = H.MultiHgraph(hg, devs)
                gq = world * nq
                Qm = H.host_empty((gq, d), np.float32)
                Qm[:] = Qall.cpu().numpy()
                for _ in range(max(1, args.warmup)):
                    mi_, md_ = mh.knn_batch_bigarray(k, Qm, ef=ef)
                t = time.perf_counter()
                for _ in range(args.steps):
                    mi_, md_ = mh.knn_batch_bigarray(k, Qm, ef=ef)
                mw = time.perf_counter() - t
                s_i = torch.empty((gq, k), dtype=torch.int32, device=dev)
                s_d = torch.empty((gq, k), dtype=torch.float32, device=dev)
                H.search_batch_device(hg, Qall.data_ptr(), gq, d, ef, k, s_i.data_ptr(), s_d.data_ptr(), 0, 0, 0, stream.cuda_stream)
                torch.cuda.synchronize()
                cnt_ = mh.debug_counters()
                one_process = {"value": round(gq * args.steps / mw, 1), "unit": "queries/s", "n_gpus": world, "global_batch": gq,
                               "ms_per_step": round(1e3 * mw / args.steps, 4), "scaling": "weak", "devices": devs,
                               "exchange": "rccl" if cnt_["peer_copies"] == 0 else "device-to-device copies (replicas share a device)",
                               "equals_single_device": bool(np.array_equal(mi_, s_i.cpu().numpy()) and
                                                            np.array_equal(md_.view(np.uint32), s_d.cpu().numpy().view(np.uint32))),
                               "what": "hnsw_multi_search_batch from ONE process (rank 0; the other ranks wait on the host): host matrices in "
                                       "and out, %d contiguous shards, per-shard search, one exchange, D2H from one device" % world}
                log("one process, %d devices %s: %.0f q/s, %.3f ms/step, equals one device: %s" %
                    (world, devs, one_process["value"], one_process["ms_per_step"], one_process["equals_single_device"]))
                mh.release()
                del s_i, s_d, Qm
            except Exception as e:     # never lose the line to a secondary leg
                one_process = {"skipped": "failed: %r" % (e,)}
                log("one-process leg failed: %r" % (e,))
            torch.cuda.set_device(gpu)     # (the library leaves the process on the last device it touched)
        dist.barrier(group=host_pg) if host_pg is not None else dist.barrier()
    if fp32_leg is not None:
        checks["byte_rows_equal_float32_rows"] = bool(np.array_equal(fp32_leg["ids"], got) and
                                                      np.array_equal(fp32_leg["dist"].view(np.uint32), got_dist.view(np.uint32)))
    ef_ok, qps_ok = None, None

    # ---- the drop-in call: what Ohnsw.knn_batch_bigarray becomes (host matrices in and out, one
    #      synchronous call per batch, benchmark/benchmark.ml:89-96), PCIe copies included ----
    drop_in = None
    functor_result = None
    if world == 1 and rank == 0:
        Qh = Qd.cpu().numpy()
        reps = max(5, min(args.steps, 10))
        hi_, hd_ = H.Ohnsw.knn_batch_bigarray(hg, k, Qh, ef=ef)       # warm-up, and the exact result of every query
        checks["device_call_equals_drop_in"] = bool(np.array_equal(hi_, got) and np.array_equal(hd_.view(np.uint32), got_dist.view(np.uint32)))
        # what the timed headline steps themselves left in the caller's matrices (float32-row leg last: same bits)
        if not multi:
            search(ef, slot=1, j=host_last_j)     # the batch the last timed step searched
            torch.cuda.synchronize()
            checks["headline_steps_results_equal_device_call"] = bool(np.array_equal(host_last[0], ids_v[1].cpu().numpy()) and
                                                                      np.array_equal(host_last[1].view(np.uint32), dist_v[1].cpu().numpy().view(np.uint32)))

        def timed_calls(fn):
            ts = []
            for _ in range(reps):
                t = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t)
            ts.sort()
            return ts[len(ts) // 2], ts[0], ts[-1]

        def leg(med, lo, hi, what):
            return {"value": round(nq / med, 1), "unit": "queries/s", "ms_per_batch": round(1e3 * med, 4),
                    "ms_min": round(1e3 * lo, 4), "ms_max": round(1e3 * hi, 4), "what": what}

        # (a) as the reference's benchmark calls it: fresh pageable matrices (the runtime stages the copies)
        sync_p = timed_calls(lambda: H.Ohnsw.knn_batch_bigarray(hg, k, Qh, ef=ef))
        # (b) the caller's query and result matrices are page-locked (hnsw_host_alloc / hnsw_host_register; an OCaml benchmark
        #     loop passes the same Bigarrays again and again) and the results land in them
        Qp = H.host_empty((nq, d), np.float32)
        Qp[:] = Qh
        oi_ = H.host_empty((nq, k), np.int32)
        od_ = H.host_empty((nq, k), np.float32)
        Qh_pageable, Qh = Qh, Qp
        H.Ohnsw.knn_batch_bigarray(hg, k, Qh, ef=ef, out=(oi_, od_))
        checks["registered_arrays_equal_device_call"] = bool(np.array_equal(oi_, got) and np.array_equal(od_.view(np.uint32), got_dist.view(np.uint32)))
        sync_r = timed_calls(lambda: H.Ohnsw.knn_batch_bigarray(hg, k, Qh, ef=ef, out=(oi_, od_)))
        # (c) the same call in two halves (hnsw_search_submit / hnsw_search_wait), two requests in flight
        inflight = []
        for _ in range(6):      # warm-up: every pooled request has its device buffers before the clock starts
            inflight.append(H.submit(hg, Qh, ef, k))
            if len(inflight) > 2:
                inflight.pop(0).wait(out=(oi_, od_))

        def one_submit_wait():
            inflight.append(H.submit(hg, Qh, ef, k))
            inflight.pop(0).wait(out=(oi_, od_))
        sub = timed_calls(one_submit_wait)
        while inflight:
            inflight.pop(0).wait(out=(oi_, od_))
        # (d) the functor module's call (Hnsw.Ba.knn_batch, lib/hnsw.ml:763-777: accept rule of Hnsw_algo.Search, distances only,
        #     +inf filled): same protocol, same matrices
        fi_ = H.host_empty((nq, k), np.int32)
        fd_ = H.host_empty((nq, k), np.float32)
        H._search(hg, Qh, ef, k, H.FILL_BA, sem=H.SEM_FUNCTOR, out=(fi_, fd_))
        functor_result = (fi_.copy(), fd_.copy())
        sync_f = timed_calls(lambda: H._search(hg, Qh, ef, k, H.FILL_BA, sem=H.SEM_FUNCTOR, out=(fi_, fd_)))
        drop_in = {"synchronous": leg(*sync_r, "the headline protocol again, as the median of %d single calls: hnsw_search_batch, the caller's query / result "
                                               "matrices page-locked (hnsw_host_alloc / hnsw_host_register: queries read and results written by the device directly) -> ordering pre-pass + search kernel, one blocking "
                                               "call per %d-query batch = the body of Ohnsw.knn_batch_bigarray" % (reps, nq)),
                   "synchronous_pageable": leg(*sync_p, "the same call on fresh pageable matrices (copies staged by the runtime)"),
                   "submit_wait_2_in_flight": leg(*sub, "hnsw_search_submit / hnsw_search_wait, two requests in flight, registered matrices"),
                   "functor_api": leg(*sync_f, "the body of Hnsw.Ba.knn_batch: the same blocking call with the functor module's accept rule "
                                               "(Hnsw_algo.Search: a neighbour AT max(W).d is still expanded) and its +inf fill, registered matrices"),
                   "batches_timed": reps, "statistic": "median of the per-call wall times (min / max beside it)"}
        log("drop-in (host buffers): synchronous %.0f q/s (%.3f ms/batch; pageable %.3f ms), submit/wait x2 %.0f q/s (%.3f ms/batch)" %
            (nq / sync_r[0], 1e3 * sync_r[0], 1e3 * sync_p[0], nq / sub[0], 1e3 * sub[0]))
    if rank == 0:
        ns = min(1000, nq)
        gt = brute_force_topk(Xd, Qd[:ns], k)
        rec = recall_ids(got[:ns], gt)
        checks["recall_at_10"] = round(rec, 4)
        # the reference's own definition (benchmark/dataset.ml:105-127): share of the returned distances
        # that are <= the true k-th distance + 1e-8
        import ocaml_hnsw_amd.dataset as D
        gt_t = torch.from_numpy(gt).to(dev)
        true_d = (Xd[gt_t] - Qd[:ns, None, :]).double().pow(2).sum(-1).sqrt().sort(dim=1).values.float().cpu().numpy()
        checks["recall_distance_threshold"] = round(D.Recall.compute(true_d, got_dist[:ns]), 4)
        log("recall@10 at ef=%d: %.4f" % (ef, rec))
    if world == 1 and checks["recall_at_10"] < 0.95:
        # do not tune the data to the gate -- report the ef that reaches it alongside
        for ef2 in EF_LADDER:
            search(ef2)
            torch.cuda.synchronize()
            r2 = recall_ids(ids_d.cpu().numpy()[:ns], gt)
            if r2 >= 0.95:
                w2, _, _ = timed_host(ef2, max(3, args.steps // 2), 1)       # the headline protocol at the ef that meets the gate
                ef_ok, qps_ok = ef2, nq * max(3, args.steps // 2) / w2
                checks["ef_for_recall_0.95"] = ef2
                checks["recall_at_that_ef"] = round(r2, 4)
                checks["qps_at_that_ef"] = round(qps_ok, 1)
                log("recall@10 >= 0.95 first reached at ef=%d (%.4f): %.0f q/s" % (ef2, r2, qps_ok))
                break

    # ---- secondary data point (N = 1): a harder SIFT-like set (256 blobs, sigma 40).  The C2 recipe of
    #      SURVEY 8d (4096 blobs, sigma 25) is benign: recall 1.0 at ef 128 with ~950 evaluations per query,
    #      where real SIFT1M needs 2500-3500.  Same n, d, M, efConstruction, batch; never `value`. ----
    secondary = None
    hard_idx = None      # (index file, ef) of the harder set at its recall gate, for the live counter passes
    if world == 1 and rank == 0 and not args.dataset and not args.no_secondary:
        t0 = time.time()
        X2d = make_sift_like(n, d, seed=1, device=dev, n_centres=256, sigma=40.0)
        Q2d = make_sift_like(nq, d, seed=2, device=dev, n_centres=256, sigma=40.0)
        X2 = X2d.cpu().numpy()
        hg2 = H.Ohnsw.build_batch_bigarray(X2, args.M, args.efc, seed=1, device=gpu)

        def search2(ef_, counters=False):
            H.search_batch_device(hg2, Q2d.data_ptr(), nq, d, ef_, k, ids_v[0].data_ptr(), dist_v[0].data_ptr(),
                                  nd_d.data_ptr() if counters else 0, nh_d.data_ptr() if counters else 0,
                                  st_d.data_ptr(), stream.cuda_stream)

        def timed2(ef_, steps):
            """-> (seconds per step without event records, search kernel ms, pre-pass ms from an instrumented pass of the same steps)"""
            search2(ef_)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(steps):
                search2(ef_)
            torch.cuda.synchronize()
            w = (time.perf_counter() - t) / steps
            hg2.set_option("time_kernels", 1)
            hg2.kernel_times()
            for _ in range(steps):
                search2(ef_)
            torch.cuda.synchronize()
            sm, pm, _ = hg2.kernel_times()
            hg2.set_option("time_kernels", 0)
            return w, sm, pm

        Q2h = H.host_empty((nq, d), np.float32)
        Q2h[:] = Q2d.cpu().numpy()
        h2i, h2d = H.host_empty((nq, k), np.int32), H.host_empty((nq, k), np.float32)

        def timed2_host(ef_, steps):
            """the headline protocol on this set: hnsw_search_batch, registered host matrices in and out"""
            H.Ohnsw.knn_batch_bigarray(hg2, k, Q2h, ef=ef_, out=(h2i, h2d))
            t = time.perf_counter()
            for _ in range(steps):
                H.Ohnsw.knn_batch_bigarray(hg2, k, Q2h, ef=ef_, out=(h2i, h2d))
            return (time.perf_counter() - t) / steps

        ns2 = min(1000, nq)
        gt2 = brute_force_topk(X2d, Q2d[:ns2], k)
        steps2 = max(3, args.steps // 2)
        w2h = timed2_host(ef, steps2)
        w2, sm2, pm2 = timed2(ef, steps2)
        search2(ef, counters=True)
        torch.cuda.synchronize()
        got2, got2_d = ids_v[0].cpu().numpy(), dist_v[0].cpu().numpy()
        rec2 = recall_ids(got2[:ns2], gt2)
        nd2, nh2 = float(nd_d.float().mean().item()), float(nh_d.float().mean().item())
        src2 = "gpu counters (include re-evaluations)"
        sec_checks = {"recall_at_10": round(rec2, 4), "tie_overflow_flagged": int((st_d & 1).sum().item())}
        nu2 = None
        if not args.no_cpu:
            from oracle import oracle as o
            hg2.export()
            s2 = min(500, nq)
            sp2 = o.Space.l2(X2, arith=o.TREE16)
            g2 = o.Graph(hg2.n, hg2.entry_point, hg2.deg0, hg2.nbr0, hg2.upper)
            oi2, od2, ond2, onh2, onu2 = o.Ohnsw.knn_batch_bigarray(g2, sp2, Q2d[:s2].cpu().numpy(), k=k, ef=ef, ties=o.TIES_CANONICAL, split=True)
            nd2, nh2, nu2 = float(ond2.mean()), float(onh2.mean()), float(onu2.mean())
            src2 = "oracle counters on %d queries" % s2
            sec_checks["parity_queries"] = s2
            sec_checks["parity_ids_equal"] = bool(np.array_equal(oi2, got2[:s2]))
            sec_checks["parity_dist_bits_equal"] = bool(np.array_equal(od2.view(np.uint32), got2_d[:s2].view(np.uint32)))
            del sp2, g2
        S2 = 2 * args.M
        rb2 = hg2.row_bytes()
        bq2_total = nd2 * (rb2 + 4) + nh2 * 4 * S2 + 4 * d + 8 * k
        if pm2 > 0 and nu2 is not None:     # ordered launch: the descent ran in its own kernel
            bq2 = (nd2 - nu2) * (rb2 + 4) + nh2 * 4 * S2 + 4 * d + 8 * k + 16
            kms2 = sm2
        else:
            bq2, kms2 = bq2_total, sm2 + pm2
        ach2 = bq2 * nq / (kms2 * 1e-3) / 1e9
        secondary = {"workload": "harder SIFT-like: n=%d d=%d clustered ints 0..218, 256 blobs, sigma 40; M=%d efConstruction=%d, ef=%d k=%d, %d queries"
                                 % (n, d, args.M, args.efc, ef, k, nq),
                     "value": round(nq / w2, 1), "unit": "queries/s", "ms_per_step": round(1e3 * w2, 4), "steps": steps2,
                     "value_host_protocol": round(nq / w2h, 1), "ms_per_step_host_protocol": round(1e3 * w2h, 4),
                     "protocols": "`value`: queries resident in HBM, results left there (hnsw_search_batch_device); `value_host_protocol`: the headline's "
                                  "protocol (hnsw_search_batch, registered host matrices in and out)",
                     "roofline": {"bound": "hbm", "achieved": round(ach2, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(ach2 / HBM_PEAK_GBS, 4), "kernel_ms": round(kms2, 4), "bytes_per_query": round(bq2, 1),
                                  "row_bytes": rb2,
                                  "n_dist_per_query": round(nd2, 1), "n_hops_per_query": round(nh2, 1), "counters": src2},
                     "checks": sec_checks}
        if rb2 == d:     # the same set through its float32 rows
            hg2.set_option("byte_rows", 0)
            wf, smf, pmf = timed2(ef, steps2)
            hg2.set_option("byte_rows", 1)
            bqf = (nd2 - (nu2 or 0)) * (4 * d + 4) + nh2 * 4 * S2 + 4 * d + 8 * k + 16
            secondary["float32_rows"] = {"value": round(nq / wf, 1), "unit": "queries/s", "ms_per_step": round(1e3 * wf, 4),
                                         "kernel_ms": round(smf, 4), "frac": round(bqf * nq / (smf * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if rec2 < 0.95:
            for ef2 in EF_LADDER:
                search2(ef2)
                torch.cuda.synchronize()
                r2 = recall_ids(ids_v[0].cpu().numpy()[:ns2], gt2)
                if r2 >= 0.95:
                    w3h = timed2_host(ef2, steps2)
                    w3, sm3, pm3 = timed2(ef2, steps2)
                    gate = {"ef": ef2, "recall_at_10": round(r2, 4), "value": round(nq / w3h, 1),
                            "unit": "queries/s", "ms_per_step": round(1e3 * w3h, 4),
                            "device_resident_value": round(nq / w3, 1), "device_resident_ms_per_step": round(1e3 * w3, 4),
                            "what": "`value`: the headline's protocol (host matrices in and out) at the smallest ef of the "
                                    "ladder that reaches recall@10 >= 0.95 on this set"}
                    # the kernel that serves this point (W in three registers for ef 129..192, four up to 256) against the HBM line, as the headline's
                    search2(ef2, counters=True)
                    torch.cuda.synchronize()
                    g_nd3 = nd_d.cpu().numpy().astype(np.int64)
                    nd3, nh3, nu3 = float(g_nd3.mean()), float(nh_d.float().mean().item()), None
                    src3 = "gpu counters (include re-evaluations)"
                    if not args.no_cpu:
                        sp3 = o.Space.l2(X2, arith=o.TREE16)
                        g3 = o.Graph(hg2.n, hg2.entry_point, hg2.deg0, hg2.nbr0, hg2.upper)
                        s3 = min(500, nq)
                        oi3, od3, ond3, onh3, onu3 = o.Ohnsw.knn_batch_bigarray(g3, sp3, Q2d[:s3].cpu().numpy(), k=k, ef=ef2, ties=o.TIES_CANONICAL, split=True)
                        gate["checks"] = {"parity_queries": s3, "parity_ids_equal": bool(np.array_equal(oi3, ids_v[0].cpu().numpy()[:s3])),
                                          "parity_dist_bits_equal": bool(np.array_equal(od3.view(np.uint32), dist_v[0].cpu().numpy()[:s3].view(np.uint32))),
                                          "gpu_reevaluation_overhead": round(float(g_nd3[:s3].mean() / max(ond3.mean(), 1) - 1), 4)}
                        nd3, nh3, nu3 = float(ond3.mean()), float(onh3.mean()), float(onu3.mean())
                        src3 = "oracle counters on %d queries" % s3
                        del sp3, g3
                    l03 = pm3 > 0 and nu3 is not None

                    def bq3_of(rb_):
                        return (nd3 - (nu3 if l03 else 0.0)) * (rb_ + 4) + nh3 * 4 * S2 + 4 * d + 8 * k + (16 if l03 else 0)
                    kms3 = sm3 if (l03 or pm3 <= 0) else sm3 + pm3
                    gate["roofline"] = {"bound": "hbm", "kernel": search_kernel_name(d, ef2, 0, 0, 2 if rb2 == d else -1), "kernel_ms": round(kms3, 4),
                                        "prepass_ms": round(pm3, 4), "bytes_per_query": round(bq3_of(rb2), 1), "row_bytes": rb2,
                                        "achieved": round(bq3_of(rb2) * nq / (kms3 * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(bq3_of(rb2) * nq / (kms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                        "n_dist_per_query": round(nd3, 1), "n_hops_per_query": round(nh3, 1), "counters": src3}
                    if rb2 == d:     # ... and through its float32 rows at this ef
                        hg2.set_option("byte_rows", 0)
                        w3fh = timed2_host(ef2, steps2)
                        w3f, sm3f, pm3f = timed2(ef2, steps2)
                        blk3f = hg2.visited_blocks(ef2)      # float32 rows, W in four registers: the handle may take the bitmap blocks here
                        search2(ef2, counters=True)
                        torch.cuda.synchronize()
                        nd3f = float(nd_d.float().mean().item())
                        hg2.set_option("byte_rows", 1)
                        kms3f = sm3f if (l03 or pm3f <= 0) else sm3f + pm3f
                        gate["float32_rows"] = {"value": round(nq / w3fh, 1), "unit": "queries/s", "ms_per_step": round(1e3 * w3fh, 4),
                                                "device_resident_value": round(nq / w3f, 1), "kernel": search_kernel_name(d, ef2, 0, 0, -1, blk3f),
                                                "visited": ("bitmap blocks, 2^%d slots" % blk3f) if blk3f else "tag cache",
                                                "gpu_evaluations_per_query": round(nd3f, 1),
                                                "kernel_ms": round(kms3f, 4), "bytes_per_query": round(bq3_of(4 * d), 1),
                                                "frac": round(bq3_of(4 * d) * nq / (kms3f * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None}
                    secondary["at_recall_0.95"] = gate
                    if not args.no_pmc and ef2 != ef:
                        import tempfile
                        hard_dir = tempfile.mkdtemp(prefix="hnsw_pmc_hard_", dir="/tmp")
                        hard_idx = (os.path.join(hard_dir, "hard.idx"), ef2)
                        hg2.save(hard_idx[0])      # for the live counter passes further down
                    break
        log("secondary (256 blobs, sigma 40): %.0f q/s at ef=%d, recall@10 %.4f, %.0f evaluations/query, frac %.3f (%.1fs)" %
            (nq / w2, ef, rec2, nd2, ach2 / HBM_PEAK_GBS, time.time() - t0))
        hg2.release()
        del hg2
        # Is the batched device builder the reason this set misses the gate at ef 128?  The same recipe at n = 100 000,
        # built twice: default batching, and one node at a time (max_batch = 1 = Ohnsw.insert, lib/ohnsw.ml:766-837,
        # pinned link for link against the reference in tests/test_gpu_build.py).  (At n = 1 M the sequential build
        # takes 12 minutes; there, 32 times smaller batches change recall@10 at ef 128 by -0.009: tools/builder_quality.py.)
        if not args.no_builder_check:
            t0 = time.time()
            nb_, nqb_ = min(100_000, n), 2000
            Xb, Qb = X2[:nb_], Q2d[:nqb_].cpu().numpy()
            gtb = brute_force_topk(X2d[:nb_], Q2d[:nqb_], k)
            br = {}
            for name_, kw_ in (("batched", {}), ("sequential", {"max_batch": 1})):
                hb = H.Ohnsw.build_batch_bigarray(Xb, args.M, args.efc, seed=1, device=gpu, **kw_)
                br[name_] = {"ef_%d" % e_: round(recall_ids(H.Ohnsw.knn_batch_bigarray(hb, k, Qb, ef=e_)[0], gtb), 4) for e_ in (16, 32, 64, 128)}
                hb.release()
            secondary["builder_recall"] = {"n": nb_, "queries": nqb_, "recall_at_10": br,
                                           "what": "same recipe at n = %d: graph built with the default batching vs one node at a time "
                                                   "(= Ohnsw.insert); within 0.01 at every ef and 0.002 from ef 32 on, so the batching is not what costs recall" % nb_}
            log("builder check (n=%d): recall@10 batched %s, sequential %s (%.0fs)" % (nb_, br["batched"], br["sequential"], time.time() - t0))
        del X2d, Q2d, X2

    # ---- the other single-GPU configurations of BASELINE.json on the same driver-run line: C5 (DEEP10M shape, "the honest
    #      HBM run" of SURVEY 8d: 6.4 GB of index, nothing cached) and C3 (GloVe-1.2M shape, inner product, k = 100).
    #      Synthetic unit vectors at the named shape; graph built on this GPU; >= 5 timed steps of the 10 k batch with the
    #      queries resident; algorithmic bytes from the oracle's exact counters on a sample of the same batch, which is
    #      also checked bit for bit.  Never `value`. ----
    others = None
    if world == 1 and rank == 0 and not args.dataset and not args.no_others:
        others = {}
        try:
            import psutil
            free_gb = psutil.virtual_memory().available / 2 ** 30
        except Exception:
            free_gb = None

        def unit_vectors(n_, d_, seed_):
            g_ = torch.Generator(device=dev)
            g_.manual_seed(seed_)
            out_ = np.empty((n_, d_), np.float32)
            for s_ in range(0, n_, 1 << 20):
                m_ = min(1 << 20, n_ - s_)
                x_ = torch.randn((m_, d_), generator=g_, device=dev)
                out_[s_:s_ + m_] = (x_ / x_.norm(dim=1, keepdim=True)).cpu().numpy()
            return out_

        _cc = [float(x_) for x_ in os.environ.get("BENCH_C3_CLUSTER", "256,1.5").split(",")]

        def clustered_unit_vectors(n_, d_, seed_, n_centres=int(_cc[0]), spread=_cc[1]):
            """unit vectors around `n_centres` random directions (word-embedding-like: a low intrinsic dimension), so that
            recall at the configuration's ef means something -- the prescribed N(0,1) data is structureless.  Parameters
            fixed once from three tries (1024 / 1.0: recall@100 1.00, 3.1 k evaluations per query; 256 / 1.5: 0.98, 8.2 k;
            4096 / 2.0: 0.49, 16 k = structureless again), not tuned to the gate; BENCH_C3_CLUSTER overrides them"""
            g_ = torch.Generator(device=dev)
            g_.manual_seed(4321)                      # centres shared by base and query sets
            cen = torch.randn((n_centres, d_), generator=g_, device=dev)
            cen = cen / cen.norm(dim=1, keepdim=True)
            g_.manual_seed(seed_)
            out_ = np.empty((n_, d_), np.float32)
            for s_ in range(0, n_, 1 << 20):
                m_ = min(1 << 20, n_ - s_)
                idx_ = torch.randint(0, n_centres, (m_,), generator=g_, device=dev)
                x_ = cen[idx_] + spread * torch.randn((m_, d_), generator=g_, device=dev) / (d_ ** 0.5)
                out_[s_:s_ + m_] = (x_ / x_.norm(dim=1, keepdim=True)).cpu().numpy()
            return out_

        def exact_topk(Xh_, Qd_, k_, metric_):
            """exact ground truth (ids) of the first rows of Qd_ over the host table Xh_, in blocks on the GPU"""
            best_v = best_i = None
            for s_ in range(0, Xh_.shape[0], 1 << 20):
                xb = torch.from_numpy(Xh_[s_:s_ + (1 << 20)]).to(dev)
                sc = Qd_ @ xb.T if metric_ else -((Qd_ * Qd_).sum(1)[:, None] - 2.0 * (Qd_ @ xb.T) + (xb * xb).sum(1)[None, :])
                v_, i_ = torch.topk(sc, min(k_, xb.shape[0]), dim=1, largest=True)
                i_ = i_ + s_
                if best_v is None:
                    best_v, best_i = v_, i_
                else:
                    v2 = torch.cat([best_v, v_], 1); i2 = torch.cat([best_i, i_], 1)
                    o_ = torch.topk(v2, k_, dim=1, largest=True).indices
                    best_v, best_i = torch.gather(v2, 1, o_), torch.gather(i2, 1, o_)
                del xb, sc
            return best_i.cpu().numpy()

        def other_config(tag, n_, d_, metric_, M_, efc_, ef_, k_, seed_, n_sample, ceiling, kind="unit"):
            t0_ = time.time()
            gen_ = clustered_unit_vectors if kind == "clustered" else unit_vectors
            Xo = gen_(n_, d_, seed_)
            Qo = gen_(nq, d_, seed_ + 100)
            hgo = H.Ohnsw.build_batch_bigarray(Xo, M_, efc_, seed=1, metric=metric_, device=gpu)
            build_s_ = time.time() - t0_
            Qod = torch.from_numpy(Qo).to(dev)
            io = torch.empty((nq, k_), dtype=torch.int32, device=dev)
            do = torch.empty((nq, k_), dtype=torch.float32, device=dev)
            ndo = torch.zeros(nq, dtype=torch.int32, device=dev)
            nho = torch.zeros(nq, dtype=torch.int32, device=dev)
            sto = torch.zeros(nq, dtype=torch.int32, device=dev)

            def go(c_=False):
                H.search_batch_device(hgo, Qod.data_ptr(), nq, d_, ef_, k_, io.data_ptr(), do.data_ptr(),
                                      ndo.data_ptr() if c_ else 0, nho.data_ptr() if c_ else 0, sto.data_ptr(), stream.cuda_stream)
            go(True)
            torch.cuda.synchronize()
            gi, gd = io.cpu().numpy(), do.cpu().numpy()
            nd_, nh_ = float(ndo.float().mean().item()), float(nho.float().mean().item())
            steps_ = 5
            hgo.set_option("time_kernels", 1)
            hgo.kernel_times()
            ts_ = []
            for _ in range(steps_):
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record(stream); go(); b_.record(stream); torch.cuda.synchronize()
                ts_.append(a_.elapsed_time(b_))
            sm_, pm_, _ = hgo.kernel_times()
            hgo.set_option("time_kernels", 0)
            ts_.sort()
            med_ = ts_[len(ts_) // 2]
            # where the handle chose the bitmap blocks: the same index, batch and steps with the tag cache (round 4's structure), for the record
            tags_ = None
            if hgo.visited_blocks(ef_) and not os.environ.get("BENCH_NO_TAGS_AB"):      # (tools/profile_bench.sh sets it: one workload per kernel line)
                hgo.set_option("visited_blocks", 0)
                go(True)
                torch.cuda.synchronize()
                same_ = bool(np.array_equal(io.cpu().numpy(), gi) and np.array_equal(do.cpu().numpy().view(np.uint32), gd.view(np.uint32)))
                ndt_ = float(ndo.float().mean().item())
                tt_ = []
                for _ in range(3):
                    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a_.record(stream); go(); b_.record(stream); torch.cuda.synchronize()
                    tt_.append(a_.elapsed_time(b_))
                tt_.sort()
                tags_ = {"ms_per_step": round(tt_[1], 4), "value": round(nq / (tt_[1] * 1e-3), 1), "gpu_evaluations_per_query": round(ndt_, 1),
                         "gpu_evaluations_per_query_with_blocks": round(nd_, 1), "same_ids_and_distance_bits": same_,
                         "what": "option visited_blocks = 0 on the same index: the tag cache of rounds 1-4"}
                hgo.set_option("visited_blocks", -1)
                go(True)                                   # (the counters below are the default structure's)
                torch.cuda.synchronize()
            ck = {"tie_overflow_flagged": int((sto & 1).sum().item())}
            nrec_ = min(300, nq)
            ck["recall_at_k"] = round(recall_ids(gi[:nrec_], exact_topk(Xo, Qod[:nrec_], k_, metric_)), 4)      # id-set recall@k, exact ground truth
            ck["recall_queries"] = nrec_
            src_, nu_ = "gpu counters (include re-evaluations)", None
            if not args.no_cpu:
                from oracle import oracle as o
                hgo.export()
                sel = np.random.default_rng(0).choice(nq, n_sample, replace=False)
                spo = (o.Space.ip if metric_ else o.Space.l2)(Xo, arith=o.TREE16)
                go_ = o.Graph(hgo.n, hgo.entry_point, hgo.deg0, hgo.nbr0, hgo.upper)
                oi_, od_, ond_, onh_, onu_ = o.Ohnsw.knn_batch_bigarray(go_, spo, Qo[sel], k=k_, ef=ef_, ties=o.TIES_CANONICAL, split=True)
                ck["parity_queries"] = int(n_sample)
                ck["parity_ids_equal"] = bool(np.array_equal(oi_, gi[sel]))
                ck["parity_dist_bits_equal"] = bool(np.array_equal(od_.view(np.uint32), gd[sel].view(np.uint32)))
                ck["gpu_reevaluation_overhead"] = round(float(ndo.cpu().numpy()[sel].mean() / max(ond_.mean(), 1) - 1), 4)
                nd_, nh_, nu_ = float(ond_.mean()), float(onh_.mean()), float(onu_.mean())
                src_ = "oracle counters on %d queries of the batch" % n_sample
                del spo, go_
            So = 2 * M_
            rbo = hgo.row_bytes()
            fmt_ = int(hgo.info().row_format)
            blk_ = hgo.visited_blocks(ef_)
            kern = search_kernel_name(d_, ef_, metric_, 0, fmt_ if fmt_ else -1, blk_)
            ordered_ = pm_ > 0
            l0 = ordered_ and nu_ is not None
            bq_ = (nd_ - (nu_ if l0 else 0.0)) * (rbo + 4) + nh_ * 4 * So + 4 * d_ + 8 * k_ + (16 if l0 else 0)
            kms_ = sm_ if (l0 or not ordered_) else sm_ + pm_
            ach_ = bq_ * nq / (kms_ * 1e-3) / 1e9
            res_ = {"workload": "%s: n=%d d=%d %s, %s (synthetic), M=%d efConstruction=%d (built on this GPU in %.0f s incl. data), "
                                "ef=%d k=%d, %d queries resident in HBM" % (tag, n_, d_, "inner product" if metric_ else "L2",
                                                                            ("unit vectors clustered around %d random directions (spread %.2f)" % (int(_cc[0]), _cc[1])) if kind == "clustered"
                                                                            else "N(0,1) unit vectors as BASELINE.md prescribes: structureless, recall is inherently low",
                                                                            M_, efc_, build_s_, ef_, k_, nq),
                    "value": round(nq / (med_ * 1e-3), 1), "unit": "queries/s", "ms_per_step": round(med_, 4),
                    "ms_min": round(ts_[0], 4), "ms_max": round(ts_[-1], 4), "steps": steps_, "statistic": "median",
                    "roofline": {"bound": "hbm", "achieved": round(ach_, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(ach_ / HBM_PEAK_GBS, 4), "kernel": kern, "kernel_ms": round(kms_, 4),
                                 "row_format": {0: "float32 rows", 2: "byte rows", 3: "split rows: whole 128-byte lines per row + the last 16 / 32 bytes beside the "
                                                "neighbour in the layer-0 adjacency (hnsw_rows_split.hip)"}.get(fmt_),
                                 "measured_gather_ceiling": dict(ceiling, frac_of_it=round(ach_ / (1e3 * ceiling["TBps"]), 4)),
                                 "prepass_ms": round(pm_, 4), "bytes_per_query": round(bq_, 1), "row_bytes": rbo,
                                 "n_dist_per_query": round(nd_, 1), "n_hops_per_query": round(nh_, 1), "counters": src_,
                                 "visited": ("bitmap blocks over locality codes, 2^%d slots of 256 codes (the handle's own measurement: option visited_blocks -1)" % blk_) if blk_
                                            else "tag cache (the handle's own measurement where the shape has a choice: option visited_blocks -1)",
                                 "index_bytes": int(hgo.info().device_bytes) if hasattr(hgo, "info") else None},
                    "checks": ck}
            if tags_:
                res_["with_the_tag_cache"] = tags_
            log("%s: %.0f q/s, %.3f ms/step (kernel %.3f ms), %.0f evaluations/query, frac %.3f, parity %s (%.0fs)" %
                (tag, nq / (med_ * 1e-3), med_, kms_, nd_, ach_ / HBM_PEAK_GBS, ck.get("parity_ids_equal"), time.time() - t0_))
            hgo.release()
            del Xo, Qo, Qod
            return res_

        try:
            others["C3"] = other_config("C3 GloVe-1.2M shape", 1_183_514, 100, 1, 32, 200, 256, 100, 2, 200,
                                        {"TBps": 6.84, "what": "random whole 384-byte rows (the split layout's main rows) from a 1.07 GB table, independent "
                                                                "requests, 4-8 waves/SIMD: tools/gather_ceiling.hip, profiles/r04_gather_ceiling.txt (6.22 from 4.3 GB; "
                                                                "this index is 2.5 GB).  A search may exceed it: its walks re-read popular rows from L2 / "
                                                                "Infinity Cache, the ceiling's requests are uniform"})
        except Exception as e:   # never lose the headline line to a secondary leg
            others["C3"] = {"skipped": "failed: %r" % (e,)}
        try:     # the same shape on data with structure: what "q/s at recall >= 0.95" means for the inner-product path
            if args.no_clustered:
                raise RuntimeError("--no-clustered")
            others["C3_clustered"] = other_config("C3 GloVe-1.2M shape, clustered", 1_183_514, 100, 1, 32, 200, 256, 100, 12, 200,
                                                  {"TBps": 6.84, "what": "as C3"}, kind="clustered")
        except Exception as e:
            others["C3_clustered"] = {"skipped": "failed: %r" % (e,)}
        if free_gb is not None and free_gb < 24:
            others["C5"] = {"skipped": "needs about 12 GB of host memory for the vectors and the exported graph; %.1f GB free" % free_gb}
        else:
            try:
                others["C5"] = other_config("C5 DEEP10M shape", 10_000_000, 96, 0, 32, 200, 512, 10, 3, 100,
                                            {"TBps": 6.44, "what": "random whole 384-byte rows from a 6.4 GB table, independent requests, the same at 4, 5 and 8 "
                                                                    "waves/SIMD and 1-8 batches in flight: tools/gather_ceiling.hip, profiles/r04_gather_ceiling.txt"})
            except Exception as e:
                others["C5"] = {"skipped": "failed: %r" % (e,)}
            try:     # ... and C5's shape on clustered vectors (recall at ef 512 then means something for the L2 path at 10 M too)
                if args.no_clustered:
                    raise RuntimeError("--no-clustered")
                others["C5_clustered"] = other_config("C5 DEEP10M shape, clustered", 10_000_000, 96, 0, 32, 200, 512, 10, 13, 100,
                                                      {"TBps": 6.44, "what": "as C5"}, kind="clustered")
            except Exception as e:
                others["C5_clustered"] = {"skipped": "failed: %r" % (e,)}

    # ---- C1 (BASELINE.json configs[0]: the reference's own CPU-runnable case -- 10 k random fp32 vectors, d = 32, M 8,
    #      efConstruction 100, ef 32, k 10, 1000 queries): the CPU restatement does the whole of it, build included (the
    #      oracle's restatement of Ohnsw.insert), as benchmark/benchmark.ml does; the GPU searches the SAME graph and must
    #      return the same bits.  A parity / context point, never `value`. ----
    c1 = None
    if world == 1 and rank == 0 and not args.no_cpu and not args.dataset and not args.no_others:
        from oracle import oracle as o
        t0 = time.time()
        rng1 = np.random.default_rng(0)
        X1 = rng1.uniform(-1.0, 1.0, size=(10_000, 32)).astype(np.float32)       # Lacaml Mat.random's default range, benchmark/dataset.ml:48
        Q1 = rng1.uniform(-1.0, 1.0, size=(1_000, 32)).astype(np.float32)
        sp1 = o.Space.l2(X1, arith=o.SEQ_F32)
        t = time.perf_counter()
        g1 = o.build_ohnsw(sp1, 8, 100, seed=0, ties=o.TIES_CANONICAL)
        build1 = time.perf_counter() - t
        t = time.perf_counter()
        ci1, cd1 = o.Ohnsw.knn_batch_bigarray(g1, sp1, Q1, k=10, ef=32, ties=o.TIES_CANONICAL)
        cpu1 = time.perf_counter() - t
        hg1 = H.Hgraph(X1, g1.deg0, g1.nbr0, g1.upper, entry_point=g1.entry_point, id_base=0, max_degree=8).to_device(gpu)
        gi1, gd1 = H.Ohnsw.knn_batch_bigarray(hg1, 10, Q1, ef=32)
        ts1 = []
        for _ in range(7):
            t = time.perf_counter()
            H.Ohnsw.knn_batch_bigarray(hg1, 10, Q1, ef=32)
            ts1.append(time.perf_counter() - t)
        ts1.sort()
        sp1k = o.Space.l2(X1, arith=o.TREE16)                                     # the kernel's summation order: bit parity
        ki1, kd1 = o.Ohnsw.knn_batch_bigarray(g1, sp1k, Q1, k=10, ef=32, ties=o.TIES_CANONICAL)
        gt1 = brute_force_topk(torch.from_numpy(X1).to(dev), torch.from_numpy(Q1).to(dev), 10)
        c1 = {"workload": "C1: 10 000 x 32 uniform fp32, M=8 efConstruction=100 (graph built by the CPU restatement of Ohnsw.insert), ef=32 k=10, 1000 queries",
              "cpu_restatement": {"value": round(1000 / cpu1, 1), "unit": "queries/s", "cores": 1, "build_s": round(build1, 2),
                                  "what": "single-thread C restatement of Ohnsw.build_batch_bigarray + knn_batch_bigarray, the reference's arithmetic"},
              "gpu_same_graph": {"value": round(1000 / ts1[len(ts1) // 2], 1), "unit": "queries/s", "ms_per_batch": round(1e3 * ts1[len(ts1) // 2], 4),
                                 "what": "hnsw_search_batch on the same graph, pageable host matrices (1000 queries: a latency point, 1/8 of the chip's wave slots)"},
              "recall_at_10": round(recall_ids(gi1, gt1), 4),
              "checks": {"gpu_ids_equal_cpu_reference_arithmetic": bool(np.array_equal(gi1, ci1)),
                         "gpu_bits_equal_oracle_kernel_order": bool(np.array_equal(gi1, ki1) and np.array_equal(gd1.view(np.uint32), kd1.view(np.uint32))),
                         "max_rel_distance_error_vs_reference_arithmetic": float(np.nanmax(np.abs(gd1 - cd1) / np.maximum(np.abs(cd1), 1e-30)))}}
        hg1.release()
        log("C1: cpu restatement %.0f q/s (build %.1f s), gpu on the same graph %.0f q/s, recall@10 %.3f, ids equal %s (%.1fs)" %
            (c1["cpu_restatement"]["value"], build1, c1["gpu_same_graph"]["value"], c1["recall_at_10"],
             c1["checks"]["gpu_ids_equal_cpu_reference_arithmetic"], time.time() - t0))
        if others is not None:
            others["C1"] = c1

    # ---- bench_dist counterpart (bench_dist/bench_dist.ml:8-33: 1 M calls of distance_l2 at d = 784, checksum, s/call,
    #      calls/s): 1 M gathered distances over random rows of a 1 M x 784 table, one launch ----
    bench_dist = None
    if world == 1 and rank == 0 and not args.no_bench_dist:
        t0 = time.time()
        dd, nn, nqd, md = 784, 1_000_000, 1024, 1024
        gd_ = torch.Generator(device=dev)
        gd_.manual_seed(dd)
        Xdist = torch.rand((nn, dd), generator=gd_, device=dev).cpu().numpy()
        hgd = H.Hgraph(Xdist, np.zeros(nn, np.int32), np.full((nn, 2), -1, np.int32), entry_point=0).to_device(gpu)
        Qdist = torch.rand((nqd, dd), generator=gd_, device=dev)
        idd = torch.randint(0, nn, (nqd, md), generator=gd_, device=dev, dtype=torch.int32)
        outd = torch.empty((nqd, md), dtype=torch.float32, device=dev)
        Ld = H.load()

        def go_d():
            rc_ = Ld.hnsw_distance_batch_device(hgd.handle, Qdist.data_ptr(), nqd, dd, idd.data_ptr(), md, outd.data_ptr(), stream.cuda_stream)
            assert rc_ == 0, Ld.hnsw_last_error()
        go_d()
        torch.cuda.synchronize()
        tsd = []
        for _ in range(7):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record(stream); go_d(); b_.record(stream); torch.cuda.synchronize()
            tsd.append(a_.elapsed_time(b_))
        tsd.sort()
        msd = tsd[len(tsd) // 2]
        pairs = nqd * md
        # spot check against fp64 on the host: the north star's 1e-5 relative
        jq, jm = np.arange(0, nqd, 97), np.arange(0, md, 89)
        got_d = outd.cpu().numpy()[np.ix_(jq, jm)]
        ids_h = idd.cpu().numpy()[np.ix_(jq, jm)]
        qh = Qdist.cpu().numpy()[jq].astype(np.float64)
        want_d = np.sqrt(((Xdist[ids_h].astype(np.float64) - qh[:, None, :]) ** 2).sum(-1))
        bench_dist = {"d": dd, "n": nn, "pairs": pairs, "ms": round(msd, 4), "ms_min": round(tsd[0], 4), "ms_max": round(tsd[-1], 4),
                      "s_per_call": msd * 1e-3 / pairs, "calls_per_s": round(pairs / (msd * 1e-3), 1),
                      "gathered_TBps": round(pairs * 4 * dd / (msd * 1e-3) / 1e12, 3), "frac_of_hbm_peak": round(pairs * 4 * dd / (msd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "checksum": float(outd.double().sum()), "max_rel_err_vs_fp64_sample": float(np.max(np.abs(got_d - want_d) / want_d)),
                      "what": "hnsw_distance_batch_device: 1024 queries x 1024 random ids = 1 M (query, row) pairs per launch, 3136-byte rows "
                              "(bench_dist/bench_dist.ml:22-33 times 1 M single calls of Ohnsw.distance_l2 at this d)"}
        log("bench_dist d=%d: %.3g calls/s, %.2f TB/s gathered, max rel err %.2g (%.0fs)" %
            (dd, bench_dist["calls_per_s"], bench_dist["gathered_TBps"], bench_dist["max_rel_err_vs_fp64_sample"], time.time() - t0))
        hgd.release()
        del Xdist, Qdist, idd, outd

    # ---- live counter passes: rocprofv3 --pmc around a child of this script that loads the SAME index (saved to a
    #      temporary file) and runs the SAME batch; one pass per counter group (MI355X_MICROARCH.md: separate --pmc
    #      passes; FETCH_SIZE x 2 on gfx950).  Gives roofline.traffic and roofline.issue of THIS run. ----
    pmc = {}
    if world == 1 and rank == 0 and not args.no_pmc and not args.dataset:
        import csv
        import glob
        import shutil
        import subprocess
        import tempfile
        t0 = time.time()
        rp = shutil.which("rocprofv3")
        under_profiler = any(k_.startswith(("ROCPROF", "ROCP_")) for k_ in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        if rp is None:
            pmc = {"skipped": "rocprofv3 not on PATH"}
        elif under_profiler:
            pmc = {"skipped": "this process is itself being profiled: no nested rocprofv3 passes"}
        else:
            tmpd = tempfile.mkdtemp(prefix="hnsw_pmc_", dir="/tmp")
            try:
                idx_file = os.path.join(tmpd, "c2.idx")
                hg.save(idx_file)
                groups = {"inst": ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"],
                          "fetch": ["FETCH_SIZE"]}
                env_ = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT)
                phases = [("head", idx_file, ef)] + ([("hard", hard_idx[0], hard_idx[1])] if hard_idx else [])
                for phase, file_, ef_p in phases:
                    for gname, counters in groups.items():
                        outd = os.path.join(tmpd, phase + "_" + gname)
                        cmd = [rp, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", outd, "--", sys.executable,
                               os.path.abspath(__file__), "--pmc-child", file_, "--pmc-phase", phase, "--nq", str(nq), "--d", str(d),
                               "--ef", str(ef_p), "--k", str(k)]
                        r_ = subprocess.run(cmd, cwd="/tmp", env=env_, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
                        if r_.returncode != 0:
                            pmc.setdefault("failed", {})[phase + "_" + gname] = r_.stderr.decode(errors="replace")[-300:]
                            continue
                        rows_ = []
                        for f_ in glob.glob(os.path.join(outd, "**", "*counter_collection.csv"), recursive=True):
                            rows_ += list(csv.DictReader(open(f_)))
                        # per kernel instance AND launch size: only the nq-query launches (warm_up's one-query dispatches and the
                        # handle's 256-query visited-structure measurement run the same instances)
                        for (kn, _nq), cs in counter_means(rows_, nq).items():
                            pmc.setdefault(phase, {}).setdefault(kn, {}).update(cs)
            except Exception as e:
                pmc = {"skipped": "live counter pass failed: %r" % (e,)}
            finally:
                shutil.rmtree(tmpd, ignore_errors=True)
                if hard_idx:
                    shutil.rmtree(os.path.dirname(hard_idx[0]), ignore_errors=True)
        log("live rocprofv3 counter passes: %s (%.0fs)" % ({k_: (sorted(v_) if isinstance(v_, dict) else v_) for k_, v_ in pmc.items()}, time.time() - t0))

    def live_counters(phase, kernel, kernel_ms_):
        """(traffic bytes per launch, issue dict) of one search-kernel instance from this run's counter passes"""
        c_ = (pmc.get(phase) or {}).get((kernel or "").replace(" ", "")) if isinstance(pmc.get(phase), dict) else None
        if not c_:
            return None, None
        traffic_ = int(2 * c_["FETCH_SIZE"] * 1024) if "FETCH_SIZE" in c_ else None     # gfx950: FETCH_SIZE (KB) counts 128-B requests at 64 B
        issue_ = None
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count          # 256 on MI355X
        n_simd, n_se = 4 * n_cu, max(1, n_cu // 8)                                   # four SIMDs per CU; shader engines of 8 CUs (32)
        if "SQ_INSTS_VALU" in c_ and c_.get("SQ_BUSY_CYCLES"):
            cyc = c_["SQ_BUSY_CYCLES"] / float(n_se)     # summed over the chip's shader engines
            tot = c_["SQ_INSTS_VALU"] + c_["SQ_INSTS_SALU"] + c_.get("SQ_INSTS_LDS", 0) + c_.get("SQ_INSTS_VMEM_RD", 0)
            # capacities measured with tools/issue_latency.hip at 8 waves per SIMD on every CU (profiles/r03_issue_latency.txt):
            # a SIMD issues one simple vector wave-instruction per ~2.4 cycles (v_dot4 / v_mad_u64 per ~4.3), a CU one scalar
            # instruction per cycle for its four SIMDs; a wave alone issues at most one instruction of any kind per 4 cycles
            issue_ = {"valu": round(c_["SQ_INSTS_VALU"] * 2.4 / (n_simd * cyc), 4), "salu": round(c_["SQ_INSTS_SALU"] / (n_cu * cyc), 4),
                      "what": "share of the issue capacity the launch had, from measured capacities (tools/issue_latency.hip): vector = "
                              "wave-instructions x 2.4 cycles / (1024 SIMDs x kernel cycles) (a lower bound: dot products and 64-bit "
                              "multiply-adds take 4.3); scalar = instructions / (256 CUs x kernel cycles) (one scalar issue per cycle "
                              "per CU, shared by its four SIMDs)",
                      "SQ_INSTS_VALU": c_["SQ_INSTS_VALU"], "SQ_INSTS_SALU": c_["SQ_INSTS_SALU"], "SQ_INSTS_LDS": c_.get("SQ_INSTS_LDS"),
                      "SQ_INSTS_VMEM_RD": c_.get("SQ_INSTS_VMEM_RD"), "kernel_cycles": round(cyc, 1),
                      "clock_GHz": round(cyc / (kernel_ms_ * 1e6), 3) if kernel_ms_ else None,
                      "instructions_per_hop": None, "wave_occupancy": round(c_["SQ_WAVE_CYCLES"] * 4.0 / (8 * n_simd * cyc), 4) if c_.get("SQ_WAVE_CYCLES") else None,
                      "instructions_per_dispatch": tot,
                      "source": "rocprofv3 --pmc pass of this run (a child process on the same index and batch, 5 dispatches)"}
        return traffic_, issue_

    if secondary and isinstance(secondary.get("at_recall_0.95"), dict) and "roofline" in secondary["at_recall_0.95"]:
        g_ = secondary["at_recall_0.95"]
        for tgt in (g_["roofline"], g_.get("float32_rows")):
            if tgt is not None and tgt.get("traffic") is None:
                tr_, is_ = live_counters("hard", tgt.get("kernel"), tgt["kernel_ms"])
                tgt["traffic"] = tr_
                tgt["traffic_source"] = "live rocprofv3 FETCH_SIZE pass of this run x 2 (gfx950)" if tr_ is not None else None
                if is_ is not None and tgt is g_["roofline"]:
                    is_["instructions_per_hop"] = round(is_["instructions_per_dispatch"] / (nq * tgt["n_hops_per_query"]), 1)
                    tgt["issue"] = is_

    # ---- algorithmic bytes from the CPU oracle's counters on the same graph/queries,
    #      parity spot-check, and the CPU baseline (rank 0) ----
    roofline, cpu_baseline = None, None
    if rank == 0:
        S = 2 * args.M
        n_dist_mean, n_hops_mean = float(gpu_nd.mean()), float(gpu_nh.mean())
        n_upper_mean = None
        src = "gpu counters (include re-evaluations)"
        if not args.no_cpu:
            from oracle import oracle as o
            sample = min(args.cpu_sample if world == 1 else 500, nq)
            Qs = Qd[:sample].cpu().numpy()
            sp = o.Space.l2(X, arith=o.TREE16)
            g = o.Graph(hg.n, hg.entry_point, hg.deg0, hg.nbr0, hg.upper)
            t = time.perf_counter()
            oids, odist, ond, onh, onu = o.Ohnsw.knn_batch_bigarray(g, sp, Qs, k=k, ef=ef, ties=o.TIES_CANONICAL, split=True)
            cpu_s = time.perf_counter() - t
            n_dist_mean, n_hops_mean = float(ond.mean()), float(onh.mean())
            n_upper_mean = float(onu.mean())
            src = "oracle counters on %d queries" % sample
            checks["parity_queries"] = sample
            checks["parity_ids_equal"] = bool(np.array_equal(oids, got[:sample]))
            checks["parity_dist_bits_equal"] = bool(np.array_equal(odist.view(np.uint32), got_dist[:sample].view(np.uint32)))
            checks["gpu_reevaluation_overhead"] = round(float(gpu_nd[:sample].mean() / max(ond.mean(), 1) - 1), 4)
            if functor_result is not None:      # the functor module's rule on the same sample (ties in the (d, id) order)
                fs = min(sample, 500)
                ofd, ofi = o.Functor.knn_batch(g, sp, Qs[:fs], ef, k, ties=o.TIES_CANONICAL, with_ids=True)
                checks["functor_parity_ids_equal"] = bool(np.array_equal(ofi, functor_result[0][:fs]))
                checks["functor_parity_dist_bits_equal"] = bool(np.array_equal(ofd.view(np.uint32), functor_result[1][:fs].view(np.uint32)))
            if world == 1:
                # the baseline is timed with the REFERENCE's arithmetic (Lacaml-style sequential fp32 sum, sqrt in
                # double: oracle SEQ_F32), not with the kernel's summation order used for the parity leg above
                sp_ref = o.Space.l2(X, arith=o.SEQ_F32)
                Qall_h = Qd.cpu().numpy()
                ref_ts = []
                for _ in range(3):          # three passes, the median (one pass moved 3.7 -> 3.0 k q/s between two boxes in round 4)
                    t = time.perf_counter()
                    rids, rdist = o.Ohnsw.knn_batch_bigarray(g, sp_ref, Qall_h, k=k, ef=ef, ties=o.TIES_CANONICAL)
                    ref_ts.append(time.perf_counter() - t)
                ref_ts.sort()
                ref_s = ref_ts[1]
                # integer-valued data: every summation order is exact, so this leg must reproduce the GPU bit for bit too
                checks["cpu_reference_arithmetic_ids_equal"] = bool(np.array_equal(rids, got))
                ncores = host_cores()
                t = time.perf_counter()
                mids, _ = o.knn_batch_all_cores(g, sp_ref, Qall_h, k, ef, ncores)
                mt_s = time.perf_counter() - t
                checks["cpu_all_cores_ids_equal"] = bool(np.array_equal(mids, got))
                cpu_baseline = {"value": round(nq / ref_s, 1), "unit": "queries/s", "cores": 1, "kind": "port",
                                "passes": 3, "statistic": "median", "value_min": round(nq / ref_ts[2], 1), "value_max": round(nq / ref_ts[0], 1),
                                "sample": "all %d queries of one batch, median of 3 passes (%.1f s), 1 thread, C restatement (not OCaml)" % (nq, ref_s),
                                "sample_note": "C restatement of Ohnsw.knn_batch_bigarray with the reference's arithmetic (sequential fp32 sum, sqrt in "
                                               "double), same graph, ef=%d k=%d; the reference is single-threaded" % (ef, k),
                                "all_cores": {"value": round(nq / mt_s, 1), "cores": ncores,
                                              "sample": "all %d queries split over %d host threads" % (nq, ncores)}}
                log("cpu restatement (reference arithmetic): %.1f q/s single-thread, %.1f q/s on %d threads" % (nq / ref_s, nq / mt_s, ncores))
            log("parity (kernel summation order) on %d queries: ids=%s dist=%s" %
                (sample, checks["parity_ids_equal"], checks["parity_dist_bits_equal"]))
        # B_q = n_dist*(4d+4) + n_hops*4S + 4d + 8k   
        # B_q = n_dist*(4d+4) + n_hops*4S + 4d + 8k   for the whole query.  When the batch was
        # ordered longest-first the descent ran in its own kernel: the search kernel then does the layer-0
        # part of it (the evaluations after the descent, the hops' adjacency rows, the query, the results,
        # 16 B of hand-over per query), and its own duration is what the library's HIP events measured.
        # A row is 4d bytes as float32 and d bytes as a byte row (what this launch read: `row_bytes`).
        def bq_of(rb, layer0_only):
            nd_ = n_dist_mean - (n_upper_mean if layer0_only else 0.0)
            return nd_ * (rb + 4) + n_hops_mean * 4 * S + 4 * d + 8 * k + (16 if layer0_only else 0)

        def traffic_of(name, ordered_):
            tp = os.path.join(ROOT, "profiles", "traffic.json")
            try:
                tj = json.load(open(tp))
                if tj.get("workload") == "C2" and tj.get("nq") == nq and tj.get("ef") == ef and bool(tj.get("ordered")) == ordered_:
                    return tj.get("kernels", {}).get(name, {}).get("hbm_bytes_per_launch")
            except Exception:
                pass
            return None

        bq_total = bq_of(row_bytes, False)
        ordered = prepass_ms > 0
        if ordered and n_upper_mean is not None:
            bq, kernel_ms, kname = bq_of(row_bytes, True), search_ms, kernel_name(byte_rows)
        elif ordered:      # no oracle counters: the pre-pass and the search kernel together
            bq, kernel_ms, kname = bq_total, search_ms + prepass_ms, "hnsw_descent_kernel + radix sort + " + kernel_name(byte_rows)
        else:
            bq, kernel_ms, kname = bq_total, search_ms, kernel_name(byte_rows)
        achieved = bq * nq / (kernel_ms * 1e-3) / 1e9
        traffic, issue = live_counters("head", kernel_name(byte_rows), kernel_ms)
        traffic_src = "live rocprofv3 FETCH_SIZE pass of this run x 2 (gfx950)" if traffic is not None else None
        if traffic is None:
            traffic = traffic_of(kname, ordered)
            traffic_src = "profiles/traffic.json (static)" if traffic is not None else None
        if issue is not None and n_hops_mean:
            issue["instructions_per_hop"] = round(issue["instructions_per_dispatch"] / (nq * n_hops_mean), 1)
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src, "issue": issue,
                    "kernel": kname, "kernel_ms": round(kernel_ms, 4), "row_bytes": row_bytes,
                    "bytes_per_query": round(bq, 1), "n_dist_per_query": round(n_dist_mean, 1),
                    "n_hops_per_query": round(n_hops_mean, 1), "counters": src,
                    "kernel_ms_source": "HIP events the library records on its launch stream around its own launches (option time_kernels), mean "
                                        "over the %d timed steps of `value` themselves (the same steps without the event records: %.4f ms per step)"
                                        % (args.steps, 1e3 * wall_plain / args.steps),
                    "step": {"host_call_ms": round(1e3 * wall / args.steps, 4), "host_call_ms_stats": host_stats,
                             "host_call_ms_without_kernel_events": round(1e3 * wall_plain / args.steps, 4),
                             "device_call_ms": round(kern_ms, 4), "device_call_ms_stats": headline_steps, "prepass_ms": round(prepass_ms, 4),
                             "kernel_ms_in_device_resident_steps": round(dev_lib["search_ms"], 4),
                             "prepass": "hnsw_descent_kernel + radix sort (longest-first ordering)" if ordered else None,
                             "bytes_per_query_whole_path": round(bq_total, 1),
                             "n_dist_before_layer0_per_query": None if n_upper_mean is None else round(n_upper_mean, 1),
                             "achieved_whole_path": round(bq_total * nq / (kern_ms * 1e-3) / 1e9, 1)}}
        if latency_floor is not None:
            # what a reader needs beside `frac` when the launch is bound by its longest walk, not by HBM
            latency_floor["kernel_ms"] = round(kernel_ms, 4)
            latency_floor["kernel_over_floor"] = round(kernel_ms / latency_floor["floor_ms"], 3) if latency_floor["floor_ms"] else None
            latency_floor["wave_occupancy"] = (issue or {}).get("wave_occupancy")
            if big:      # what the chip sustains per 10000 queries when every wave slot stays busy (the 100 k launch, whole step)
                latency_floor["full_load_ms_per_batch"] = round(big["ms_per_step"] * nq / big["queries_per_gpu"], 4)
            roofline["latency_floor"] = latency_floor
        if byte_rows:
            roofline["note"] = ("byte rows: every value of this data set is an integer in 0..255, so the knn kernel gathers d-byte rows "
                                "(a quarter of the float32 bytes, same arithmetic, bit-identical results); the launch is then bound by "
                                "the latency of a hop, not by HBM -- the float32-row kernel on the same batch is in `float32_rows`")
        if fp32_leg is not None:
            f_ord = fp32_leg["prepass_ms"] > 0
            f_l0 = f_ord and n_upper_mean is not None
            f_bq = bq_of(4 * d, f_l0)
            f_ms = fp32_leg["search_ms"] if (f_l0 or not f_ord) else fp32_leg["search_ms"] + fp32_leg["prepass_ms"]
            f_ach = f_bq * nq / (f_ms * 1e-3) / 1e9
            roofline["float32_rows"] = {"value": round(nq * args.steps / fp32_leg["wall"], 1), "unit": "queries/s",
                                        "ms_per_step": round(1e3 * fp32_leg["wall"] / args.steps, 4),
                                        "kernel": kernel_name(False), "kernel_ms": round(f_ms, 4), "prepass_ms": round(fp32_leg["prepass_ms"], 4),
                                        "bytes_per_query": round(f_bq, 1), "achieved": round(f_ach, 1), "peak": HBM_PEAK_GBS, "unit_bw": "GB/s",
                                        "frac": round(f_ach / HBM_PEAK_GBS, 4),
                                        "traffic": live_counters("head", kernel_name(False), f_ms)[0] if live_counters("head", kernel_name(False), f_ms)[0] is not None else traffic_of(kernel_name(False), f_ord),
                                        "issue": live_counters("head", kernel_name(False), f_ms)[1],
                                        "what": "option byte_rows = 0: the same index, batch and steps through the float32 rows"}

    if rank == 0:
        gate_ok = checks.get("recall_at_10", 0) >= 0.95
        harder_gate = None
        if secondary:
            harder_gate = ({"ef": ef, "recall_at_10": secondary["checks"]["recall_at_10"], "value": secondary.get("value_host_protocol"),
                            "device_resident_value": secondary["value"], "unit": "queries/s"}
                           if secondary["checks"]["recall_at_10"] >= 0.95 else secondary.get("at_recall_0.95"))
            if harder_gate is secondary.get("at_recall_0.95"):
                secondary["at_recall_0.95"] = "see harder_set_at_recall_gate"      # embedded once
        fl = roofline.get("float32_rows") if roofline else None
        # which synthetic set is the closer stand-in for SIFT1M (about 2.5-3 k evaluations per query at ef 128)
        nd_head = roofline["n_dist_per_query"] if roofline else None
        nd_hard = secondary["roofline"]["n_dist_per_query"] if secondary else None
        out = {
            "metric": "queries/sec at recall@10>=0.95, SIFT1M d=128 ef=128 k=10",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * wall / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # the arithmetic of the timed kernel, not a precision claim: byte rows are gathered as uint8 and summed with
            # v_dot4_u32_u8 (exact: every partial sum < 2^24, the float32 result bit for bit); float data computes in f32
            "dtype": "u8" if byte_rows else "f32",
            "dtype_note": "u8 rows, u32 dot products (exact) -> f32 distances" if byte_rows else "f32",
            "data": ("file:" + os.path.basename(os.path.normpath(args.dataset))) if args.dataset else "synthetic",
            "config": {"workload": "C2: %s (n=%d d=%d), M=%d efC=%d (built on the GPU), ef=%d k=%d, %d queries/GPU/step, "
                                   "replicated index%s; easier than SIFT1M (%s evals/query vs ~2.5-3k): see harder_set_at_recall_gate (%s)"
                                   % ("vectors from " + args.dataset if args.dataset else "SIFT1M-shaped synthetic, clustered ints 0..218 as fp32",
                                      n, d, args.M, args.efc, ef, k, nq,
                                      ", RCCL all-gather of results" if world > 1 else "", nd_head, nd_hard),
                       "n": n, "d": d, "M": args.M, "ef_construction": args.efc, "ef": ef, "k": k,
                       "queries_per_gpu": nq, "global_batch": world * nq, "parallelism": "replica x%d" % world,
                       "batches_rotated": NB,
                       "rows": "bytes (lossless copy of integer-valued float32 data)" if byte_rows else "float32",
                       "headline_set": {"n_dist_per_query": nd_head, "recall_at_10": checks.get("recall_at_10"), "ef": ef},
                       "harder_set": (None if not secondary else
                                      {"n_dist_per_query": nd_hard, "recall_at_10": secondary["checks"]["recall_at_10"], "ef": ef,
                                       "ef_at_recall_gate": (harder_gate or {}).get("ef"),
                                       "closer_to_SIFT1M": True})},
            "cold_first_call_ms": None if not cold else cold["first_call_ms"],
            "cold": cold,
            "protocol": ("one synchronous call per %d-query batch with HOST matrices in and out, as benchmark/benchmark.ml:89-96 times knn_batch "
                         "and SURVEY 8d prescribes; the warm-up and the timed steps rotate through %d distinct batches (seeds 2..%d): " % (nq, NB, 1 + NB)) +
                        ("hnsw_search_batch (H2D of the queries, ordering pre-pass, search kernel, D2H of the results; the caller's matrices in "
                         "page-locked memory: hnsw_host_alloc / hnsw_host_register)" if world == 1 else
                         "per rank hnsw_search_batch_h2d (its registered query shard read by the device, ordering pre-pass, search kernel, results in "
                         "HBM), RCCL all-gather of the per-shard results (the full table resident on every GPU), D2H of the rank's own shard, stream "
                         "synchronisation") +
                        "; the rate with the queries already resident in HBM and the results left there is `device_resident`",
            "device_resident": {"value": round(qps_dev, 1), "unit": "queries/s", "ms_per_step": round(1e3 * wall_dev / args.steps, 4),
                                "ms_per_step_with_events": round(1e3 * wall_dev_i / args.steps, 4), "ms_per_step_stats_with_events": headline_steps,
                                "kernel_ms": round(dev_lib["search_ms"], 4), "prepass_ms": round(dev_lib["prepass_ms"], 4),
                                "what": "hnsw_search_batch_device: the same %d steps with the queries resident in HBM before the timed region and the "
                                        "results left in HBM%s; `value` from a pass without event records, the kernel durations from an instrumented "
                                        "pass of the same steps (five event records per step cost the stream about 0.02 ms)"
                                        % (args.steps, " (search of step i+1 overlapped with the all-gather of step i)" if world > 1 else "")},
            "bandwidth_point": big,
            "float32_rows": (None if not fl else
                             {"value": round(nq * args.steps / fp32_leg["host_wall"], 1), "unit": "queries/s",
                              "ms_per_step": round(1e3 * fp32_leg["host_wall"] / args.steps, 4), "ms_per_step_stats": fp32_leg["host_stats"],
                              "device_resident_value": fl["value"], "frac": fl["frac"], "kernel": fl["kernel"], "kernel_ms": fl["kernel_ms"],
                              "what": "option byte_rows = 0: the same index, batch and protocol through the float32 rows (the general-format kernel, "
                                      "what data that is not byte-valued takes); details in roofline.float32_rows"}),
            "harder_set_at_recall_gate": harder_gate,
            "functor_api": (None if not drop_in else
                            {"value": drop_in["functor_api"]["value"], "unit": "queries/s", "ms_per_step": drop_in["functor_api"]["ms_per_batch"],
                             "what": "Hnsw.Ba.knn_batch (lib/hnsw.ml:763-777), the functor module's entry point, through the headline's protocol: "
                                     "accept rule of Hnsw_algo.Search, same index, batch, ef and k; median of %d blocking calls" % drop_in["batches_timed"]}),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "drop_in": drop_in, "secondary": secondary,
            "recall_gate": {"threshold": 0.95, "metric": "id-set recall@10 against exact brute force",
                            "headline_set": {"ef": ef if gate_ok else checks.get("ef_for_recall_0.95"),
                                             "recall_at_10": checks.get("recall_at_10") if gate_ok else checks.get("recall_at_that_ef"),
                                             "value": round(qps, 1) if gate_ok else checks.get("qps_at_that_ef")},
                            "harder_set": _pick(harder_gate, "ef", "recall_at_10", "value")} if rank == 0 and world == 1 else None,
            "others": others, "bench_dist": bench_dist, "strong": strong, "one_process": one_process, "pipelined": pipelined, "checks": checks,
        }
        if big and roofline:
            # the 100 k launch against the same line: whole-path bytes per query x its rate (no kernel-only duration is taken there)
            big["frac"] = round(roofline["step"]["bytes_per_query_whole_path"] * big["value"] / world / 1e9 / HBM_PEAK_GBS, 4)
        # Everything measured goes to DETAIL_FILE beside this script (and to stderr); stdout carries the short driver line only
        detail_txt = json.dumps(_finite(out), allow_nan=False)
        for path_ in (os.path.join(ROOT, DETAIL_FILE), os.path.join(ROOT, "gpurun_out", DETAIL_FILE)):
            try:
                if os.path.isdir(os.path.dirname(path_)):
                    with open(path_, "w") as f_:
                        f_.write(detail_txt + "\n")
            except OSError as e:
                log("could not write %s: %r" % (path_, e))
        print("[bench-detail] " + detail_txt, file=sys.stderr, flush=True)
        line = driver_line(out)
        r_ = roofline or {}
        log("SUMMARY value %.0f q/s (%.4f ms/step, n_gpus %d); roofline frac %s (kernel %s ms, traffic %s B); cpu_baseline %s q/s; recall@10 %s; line %d bytes" %
            (out["value"], out["ms_per_step"], world, r_.get("frac"), r_.get("kernel_ms"), r_.get("traffic"),
             (cpu_baseline or {}).get("value"), checks.get("recall_at_10"), len(line)))
        _restore_stdout(saved_stdout)
        saved_stdout = None
        print(line, flush=True)
        os.dup2(2, 1)                      # whatever teardown prints does not follow the JSON line
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if saved_stdout is not None:
        os.close(saved_stdout)


if __name__ == "__main__":
    main()
