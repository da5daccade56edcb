#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
out = []
KERNELS = ("hnsw_search_kernel", "hnsw_distance_kernel", "hnsw_descent_kernel")      # the kernels whose dispatches are summarised


def wanted(name):
    return any(k in name for k in KERNELS)


def find(pattern):
    return sorted(glob.glob(os.path.join(d, pattern), recursive=True))


for f in find("trace/**/*kernel_stats.csv"):
    out.append("== kernel stats (rocprofv3 --kernel-trace --stats): %s" % os.path.relpath(f, d))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        out.append("  %-70s calls=%s total_ns=%s avg_ns=%s pct=%s" % (
            r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
for f in find("trace/**/*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if wanted(r.get("Kernel_Name", ""))]
    if rows:
        durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        r0 = rows[0]
        # resource columns as rocprofv3 names them; the compiler's own table (build.py --resources) is the
        # authority for registers: the trace reports allocation granules / accumulation offsets, not counts
        def col(*names):
            for nme in names:
                if r0.get(nme) not in (None, ""):
                    return r0.get(nme)
            return "n/a"
        # per kernel AND launch size: one bench run launches the same kernel on its 10 k batches, on the 100 k bandwidth point and
        # (option visited_blocks -1) on the 256 sample queries of the handle's own measurement -- their durations are not one average
        by_name = defaultdict(list)
        for r, du in zip(rows, durs):
            grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
            try:
                grid = "%d waves" % (int(grid) // 64)
            except ValueError:
                pass
            by_name[(r["Kernel_Name"].split("(")[0], grid)].append(du)
        for (nme, grid), ds in sorted(by_name.items()):
            out.append("== %s [%s per launch: one wave per query in the search kernels] dispatches: n=%d avg=%.1f us min=%.1f us max=%.1f us" % (nme[-60:], grid, len(ds), sum(ds) / len(ds) / 1e3, min(ds) / 1e3, max(ds) / 1e3))
        out.append("   (trace columns of the first dispatch: arch_vgpr=%s accum_vgpr=%s sgpr=%s lds=%s grid=%s wg=%s)" % (
            col("Arch_VGPR_Count", "VGPR_Count"), col("Accum_VGPR_Count"), col("SGPR_Count"), col("LDS_Block_Size", "LDS_Block_Size_v"),
            col("Grid_Size", "Grid_Size_X"), col("Workgroup_Size", "Workgroup_Size_X")))
def short(name):
    """hnsw_search_kernel<2, 8, 2, 0, 0, 2> from the demangled signature"""
    n = name.split("(")[0]
    for k in KERNELS:
        i = n.find(k)
        if i >= 0:
            return n[i:]
    return n


for pdir in find("pmc_*/"):
    for f in find(os.path.relpath(pdir, d) + "/**/*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))     # per kernel variant (a bench run launches the
        for r in csv.DictReader(open(f)):                            # byte-row and the float32-row kernel)
            if not wanted(r.get("Kernel_Name", "")):
                continue
            grid = r.get("Grid_Size") or r.get("Grid_Size_X") or ""
            try:
                grid = " [%d waves per launch]" % (int(grid) // 64)
            except ValueError:
                grid = ""
            a = acc[short(r["Kernel_Name"]) + grid][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        for kn, cs in sorted(acc.items()):
            out.append("== PMC %s (per dispatch of %s, mean over %d dispatches)" % (
                os.path.basename(os.path.dirname(pdir)), kn, max(v[1] for v in cs.values())))
            for k, (s_, c) in sorted(cs.items()):
                out.append("  %-28s %.6g" % (k, s_ / c))
print("\n".join(out))
