#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
out = []


def find(pattern):
    return sorted(glob.glob(os.path.join(d, pattern), recursive=True))


for f in find("trace/**/*kernel_stats.csv"):
    out.append("== kernel stats (rocprofv3 --kernel-trace --stats): %s" % os.path.relpath(f, d))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        out.append("  %-70s calls=%s total_ns=%s avg_ns=%s pct=%s" % (
            r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
for f in find("trace/**/*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "hnsw_search_kernel" in r.get("Kernel_Name", "")]
    if rows:
        durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        r0 = rows[0]
        out.append("== hnsw_search_kernel dispatches: n=%d avg=%.1f us min=%.1f us max=%.1f us  VGPR=%s SGPR=%s LDS=%s grid=%s wg=%s" % (
            len(durs), sum(durs) / len(durs) / 1e3, min(durs) / 1e3, max(durs) / 1e3, r0.get("VGPR_Count"),
            r0.get("SGPR_Count"), r0.get("LDS_Block_Size"), r0.get("Grid_Size"), r0.get("Workgroup_Size")))
for pdir in find("pmc_*/"):
    for f in find(os.path.relpath(pdir, d) + "/**/*counter_collection.csv"):
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "hnsw_search_kernel" not in r.get("Kernel_Name", ""):
                continue
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        if acc:
            out.append("== PMC %s (per hnsw_search_kernel dispatch, mean over %d dispatches)" % (
                os.path.basename(os.path.dirname(pdir)), max(v[1] for v in acc.values())))
            for k, (s, c) in sorted(acc.items()):
                out.append("  %-28s %.6g" % (k, s / c))
print("\n".join(out))
