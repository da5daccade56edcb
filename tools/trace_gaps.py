#!/usr/bin/env python3
"""Gaps between the launches of one search step, from a rocprofv3 --kernel-trace CSV:
   python tools/trace_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
steps = []
for i, r in enumerate(rows):
    if "hnsw_descent_kernel" in r["Kernel_Name"]:
        j = i + 1
        seq = [r]
        while j < len(rows) and "hnsw_search_kernel" not in rows[j]["Kernel_Name"]:
            seq.append(rows[j]); j += 1
        if j < len(rows):
            seq.append(rows[j]); steps.append(seq)
for seq in steps[-5:]:
    t0 = int(seq[0]["Start_Timestamp"])
    print(" | ".join("%s %.1f-%.1f us" % (s["Kernel_Name"][:28].replace("void ", ""), (int(s["Start_Timestamp"]) - t0) / 1e3, (int(s["End_Timestamp"]) - t0) / 1e3) for s in seq))
if len(steps) >= 2:
    a, b = steps[-2], steps[-1]
    print("previous search end -> next descent start: %.1f us" % ((int(b[0]["Start_Timestamp"]) - int(a[-1]["End_Timestamp"])) / 1e3))
