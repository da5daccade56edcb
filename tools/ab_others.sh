#!/bin/bash
# A/B of whole-library variants on the C3 / C5 shapes (bench.py's `others` leg), alternating on one box:
#   tools/ab_others.sh exp/a.so exp/b.so [rounds]      -> gpurun_out/ab_others.txt
# Each run is one `bench.py --no-cpu --no-secondary --no-pmc --no-bench-dist`; the lines printed are kernel ms, q/s and the
# roofline fraction per configuration, and whether the sample agreed with the oracle.
R=$(cd "$(dirname "$0")/.." && pwd)
A=$1; B=$2; ROUNDS=${3:-2}
OUT=$R/gpurun_out/ab_others.txt; mkdir -p $R/gpurun_out; : > $OUT
for r in $(seq $ROUNDS); do
  for lib in $A $B; do
    echo "== $lib (round $r)" >> $OUT
    HNSW_LIB_PATH=$R/$lib BENCH_NO_TAGS_AB=1 timeout -k 10 500 python3 $R/bench.py --no-cpu --no-secondary --no-pmc --no-bench-dist --steps 10 \
        > $R/gpurun_out/ab_others_line.json 2> $R/gpurun_out/ab_others_err.log || { echo "bench failed" >> $OUT; tail -5 $R/gpurun_out/ab_others_err.log >> $OUT; exit 1; }
    python3 - $R/bench_detail.json >> $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("  headline %.4f ms/step  %.2f M q/s  kernel %.4f ms" % (d["ms_per_step"], d["value"] / 1e6, d["roofline"]["kernel_ms"]))
f = d.get("float32_rows") or {}
if f:
    print("  float32 rows: %s" % {k: f[k] for k in ("value", "ms_per_step", "kernel_ms", "frac") if k in f})
for tag, o in (d.get("others") or {}).items():
    if isinstance(o, dict) and "value" in o:
        r, c = o.get("roofline") or {}, o.get("checks") or {}
        print("  %-13s %.4f M q/s  kernel %s ms  frac %s  parity %s/%s  evals/q %s" % (tag, o["value"] / 1e6, r.get("kernel_ms"), r.get("frac"),
              c.get("parity_ids_equal"), c.get("parity_dist_bits_equal"), r.get("n_dist_per_query")))
    else:
        print("  %-13s %s" % (tag, o))
PY
  done
done
cat $OUT
