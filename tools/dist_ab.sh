#!/bin/bash
# bench_dist counterpart under the distance kernel's tuning switches (run on the GPU box): LDS query reads kept in the loop or hoisted into
# registers (HNSW_DIST_QPIN), waves per launch (HNSW_DIST_WAVES)  -> gpurun_out/<tag>_dist_ab.txt
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_dist_ab.txt
: > $OUT
for q in 1 0; do for w in 65536 16384 8192; do
  echo "== HNSW_DIST_QPIN=$q HNSW_DIST_WAVES=$w" >> $OUT
  HNSW_DIST_QPIN=$q HNSW_DIST_WAVES=$w DIST_ONLY=784 python3 tools/bench_dist.py >> $OUT 2>&1
done; done
cat $OUT
