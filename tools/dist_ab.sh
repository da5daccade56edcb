#!/bin/bash
# bench_dist counterpart under the distance kernel's grid size (run on the GPU box): waves per launch (HNSW_DIST_WAVES)
#   -> gpurun_out/<tag>_dist_ab.txt   (round 6's A/Bs of where the query lives needed builds of their own: profiles/r06_dist_ab.txt)
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_dist_ab.txt
: > $OUT
for w in 65536 16384 8192; do
  echo "== HNSW_DIST_WAVES=$w" >> $OUT
  HNSW_DIST_WAVES=$w DIST_ONLY=784 python3 tools/bench_dist.py >> $OUT 2>&1
done
cat $OUT
