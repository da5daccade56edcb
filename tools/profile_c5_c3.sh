#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + PMC passes of the search kernel on the C5
# ("DEEP-like": n = 10 M, d = 96, M 32, ef 512) and C3 ("GloVe-like": n = 1.18 M, d = 100, inner product,
# M 32, ef 256, k 100) shapes.  The index is built once per shape and reloaded by the later passes.
# usage: tools/profile_c5_c3.sh <tag> [c5|c3|both]
set -u
TAG=${1:-r02}; WHICH=${2:-both}
export HNSW_ORDER_QUERIES=${HNSW_ORDER_QUERIES:--1}
if [ "$WHICH" = c5 ] || [ "$WHICH" = both ]; then
  export N=10000000 D=96 M=32 EFC=200 K=10 KIND=unit METRIC=0 INDEX_CACHE=/tmp/hnsw_c5.idx
  tools/profile_cmd.sh ${TAG}_c5 ${PASSES:-trace,inst,wait,fetch,tcc} python3 $PWD/tools/sweep.py 10000,512,0
  rm -f /tmp/hnsw_c5.idx
fi
if [ "$WHICH" = c3 ] || [ "$WHICH" = both ]; then
  export N=1183514 D=100 M=32 EFC=200 K=100 KIND=unit METRIC=1 INDEX_CACHE=/tmp/hnsw_c3.idx
  tools/profile_cmd.sh ${TAG}_c3 ${PASSES:-trace,inst,wait,fetch,tcc} python3 $PWD/tools/sweep.py 10000,256,0
  rm -f /tmp/hnsw_c3.idx
fi
