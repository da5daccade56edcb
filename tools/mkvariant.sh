#!/bin/bash
# Experiment build: exp/<name>.so = the in-tree objects with the headline kernel's translation units
# (hnsw_search_variants 0_0_2 / 0_0_1 and hnsw_order.hip) recompiled with extra flags.
# usage: tools/mkvariant.sh <name> [-DHNSW_...=...]...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/exp/$NAME.obj; mkdir -p $OBJ
CS=$ROOT/ocaml-hnsw_amd/csrc
BASE="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I $ROOT/include"
$BASE "$@" -DHNSW_V_METRIC=0 -DHNSW_V_SEMF=0 -DHNSW_V_FULL=2 -c $CS/hnsw_search_variants.hip -o $OBJ/hnsw_search_variants_0_0_2.o &
$BASE "$@" -DHNSW_V_METRIC=0 -DHNSW_V_SEMF=0 -DHNSW_V_FULL=1 -c $CS/hnsw_search_variants.hip -o $OBJ/hnsw_search_variants_0_0_1.o &
$BASE "$@" -c $CS/hnsw_order.hip -o $OBJ/hnsw_order.hip.o &
wait
OBJS=""
for o in $ROOT/ocaml-hnsw_amd/build/*.o; do
  b=$(basename $o)
  if [ -f $OBJ/$b ]; then OBJS="$OBJS $OBJ/$b"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/exp/$NAME.so
echo $ROOT/exp/$NAME.so
