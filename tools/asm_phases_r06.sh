#!/bin/bash
# Round 6: cycles per phase of a hop for W in THREE registers against FOUR (HNSW_NSLOT_POW2=1) at the harder set's recall gate (ef 176), and
# for the headline (two registers, ef 128), idle chip (64 queries) and 10 k batch.  Build first (container):
#   for k in 0 1 2 3; do tools/mkvariant.sh phase$k -DHNSW_ASM_PHASE=$k; done
for k in 0 1 2 3; do
  HNSW_LIB_PATH=$PWD/exp/phase$k.so PHASE=$k HARD=1 EF=176 python3 tools/asm_phases.py
  HNSW_LIB_PATH=$PWD/exp/phase$k.so PHASE=$k HARD=1 EF=176 HNSW_NSLOT_POW2=1 python3 tools/asm_phases.py
  HNSW_LIB_PATH=$PWD/exp/phase$k.so PHASE=$k python3 tools/asm_phases.py
done
