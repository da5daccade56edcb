#!/bin/bash
# Cycles per phase of a hop of the hand-scheduled loop (two-slot variant): four measurement builds (-DHNSW_ASM_PHASE=0..3),
# each sums one phase into the n_dist counter; run on the GPU box after building the variants here:
#   for k in 0 1 2 3; do tools/mkvariant.sh phase$k -DHNSW_ASM_PHASE=$k; done      (build container)
#   bash tools/asm_phases.sh                                                        (GPU box)
for k in 0 1 2 3; do HNSW_LIB_PATH=$PWD/exp/phase$k.so PHASE=$k python3 tools/asm_phases.py; done
