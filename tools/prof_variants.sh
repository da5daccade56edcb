#!/bin/bash
# PMC instruction / wait counters of library variants on the 10 k batch (run on the GPU box)
# usage: tools/prof_variants.sh <variant>...   (exp/<variant>.so)
R=$PWD
for v in "$@"; do
  AB_NQ=${AB_NQ:-10000} REPS=5 bash tools/profile_cmd.sh ab_$v ${PASSES:-inst,wait} python3 $R/tools/ab.py --one $R/exp/$v.so > gpurun_out/prof_ab_$v.out 2>&1 || exit 1
done
for v in "$@"; do echo "#### $v"; grep -h "PMC\|SQ_INSTS\|SQ_WAIT\|SQ_ACTIVE\|SQ_WAVE_CYCLES\|SQ_BUSY\|LDS_BANK" gpurun_out/prof_ab_${v}_summary.txt; done
