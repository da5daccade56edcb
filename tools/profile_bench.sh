#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + PMC passes of the default bench workload.
# usage: tools/profile_bench.sh <tag>    -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export BENCH_NO_TAGS_AB=1     # no tag-cache comparison launches on the clustered sets: every kernel line below is ONE workload
# the default bench command minus the legs that would add dispatches of the same kernel on OTHER inputs (the
# secondary data set), need the CPU oracle, or start rocprofv3 themselves: every hnsw_search_kernel dispatch below is the
# headline workload (C2) -- the harder set at the ef of its recall gate has a profile of its own: tools/profile_gate.py under
# tools/profile_cmd.sh --; BENCH_EXTRA=" " adds the C3 / C5 legs (`others`), whose kernels then show too, the clustered twins under their own names (<...,1>: Visited as bitmap blocks)
BENCH="python3 $PWD/bench.py --steps 20 --warmup 3 --no-cpu --no-secondary --no-bench-dist --no-pmc ${BENCH_EXTRA:---no-others}"
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.json 2> $OUT/trace.log || exit 1
pmc() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- $BENCH > $OUT/pmc_$name.json 2> $OUT/pmc_$name.log || echo "pmc pass $name failed" >&2
}
PASSES=${PASSES:-inst,wait,fetch,tcc,grbm}      # the counter groups to collect (each a run of its own)
has() { [[ ",$PASSES," == *",$1,"* ]]; }
has inst && pmc inst SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
has wait && pmc wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT
has fetch && pmc fetch FETCH_SIZE
has tcc && pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
has grbm && pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd - > /dev/null
python3 tools/summarize_prof.py $OUT > gpurun_out/prof_${TAG}_summary.txt 2>&1
# keep only the small stats files: gpurun_out/ is capped at 64 MiB
mkdir -p gpurun_out/prof_${TAG}_keep
find $OUT -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_${TAG}_keep/ \;
cp $OUT/*.json $OUT/*.log gpurun_out/prof_${TAG}_keep/ 2>/dev/null
rm -rf $OUT
cat gpurun_out/prof_${TAG}_summary.txt
