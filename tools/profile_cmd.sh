#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + separate PMC passes of an arbitrary python command.
# usage: tools/profile_cmd.sh <tag> <passes> python3 <script> [args...]    -> gpurun_out/prof_<tag>_summary.txt
#   passes: comma list out of trace,inst,wait,fetch,tcc,grbm   (each PMC group is a run of its own)
set -u
TAG=$1; PASSES=$2; shift 2
OUT=$PWD/gpurun_out/prof_$TAG
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
export PYTHONPATH=$REPO
cd /tmp
has() { [[ ",$PASSES," == *",$1,"* ]]; }
if has trace; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- "$@" > $OUT/trace.out 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
fi
pmc() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- "${CMD[@]}" > $OUT/pmc_$name.out 2> $OUT/pmc_$name.log || echo "pmc pass $name failed" >&2
}
CMD=("$@")
has inst && pmc inst SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
has wait && pmc wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT
has fetch && pmc fetch FETCH_SIZE
has tcc && pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
has grbm && pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $REPO
python3 tools/summarize_prof.py $OUT > gpurun_out/prof_${TAG}_summary.txt 2>&1
mkdir -p gpurun_out/prof_${TAG}_keep
find $OUT -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_${TAG}_keep/ \;
cp $OUT/*.out $OUT/*.log gpurun_out/prof_${TAG}_keep/ 2>/dev/null
rm -rf $OUT
cat gpurun_out/prof_${TAG}_summary.txt
