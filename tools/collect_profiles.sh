#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Produces, under gpurun_out/prof_<tag>/:
#   stats/   rocprofv3 --kernel-trace --stats of the default bench command
#   pmc_*/   counter passes (separate runs, kernel-trace only) of a short eager bench
# then tools/summarize_profiles.py condenses them into gpurun_out/prof_<tag>/summary/*.{csv,json} (copy those to profiles/).
set -o pipefail
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="$REPO/bench.py"
# the headline workload only (--no-full-model drops the whole-model / full-batch / MFT blocks, which launch the same kernels at other
# sizes and would blur the per-kernel averages the roofline line is checked against)
echo "== kernel-trace + stats of: python3 bench.py --no-full-model --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $BENCH --no-full-model --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log || exit 1
SHORT="--steps 3 --warmup 1 --profile-steps 0 --no-graph --no-full-model --no-cpu-baseline"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"; do
  name=$(echo $pass | tr ' ' '+')
  echo "== pmc pass: $pass"
  rocprofv3 --kernel-trace --pmc $pass -d $OUT/pmc_$name -o run --output-format csv -- python3 $BENCH $SHORT > $OUT/pmc_$name.json 2> $OUT/pmc_$name.log || { tail -5 $OUT/pmc_$name.log; echo "pass failed: $pass"; }
done
cd $REPO && python3 tools/summarize_profiles.py $OUT
