#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Produces, under gpurun_out/prof_<tag>/:
#   stats/   rocprofv3 --kernel-trace --stats of the default bench command
#   pmc_*/   counter passes (separate runs, kernel-trace only) of a short eager bench
# then tools/summarize_profiles.py condenses them into gpurun_out/prof_<tag>/summary/*.{csv,json} (copy those to profiles/), and the
# default `python3 bench.py` runs once more with the fresh counter table in place (summary/bench_C4.json).
set -o pipefail
TAG=${1:-r01}
WL=${2:-C4}          # bench.py --workload: C4 (headline), C3e / C5e (one modality's encoder stack of the MFT at configs[2] / configs[4])
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="$REPO/bench.py --workload $WL"
# the headline workload only (--no-full-model drops the whole-model / full-batch / MFT blocks, which launch the same kernels at other
# sizes and would blur the per-kernel averages the roofline line is checked against)
echo "== kernel-trace + stats of: python3 bench.py --no-full-model --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $BENCH --no-full-model --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log || exit 1
SHORT="--steps 3 --warmup 1 --profile-steps 0 --no-graph --headline-only --no-cpu-baseline"
# (the last three groups answer "how much does a kernel pull from L2 into the CUs" and "how scalar is it"; a counter name this rocprofv3
# does not know fails its own pass only)
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $pass | tr ' ' '+')
  echo "== pmc pass: $pass"
  rocprofv3 --kernel-trace --pmc $pass -d $OUT/pmc_$name -o run --output-format csv -- python3 $BENCH $SHORT > $OUT/pmc_$name.json 2> $OUT/pmc_$name.log || { tail -5 $OUT/pmc_$name.log; echo "pass failed: $pass"; }
done
cd $REPO && python3 tools/summarize_profiles.py $OUT || exit 1
# the summaries in place (on the box), then the default bench line: its roofline.traffic reads the counter table just written
cp $OUT/summary/kernel_stats.csv profiles/${TAG}_kernel_stats_bench_$WL.csv
if [ "$WL" = "C4" ]; then cp $OUT/summary/pmc_per_kernel.json profiles/${TAG}_pmc_per_kernel.json; else cp $OUT/summary/pmc_per_kernel.json profiles/${TAG}_pmc_per_kernel_$WL.json; fi
tail -1 $OUT/bench_under_rocprof.json > profiles/${TAG}_bench_${WL}_under_rocprofv3.json
if [ "$WL" = "C4" ]; then
  echo "== default bench"
  python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
  tail -1 $OUT/bench_default.json > $OUT/summary/bench_C4.json
fi
