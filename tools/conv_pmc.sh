#!/bin/bash
# PMC passes over the window-encoder micro-benchmark (runs on the GPU box)
REPO=$(pwd); OUT=$REPO/gpurun_out/conv_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $pass -d $OUT/pmc_$name -o run --output-format csv -- python3 $REPO/tools/conv_micro.py > $OUT/$name.txt 2> $OUT/$name.log || { tail -3 $OUT/$name.log; echo "pass failed: $pass"; }
done
cd $REPO && python3 tools/summarize_profiles.py $OUT | grep convpool
