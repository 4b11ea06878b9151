#!/bin/bash
# developer helper (runs ON the GPU box): the current library against experiment builds tools/bin/libmmt_exp<k>.so, interleaved, same box
# usage: tools/abx.sh <outdir> "<k1> <k2> ..." [bench flags]
OUT=$1; KS=$2; shift 2
for i in 1 2; do
  python bench.py --no-full-model --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/base_$i.json
  for k in $KS; do
    MMT_LIB_PATH=$PWD/tools/bin/libmmt_exp$k.so python bench.py --no-full-model --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/exp${k}_$i.json
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/base_*.json")+glob.glob("$OUT/exp*_*.json")):
    b=json.loads(open(f).read())
    k=b['kernel_ms_per_step']
    print('%-14s'%f.split('/')[-1], b['ms_per_step'], ' '.join('%s=%.4f'%(n.split(':')[0][:14]+n[-6:],v) for n,v in list(k.items())[:7]))
PY
