#!/bin/bash
# developer helper (runs ON the GPU box): one library build (MMT_LIB_PATH as given), bench with and without an environment switch, interleaved
# usage: tools/abenv.sh <outdir> "<VAR=value>" [bench flags]
OUT=$1; SW=$2; shift 2
for i in 1 2 3; do
  python bench.py --no-full-model --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/off_$i.json
  env $SW python bench.py --no-full-model --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/on_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/off_*.json")+glob.glob("$OUT/on_*.json")):
    b=json.loads(open(f).read())
    k=b['kernel_ms_per_step']
    print('%-14s'%f.split('/')[-1], b['ms_per_step'], b.get('with_adam',{}).get('ms_per_step'), ' '.join('%s=%.4f'%(n.split(':')[0][:14]+n[-6:],v) for n,v in list(k.items())[:7]))
PY
