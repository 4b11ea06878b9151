#!/bin/bash
# developer helper (runs ON the GPU box): the round-2 tree (tools/bin/r2tree, built by hand) and the current tree, interleaved, same box
OUT=$1; shift
for i in 1 2; do
  (cd tools/bin/r2tree && python bench.py --no-full-model --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OLDPWD/$OUT/r2_$i.json)
  python bench.py --no-full-model --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/new_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/r2_*.json")+glob.glob("$OUT/new_*.json")):
    b=json.loads(open(f).read())
    k=b['kernel_ms_per_step']
    print(f.split('/')[-1], b['ms_per_step'], ' '.join('%s=%.3f'%(n.split(':')[0][:14]+n[-6:],v) for n,v in list(k.items())[:7]))
PY
