#!/bin/bash
# developer helper (runs ON the GPU box): device-resident seed vs by-value seed, graph and eager
OUT=$1
for i in 1 2; do
  python bench.py --no-full-model --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/dev_$i.json
  MMT_DEVICE_SEED=0 python bench.py --no-full-model --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/val_$i.json
done
MMT_DEVICE_SEED=1 python bench.py --no-full-model --no-cpu-baseline --no-graph 2>/dev/null | tail -1 > $OUT/eagerdev_1.json
MMT_DEVICE_SEED=0 python bench.py --no-full-model --no-cpu-baseline --no-graph 2>/dev/null | tail -1 > $OUT/eagerval_1.json
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*_?.json")):
    b=json.loads(open(f).read())
    k=b['kernel_ms_per_step']
    print(f.split('/')[-1], b['launch'], b['ms_per_step'], ' '.join('%s=%.3f'%(n.split(':')[0][:14]+n[-6:],v) for n,v in list(k.items())[:8]))
PY
