#!/bin/bash
# developer helper (runs ON the GPU box): interleaved same-box timing of attention-core variants.  usage: tools/ab_attn.sh <outdir> <rounds> <T> name1 name2 ...
# ("base" = the product library; other names = tools/bin/libmmt_<name>.so)
OUT=$1; R=$2; T=$3; shift 3
for i in $(seq 1 $R); do for v in "$@"; do
  if [ "$v" = base ]; then L=""; else L="tools/bin/libmmt_$v.so"; fi
  MMT_LIB_PATH=$L timeout -k 5 120 python tools/attn_micro.py --p 0.1 --iters 50 --T $T 2>/dev/null | grep "attn_bwd\|attn_fwd" | awk -v n=$v '{printf "%s %s %s\n", n, $1, $2}' >> $OUT/ab_T$T.txt
done; done
python3 - <<PY
import collections
d=collections.defaultdict(list)
for l in open("$OUT/ab_T$T.txt"):
    n,k,v=l.split(); d[(n,k)].append(float(v))
for (n,k),v in sorted(d.items()): print("T=$T %-14s %-26s %s  min %.2f" % (n,k," ".join("%.2f"%x for x in v),min(v)))
PY
