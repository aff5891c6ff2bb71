#!/bin/bash
# A/B timing of mpe_wave.hip variants on one GPU box: builds nothing, expects variants/libmpe_<name>.so.
# usage: tools/ab_wave.sh "<bench args>" name1 name2 ...   ("base" = the in-tree library)
args="$1"; shift
mkdir -p gpurun_out
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="$PWD/variants/libmpe_$v.so"; fi
  MPE_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-step-mode --no-configs --no-sustained $args > gpurun_out/ab_${v}_$rep.json || exit 1
  python3 - "$v" "$rep" <<'PY'
import json,sys
v,rep=sys.argv[1:3]
d=json.loads(open('gpurun_out/ab_%s_%s.json'%(v,rep)).read().strip().splitlines()[-1])
print(v, rep, "%.4f ms" % d['roofline']['launch_ms'], "frac %.3f" % d['roofline']['frac'], flush=True)
PY
done
done
