#!/bin/bash
# A/B of single-step launches: usage ab_step.sh "<bench args>" name1 name2 ...
args="$1"; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="$PWD/variants/libmpe_$v.so"; fi
  MPE_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-configs --no-sustained $args > gpurun_out/abs_${v}_$rep.json 2>/dev/null || exit 1
  python3 - "$v" "$rep" <<'PY'
import json,sys
v,rep=sys.argv[1:3]
d=json.loads(open('gpurun_out/abs_%s_%s.json'%(v,rep)).read().strip().splitlines()[-1])
s=d['step_mode']
print(v, rep, "episode %.4f ms" % d['roofline']['launch_ms'], "| step %.2f us frac %.3f" % (s['launch_ms_hip_events']*1e3, s['roofline_frac']), flush=True)
PY
done
done
