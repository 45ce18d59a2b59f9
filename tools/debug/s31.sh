#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for cfg in "262144 11 32" "65536 11 32" "262144 14 64"; do
  set -- $cfg
  timeout 900 python tools/ab_probe.py --container mt --block $1 --bits $2 --index $3 --rounds 4 --window 100 --pairs 4 \
    --variant new --variant prev:lib/variants/libhsrans_hip_prev.so 2>> gpurun_out/s31.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$cfg', r['tag'], r['rotated_us_median'], r['warm_us_median'], r['launch']['grid'], r['launch']['block'], r['launch'].get('dynamic_groups'))
"
done
