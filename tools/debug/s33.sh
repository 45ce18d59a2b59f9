#!/bin/bash
mkdir -p gpurun_out
for cfg in "262144 11 32"; do
  set -- $cfg
  timeout 900 python tools/ab_probe.py --container mt --block $1 --bits $2 --index $3 --rounds 5 --window 100 --pairs 4 \
    --variant spread --variant swt:lib/variants/libhsrans_hip_swt.so --variant swt_ns:lib/variants/libhsrans_hip_swt_ns.so --variant s_ns:lib/variants/libhsrans_hip_s_ns.so --variant grouped::HSRANS_SPREAD=0 2>> gpurun_out/s33.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$cfg', r['tag'], r['rotated_us_median'], r['warm_us_median'], r['launch']['grid'], r['launch']['block'], r['launch'].get('spread'))
"
done
tail -3 gpurun_out/s33.err
