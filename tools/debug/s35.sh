#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "few_large" 2>&1 | grep -E "passed|failed|Error|assert" | head
for cfg in "262144 11 32" "1048576 11 32"; do
  set -- $cfg
  timeout 900 python tools/ab_probe.py --container mt --block $1 --bits $2 --index $3 --rounds 5 --window 100 --pairs 4 \
    --variant weighted --variant equal:lib/variants/libhsrans_hip_noearly.so --variant grouped::HSRANS_SPREAD=0 2>> gpurun_out/s35.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$cfg', r['tag'], r['rotated_us_median'], r['warm_us_median'], r['rotated_us'], r['launch'].get('spread'))
"
done
tail -3 gpurun_out/s35.err
