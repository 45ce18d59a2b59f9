#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "few_large or dynamic_block" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15
for cfg in "262144 11 32" "1048576 11 32" "262144 10 32"; do
  set -- $cfg
  timeout 900 python tools/ab_probe.py --container mt --block $1 --bits $2 --index $3 --rounds 4 --window 100 --pairs 4 \
    --variant spread --variant grouped::HSRANS_SPREAD=0 2>> gpurun_out/s32.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$cfg', r['tag'], r['rotated_us_median'], r['warm_us_median'], r['launch']['grid'], r['launch']['block'], r['launch'].get('spread'))
"
done
tail -3 gpurun_out/s32.err
