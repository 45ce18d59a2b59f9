#!/bin/bash
# k_decode_spread by checkpoint interval (a wave's share is whole chains: finer checkpoints = finer rounding, larger index)
mkdir -p gpurun_out
for idx in 32 16 8; do
  timeout 900 python tools/ab_probe.py --container mt --block 262144 --bits 11 --index $idx --rounds 5 --window 100 --pairs 4 \
    --variant spread --variant grouped::HSRANS_SPREAD=0 2>> gpurun_out/spread_interval.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('interval $idx', r['tag'], r['rotated_us_median'], r['warm_us_median'], 'chains', r['chains'], 'plan MB', round(r['plan_bytes']/1e6,1), 'spread', r['launch'].get('spread'))
" | tee -a gpurun_out/spread_interval.txt
done
