#!/bin/bash
mkdir -p gpurun_out
for bits in 15 14 13; do
timeout 900 python tools/ab_probe.py --bits $bits --rounds 5 --window 100 --pairs 4 \
  --variant dual --variant direct::HSRANS_DUAL=0 2>> gpurun_out/s38.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$bits', r['tag'], r['rotated_us_median'], r['warm_us_median'], r['rotated_us'], r['launch']['grid'], r['launch']['block'], r['launch']['chains_per_wave'], r['launch']['table_mode'])
" | tee -a gpurun_out/s38_dual_vs_direct.txt
done
tail -3 gpurun_out/s38.err
