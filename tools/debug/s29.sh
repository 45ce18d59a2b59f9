#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/s29_grouped.jsonl
for cfg in "262144 11 32" "1048576 11 32" "524288 11 32" "262144 13 32" "262144 15 32" "262144 11 64" "131072 11 32"; do
  set -- $cfg
  timeout 900 python tools/ab_probe.py --container mt --block $1 --bits $2 --index $3 --rounds 4 --window 100 --pairs 3 \
    --variant base --variant p64::HSRANS_GROUP_PART_CHAINS=64 --variant p64ovl::HSRANS_GROUP_PART_CHAINS=64,HSRANS_GROUP_OVERLAP=1 --variant ovl::HSRANS_GROUP_OVERLAP=1 \
    2>> gpurun_out/s29.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$cfg', r['tag'], r['rotated_us_median'], r['warm_us_median'], r['launch']['grid'], r['launch']['block'], r['launch'].get('dynamic_groups'))
" | tee -a gpurun_out/s29_grouped.jsonl
done
tail -3 gpurun_out/s29.err
