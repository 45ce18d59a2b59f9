#!/bin/bash
# BASELINE config 4's kernel at 12-15 bits: bench.py --workload sharded (2^30 B mt_, 256 KiB blocks, G = 256), one line per width
mkdir -p gpurun_out
for bits in 12 13 14 15; do
  python bench.py --workload sharded --no-cpu --bits $bits --parts 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; l = d['per_rank'][0]['launch']
print('bits $bits', 'ms/step', round(d['ms_per_step'], 4), 'frac', round(r['frac'], 4), 'wall', round(r['frac_wall'], 4), 'grid', l['grid'], 'block', l['block'], 'lds', l['lds_bytes'], 'mode', l['table_mode'])
" | tee -a gpurun_out/grouped_wide.txt
done
