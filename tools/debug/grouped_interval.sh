# the sharded workload (1 GiB mt_ in 256 KiB blocks) by checkpoint interval
for g in 256 128 64 32 16; do
  timeout 300 python bench.py --workload sharded --steps 10 --no-cpu --interval $g 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('interval $g', 'value', round(r['value']), 'decode_ms', round(r['per_rank'][0]['decode_ms'],4), 'frac', round(r['roofline']['frac'],4), 'chains', r['per_rank'][0]['chains'])"
done
