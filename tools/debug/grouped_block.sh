# the sharded workload (1 GiB mt_) by block size, checkpoint every 32 groups
for b in 65536 262144 1048576 4194304; do
  timeout 300 python bench.py --workload sharded --steps 10 --no-cpu --interval 32 --block $b 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('block $b', 'value', round(r['value']), 'decode_ms', round(r['per_rank'][0]['decode_ms'],4), 'frac', round(r['roofline']['frac'],4), 'chains', r['per_rank'][0]['chains'])"
done
