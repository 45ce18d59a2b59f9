# A/B of the age-class weights on the grouped (mt_ / block_ + index) launch: 1 GiB mt_ in 256 KiB blocks (bench.py --workload sharded)
for g in 32 64; do
for w in "1328,1268,1211,1145,1018,875,665,490" "1200,1060,920,820,1200,1060,920,820" "1300,1100,900,700,1300,1100,900,700" "1000,1000,1000,1000,1000,1000,1000,1000"; do
  HSRANS_SLOT_WEIGHTS=$w timeout 300 python bench.py --workload sharded --steps 10 --no-cpu --interval $g 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('G=$g', '$w', 'value', round(r['value']), 'decode_ms', round(r['per_rank'][0]['decode_ms'],4), 'frac', round(r['roofline']['frac'],4))"
done; done
