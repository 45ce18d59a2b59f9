# A/B of the age-class weights on the grouped (mt_ / block_ + index) launch: 1 GiB mt_ in 256 KiB blocks (bench.py --workload sharded)
for w in "1350,1100,870,680,1300,1080,860,660" "1328,1268,1211,1145,1018,875,665,490" "1424,1371,1283,1165,920,768,606,464" "1200,1100,1000,900,1100,1000,900,800" "1000,1000,1000,1000,1000,1000,1000,1000" "1500,1200,900,600,1400,1100,800,500"; do
  HSRANS_SLOT_WEIGHTS=$w timeout 300 python bench.py --workload sharded --steps 10 --no-cpu 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$w', 'value', round(r['value']), 'decode_ms', r['per_rank'][0]['decode_ms'], 'frac', round(r['roofline']['frac'],4))"
done
