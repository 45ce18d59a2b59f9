# the grouped launch (mt_ 256 KiB blocks): 100 MB with G=32 and 1 GiB with G=256, for the library variants given (HSRANS_LIB)
V=hypersonic_rans_amd/lib/variants
for v in "" "$@" ""; do
  lib=${v:+$PWD/$V/libhsrans_hip_$v.so}
  for rep in 1 2; do
  HSRANS_LIB=$lib timeout 300 python bench.py --workload sharded --steps 20 --no-cpu --interval 256 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('variant [${v:-HEAD}] 1 GiB G=256:', round(r['per_rank'][0]['decode_ms'],4), round(r['roofline']['frac'],4))"
  done
done
