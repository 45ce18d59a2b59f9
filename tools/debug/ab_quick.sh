V=hypersonic_rans_amd/lib/variants
echo base; timeout 300 python tools/cold_cache.py --quick 2>/dev/null | grep case
for v in wait12 ahead2; do echo $v; HSRANS_LIB=$PWD/$V/libhsrans_hip_$v.so timeout 300 python tools/cold_cache.py --quick 2>/dev/null | grep case; done
echo "== stamps warm"; timeout 200 python tools/stamps.py --index wave 2>/dev/null | grep -E "^done|by wave|hardware"
echo "== stamps cold"; timeout 200 python tools/stamps.py --index wave --cold 4 2>/dev/null | grep -E "^done|by wave|hardware"
