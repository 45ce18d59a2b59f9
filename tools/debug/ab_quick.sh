# A/B of kernel build variants (tools/build_variants.sh name=flags ...) on one box: bash tools/debug/ab_quick.sh name1 name2 ...
V=hypersonic_rans_amd/lib/variants
echo "== base"; timeout 300 python tools/cold_cache.py --quick 2>/dev/null | grep -E "case|error"
for v in "$@"; do echo "== $v"; HSRANS_LIB=$PWD/$V/libhsrans_hip_$v.so timeout 300 python tools/cold_cache.py --quick 2>/dev/null | grep -E "case|error"; done
echo "== base again"; timeout 300 python tools/cold_cache.py --quick 2>/dev/null | grep -E "case|error"
