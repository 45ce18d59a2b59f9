# Experiments around the cold-stream penalty of the headline decode ("warm and cold").  Run through gpurun.
# Needs the A/B libraries first: tools/build_variants.sh ldnt=-DHSRANS_STREAM_LOAD_POLICY=1 ldsc1=-DHSRANS_STREAM_LOAD_POLICY=2 ldsc01=-DHSRANS_STREAM_LOAD_POLICY=3 plainst=-DHSRANS_NT_STORES=0
set -x
D=gpurun_out/dump3
mkdir -p $D
V=hypersonic_rans_amd/lib/variants
timeout 300 python tools/cold_cache.py --quick > $D/base.jsonl 2>&1
for v in ldnt ldsc1 ldsc01 plainst; do
  HSRANS_LIB=$PWD/$V/libhsrans_hip_$v.so timeout 300 python tools/cold_cache.py --quick > $D/$v.jsonl 2>&1
done
timeout 300 python tools/cold_cache.py --quick --pairs 8 > $D/pairs8.jsonl 2>&1
timeout 300 python tools/cold_cache.py --quick --slab > $D/slab.jsonl 2>&1
HSRANS_STAMPS_DUMP=$D/flush.npz timeout 300 python tools/stamps.py --index wave --cold 4 --flush --dump-launches 4 > $D/flush.txt 2>&1
(cd /tmp && rocprofv3 -L > $OLDPWD/$D/counters.txt 2>&1)
grep -c . $D/counters.txt
for f in $D/*.jsonl; do echo $f; grep case $f; done
