#!/bin/bash
# Builds A/B variants of libhsrans_hip.so that differ only in -D flags of hsrans_kernels.hip (kernel experiments):
#   tools/build_variants.sh name1="-DFLAG=1" name2="-DOTHER=0 -DX=2" ...
# -> hypersonic_rans_amd/lib/variants/libhsrans_hip_<name>.so ; select one at run time with HSRANS_LIB=<path>.
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
C=$HERE/hypersonic_rans_amd/csrc
OUT=$HERE/hypersonic_rans_amd/lib/variants
mkdir -p "$OUT" "$C/build/variants"
make -s -C "$C" -j8
for spec in "$@"; do
  name=${spec%%=*}
  flags=${spec#*=}
  eval "farr=($flags)" # (flags may carry quoted strings: -DHSRANS_STORE_POLICY='" nt"')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter --offload-arch=gfx950 "${farr[@]}" -c "$C/hsrans_kernels.hip" -o "$C/build/variants/kernels_$name.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libhsrans_hip_$name.so" "$C/build/hsrans_host.o" "$C/build/hsrans_capi.o" "$C/build/hsrans_capi_encode.o" "$C/build/hsrans_capi_index.o" "$C/build/hsrans_capi_hpipe.o" "$C/build/hsrans_capi_calibrate.o" "$C/build/hsrans_batch.o" "$C/build/hsrans_batch_deal.o" "$C/build/hsrans_comm.o" "$C/build/hsrans_dropin.o" \
    "$C/build/hsrans_cpu.o" "$C/build/variants/kernels_$name.o" "$C/build/hsrans_encode.o" -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libhsrans_hip.so
  echo "built $OUT/libhsrans_hip_$name.so ($flags)"
done
