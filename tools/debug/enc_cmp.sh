#!/bin/bash
for v in "" encserial "" encserial; do
  lib=${v:+$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_$v.so}
  echo "== ${v:-parallel build (default)}"
  HSRANS_LIB=$lib timeout 600 python tools/encode_rate.py --cpu-sample 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if r['codec'] in ('mt_ rANS32x64 16w 11', 'raw rANS32x64 16w 11'): print('  ', r['codec'], r.get('block'), r.get('ms_best'))"
done
