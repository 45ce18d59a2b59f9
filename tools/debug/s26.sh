#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python tools/ab_probe.py --container mt --block 262144 --index 32 --rounds 5 --window 100 \
  --variant base --variant p64ovl::HSRANS_GROUP_PART_CHAINS=64,HSRANS_GROUP_OVERLAP=1 \
  --variant p64ovl_w8::HSRANS_GROUP_PART_CHAINS=64,HSRANS_GROUP_OVERLAP=1,HSRANS_WAVES_PER_WG=8 \
  --variant w3p43ovl::HSRANS_GROUP_WANT_PER_CU=3,HSRANS_GROUP_PART_CHAINS=43,HSRANS_GROUP_OVERLAP=1 \
  --variant w4p32ovl::HSRANS_GROUP_WANT_PER_CU=4,HSRANS_GROUP_PART_CHAINS=32,HSRANS_GROUP_OVERLAP=1 \
  --variant w4p32ovl_w8::HSRANS_GROUP_WANT_PER_CU=4,HSRANS_GROUP_PART_CHAINS=32,HSRANS_GROUP_OVERLAP=1,HSRANS_WAVES_PER_WG=8 \
  --variant p64ovl_prio0::HSRANS_GROUP_PART_CHAINS=64,HSRANS_GROUP_OVERLAP=1,HSRANS_GROUP_PRIO=0 \
  --variant p64ovl_w12::HSRANS_GROUP_PART_CHAINS=64,HSRANS_GROUP_OVERLAP=1,HSRANS_WAVES_PER_WG=12 \
  > gpurun_out/s26_grouped.jsonl 2> gpurun_out/s26.err
python - <<'PY'
import json
for l in open("gpurun_out/s26_grouped.jsonl"):
    r = json.loads(l); print(r["tag"], r["rotated_us_median"], r["warm_us_median"], r["launch"]["grid"], r["launch"]["block"], r["launch"].get("dynamic_groups"))
PY
tail -3 gpurun_out/s26.err
