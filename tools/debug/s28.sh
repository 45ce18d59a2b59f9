#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python tools/ab_probe.py --container mt --block 262144 --index 32 --rounds 5 --window 100 \
  --variant base --variant p64ovl::HSRANS_GROUP_PART_CHAINS=64,HSRANS_GROUP_OVERLAP=1 \
  --variant p64ovl_w10::HSRANS_GROUP_PART_CHAINS=64,HSRANS_GROUP_OVERLAP=1,HSRANS_WAVES_PER_WG=10 \
  --variant p64_w10::HSRANS_GROUP_PART_CHAINS=64,HSRANS_WAVES_PER_WG=10 \
  --variant w3p43ovl_w6::HSRANS_GROUP_WANT_PER_CU=3,HSRANS_GROUP_PART_CHAINS=43,HSRANS_GROUP_OVERLAP=1,HSRANS_WAVES_PER_WG=6 \
  --variant w3p43ovl_w7::HSRANS_GROUP_WANT_PER_CU=3,HSRANS_GROUP_PART_CHAINS=43,HSRANS_GROUP_OVERLAP=1,HSRANS_WAVES_PER_WG=7 \
  --variant p64ovl_w10_prio0::HSRANS_GROUP_PART_CHAINS=64,HSRANS_GROUP_OVERLAP=1,HSRANS_WAVES_PER_WG=10,HSRANS_GROUP_PRIO=0 \
  > gpurun_out/s28_grouped.jsonl 2> gpurun_out/s28.err
python - <<'PY'
import json
for l in open("gpurun_out/s28_grouped.jsonl"):
    r = json.loads(l); print(r["tag"], r["rotated_us_median"], r["warm_us_median"], r["launch"]["grid"], r["launch"]["block"], r["launch"].get("dynamic_groups"), r["launch"]["lds_bytes"])
PY
tail -3 gpurun_out/s28.err
