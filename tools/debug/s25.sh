#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python tools/ab_probe.py --container mt --block 262144 --index 32 --rounds 5 --window 100 \
  --variant base --variant ovl::HSRANS_GROUP_OVERLAP=1 --variant p64::HSRANS_GROUP_PART_CHAINS=64 --variant p64ovl::HSRANS_GROUP_PART_CHAINS=64,HSRANS_GROUP_OVERLAP=1 \
  --variant fine::HSRANS_GROUP_FINE_SPLIT=1 --variant fineovl::HSRANS_GROUP_FINE_SPLIT=1,HSRANS_GROUP_OVERLAP=1 --variant static::HSRANS_GROUP_STATIC=1 \
  > gpurun_out/s25_grouped.jsonl 2> gpurun_out/s25.err
python - <<'PY'
import json
for l in open("gpurun_out/s25_grouped.jsonl"):
    r = json.loads(l); print(r["tag"], r["rotated_us_median"], r["warm_us_median"], r["launch"]["grid"], r["launch"]["block"], r["launch"].get("dynamic_groups"))
PY
tail -3 gpurun_out/s25.err
