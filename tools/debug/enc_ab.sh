#!/bin/bash
# GPU encoders: parity tests, then the rates (tools/encode_rate.py)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "encoder or encode" 2>&1 | grep -E "passed|failed|Error" | head
timeout 900 python tools/encode_rate.py --cpu-sample 0 > gpurun_out/enc_ab.jsonl 2> gpurun_out/enc_ab.err
python - <<'PY'
import json
for l in open("gpurun_out/enc_ab.jsonl"):
    r=json.loads(l); print("  ", r["codec"], r.get("block"), r.get("ms_best"), r.get("GB_s_best", r.get("MB_s_best")), r.get("with_plan_G32_ms_best"), r.get("round_trip_bit_exact"))
PY
