#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "encoder or encode" 2>&1 | grep -E "passed|failed|Error" | head
timeout 900 python tools/encode_rate.py > gpurun_out/s37_enc.jsonl 2> gpurun_out/s37_enc.err
HSRANS_ENC_WAVE_HISTOGRAM=1 timeout 900 python tools/encode_rate.py --cpu-sample 0 2>/dev/null | head -5 > gpurun_out/s37_enc_wavehist.jsonl
python - <<'PY'
import json
for f in ("gpurun_out/s37_enc.jsonl","gpurun_out/s37_enc_wavehist.jsonl"):
    print(f)
    for l in open(f):
        r=json.loads(l); print("  ", r["codec"], r.get("block"), r.get("ms_best"), r.get("GB_s_best", r.get("MB_s_best")), r.get("with_plan_G32_ms_best"))
PY
