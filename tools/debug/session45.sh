#!/bin/bash
mkdir -p gpurun_out/s45
python - <<'PY' > /dev/null 2>&1
import sys; sys.path.insert(0, '.')
from hypersonic_rans_amd import synth
synth.enwik8_shaped(100_000_000).tofile('/tmp/zipf100.bin')
PY
hypersonic_rans_amd/bin/hsrans_harness /tmp/zipf100.bin --bits 11 --runs 1 --decode-runs 6 > gpurun_out/s45/full.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_harness.py -m gpu -x -q -k "pipelin or host or config5 or harness or hpipe" > gpurun_out/s45/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s45/pytest.log
python tools/host_pipeline_rate.py > gpurun_out/s45/host_pipeline_1gib.jsonl 2>/dev/null
