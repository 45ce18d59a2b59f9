#!/bin/bash
OUT=gpurun_out/r03
mkdir -p $OUT gpurun_out/s47
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s47/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s47/pytest.log
HSRANS_HPIPE_DIRECT=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "pipelin or host or config5 or hpipe" > gpurun_out/s47/pytest_direct.log 2>&1; echo "rc=$?" >> gpurun_out/s47/pytest_direct.log
python - <<'PY' > /dev/null 2>&1
import sys; sys.path.insert(0, '.')
from hypersonic_rans_amd import synth
synth.enwik8_shaped(100_000_000).tofile('/tmp/zipf100.bin')
PY
hypersonic_rans_amd/bin/hsrans_harness /tmp/zipf100.bin --bits 11 --runs 2 --decode-runs 8 --test > $OUT/harness_100mb_11bit.txt 2>&1
hypersonic_rans_amd/bin/hsrans_harness /tmp/zipf100.bin --bits 14 --only "(raw)" --runs 1 --decode-runs 8 --test > $OUT/harness_100mb_14bit_raw.txt 2>&1
python tools/host_pipeline_rate.py > $OUT/host_pipeline_1gib.jsonl 2> $OUT/pipeline.err
HSRANS_HPIPE_DIRECT=1 python tools/host_pipeline_rate.py >> $OUT/host_pipeline_1gib.jsonl 2>> $OUT/pipeline.err
python bench.py --workload host > $OUT/bench_host_line.json 2>/dev/null
