#!/bin/bash
mkdir -p gpurun_out/s8
python -m pytest tests/test_gpu_calibrate.py tests/test_bench_contract.py -m gpu -x -q > gpurun_out/s8/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s8/pytest.log
python - <<'PY' > /dev/null 2>&1
import sys; sys.path.insert(0, '.')
from hypersonic_rans_amd import synth
synth.enwik8_shaped(100_000_000).tofile('/tmp/zipf100.bin')
PY
HSRANS_HPIPE_TRACE=1 hypersonic_rans_amd/bin/hsrans_harness /tmp/zipf100.bin --bits 11 --runs 1 --decode-runs 3 --only "64 16w (raw)" > gpurun_out/s8/harness_trace_raw.txt 2>&1
HSRANS_HPIPE_TRACE=1 hypersonic_rans_amd/bin/hsrans_harness /tmp/zipf100.bin --bits 11 --runs 1 --decode-runs 3 --only "64 16w (independent" > gpurun_out/s8/harness_trace_mt.txt 2>&1
for rep in 1 2 3; do
python bench.py --no-cpu --no-single > gpurun_out/s8/bench_cal_$rep.json 2> gpurun_out/s8/bench_cal_$rep.err
python bench.py --no-cpu --no-single --no-calibrate > gpurun_out/s8/bench_nocal_$rep.json 2> gpurun_out/s8/bench_nocal_$rep.err
done
tail -3 gpurun_out/s8/pytest.log
