#!/bin/bash
mkdir -p gpurun_out/s46
rm -f gpurun_out/s46/*
python - <<'PY' > /dev/null 2>&1
import sys; sys.path.insert(0, '.')
from hypersonic_rans_amd import synth
synth.enwik8_shaped(100_000_000).tofile('/tmp/zipf100.bin')
PY
HSRANS_HPIPE_STAGED=1 hypersonic_rans_amd/bin/hsrans_harness /tmp/zipf100.bin --bits 11 --runs 1 --decode-runs 6 > gpurun_out/s46/full_staged.txt 2>&1
