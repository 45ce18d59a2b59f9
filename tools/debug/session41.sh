#!/bin/bash
# refresh: 32-state counters, configuration sweep, harness rows
OUT=gpurun_out/r03
mkdir -p $OUT
bash tools/profile.sh r03_s32 --states 32
python tools/sweep_configs.py > $OUT/config_sweep.jsonl 2> $OUT/sweep.err
HSRANS_TABLE_SPILL=1 python tools/sweep_configs.py --only-raw --tag "HSRANS_TABLE_SPILL=1 (tables left in global memory)" >> $OUT/config_sweep.jsonl 2>> $OUT/sweep.err
HSRANS_DUAL=0 python tools/sweep_configs.py --only-raw --tag "HSRANS_DUAL=0 (one chain per wave at 13-15 bits)" >> $OUT/config_sweep.jsonl 2>> $OUT/sweep.err
python - <<'PY' > /tmp/zipf100.bin.log 2>&1
import sys; sys.path.insert(0, '.')
from hypersonic_rans_amd import synth
synth.enwik8_shaped(100_000_000).tofile('/tmp/zipf100.bin')
PY
hypersonic_rans_amd/bin/hsrans_harness /tmp/zipf100.bin --bits 11 --runs 2 --decode-runs 8 --test > $OUT/harness_100mb_11bit.txt 2>&1
hypersonic_rans_amd/bin/hsrans_harness /tmp/zipf100.bin --bits 14 --only "(raw)" --runs 1 --decode-runs 8 --test > $OUT/harness_100mb_14bit_raw.txt 2>&1
tail -2 $OUT/harness_100mb_11bit.txt
