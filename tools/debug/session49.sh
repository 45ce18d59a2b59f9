#!/bin/bash
OUT=gpurun_out/r03
mkdir -p $OUT
python tools/sweep_configs.py > $OUT/config_sweep.jsonl 2> $OUT/sweep.err
HSRANS_TABLE_SPILL=1 python tools/sweep_configs.py --only-raw --tag "HSRANS_TABLE_SPILL=1 (tables left in global memory)" >> $OUT/config_sweep.jsonl 2>> $OUT/sweep.err
HSRANS_DUAL=0 python tools/sweep_configs.py --only-raw --tag "HSRANS_DUAL=0 (one chain per wave at 13-15 bits)" >> $OUT/config_sweep.jsonl 2>> $OUT/sweep.err
