#!/bin/bash
mkdir -p gpurun_out/s11
E=HSRANS_DIRECT_TAIL_PIECES
F=HSRANS_DIRECT_TAIL_PERMILLE
M=HSRANS_DIRECT_STEAL
python tools/ab_probe.py --rounds 4 --variant base --variant k2_whole::$E=2,$F=300,$M=0 --variant k2_owner::$E=2,$F=300,$M=1 --variant k2_noreq::$E=2,$F=300,$M=3 \
   --variant k2_nostore::$E=2,$F=300,$M=4 --variant k2_segonly::$E=2,$F=300,$M=5 > gpurun_out/s11/tails.jsonl 2> gpurun_out/s11/err.txt
cut -c1-230 gpurun_out/s11/tails.jsonl; grep -v amdgpu.ids gpurun_out/s11/err.txt | tail -5
