#!/bin/bash
mkdir -p gpurun_out/s13
E=HSRANS_DIRECT_TAIL_PIECES
F=HSRANS_DIRECT_TAIL_PERMILLE
M=HSRANS_DIRECT_STEAL
python tools/ab_probe.py --rounds 4 --variant base --variant k1_seg::$E=1,$F=250,$M=5 --variant k1_owner::$E=1,$F=250,$M=1 --variant k1_full::$E=1,$F=250,$M=2 \
   --variant k2_owner::$E=2,$F=300,$M=1 --variant k2_full::$E=2,$F=300,$M=2 --variant k3_full::$E=3,$F=450,$M=2 --variant k1f150_full::$E=1,$F=150,$M=2 > gpurun_out/s13/tails.jsonl 2> gpurun_out/s13/err.txt
cut -c1-230 gpurun_out/s13/tails.jsonl; grep -v amdgpu.ids gpurun_out/s13/err.txt | tail -5
