#!/bin/bash
mkdir -p gpurun_out/s10
E=HSRANS_DIRECT_TAIL_PIECES
F=HSRANS_DIRECT_TAIL_PERMILLE
python tools/ab_probe.py --rounds 5 --variant base --variant k1f150::$E=1,$F=150 --variant k1f250::$E=1,$F=250 --variant k2f200::$E=2,$F=200 --variant k2f300::$E=2,$F=300 \
   --variant k3f300::$E=3,$F=300 --variant k3f450::$E=3,$F=450 > gpurun_out/s10/tails.jsonl 2> gpurun_out/s10/err.txt
cut -c1-230 gpurun_out/s10/tails.jsonl; grep -v amdgpu.ids gpurun_out/s10/err.txt | tail -5
