#!/bin/bash
mkdir -p gpurun_out/s14
E=HSRANS_DIRECT_TAIL_PIECES
F=HSRANS_DIRECT_TAIL_PERMILLE
M=HSRANS_DIRECT_STEAL
python tools/ab_probe.py --rounds 4 --variant base --variant k1f80::$E=1,$F=80 --variant k1f80_owner::$E=1,$F=80,$M=1 --variant k2f160::$E=2,$F=160 --variant k3f240::$E=3,$F=240 \
   --variant k1f120::$E=1,$F=120 > gpurun_out/s14/tails.jsonl 2> gpurun_out/s14/err.txt
export HSRANS_DEBUG_STAMPS=1
HSRANS_DIRECT_TAIL_PIECES=1 HSRANS_DIRECT_TAIL_PERMILLE=80 python tools/rot_probe.py --tag k1f80_stamps > gpurun_out/s14/stamps.jsonl 2>> gpurun_out/s14/err.txt
HSRANS_DIRECT_TAIL_PIECES=2 HSRANS_DIRECT_TAIL_PERMILLE=160 python tools/rot_probe.py --tag k2f160_stamps >> gpurun_out/s14/stamps.jsonl 2>> gpurun_out/s14/err.txt
python tools/rot_probe.py --tag base_stamps >> gpurun_out/s14/stamps.jsonl 2>> gpurun_out/s14/err.txt
