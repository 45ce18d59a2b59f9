#!/bin/bash
mkdir -p gpurun_out/s12
O=gpurun_out/s12/seg.jsonl
export HSRANS_DEBUG_STAMPS=1
python tools/rot_probe.py --tag base > $O 2> gpurun_out/s12/err.txt
HSRANS_DIRECT_TAIL_PIECES=2 HSRANS_DIRECT_TAIL_PERMILLE=300 HSRANS_DIRECT_STEAL=0 python tools/rot_probe.py --tag k2_whole >> $O 2>> gpurun_out/s12/err.txt
HSRANS_DIRECT_TAIL_PIECES=2 HSRANS_DIRECT_TAIL_PERMILLE=300 HSRANS_DIRECT_STEAL=5 python tools/rot_probe.py --tag k2_segonly >> $O 2>> gpurun_out/s12/err.txt
HSRANS_DIRECT_TAIL_PIECES=2 HSRANS_DIRECT_TAIL_PERMILLE=300 HSRANS_DIRECT_STEAL=2 python tools/rot_probe.py --tag k2_full >> $O 2>> gpurun_out/s12/err.txt
cut -c1-1200 $O
