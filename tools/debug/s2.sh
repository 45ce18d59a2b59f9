#!/bin/bash
mkdir -p gpurun_out/s2
python tools/settle_probe.py > gpurun_out/s2/settle.txt 2> gpurun_out/s2/settle.err
HSRANS_DEBUG_STAMPS=1 python tools/settle_probe.py --windows 10 > gpurun_out/s2/settle_stamps.txt 2> gpurun_out/s2/settle_stamps.err
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > gpurun_out/s2/hwmon_ls.txt 2>&1
rocm-smi --showclocks --showpower > gpurun_out/s2/smi.txt 2>&1
tail -3 gpurun_out/s2/settle.txt | cut -c1-600
