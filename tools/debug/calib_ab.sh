#!/bin/bash
# calibration under sustained rotated launches (this build) against the warm single-buffer calibration (variant `head`), headline workload
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_calibrate.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | head -5
for rep in 1 2; do
timeout 900 python tools/ab_probe.py --rounds 6 --window 150 --pairs 4 --calibrate \
  --variant newcal --variant oldcal:lib/variants/libhsrans_hip_head.so 2>> gpurun_out/calib.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['tag'], r['rotated_us_median'], r['warm_us_median'], r['rotated_us'], r['calibration'])
" | tee -a gpurun_out/calib_ab.txt
done
timeout 900 python tools/ab_probe.py --rounds 6 --window 150 --pairs 4 --variant nocal 2>> gpurun_out/calib.err | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['tag'], r['rotated_us_median'], r['warm_us_median'], r['rotated_us'], r['launch']['class_weights'])
" | tee -a gpurun_out/calib_ab.txt
tail -3 gpurun_out/calib.err
