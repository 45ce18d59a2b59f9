#!/bin/bash
mkdir -p gpurun_out/s19
for v in symtab symsmall; do
V=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_$v.so
HSRANS_LIB=$V timeout 300 python tools/sweep_configs.py --bits 14,15 --tag $v > gpurun_out/s19/sweep_$v.jsonl 2> gpurun_out/s19/err_$v.txt
done
V=$PWD/hypersonic_rans_amd/lib/variants/libhsrans_hip_symsmall.so
HSRANS_LIB=$V HSRANS_DUAL_WAVES=16 timeout 300 python tools/sweep_configs.py --bits 14 --tag symsmall16 > gpurun_out/s19/sweep_symsmall16.jsonl 2> gpurun_out/s19/err_symsmall16.txt
HSRANS_LIB=$V timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "15_bit or raw_indexed or one_chain_per_wave or random_sweep_decode" > gpurun_out/s19/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/s19/pytest.log
tail -3 gpurun_out/s19/pytest.log
