#!/bin/bash
mkdir -p gpurun_out/s9
python tools/host_decoder_vs_reference.py --size 100000000 --budget 2.0 --cases 32:11,32:12,32:13,32:14,32:15,64:11,64:12,64:13,64:14,64:15 > gpurun_out/s9/host_decoder.jsonl 2> gpurun_out/s9/host_decoder.err
HSRANS_CPU_WIDE_MODE=2 python tools/host_decoder_vs_reference.py --size 100000000 --budget 1.5 --cases 32:15,64:13,64:14,64:15 > gpurun_out/s9/host_decoder_byslot.jsonl 2>> gpurun_out/s9/host_decoder.err
HSRANS_CPU_WIDE_MODE=0 python tools/host_decoder_vs_reference.py --size 100000000 --budget 1.5 --cases 32:13,32:14 > gpurun_out/s9/host_decoder_compact.jsonl 2>> gpurun_out/s9/host_decoder.err
python -m pytest tests/test_cpu_decoder.py -q > gpurun_out/s9/pytest.log 2>&1
