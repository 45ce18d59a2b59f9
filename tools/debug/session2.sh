#!/bin/bash
mkdir -p gpurun_out/s2
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/s2/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s2/pytest.log
python tools/host_pipeline_rate.py > gpurun_out/s2/host_pipeline.jsonl 2> gpurun_out/s2/host_pipeline.err
HSRANS_HPIPE_STAGED=1 python tools/host_pipeline_rate.py > gpurun_out/s2/host_pipeline_staged.jsonl 2> gpurun_out/s2/host_pipeline_staged.err
python tools/host_decoder_vs_reference.py --size 100000000 --budget 2.0 --cases 32:11,32:14,64:11,64:14,32:15,64:15,32:12,64:13 > gpurun_out/s2/host_decoder.jsonl 2> gpurun_out/s2/host_decoder.err
tail -3 gpurun_out/s2/pytest.log
