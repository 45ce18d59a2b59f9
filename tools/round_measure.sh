#!/bin/bash
# Everything DESIGN.md §8 quotes, in one go on the GPU box:  bash tools/round_measure.sh <tag>   (outputs under gpurun_out/<tag>/)
# (the two stamps files need the diagnostic library: make -C hypersonic_rans_amd/csrc stamps, before gpurun)
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
python bench.py --workload sharded > $OUT/bench_sharded_line.json 2> $OUT/bench_sharded.err
python bench.py --workload sharded --no-cpu --block 65536 --interval 64 > $OUT/bench_sharded_64k_line.json 2> $OUT/bench_sharded_64k.err
python bench.py --workload sharded --no-cpu --size 100000000 --block 262144 --interval 32 > $OUT/bench_sharded_100mb_256k_line.json 2> $OUT/bench_sharded_100mb_256k.err   # few, large blocks: k_decode_spread
python tools/batch_probe.py --batch-index --out $OUT/batch_probe.jsonl > /dev/null 2> $OUT/batch_probe.err
python tools/small_streams_probe.py --out $OUT/small_streams.jsonl > /dev/null 2> $OUT/small_streams.err
python tools/ratio_sweep.py --out $OUT/ratio_sweep.jsonl > /dev/null 2> $OUT/ratio_sweep.err
python tools/first_decode_rate.py > $OUT/first_decode_100mb.jsonl 2> $OUT/first_decode.err
python tools/host_loop_rate.py > $OUT/host_loop_index_cache.jsonl 2> $OUT/host_loop.err
python tools/sweep_configs.py > $OUT/config_sweep.jsonl 2> $OUT/sweep.err
HSRANS_TABLE_SPILL=1 python tools/sweep_configs.py --only-raw --tag "HSRANS_TABLE_SPILL=1 (tables left in global memory)" >> $OUT/config_sweep.jsonl 2>> $OUT/sweep.err
HSRANS_DUAL=0 python tools/sweep_configs.py --only-raw --tag "HSRANS_DUAL=0 (one chain per wave at 13-15 bits)" >> $OUT/config_sweep.jsonl 2>> $OUT/sweep.err
python - <<'PY' > /tmp/zipf100.bin.log 2>&1
import sys; sys.path.insert(0, '.')
from hypersonic_rans_amd import synth
synth.enwik8_shaped(100_000_000).tofile('/tmp/zipf100.bin')
PY
hypersonic_rans_amd/bin/hsrans_harness /tmp/zipf100.bin --bits 11 --runs 2 --decode-runs 8 --test > $OUT/harness_100mb_11bit.txt 2>&1
hypersonic_rans_amd/bin/hsrans_harness /tmp/zipf100.bin --bits 14 --only "(raw)" --runs 1 --decode-runs 8 --test > $OUT/harness_100mb_14bit_raw.txt 2>&1
python tools/host_pipeline_rate.py > $OUT/host_pipeline_1gib.jsonl 2> $OUT/pipeline.err
HSRANS_HPIPE_DIRECT=1 python tools/host_pipeline_rate.py >> $OUT/host_pipeline_1gib.jsonl 2>> $OUT/pipeline.err
python tools/host_decoder_vs_reference.py --size 100000000 --budget 2.0 --cases 32:11,32:12,32:13,32:14,32:15,64:11,64:12,64:13,64:14,64:15 > $OUT/host_decoder_vs_reference.jsonl 2> $OUT/host_decoder.err
python tools/stamps_grouped.py 2>/dev/null | grep -v amdgpu > $OUT/stamps_grouped_1gib.txt
python tools/encode_rate.py > $OUT/encode_rate_100mb.jsonl 2> $OUT/encode.err
HSRANS_DEBUG_STAMPS=1 python tools/encode_phase_probe.py 2>&1 | grep -v amdgpu > $OUT/encode_phases.txt   # per-block phase stamps of k_encode_blocks by blocks in flight
timeout 120 tools/microbench/lone_wave > $OUT/encoder_lone_wave.txt 2>&1                                  # what one wavefront alone on its SIMD pays per instruction
python tools/spread_by_interval.py > $OUT/spread_by_interval.txt 2> $OUT/spread.err                        # mt_ decode against block size and index interval
python tools/shard_projection.py --interval 256 64 16 > $OUT/shard_projection.jsonl 2> $OUT/shard_projection.err   # every rank's GPU side of world 2 / 4 / 8 on this one GPU
HSRANS_SHARD_ONE_LAUNCH=0 python tools/shard_projection.py --interval 64 --worlds 8 --parts 4 --label a-launch-per-sub-run >> $OUT/shard_projection.jsonl 2>> $OUT/shard_projection.err
python tools/stamps_dealt.py --size 134217728 --interval 16 --calibrate 2>/dev/null | grep -v "amdgpu\|encode stamps" > $OUT/stamps_dealt_128mib.txt      # a rank of 8: per-class timeline of the dealt launch
python tools/stamps_dealt.py --size 100000000 --interval 32 --calibrate 2>/dev/null | grep -v "amdgpu\|encode stamps" > $OUT/stamps_dealt_100mb.txt
python tools/stamps_grouped.py --size 134217728 --interval 256 2>/dev/null | grep -v amdgpu > $OUT/grouped_stamps.txt                                     # the same rank on the grouped launch: older / younger workgroup of a CU
python tools/stamps_grouped.py --size 100000000 --block 65536 --interval 64 2>/dev/null | grep -v amdgpu >> $OUT/grouped_stamps.txt
tools/microbench/wait_value > $OUT/wait_value.txt 2>&1                                                                                                        # hipStreamWaitValue32 by memory kind
python tools/cold_cache.py > $OUT/cold_cache.jsonl 2> $OUT/cold.err
timeout 300 tools/microbench/stream_pattern > $OUT/stream_pattern.txt 2>&1
python tools/stamps.py --index wave 2>/dev/null | grep -v amdgpu > $OUT/stamps_wave_warm.txt
python tools/stamps.py --index wave --cold 4 2>/dev/null | grep -v amdgpu > $OUT/stamps_wave_cold.txt
tail -c 400 $OUT/bench_line.json; echo; tail -3 $OUT/harness_100mb_11bit.txt
