#!/bin/bash
# Regenerates every measurement the round's documents quote (run through gpurun from the repo root):
#   bash tools/regen_round.sh r04
# then, back in the container:  python tools/collect_round.py r04   (condenses gpurun_out/prof_<tag>*/ and copies gpurun_out/<tag>/* into profiles/)
# ONE regeneration per round: the numbers are whatever this box gives.
TAG=${1:-r06}
bash tools/profile.sh $TAG                                                                   # headline, 4 pairs rotated: trace + counters
ONLY_TRACE=1 bash tools/profile.sh ${TAG}_warm --pairs 1                                     # one pair replayed: trace
bash tools/profile.sh ${TAG}_b15 --bits 15                                                   # BASELINE config 3: 15-bit histogram (k_decode_dual<4>): trace + counters
ONLY_TRACE=1 bash tools/profile.sh ${TAG}_b14 --bits 14
bash tools/profile.sh ${TAG}_batch --one-launch                                              # the 4 streams decoded by ONE launch (k_decode_batch<3>): trace + counters
bash tools/profile.sh ${TAG}_s32 --states 32                                                 # rANS32x32: trace + counters (r04's were empty)
STEPS=20 bash tools/profile.sh ${TAG}_1gib --size 1073741824 --pairs 1                       # BASELINE config 2 at 2^30 bytes: trace + counters
STEPS=10 bash tools/profile.sh ${TAG}_sharded --workload sharded                             # BASELINE config 4 shape (mt_, 1 GiB, 256 KiB blocks): trace + counters
# the 100 MB-class mt_ launches (there were no counters for them): 64 KiB blocks = k_decode_grouped, 256 KiB blocks = k_decode_dealt
STEPS=20 bash tools/profile.sh ${TAG}_grouped_100mb --workload sharded --size 100000000 --block 65536 --interval 64
STEPS=20 bash tools/profile.sh ${TAG}_dealt_100mb --workload sharded --size 100000000 --block 262144 --interval 32
# the mt_ GPU encoder's kernels (100 MB, 64 KiB blocks): trace only
mkdir -p gpurun_out/prof_${TAG}_enc; TMPDIR=/tmp timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_enc/trace -- python3 tools/debug/enc_once.py > gpurun_out/prof_${TAG}_enc/trace.log 2>&1
mkdir -p gpurun_out/$TAG; cp $(find gpurun_out/prof_${TAG}_enc -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/encode_kernel_stats.csv
# (on the box: the counter runs above condensed into profiles/ BEFORE bench.py runs, so that its line finds counters of these very kernels
# and reports roofline.traffic / issue_bound / lds_bound from the same box; back in the container collect_round.py writes the same files again)
python tools/collect_round.py $TAG > /dev/null 2>&1
bash tools/round_measure.sh $TAG
