#!/bin/bash
# Regenerates every measurement the round's documents quote (run through gpurun from the repo root):
#   bash tools/regen_round.sh r03
# then, back in the container:  python tools/collect_round.py r03   (condenses gpurun_out/prof_<tag>*/ and copies gpurun_out/<tag>/* into profiles/)
TAG=${1:-r03}
bash tools/profile.sh $TAG                                                                   # headline, 4 pairs rotated: trace + counters
bash tools/profile.sh ${TAG}_warm --pairs 1                                                  # one pair replayed: trace + counters
ONLY_TRACE=1 STEPS=20 bash tools/profile.sh ${TAG}_1gib --size 1073741824 --pairs 1          # BASELINE config 2 at 2^30 bytes
ONLY_TRACE=1 STEPS=10 bash tools/profile.sh ${TAG}_7gib --size 7516192768 --pairs 1          # 7 GiB raw stream in one launch
STEPS=10 bash tools/profile.sh ${TAG}_sharded --workload sharded                             # BASELINE config 4 shape (mt_, 1 GiB, 256 KiB blocks): trace + counters
bash tools/round_measure.sh $TAG
timeout 300 tools/microbench/rw_pattern > gpurun_out/$TAG/rw_pattern.txt 2>&1
