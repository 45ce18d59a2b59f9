# Re-fits the run lengths of the uniform-interval (persistent) launches.  Run through gpurun.
O=gpurun_out/tune; mkdir -p $O
t() { name=$1; shift; timeout 400 python tools/tune_weights.py "$@" > $O/$name.txt 2>&1; echo "== $name: $*"; tail -3 $O/$name.txt; }
t s11 --bits 11 --index 32 --iters 6 --var HSRANS_SLOT_WEIGHTS --start 1424,1371,1283,1165,920,768,606,464
t s13 --bits 13 --index 32 --iters 5 --var HSRANS_SLOT_WEIGHTS4 --start 1097,1053,977,873,1098,1053,977,873
t s15 --bits 15 --index 32 --iters 5 --var HSRANS_SLOT_WEIGHTS --start 1192,1159,1120,1072,976,907,829,745
