#!/bin/bash
# Run ON THE GPU BOX: L2 (TCC) hit / miss of the bench command with variants/libmipt_<name>.so, per kernel family.
# The variant is selected with MIPT_LIBRARY (renderer.load_library), inherited by the profiled process: the tree's product library is
# never overwritten, so a timeout or a failing run cannot leave a tuning build behind.
# usage: tools/pmc_l2_variant.sh <name> [steps]
NAME=$1; STEPS=${2:-6}
export TMPDIR=/tmp
export MIPT_LIBRARY=$PWD/variants/libmipt_$NAME.so
[ -f "$MIPT_LIBRARY" ] || { echo "no $MIPT_LIBRARY"; exit 1; }
OUT=gpurun_out/pmc_l2_$NAME
mkdir -p $OUT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -- python3 bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt; cat $OUT/summary.txt; grep "^{" $OUT/bench.log | tail -1 | cut -c1-200
