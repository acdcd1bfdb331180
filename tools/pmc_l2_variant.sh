#!/bin/bash
# Run ON THE GPU BOX: L2 (TCC) hit / miss of the bench command with variants/libmipt_<name>.so swapped in, per kernel family.
# usage: tools/pmc_l2_variant.sh <name> [steps]
NAME=$1; STEPS=${2:-6}
export TMPDIR=/tmp
cp gltf_renderer_amd/libmipt.so /tmp/orig_$NAME.so
cp variants/libmipt_$NAME.so gltf_renderer_amd/libmipt.so
OUT=gpurun_out/pmc_l2_$NAME
mkdir -p $OUT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -- python3 bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1
cp /tmp/orig_$NAME.so gltf_renderer_amd/libmipt.so
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt; cat $OUT/summary.txt; grep "^{" $OUT/bench.log | tail -1 | cut -c1-200
