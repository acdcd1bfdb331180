#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + two PMC passes of the default bench command,
# then summarise into gpurun_out/prof_summary (copied to profiles/ by hand after review).
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ARGS=${@:---steps 10 --warmup 2 --no-cpu-baseline}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1 || { tail -20 $OUT/bench_trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS --no-roofline > $OUT/bench_fetch.log 2>&1 || { tail -20 $OUT/bench_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS --no-roofline > $OUT/bench_write.log 2>&1 || { tail -20 $OUT/bench_write.log; exit 1; }
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py $ARGS --no-roofline > $OUT/bench_l2.log 2>&1 || { tail -20 $OUT/bench_l2.log; }
python3 tools/summarize_profile.py $OUT $TAG
