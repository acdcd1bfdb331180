#!/bin/bash
# Run ON THE GPU BOX (through gpurun): one plain bench run (the JSON line), one kernel-trace pass and three PMC passes
# of the same bench command, then a summary into gpurun_out/prof_<tag>/ (copied into profiles/ after review).
# PMC passes are separate from the trace pass, as the profiling guide requires (FETCH_SIZE and WRITE_SIZE do not fit
# one pass; no --pmc together with the trace domains).
# usage: tools/profile_gpu.sh <tag> [steps] [warmup] [extra bench args...]
set -o pipefail
TAG=${1:-r01}; STEPS=${2:-10}; WARM=${3:-2}; shift 3 2>/dev/null
EXTRA="$@"
ARGS="--steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline $EXTRA"
FRAMES=$((STEPS + WARM))
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps $STEPS --warmup $WARM $EXTRA > $OUT/bench_plain.log 2>&1 || { tail -20 $OUT/bench_plain.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1 || { tail -20 $OUT/bench_trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.log 2>&1 || { tail -20 $OUT/bench_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.log 2>&1 || { tail -20 $OUT/bench_write.log; exit 1; }
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py $ARGS > $OUT/bench_l2.log 2>&1 || { tail -20 $OUT/bench_l2.log; }
python3 tools/summarize_profile.py $OUT $TAG $FRAMES $FRAMES
echo "plain bench line:"; grep "^{" $OUT/bench_plain.log | tail -1
