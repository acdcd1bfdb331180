#!/bin/bash
# Run ON THE GPU BOX: A/B the tuning builds named on the command line (variants/libmipt_<name>.so, "base" = the in-tree library)
# through bench.py with MIPT_LIBRARY; prints Mrays/s, ms per 1-spp frame, single-sample launch and the stage split.
# usage: tools/ab.sh [-c config] [-a "extra bench args"] base name1 name2 ...
CFG=sponza; EXTRA=""
while getopts "c:a:" o; do case $o in c) CFG=$OPTARG;; a) EXTRA=$OPTARG;; esac; done; shift $((OPTIND - 1))
mkdir -p gpurun_out
for n in "$@"; do
  if [ "$n" = base ]; then unset MIPT_LIBRARY; else export MIPT_LIBRARY=$PWD/variants/libmipt_$n.so; fi
  timeout -k 5 150 python bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline $EXTRA > gpurun_out/ab_$n.log 2>&1 || { echo "$n FAILED"; tail -3 gpurun_out/ab_$n.log; continue; }
  grep "^{" gpurun_out/ab_$n.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; st = d.get('roofline', {}).get('stages', {})
print('%-14s %8.1f Mrays/s  %.3f ms/1spp  single %.3f ms  trav %.2f  shade %.2f  nodes/ray %s' % ('$n', d['value'], c['ms_per_1spp_frame'], c.get('ms_single_sample_launch', 0),
      st.get('traversal', {}).get('ms_per_launch', 0), st.get('shade', {}).get('ms_per_launch', 0), d.get('roofline', {}).get('nodes_per_ray')))"
done
