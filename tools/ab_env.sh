#!/bin/bash
# Run ON THE GPU BOX: A/B one environment variable of the library through bench.py (the in-tree library both times).
# usage: tools/ab_env.sh VAR value1 value2 [value1 value2 ...]
VAR=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  env $VAR=$v timeout -k 5 200 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-pmc-live > gpurun_out/abenv_${VAR}_$v.log 2>&1 || { echo "$VAR=$v FAILED"; tail -3 gpurun_out/abenv_${VAR}_$v.log; continue; }
  grep "^{" gpurun_out/abenv_${VAR}_$v.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; st = d.get('roofline', {}).get('stages', {})
print('%-18s %8.1f Mrays/s  %.3f ms/1spp  single %.3f ms  trav %.2f  shade %.2f  nodes/ray %s' % ('$VAR=$v', d['value'], c['ms_per_1spp_frame'], c.get('ms_single_sample_launch', 0),
      st.get('traversal', {}).get('ms_per_launch', 0), st.get('shade', {}).get('ms_per_launch', 0), d.get('roofline', {}).get('nodes_per_ray')))"
done
