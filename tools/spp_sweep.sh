#!/bin/bash
# Run ON THE GPU BOX: stage times per sample of the bench scene by samples per launch (what a small launch loses against a batch).
mkdir -p gpurun_out
for S in "$@"; do
  timeout -k 5 200 python bench.py --spp $S --steps 12 --warmup 3 --no-cpu-baseline --no-pmc-live > gpurun_out/spp_$S.log 2>&1 || { echo "spp $S FAILED"; continue; }
  grep "^{" gpurun_out/spp_$S.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; st = d['roofline']['stages']; k = d['roofline']['kernels']
S = float($S)
print('spp %2d: %7.1f Mrays/s  %.3f ms per sample | per sample: trace %.3f  traverse %.3f  shade %.3f  gen+resolve %.3f' % ($S, d['value'], c['ms_per_1spp_frame'],
      k['k_wf_trace']['ms_per_launch'] / S, k['k_wf_traverse']['ms_per_launch'] / S, st['shade']['ms_per_launch'] / S, st['generate+resolve']['ms_per_launch'] / S))"
done
