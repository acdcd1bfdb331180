#!/bin/bash
# Run ON THE GPU BOX: bench every variants/libmipt_*.so (built with different -D tuning macros) in turn.
cp gltf_renderer_amd/libmipt.so /tmp/orig.so
for f in variants/libmipt_*.so; do
  cp "$f" gltf_renderer_amd/libmipt.so
  timeout -k 5 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['config']['ms_per_1spp_frame'])"
done
cp /tmp/orig.so gltf_renderer_amd/libmipt.so
