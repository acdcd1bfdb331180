#!/bin/bash
# Run ON THE GPU BOX: traversal occupancy sweep -- LDS stack depth variants (variants/libmipt_st<N>.so) x stage grid sizes.
cp gltf_renderer_amd/libmipt.so /tmp/orig.so
run() { timeout -k 5 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --stage-blocks $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 blocks $2:', d['value'], d['config']['ms_per_1spp_frame'])"; }
for b in 1536 2048; do run base $b; done
for v in st16 st12; do
  cp variants/libmipt_$v.so gltf_renderer_amd/libmipt.so
  for b in 1536 2048 2560; do run $v $b; done
done
cp /tmp/orig.so gltf_renderer_amd/libmipt.so
