#!/bin/bash
# Run ON THE GPU BOX: tools/builder_probe.py <scene> with every variants/libmipt_*.so swapped in.
cp gltf_renderer_amd/libmipt.so /tmp/orig_pv.so
for f in variants/libmipt_*.so; do
  cp "$f" gltf_renderer_amd/libmipt.so
  echo "== $f"; timeout -k 5 200 python tools/builder_probe.py ${1:-sponza} 2>&1 | grep "reins \|differ"
done
cp /tmp/orig_pv.so gltf_renderer_amd/libmipt.so
