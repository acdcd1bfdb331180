#!/bin/bash
# Run ON THE GPU BOX: tools/builder_probe.py <scene> with every variants/libmipt_*.so, selected through MIPT_LIBRARY (the tree's product
# library is never overwritten).
for f in variants/libmipt_*.so; do
  echo "== $f"; MIPT_LIBRARY=$PWD/$f timeout -k 5 200 python tools/builder_probe.py ${1:-sponza} 2>&1 | grep "reins \|differ"
done
