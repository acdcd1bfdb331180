#!/bin/bash
# Build variants/libmipt_<name>.so: the library with extra -D tuning macros on the path-tracing translation units (A/B runs on the
# GPU box with tools/bench_variants.sh).  usage: tools/build_variant.sh <name> [-DFOO=1 ...]
set -e
NAME=$1; shift
cd "$(dirname "$0")/../gltf_renderer_amd/csrc"
mkdir -p ../../variants /tmp/variant_$NAME
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-unused-value -I../../include $@"
for f in pt_wavefront pt_kernel accel; do
  X="-ffp-contract=off"; [ $f != accel ] && X="-fno-slp-vectorize -ffp-contract=off"      # as in the Makefile: no contraction anywhere, the path-tracing kernels also without the SLP vectoriser
  /opt/rocm/bin/hipcc $FLAGS $X -c $f.hip -o /tmp/variant_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../variants/libmipt_$NAME.so /tmp/variant_$NAME/pt_wavefront.o /tmp/variant_$NAME/pt_kernel.o /tmp/variant_$NAME/accel.o \
  mipt_api.o sort_scan.o envmap.o skin_tonemap.o exchange.o host/image_decode.o host/gltf_scene.o -ldl
echo built variants/libmipt_$NAME.so
