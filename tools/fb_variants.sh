#!/bin/bash
# Developer: build libmivi variants that differ in kernels_fullrank_batch.hip's compile-time knobs (here, CPU box) into tools/bin/libmivi_<tag>.so;
# on the GPU box: tools/fb_variants.sh run <tag...> copies each over libmivi.so in turn and runs tools/fb_lane_curve.py.
cd "$(dirname "$0")/../advancedvi.jl_amd/csrc"
if [ "$1" = build ]; then
  shift
  mkdir -p ../../tools/bin
  for spec in "$@"; do   # tag:flags
    tag=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc $flags -c kernels_fullrank_batch.hip -o /tmp/fb_$tag.o || exit 1
    objs=$(ls *.o | grep -v kernels_fullrank_batch.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/fb_$tag.o -o ../../tools/bin/libmivi_$tag.so || exit 1
    echo built $tag
  done
else
  shift
  cp ../libmivi.so /tmp/libmivi_keep.so
  for tag in "$@"; do
    cp ../../tools/bin/libmivi_$tag.so ../libmivi.so
    echo "== $tag"
    python ../../tools/fb_lane_curve.py ${LANES:-20 50} 2>&1 | grep -v amdgpu
  done
  cp /tmp/libmivi_keep.so ../libmivi.so
fi
