#!/bin/bash
# compile-time work-skipping knock-outs of k_fb_vjp (-DFB_VK: 16 no DMA, 32 no MFMA, 64 no LDS fragment reads, 128 no epilogue); shipped schedule otherwise
cp advancedvi.jl_amd/libmivi.so /tmp/ship.so
for k in 0 16 32 64 128 96 112 240 224 0; do
  cp tools/bin/libmivi_vk$k.so advancedvi.jl_amd/libmivi.so
  echo "== FB_VK=$k"; python tools/fb_lane_curve.py 20 50 2>&1 | tail -2 | sed 's/|.*//'
done
cp /tmp/ship.so advancedvi.jl_amd/libmivi.so
