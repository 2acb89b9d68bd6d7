#!/bin/bash
# developer build only: work-skipping knock-outs of the dense product (1 no DMA, 2 no MFMA, 4 no LDS reads, 8 no barriers)
for k in 0 6 14 1 5 7 2 4; do
  echo "== MIVI_FB_KNOCK=$k"; MIVI_FB_KNOCK=$k python tools/fb_lane_curve.py --dense 16 32 64 2>&1 | tail -3 | sed 's/|.*//'
done
