#!/bin/bash
# TIMING ONLY (wrong values): the dense product without its epilogue's re-read of R from memory -- how much would taking l from registers return?
for v in ship nor ship nor; do cp tools/bin/libmivi_$v.so advancedvi.jl_amd/libmivi.so; echo "== $v"; python tools/fb_lane_curve.py --dense 16 20 50 64 2>&1 | tail -4 | sed 's/|.*//'; done
cp tools/bin/libmivi_ship.so advancedvi.jl_amd/libmivi.so
