cd advancedvi.jl_amd; cp libmivi.so libmivi_new.so
for r in 1 2 3; do
  cp libmivi_head.so libmivi.so; echo -n "head: "; python ../tools/fb_lane_curve.py 20 50 2>/dev/null | grep -v amdgpu | awk '{printf "%s %s %s %s %s %s %s %s | ", $3,$4,$5,$6,$7,$8,$9,$10}'; echo
  cp libmivi_new.so libmivi.so;  echo -n "new : "; python ../tools/fb_lane_curve.py 20 50 2>/dev/null | grep -v amdgpu | awk '{printf "%s %s %s %s %s %s %s %s | ", $3,$4,$5,$6,$7,$8,$9,$10}'; echo
done
cp libmivi_new.so libmivi.so
