#!/bin/bash
# same-box A/B: the draws in both orientations (e0) against the draws once + transposing VJP reads (e1)
line() { python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])"; }
for r in 1 2 3; do
  for v in e0 e1; do
    cp tools/bin/libmivi_$v.so advancedvi.jl_amd/libmivi.so
    echo "$v round $r: driver $(line --steps 20 --warmup 5) | ns $(line) | dense $(line --workload ns_dense) | stl $(line --workload ns_stl)"
  done
done
for v in e0 e1; do
  cp tools/bin/libmivi_$v.so advancedvi.jl_amd/libmivi.so
  echo "== $v"; python tools/fb_lane_curve.py 20 50 2>&1 | tail -2
done
