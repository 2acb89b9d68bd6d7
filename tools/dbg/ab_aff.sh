#!/bin/bash
# same-box A/B: XCD-affine product / draw tables (FBX_AFF bit 0) and write-back plane stores (bit 1)
line() { python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])"; }
for r in 1 2; do
  for v in 0 1 2 3; do
    echo "FBX_AFF=$v round $r: driver $(FBX_AFF=$v line --steps 20 --warmup 5) | ns $(FBX_AFF=$v line) | stl $(FBX_AFF=$v line --workload ns_stl)"
  done
done
for v in 0 3; do echo "== FBX_AFF=$v"; FBX_AFF=$v python tools/fb_lane_curve.py 16 20 24 50 2>&1 | tail -4 | sed 's/|.*//'; done
FBX_AFF=3 timeout 900 python -m pytest tests/test_gpu_each.py tests/test_gpu_batches.py tests/test_gpu_engine_fuzz.py tests/test_gpu_objective_engine.py -m gpu -x -q 2>&1 | tail -2
