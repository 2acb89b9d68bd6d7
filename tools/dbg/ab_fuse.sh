#!/bin/bash
# same-box A/B: product + VJP of a step as ONE launch (FBX_FUSE=1, k_fb_pv) against the two launches
line() { timeout 300 python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])"; }
FBX_FUSE=1 timeout 600 python -m pytest tests/test_gpu_each.py tests/test_gpu_batches.py -m gpu -x -q -k "not dense and not stl and not Dense" 2>&1 | tail -3
for r in 1 2 3; do
  for v in 0 1; do
    echo "FBX_FUSE=$v round $r: driver $(FBX_FUSE=$v line --steps 20 --warmup 5) | ns $(FBX_FUSE=$v line)"
  done
done
