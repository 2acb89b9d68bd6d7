#!/bin/bash
python tools/fb_lane_curve.py 8 16 20 32 50 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_each.py tests/test_gpu_batches.py tests/test_gpu_engine_fuzz.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('driver', j['value'], j['ms_per_step'], j['roofline']['frac'])"; done
python bench.py --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('ns', j['value'], j['ms_per_step'])"
python bench.py --no-cpu-baseline --no-also --workload ns_dense 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('ns_dense', j['value'], j['ms_per_step'])"
