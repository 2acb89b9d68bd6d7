#!/bin/bash
# A/B of the dense product's column-pair workgroups (MIVI_FB_PAIR = 4 / 6 ring slots) against the shipped 128 x 128 tiles
mkdir -p gpurun_out
for v in 0 4 6; do
  echo "== FBX_PAIR=$v"; FBX_PAIR=$v python tools/fb_lane_curve.py --dense 16 20 32 48 50 64 2>&1 | tail -6
done
for v in 4 6; do
  echo "== tests FBX_PAIR=$v"; FBX_PAIR=$v timeout 900 python -m pytest tests/test_gpu_each.py tests/test_gpu_batches.py tests/test_gpu_engine_fuzz.py -m gpu -x -q -k "dense or fuzz or Dense" 2>&1 | tail -4
done
for v in 0 4 6; do
  echo "== bench ns_dense FBX_PAIR=$v"; FBX_PAIR=$v python bench.py --no-cpu-baseline --no-also --workload ns_dense 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])"
done
