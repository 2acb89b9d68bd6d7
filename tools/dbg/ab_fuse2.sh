#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
FBX_FUSE=1 timeout 600 python -m pytest tests/test_gpu_each.py -m gpu -x -q -k "test_two_engine_steps_and_the_last_estimate_entry" 2>&1 | grep -E "Error|assert|Mismatch|rel|tol|status" | head -12
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t_f
FBX_FUSE=1 rocprofv3 --kernel-trace --stats -d /tmp/t_f -o run -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --concurrent 1 > /tmp/t.log 2>&1
python $REPO/tools/rocpd_stats.py $(find /tmp/t_f -name '*.db' | head -1) 2>/dev/null | grep -E "k_fb_" | cut -c1-140
