#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the C3 workload (LogReg n=1e6, D=512, full-rank, 128 samples), top kernels only
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/c3p
rocprofv3 --kernel-trace --stats -d /tmp/c3p -o run -- python $REPO/bench.py --no-cpu-baseline --concurrent 1 --workload c3 --steps 20 --warmup 5 > /tmp/c3p.log 2>&1
python $REPO/tools/rocpd_stats.py $(find /tmp/c3p -name '*.db' | head -1) | cut -c1-150 | sed -n 1,9p
tail -1 /tmp/c3p.log | cut -c1-200
