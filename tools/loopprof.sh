#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of tools/loop_bench.py (device-resident Adam loops) -> gpurun_out/summ/<tag>_loop_kernel_stats.md
set -u
TAG=${1:-r0x}
REPO=$(pwd)
OUT=$REPO/gpurun_out/summ
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_loop
rocprofv3 --kernel-trace --stats -d /tmp/prof_loop -o run -- python $REPO/tools/loop_bench.py > /tmp/prof_loop.log 2>&1
db=$(find /tmp/prof_loop -name '*.db' | head -1)
{ echo "# $TAG: rocprofv3 --kernel-trace --stats -- python tools/loop_bench.py   (mivi_optimize_steps: Adam + ClipScale, 11 x 1000 steps per problem)"; echo;
  python $REPO/tools/rocpd_stats.py $db; echo; echo '```'; grep "steps/s" /tmp/prof_loop.log; echo '```'; } > $OUT/${TAG}_loop_kernel_stats.md
cat $OUT/${TAG}_loop_kernel_stats.md
