#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of tools/stein_bench.py (mivi_gauss_expected_grad_hess) -> gpurun_out/summ/
set -u
TAG=${1:-r01_x}
REPO=$(pwd)
OUT=$REPO/gpurun_out/summ
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_stein
rocprofv3 --kernel-trace --stats -d /tmp/prof_stein -o run -- python $REPO/tools/stein_bench.py ns > /tmp/prof_stein.log 2>&1
db=$(find /tmp/prof_stein -name '*.db' | head -1)
{ echo "# $TAG: rocprofv3 --kernel-trace --stats -- python tools/stein_bench.py ns   (d=1024, n=256, diagonal-Gaussian target, f32; 520 calls of mivi_gauss_expected_grad_hess + 500 ELBO estimates)"; echo;
  python $REPO/tools/rocpd_stats.py $db; echo; echo '```'; grep "grad+hess" /tmp/prof_stein.log; echo '```'; } > $OUT/${TAG}_stein_kernel_stats.md
cat $OUT/${TAG}_stein_kernel_stats.md
