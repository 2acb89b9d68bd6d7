#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats per workload + the two PMC traffic passes,
# summarised on the box; only the small summaries land in gpurun_out/summ/ (copy them to profiles/<tag>_*).
# usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r01_x}
REPO=$(pwd)
OUT=$REPO/gpurun_out/summ
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-also --concurrent 1"
for w in ns c2 ns_stl ns_dense c3 c5; do
  steps=400; [ $w = c3 ] && steps=20; [ $w = ns_stl ] && steps=100
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o run -- $BENCH --workload $w --steps $steps --warmup 20 > /tmp/prof_$w.log 2>&1
  db=$(find /tmp/prof_$w -name '*.db' | head -1)
  { echo "# $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --concurrent 1 --workload $w --steps $steps --warmup 20"; echo;
    python $REPO/tools/rocpd_stats.py $db; } > $OUT/${TAG}_${w}_kernel_stats.md
done
# the driver's own protocol (--steps 20 --warmup 5: one 20-lane step per call): the in-chain averages of the scored shape
rm -rf /tmp/prof_ns20
rocprofv3 --kernel-trace --stats -d /tmp/prof_ns20 -o run -- $BENCH --workload ns --steps 20 --warmup 5 > /tmp/prof_ns20.log 2>&1
db=$(find /tmp/prof_ns20 -name '*.db' | head -1)
{ echo "# $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-also --concurrent 1 --workload ns --steps 20 --warmup 5 (the driver's protocol)"; echo;
  python $REPO/tools/rocpd_stats.py $db; } > $OUT/${TAG}_ns20_kernel_stats.md
{ echo "# $TAG: idle time between consecutive batch-engine dispatches of the same trace (tools/rocpd_gaps.py)"; echo; python $REPO/tools/rocpd_gaps.py $db; } > $OUT/${TAG}_ns20_gaps.md
# counter calibration on this library's access patterns (tools/ubench_fetchcal.hip), separate passes
[ -x $REPO/tools/bin/ubench_fetchcal.exe ] || { mkdir -p $REPO/tools/bin; /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $REPO/tools/ubench_fetchcal.hip -o $REPO/tools/bin/ubench_fetchcal.exe; }
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/cal_$c -o run -- $REPO/tools/bin/ubench_fetchcal.exe > /tmp/cal_$c.log 2>&1
done
( cd $REPO && python tools/pmc_calibrate.py $(find /tmp/cal_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/cal_WRITE_SIZE -name '*.db' | head -1) 2147483648 \
  "$TAG: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- tools/bin/ubench_fetchcal.exe (2 GiB per launch, separate passes)" > /dev/null; cp profiles/pmc_calibration.json $OUT/ )
# L2-side requests of the north-star kernels (the operand re-fetch that never reaches the memory-side counters), own pass per counter
rocprofv3 --list-avail 2>/dev/null | grep -oE "TCC_REQ_sum|TCC_READ_sum|TCC_HIT_sum|TCC_MISS_sum|TCP_TCC_READ_REQ_sum" | sort -u > /tmp/avail_l2.txt
for c in $(cat /tmp/avail_l2.txt); do
  rm -rf /tmp/l2_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/l2_$c -o run -- python $REPO/bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-also --concurrent 1 > /tmp/l2_$c.log 2>&1
  db=$(find /tmp/l2_$c -name '*.db' | head -1)
  [ -n "$db" ] && { echo "# $TAG: rocprofv3 --kernel-trace --pmc $c -- python bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-also --concurrent 1"; echo;
    python $REPO/tools/rocpd_pmc.py $db; } > $OUT/${TAG}_ns_pmc_${c}.md
done
# PMC passes (own runs, kernel-trace only), NS default bench (which also runs C2 as `also`)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o run -- python $REPO/bench.py --steps 200 --warmup 100 --no-cpu-baseline --concurrent 1 > /tmp/pmc_$c.log 2>&1
  db=$(find /tmp/pmc_$c -name '*.db' | head -1)
  { echo "# $TAG: rocprofv3 --kernel-trace --pmc $c -- python bench.py --steps 200 --warmup 100 --no-cpu-baseline --concurrent 1"; echo;
    python $REPO/tools/rocpd_pmc.py $db; } > $OUT/${TAG}_ns_pmc_${c}.md
done
for c in FETCH_SIZE WRITE_SIZE; do   # ... and at the driver's --steps 20 (20 lanes per launch)
  rm -rf /tmp/pmc20_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc20_$c -o run -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --concurrent 1 > /tmp/pmc20_$c.log 2>&1
done
cd $REPO
python tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1) \
  "$TAG: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --steps 200 --warmup 100 --no-cpu-baseline --concurrent 1 (two separate passes); batch-engine kernels also at --steps 20 --warmup 5 (20 lanes per launch)" \
  $(find /tmp/pmc20_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc20_WRITE_SIZE -name '*.db' | head -1) > /dev/null
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
# the sharded step on one GPU (world 1, exchange forced): kernel stats of the peer-to-peer route
rm -rf /tmp/prof_dist
MIVI_FORCE_DIST=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_dist -o run -- python $REPO/bench.py --no-cpu-baseline --steps 400 --warmup 40 > /tmp/prof_dist.log 2>&1
db=$(find /tmp/prof_dist -name '*.db' | head -1)
{ echo "# $TAG: MIVI_FORCE_DIST=1 rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 400 --warmup 40 (world 1, peer-to-peer exchange forced)"; echo;
  python $REPO/tools/rocpd_stats.py $db; } > $OUT/${TAG}_dist_forced_kernel_stats.md
MIVI_FORCE_DIST=1 python bench.py --no-cpu-baseline --steps 400 --warmup 40 2>/dev/null | tail -1 > $OUT/${TAG}_bench_dist_forced.json
# ... and the peer-to-peer exchange kernel's route (round 6: the automatic route of an engine shape is the batch engine + one all-reduce per step)
MIVI_FORCE_DIST=1 MIVI_DIST_MODE=p2p python bench.py --no-cpu-baseline --no-also --steps 400 --warmup 40 2>/dev/null | tail -1 > $OUT/${TAG}_bench_dist_forced_p2p.json
# the launch-free optimisation loops (round 4): kernel stats of one call each -- default algorithm settings and Adam, the three families of loops
rm -rf /tmp/prof_loops
rocprofv3 --kernel-trace --stats -d /tmp/prof_loops -o run -- python $REPO/tools/loop_rules_bench.py 0,1024,256 1,1024,8 1,10,1 > /tmp/prof_loops.log 2>&1
db=$(find /tmp/prof_loops -name '*.db' | head -1)
{ echo "# $TAG: rocprofv3 --kernel-trace --stats -- python tools/loop_rules_bench.py 0,1024,256 1,1024,8 1,10,1 (mivi_optimize_loop, 4 x 500 steps per launch-free kernel call)"; echo;
  python $REPO/tools/rocpd_stats.py $db; echo; echo '```'; cat /tmp/prof_loops.log | grep -v amdgpu.ids; echo '```'; } > $OUT/${TAG}_loops_kernel_stats.md
# un-profiled bench lines
for w in ns c2 ns_dense ns_stl c3 c5; do
  python bench.py --workload $w $( [ $w = c3 ] && echo "--steps 100 --warmup 10" ) 2>/dev/null | tail -1 > $OUT/${TAG}_bench_$w.json
done
# the driver's own command (its BENCH_rNN.json protocol), un-profiled
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_protocol.json
ls -la $OUT
