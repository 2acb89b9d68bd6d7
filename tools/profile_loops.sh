# On the GPU box: kernel stats of the launch-free optimisation loops and of the logistic-regression loops + the two bench lines (default, driver protocol) -> gpurun_out/summ/r04_e_* (copied to profiles/)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/summ; mkdir -p $OUT; export TMPDIR=/tmp; TAG=r04_e
cd /tmp; rm -rf /tmp/prof_loops
rocprofv3 --kernel-trace --stats -d /tmp/prof_loops -o run -- python $REPO/tools/loop_rules_bench.py 0,1024,256 1,1024,8 1,10,1 > /tmp/prof_loops.log 2>&1
db=$(find /tmp/prof_loops -name '*.db' | head -1)
{ echo "# $TAG: rocprofv3 --kernel-trace --stats -- python tools/loop_rules_bench.py 0,1024,256 1,1024,8 1,10,1 (mivi_optimize_loop, 4 x 500 steps per launch-free kernel call)"; echo;
  python $REPO/tools/rocpd_stats.py $db; echo; echo '```'; grep "^family" /tmp/prof_loops.log; echo '```'; } > $OUT/${TAG}_loops_kernel_stats.md
rm -rf /tmp/prof_lr
rocprofv3 --kernel-trace --stats -d /tmp/prof_lr -o run -- python $REPO/tools/logreg_loop_bench.py 1,208,61,1 0,208,61,1 0,1000,33,16 > /tmp/prof_lr.log 2>&1
db=$(find /tmp/prof_lr -name '*.db' | head -1)
{ echo "# $TAG: rocprofv3 --kernel-trace --stats -- python tools/logreg_loop_bench.py 1,208,61,1 0,208,61,1 0,1000,33,16 (README-sized: k_lr_small_loop; BASELINE configs[0]: the hipGraph of launches)"; echo;
  python $REPO/tools/rocpd_stats.py $db | head -24; echo; echo '```'; grep "^family" /tmp/prof_lr.log; echo '```'; } > $OUT/${TAG}_logreg_loops_kernel_stats.md
cd $REPO
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_protocol.json
python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_ns_default.json
