set -u
TAG=r03_j; REPO=$(pwd); OUT=$REPO/gpurun_out/summ; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-also --concurrent 1"
for w in c2 c5; do
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o run -- $BENCH --workload $w --steps 400 --warmup 20 > /tmp/prof_$w.log 2>&1
  db=$(find /tmp/prof_$w -name '*.db' | head -1)
  { echo "# $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --concurrent 1 --workload $w --steps 400 --warmup 20"; echo; python $REPO/tools/rocpd_stats.py $db; } > $OUT/${TAG}_${w}_kernel_stats.md
done
cd $REPO
for w in c2 c5; do python bench.py --workload $w 2>/dev/null | tail -1 > $OUT/${TAG}_bench_$w.json; done
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_driver_protocol.json
