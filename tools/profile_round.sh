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
  steps=200; [ $w = c3 ] && steps=20; [ $w = ns_stl ] && steps=100
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o run -- $BENCH --workload $w --steps $steps --warmup 20 > /tmp/prof_$w.log 2>&1
  db=$(find /tmp/prof_$w -name '*.db' | head -1)
  { echo "# $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --concurrent 1 --workload $w --steps $steps --warmup 20"; echo;
    python $REPO/tools/rocpd_stats.py $db; } > $OUT/${TAG}_${w}_kernel_stats.md
done
# PMC passes (own runs, kernel-trace only), NS default bench (which also runs C2 as `also`)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o run -- python $REPO/bench.py --steps 200 --warmup 100 --no-cpu-baseline --concurrent 1 > /tmp/pmc_$c.log 2>&1
  db=$(find /tmp/pmc_$c -name '*.db' | head -1)
  { echo "# $TAG: rocprofv3 --kernel-trace --pmc $c -- python bench.py --steps 200 --warmup 100 --no-cpu-baseline --concurrent 1"; echo;
    python $REPO/tools/rocpd_pmc.py $db; } > $OUT/${TAG}_ns_pmc_${c}.md
done
cd $REPO
python tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1) \
  "$TAG: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --steps 200 --warmup 100 --no-cpu-baseline --concurrent 1 (two separate passes)" > /dev/null
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
# un-profiled bench lines
for w in ns c2 ns_dense ns_stl c3 c5; do
  python bench.py --workload $w $( [ $w = c3 ] && echo "--steps 100 --warmup 10" ) 2>/dev/null | tail -1 > $OUT/${TAG}_bench_$w.json
done
ls -la $OUT
