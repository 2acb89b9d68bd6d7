REPO=$(pwd); export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/c5p
rocprofv3 --kernel-trace --stats -d /tmp/c5p -o run -- python $REPO/bench.py --workload c5 --steps 300 --warmup 20 --no-cpu-baseline --concurrent 1 > /tmp/c5p.log 2>&1
python $REPO/tools/rocpd_stats.py $(find /tmp/c5p -name '*.db' | head -1) | cut -c1-150 | head -8
