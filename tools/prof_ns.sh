#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel-trace of the north-star bench line -> gpurun_out/prof_ns_stats.md
REPO=$(pwd)
export TMPDIR=/tmp
W=${1:-ns}
cd /tmp
rm -rf /tmp/prof_$W
rocprofv3 --kernel-trace --stats -d /tmp/prof_$W -o run -- python $REPO/bench.py --no-cpu-baseline --no-also --concurrent 1 --workload $W --steps 400 --warmup 100 > /tmp/prof_$W.log 2>&1
db=$(find /tmp/prof_$W -name '*.db' | head -1)
python $REPO/tools/rocpd_stats.py $db > $REPO/gpurun_out/prof_${W}_stats.md
tail -1 /tmp/prof_$W.log > $REPO/gpurun_out/prof_${W}_bench.json
cat $REPO/gpurun_out/prof_${W}_stats.md
