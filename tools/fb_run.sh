#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): batch-engine parity, timing and a kernel-trace summary.  usage: tools/fb_run.sh <tag> [parity] [time] [prof]
TAG=${1:-x}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    parity) python tools/fb_check.py parity > $OUT/fb_${TAG}_parity.log 2>&1; tail -15 $OUT/fb_${TAG}_parity.log;;
    time) python tools/fb_check.py time > $OUT/fb_${TAG}_time.log 2>&1; grep timing $OUT/fb_${TAG}_time.log;;
    prof) cd /tmp; rm -rf /tmp/prof_fb; rocprofv3 --kernel-trace --stats -d /tmp/prof_fb -o run -- python $REPO/tools/fb_check.py time > /tmp/prof_fb.log 2>&1
          cd $REPO; python tools/rocpd_stats.py $(find /tmp/prof_fb -name '*.db' | head -1) > $OUT/fb_${TAG}_prof.md; head -12 $OUT/fb_${TAG}_prof.md;;
  esac
done
