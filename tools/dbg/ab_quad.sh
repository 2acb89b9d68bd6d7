#!/bin/bash
# developer: A/B of the lane-batched kernels' switches with tools/dbg/chains.py (long batches back to back, isolated 20-estimate calls)
run() { echo "== $*"; env "$@" python tools/dbg/chains.py 2>&1 | grep "chunk\|isolated"; }
for r in 1 2; do
for v in "$@"; do run $v; done
done
