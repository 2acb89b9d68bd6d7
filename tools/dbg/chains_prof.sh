#!/bin/bash
# developer: rocprofv3 timeline window of tools/dbg/chains.py (argv: first dispatch (negative: from the end), count)
export TMPDIR=/tmp; R=$(pwd); cd /tmp; rm -rf /tmp/pc
rocprofv3 --kernel-trace -d /tmp/pc -o run -- python $R/tools/dbg/chains.py > /tmp/pc.log 2>&1
grep "chunk\|isolated" /tmp/pc.log
python $R/tools/dbg/seq_window.py $(find /tmp/pc -name "*.db" | head -1) ${1:-600} ${2:-40}
