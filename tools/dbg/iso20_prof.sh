#!/bin/bash
# developer: rocprofv3 timeline of isolated 20-estimate calls (tools/dbg/iso20.py); prints the last ~30 dispatches
export TMPDIR=/tmp; R=$(pwd); cd /tmp; rm -rf /tmp/pi
rocprofv3 --kernel-trace -d /tmp/pi -o run -- python $R/tools/dbg/iso20.py > /tmp/pi.log 2>&1
grep "call us" /tmp/pi.log
DB=$(find /tmp/pi -name "*.db" | head -1)
N=$(python - <<PY
import sqlite3
con=sqlite3.connect("$DB")
t=[r[0] for r in con.execute("select name from sqlite_master where type='table'") if r[0].startswith("rocpd_kernel_dispatch")][0]
print(con.execute(f"select count(*) from {t}").fetchone()[0]-${1:-26})
PY
)
python $R/tools/dbg/seq_durations.py $DB $N | head -${1:-26}
