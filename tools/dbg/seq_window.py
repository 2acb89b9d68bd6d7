# developer: a window of consecutive kernel dispatches (start offset, duration, queue) from a rocprofv3 rocpd database: argv[1] = db, argv[2] = first dispatch, argv[3] = count
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
dcol = [r[1] for r in con.execute(f"pragma table_info({disp})")]
scol = [r[1] for r in con.execute(f"pragma table_info({sym})")]
nc = "kernel_name" if "kernel_name" in scol else "display_name"
qc = "queue_id" if "queue_id" in dcol else ("stream_id" if "stream_id" in dcol else "0")
rows = list(con.execute(f"select s.{nc}, d.start, d.end, d.{qc} from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"))
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if k0 < 0: k0 = len(rows) + k0
t0 = rows[k0][1]
short = lambda s: "prodQ" if "prod32q" in s else "vjpS" if "vjp32s" in s else "prod" if "prod32" in s else "vjp" if "vjp32" in s else s[:14]
for nm, s, e, q in rows[k0:k0 + n]:
    print("%-6s q%-3s start %8.1f  end %8.1f  dur %6.1f" % (short(nm), q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
