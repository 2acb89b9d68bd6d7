# developer: print the start offsets / durations of the kernels of one pipelined batch from a rocprofv3 rocpd database (argv[1]); argv[2] = first dispatch to print
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scol = [r[1] for r in con.execute(f"pragma table_info({sym})")]
nc = "kernel_name" if "kernel_name" in scol else "display_name"
rows = list(con.execute(f"select s.{nc}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"))
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) - 140
t0 = rows[k0][1]
short = lambda n: "XCHG" if "p2p_exchange" in n else "hand" if "handover" in n else "prodQ" if "prod32q" in n else "prod" if "prod32" in n else "vjpS" if "vjp32s" in n else "vjp " if "vjp32" in n else n[:12]
for n, s, e in rows[k0:k0 + 75]:
    print("%-5s start %8.1f  dur %7.1f" % (short(n), (s - t0) / 1e3, (e - s) / 1e3))
