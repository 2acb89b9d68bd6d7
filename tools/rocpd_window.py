import sqlite3,sys
db=sys.argv[1]
con=sqlite3.connect(db)
cur=con.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]
ks=[t for t in tabs if 'kernel_symbol' in t][0]
cols=[r[1] for r in cur.execute(f"pragma table_info({kd})")]
print(cols)
rows=cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
# print a window of 30 consecutive dispatches in steady state
n=len(rows); w=rows[n//2:n//2+14]
t0=w[0][1]
for name,st,en in w:
    print(f"{(st-t0)/1e3:9.2f} {(en-st)/1e3:8.2f} us  {name[:60]}")
