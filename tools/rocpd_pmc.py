#!/usr/bin/env python
"""Per-kernel PMC summary from a rocprofv3 rocpd database: for each counter the SUM over all hardware
instances of one dispatch, averaged over dispatches.  Usage: rocpd_pmc.py results.db [out.md]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    con = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    g = lambda p: [t for t in tabs if t.startswith(p)][0]
    pmc, info, disp, sym = g("rocpd_pmc_event"), g("rocpd_info_pmc"), g("rocpd_kernel_dispatch"), g("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, i.name, d.id, sum(p.value), count(*) from {pmc} p join {info} i on p.pmc_id=i.id "
         f"join {disp} d on p.event_id=d.event_id join {sym} s on d.kernel_id=s.id group by 1,2,3")
    acc = defaultdict(list)
    inst = {}
    for k, n, _, v, c in con.execute(q):
        acc[(k, n)].append(v)
        inst[(k, n)] = c
    dur = {k: (a, c) for k, a, c in con.execute(
        f"select s.kernel_name, avg(d.end-d.start), count(*) from {disp} d join {sym} s on d.kernel_id=s.id group by 1")}
    lines = ["| kernel | counter | per-dispatch sum (avg) | instances | dispatches | avg_ns |", "|---|---|---|---|---|---|"]
    for (k, n), vs in sorted(acc.items()):
        if k.startswith("__amd_rocclr"):
            continue
        lines.append(f"| `{k[:70]}` | {n} | {sum(vs) / len(vs):.1f} | {inst[(k, n)]} | {len(vs)} | {dur[k][0]:.0f} |")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(f"# rocprofv3 --pmc summary of {sys.argv[1]}\n\n{txt}\n")


if __name__ == "__main__":
    main()
