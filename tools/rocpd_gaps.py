#!/usr/bin/env python
"""Kernel-to-kernel gaps of a rocprofv3 kernel trace (rocpd sqlite): for every consecutive pair of dispatches (by start time) the idle time
between the first one's end and the second one's start, averaged per (previous kernel -> next kernel) pair; and the busy / idle split of the
densest window.  Usage: rocpd_gaps.py results.db [name-substring the window is restricted to, default k_fb_]"""
import sqlite3
import sys


def short(n):
    n = n.split("(")[0]
    for p in ("_ZN4mivi", "_Z"):
        if n.startswith(p):
            n = n[len(p):]
    return n[:40]


def main():
    con = sqlite3.connect(sys.argv[1])
    sub = sys.argv[2] if len(sys.argv) > 2 else "k_fb_"
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(con.execute(f"select s.kernel_name, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"))
    pairs = {}
    busy = idle = 0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        if sub not in n0 or sub not in n1:
            continue
        g = s1 - e0
        if g > 50000:     # a host-side pause (a synchronize between timed calls), not a launch gap
            continue
        pairs.setdefault((short(n0), short(n1)), []).append(g)
        busy += e0 - s0
        idle += max(g, 0)
    print("| previous -> next | pairs | avg gap ns | min | median | max |")
    print("|---|---|---|---|---|---|")
    for (a, b), gs in sorted(pairs.items(), key=lambda kv: -len(kv[1])):
        gs.sort()
        print(f"| `{a}` -> `{b}` | {len(gs)} | {sum(gs) / len(gs):.0f} | {gs[0]} | {gs[len(gs) // 2]} | {gs[-1]} |")
    tot = busy + idle
    if tot:
        print(f"\nbusy {busy / 1e3:.0f} us, idle between kernels {idle / 1e3:.0f} us ({100.0 * idle / tot:.1f} % of the chain)")


if __name__ == "__main__":
    main()
