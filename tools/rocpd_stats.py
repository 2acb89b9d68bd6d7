#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table:
calls, total/avg/min/max duration (ns), % of GPU kernel time.  Usage: rocpd_stats.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in con.execute(f"pragma table_info({disp})")]
    scol = [r[1] for r in con.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
    q = (f"select s.{name_col}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc")
    rows = list(con.execute(q))
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total_ns | avg_ns | min_ns | max_ns | pct |", "|---|---|---|---|---|---|---|"]
    for n, c, t, a, mn, mx in rows:
        lines.append(f"| `{n[:110]}` | {c} | {t} | {a:.0f} | {mn} | {mx} | {100.0 * t / tot:.2f} |")
    # the batch engine's kernels per GRID: the estimates (lanes) a dispatch carried are derived from its grid (tools/fb_grid.py)
    try:
        import os
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import fb_grid
        gx, gy, wx, wy = fb_grid.disp_cols(con, disp)
        d_, M_ = int(os.environ.get("FB_D", "1024")), int(os.environ.get("FB_M", "256"))
        if gx and wx:
            q2 = (f"select s.{name_col}, d.{gx} / d.{wx}, d.{gy} / d.{wy}, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                  f"from {disp} d join {sym} s on d.kernel_id = s.id where s.{name_col} like '%k_fb_%' group by 1, 2, 3 order by 1, 2, 3")
            rows2 = list(con.execute(q2))
            prod_l = {fb_grid.lanes_of(n, x, y, d_, M_) for n, x, y, *_ in rows2 if "k_fb_prod" in n}
            if rows2:
                lines += ["", "Batch-engine kernels by grid (lanes = estimates per launch, from the grid):", "",
                          "| kernel | workgroups | lanes | calls | avg_ns | min_ns | max_ns |", "|---|---|---|---|---|---|---|"]
                for n, x, y, c, a, mn, mx in rows2:
                    L = fb_grid.lanes_of(n, x, y, d_, M_)
                    if "k_fb_eps" in n and L is not None and (L - fb_grid.eps_riders(d_, M_)) in prod_l:
                        L -= fb_grid.eps_riders(d_, M_)
                    lines.append(f"| `{n[:110]}` | {x}x{y} | {L} | {c} | {a:.0f} | {mn} | {mx} |")
    except Exception as e:   # noqa: BLE001
        lines += ["", f"(per-grid table unavailable: {e})"]
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(f"# rocprofv3 --kernel-trace --stats summary of {db}\n\n" + txt + "\n")


if __name__ == "__main__":
    main()
