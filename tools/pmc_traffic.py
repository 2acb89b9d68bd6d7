#!/usr/bin/env python
"""profiles/pmc_traffic.json from two rocprofv3 passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs)."""
import json, sqlite3, sys
def per_kernel(db, counter):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    g = lambda p: [t for t in tabs if t.startswith(p)][0]
    pmc, info, disp, sym = g("rocpd_pmc_event"), g("rocpd_info_pmc"), g("rocpd_kernel_dispatch"), g("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, d.id, sum(p.value) from {pmc} p join {info} i on p.pmc_id=i.id join {disp} d on "
         f"p.event_id=d.event_id join {sym} s on d.kernel_id=s.id where i.name='{counter}' group by 1,2")
    acc = {}
    for k, _, v in con.execute(q):
        acc.setdefault(k, []).append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}
fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)",
       "unit": "KiB per launch", "kernels": {k: {"fetch_kib": fetch.get(k, 0.0), "write_kib": write.get(k, 0.0)}
                                              for k in sorted(set(fetch) | set(write)) if not k.startswith("__amd")}}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
