#!/usr/bin/env python
"""profiles/pmc_calibration.json: what rocprofv3's FETCH_SIZE / WRITE_SIZE report for a KNOWN byte count in this library's access
patterns (tools/ubench_fetchcal.hip).  usage: pmc_calibrate.py fetch.db write.db bytes_per_launch [source]"""
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
nbytes = float(sys.argv[3])
def pick(tab, key):
    for k, v in tab.items():
        if key in k:
            return v * 1024.0
    return None
out = {"source": sys.argv[4] if len(sys.argv) > 4 else "tools/ubench_fetchcal.exe under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)",
       "bytes_per_launch": nbytes, "patterns": {}}
for name, key, tab in (("lds16", "k_cal_lds16", fetch), ("ld16", "k_cal_ld16", fetch), ("ld4", "k_cal_ld4", fetch), ("st16_wt", "k_cal_st16", write)):
    c = pick(tab, key)
    out["patterns"][name] = None if not c else dict(counter_bytes=c, true_over_counter=nbytes / c)
json.dump(out, open("profiles/pmc_calibration.json", "w"), indent=1)
print(json.dumps(out, indent=1))
