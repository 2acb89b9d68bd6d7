#!/usr/bin/env python
"""profiles/pmc_traffic.json from two rocprofv3 passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs), with the gfx950 correction
of MI355X_MICROARCH.md applied per kernel: FETCH_SIZE under-reports wide loads, by the factor profiles/pmc_calibration.json measured on
this library's own access patterns (tools/ubench_fetchcal.hip: lds16 = 16 B/lane direct-to-LDS loads, ld16 = 16 B/lane register loads,
ld4 = 4 B/lane loads).  Also: traffic_over_algorithmic for the kernels whose algorithmic bytes SURVEY.md 8d fixes."""
import json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fb_grid
FB_D, FB_M = int(os.environ.get("FB_D", "1024")), int(os.environ.get("FB_M", "256"))
def per_kernel_grid(db, counter):
    """{kernel: {lanes: (avg counter sum per dispatch, dispatches, avg_ns)}} for the batch-engine kernels, lanes from the dispatch's grid."""
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    g = lambda p: [t for t in tabs if t.startswith(p)][0]
    pmc, info, disp, sym = g("rocpd_pmc_event"), g("rocpd_info_pmc"), g("rocpd_kernel_dispatch"), g("rocpd_info_kernel_symbol")
    gx, gy, wx, wy = fb_grid.disp_cols(con, disp)
    if not (gx and wx):
        return {}
    q = (f"select s.kernel_name, d.id, d.{gx} / d.{wx}, d.{gy} / d.{wy}, d.end - d.start, sum(p.value) from {pmc} p join {info} i on p.pmc_id=i.id join {disp} d on "
         f"p.event_id=d.event_id join {sym} s on d.kernel_id=s.id where i.name='{counter}' and s.kernel_name like '%k_fb_%' group by 1,2")
    rows = list(con.execute(q))
    prod_l = {fb_grid.lanes_of(k, x, y, FB_D, FB_M) for k, _, x, y, _, _ in rows if "k_fb_prod" in k}
    acc = {}
    for k, _, x, y, ns, v in rows:
        L = fb_grid.lanes_of(k, x, y, FB_D, FB_M)
        if "k_fb_eps" in k and L is not None and (L - fb_grid.eps_riders(FB_D, FB_M)) in prod_l:
            L -= fb_grid.eps_riders(FB_D, FB_M)
        if L:
            acc.setdefault(k, {}).setdefault(L, []).append((v, ns))
    return {k: {L: (sum(a for a, _ in vs) / len(vs), len(vs), sum(b for _, b in vs) / len(vs)) for L, vs in by.items()} for k, by in acc.items()}
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
# which load flavour dominates a kernel's fetches
PATTERN = (("k_fb_prod", "lds16"), ("k_fb_vjp", "lds16"), ("k_fr_prod32", "lds16"), ("k_fr_prod64", "lds16"), ("k_fr_vjp32", "lds16"), ("k_fr_vjp64", "lds16"), ("k_stl_solve", "lds16"),
           ("k_stl_update", "lds16"), ("k_lr_logits_planes", "lds16"), ("k_lr_xtr_planes", "lds16"), ("k_lr_", "ld16"), ("k_p2p_exchange", "ld16"))
# algorithmic KiB per launch at the north star (d = 1024, n_mc = 256, f32): in + out
d, M = 1024, 256
_prod = (d * (d + 1) // 2 * 4 + d * M * 4 + d * M * 4 + d * M * 4) / 1024.0   # tril(C) + eps in, W + eps(t+1) out
_vjp = (2 * d * M * 4 + d * d * 4) / 1024.0                                  # W + eps in, dense dC out
ALGO = {"k_fr_prod32ILi0": _prod, "k_fr_vjp32ILb0": _vjp,
        # the lane-batched launches of the timed region: four estimates per launch (tril(C) is shared by the lanes)
        "k_fr_prod32mILi0": (d * (d + 1) // 2 * 4 + 4 * 3 * d * M * 4) / 1024.0, "k_fr_vjp32mILb0": 4 * _vjp,
        "k_fr_prod32q": (d * (d + 1) // 2 * 4 + 4 * 3 * d * M * 4) / 1024.0, "k_fr_vjp32s": 4 * _vjp}
# the batch engine's launches (kernels_fullrank_batch.hip): algorithmic bytes per launch of L lanes = SURVEY 8d's f32 figures per estimate
# (tril(C) shared by the lanes); L comes from every dispatch's GRID (tools/fb_grid.py), entries are kept per lane count under "by_lanes"
def algo_fb(kernel, L):
    if "k_fb_prod" in kernel:
        return (d * (d + 1) // 2 * 4 + L * 2 * d * M * 4) / 1024.0   # tril(C) + L x (eps in, W out)
    if "k_fb_vjp" in kernel:
        return L * _vjp                                               # L x (W + eps in, dense dC out)
    if "k_fb_eps" in kernel:
        return L * (d * M * 4) / 1024.0                               # L x eps out (the draw: not part of SURVEY 8d's bytes)
    return None
cal = {}
try:
    cal = json.load(open("profiles/pmc_calibration.json")).get("patterns", {})
except (OSError, ValueError):
    pass
# argv: fetch.db write.db [source] [fetch2.db write2.db ...]: further database pairs (other bench commands, e.g. the driver's --steps 20) only add
# lane counts to the batch-engine kernels' by_lanes tables
fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
fetch_g, write_g = per_kernel_grid(sys.argv[1], "FETCH_SIZE"), per_kernel_grid(sys.argv[2], "WRITE_SIZE")
for i in range(4, len(sys.argv) - 1, 2):
    for dst, db, cn in ((fetch_g, sys.argv[i], "FETCH_SIZE"), (write_g, sys.argv[i + 1], "WRITE_SIZE")):
        for k, by in per_kernel_grid(db, cn).items():
            for L, v in by.items():
                dst.setdefault(k, {}).setdefault(L, v)
kern = {}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("__amd"):
        continue
    pat = next((p for key, p in PATTERN if key in k), "ld4")
    f = (cal.get(pat) or {}).get("true_over_counter") or 1.0
    fw = (cal.get("st16_wt") or {}).get("true_over_counter") or 1.0
    e = {"fetch_kib": fetch.get(k, 0.0) * f, "write_kib": write.get(k, 0.0) * fw, "fetch_kib_raw": fetch.get(k, 0.0), "write_kib_raw": write.get(k, 0.0),
         "load_pattern": pat, "fetch_correction": f, "write_correction": fw}
    for key, alg in ALGO.items():
        if key in k:
            e["algorithmic_kib"] = alg
            e["traffic_over_algorithmic"] = (e["fetch_kib"] + e["write_kib"]) / alg
    if "k_fb_" in k:
        by = {}
        for L in sorted(set(fetch_g.get(k, {})) | set(write_g.get(k, {}))):
            fk, nf_, ns_f = fetch_g.get(k, {}).get(L, (0.0, 0, 0.0))
            wk, nw_, ns_w = write_g.get(k, {}).get(L, (0.0, 0, 0.0))
            alg = algo_fb(k, L)
            by[str(L)] = {"lanes_per_launch": L, "fetch_kib": fk * f, "write_kib": wk * fw, "dispatches": [nf_, nw_], "avg_ns": [ns_f, ns_w],
                          "algorithmic_kib": alg, "traffic_over_algorithmic": (fk * f + wk * fw) / alg if alg else None}
        e["by_lanes"] = by
    kern[k] = e
out = {"source": sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)",
       "unit": "KiB per launch (corrected; *_raw = the counter as reported)", "calibration": cal, "kernels": kern}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
