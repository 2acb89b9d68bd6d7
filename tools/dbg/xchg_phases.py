# developer: per-phase duration of the exchange kernel at world 1 (run under rocprofv3 --kernel-trace, then pass the .db as argv[1] to analyse)
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
if len(sys.argv) > 1:
    import sqlite3
    con = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scol = [r[1] for r in con.execute(f"pragma table_info({sym})")]
    nc = "kernel_name" if "kernel_name" in scol else "display_name"
    rows = list(con.execute(f"select d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id where s.{nc} like '%k_p2p_exchange%' order by d.start"))
    rows = rows[-(4 * 60):]          # the last 60 rounds of {1, 2, 4, 7}
    for k, name in enumerate(("push", "reduce", "unpack", "fused")):
        ds = sorted(r[1] - r[0] for r in rows[k::4])
        print("%-7s median %6.2f us  min %6.2f" % (name, ds[len(ds) // 2] / 1e3, ds[0] / 1e3))
    sys.exit(0)
import numpy as np, torch
import advancedvi_jl_amd as avi
from advancedvi_jl_amd.distributed import p2p_geometry
from tests.helpers import SEED
d, M = 1024, 256
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
prob = avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32))
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, 1, d, M, 0, SEED)
ctx.set_problem(prob)
ctx.p2p_attach([ctx.p2p_export(0, 1)])
L = ctx.partials_len; n, cn, G, vs = p2p_geometry(L, 1)
p = ctx.to_device(params)
P = ctx.empty(n).zero_(); ctx.estimate_partials(p, 5, P[:L])
v, g = ctx.empty(1), ctx.empty(ctx.params_len)
torch.cuda.synchronize()
for r in range(80):
    for ph in (1, 2, 4, 7):
        ctx.p2p_exchange(p, P, v, g, ph)
        torch.cuda.synchronize()
