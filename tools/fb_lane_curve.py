"""Developer: the batch engine's launch durations per step width (mivi_profile_batch).
usage: fb_lane_curve.py [--shape d,M] [--dense] [lane counts ...]   (default shape: the north star's 1024,256; default lanes: a sweep)
Run it under MIVI_FB_SPLIT=0 / 1 to compare the two product workgroup shapes at every width (kernels_fullrank_batch.hip fb_launch_compute)."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
import advancedvi_jl_amd as avi
from tests.helpers import SEED
args = sys.argv[1:]
d, M, dense = 1024, 256, False
if "--shape" in args:
    i = args.index("--shape"); d, M = (int(x) for x in args[i + 1].split(",")); del args[i:i + 2]
if "--dense" in args:
    dense = True; args.remove("--dense")
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
if dense:
    ctx.set_problem(avi.DenseNormalProblem(np.full(d, 5.0, np.float32), np.tril(np.eye(d) + np.ones((d, d)) / (2 * d)).astype(np.float32)))
else:
    ctx.set_problem(avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32)))
p = ctx.to_device(params)
for L in ([int(x) for x in args] or (4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 64, 80)):
    t = ctx.profile_batch(p, L, 20)
    dp = t.get("dense_product", 0.0)
    print(f"d={d} M={M} L={L:4d} eps {t['eps']:7.1f} prod {t['product']:7.1f} {'dense_prod %7.1f ' % dp if dense else ''}vjp {t['vjp']:7.1f} us | per lane: eps {t['eps']/L:5.2f} prod {t['product']/L:5.2f} vjp {t['vjp']/L:5.2f} sum {(t['eps']+t['product']+dp+t['vjp'])/L:5.2f}", flush=True)
