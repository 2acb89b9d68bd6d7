"""Developer: the batch engine's launch durations per step width (mivi_profile_batch) at the north-star shape.  argv: lane counts (default: a sweep)."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
import advancedvi_jl_amd as avi
from tests.helpers import SEED
d, M = 1024, 256
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
ctx.set_problem(avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32)))
p = ctx.to_device(params)
for L in ([int(x) for x in sys.argv[1:]] or (8, 16, 20, 24, 32, 40, 48, 64, 80, 100, 128)):
    t = ctx.profile_batch(p, L, 20)
    print(f"L={L:4d} eps {t['eps']:7.1f} prod {t['product']:7.1f} vjp {t['vjp']:7.1f} us | per lane: eps {t['eps']/L:5.2f} prod {t['product']/L:5.2f} vjp {t['vjp']/L:5.2f} sum {(t['eps']+t['product']+t['vjp'])/L:5.2f}", flush=True)
