# developer: us per estimate of the pipelined sharded batch at world 1 (p2p route), plus the pieces (mivi_profile_dist)
import os, sys, numpy as np, torch, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import advancedvi_jl_amd as avi
from tests.helpers import SEED
d, M = 1024, 256
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
prob = avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32))
params, _ = avi.destructure(q)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    ctx = avi.MiviContext(np.float32, 1, d, M, 0, SEED); ctx.set_problem(prob)
    ctx.p2p_attach([ctx.p2p_export(0, 1)])
    ctx.comm_set_route("p2p")
    p = ctx.to_device(params); v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    ctx.estimate_gradient_dist(p, 0, v, g); ctx.synchronize()
    for r in range(20): ctx.estimate_gradient_dist_n(p, 1 + 20 * r, 20, v, g)
    st.synchronize()
    t0 = time.perf_counter()
    for r in range(50): ctx.estimate_gradient_dist_n(p, 1000 + 20 * r, 20, v, g)
    st.synchronize()
    print("pipelined us/estimate %.2f" % ((time.perf_counter() - t0) / 1000 * 1e6))
    try:
        ctx.synchronize()
    except Exception as e:
        print("status:", e)
