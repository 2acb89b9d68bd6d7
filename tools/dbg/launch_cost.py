# developer: host time of one mivi_estimate_gradient_n call (hipGraphLaunch of the cached graph) against the GPU time of the batch
import numpy as np, torch, sys, time, os
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
    p = ctx.to_device(params); v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    for n in (20, 100, 400):
        ctx.estimate_gradient_n(p, 0, n, v, g); ctx.estimate_gradient_n(p, n, n, v, g); st.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2: ctx.estimate_gradient_n(p, 2 * n, n, v, g)
        st.synchronize()
        host, tot = [], []
        for r in range(10):
            st.synchronize()
            t0 = time.perf_counter(); ctx.estimate_gradient_n(p, 3 * n + r * n, n, v, g); t1 = time.perf_counter(); st.synchronize(); t2 = time.perf_counter()
            host.append((t1 - t0) * 1e6); tot.append((t2 - t0) * 1e6)
        host.sort(); tot.sort()
        print("n = %d: host time of the call %.0f us (%.2f per estimate), call + sync %.0f us (%.2f per estimate)" % (n, host[5], host[5] / n, tot[5], tot[5] / n))
