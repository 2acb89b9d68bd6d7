# developer: isolated batches of n estimates (median us per estimate) -- run under MIVI_CHAINS=4 / 8 to place the four / eight-context switch
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
    out = []
    for n in (8, 10, 12, 16, 20, 24, 32, 40, 48):
        ctx.estimate_gradient_n(p, 0, n, v, g); st.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1: ctx.estimate_gradient_n(p, 5, n, v, g)
        st.synchronize()
        ts = []
        for r in range(30):
            t0 = time.perf_counter(); ctx.estimate_gradient_n(p, 25 + n * r, n, v, g); st.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e6)
        ts.sort(); out.append("%d: %.2f" % (n, ts[15]))
    print(os.environ.get("MIVI_CHAINS"), " ".join(out))
