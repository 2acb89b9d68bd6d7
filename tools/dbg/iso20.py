# developer: isolated 20-estimate calls (the driver's protocol) for a rocprofv3 timeline
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
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2: ctx.estimate_gradient_n(p, 5, 20, v, g)
    st.synchronize()
    ts = []
    for r in range(10):
        t0 = time.perf_counter(); ctx.estimate_gradient_n(p, 25 + 20 * r, 20, v, g); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
        time.sleep(0.0005)
    print("call us:", [round(t) for t in ts])
