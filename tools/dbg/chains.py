import numpy as np, torch, sys, time
sys.path.insert(0, '/root/repo')
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
    import os
    if os.environ.get("FIRST"):
        ctx.estimate_gradient_n(p, 5, 20, v, g); torch.cuda.synchronize()
        if os.environ.get("HEAT"):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < float(os.environ["HEAT"]): ctx.estimate_gradient_n(p, 5, 20, v, g)
            torch.cuda.synchronize()
        ts = []
        for r in range(6):
            t0 = time.perf_counter(); ctx.estimate_gradient_n(p, 25 + 20 * r, 20, v, g); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
        print("fresh graph, calls in order:", ["%.1f" % t for t in ts])
    for chunk in (100, 20):
        ctx.estimate_gradient_n(p, 0, chunk, v, g); ctx.estimate_gradient_n(p, chunk, chunk, v, g); st.synchronize()
        n = 2000 // chunk
        t0 = time.perf_counter()
        for r in range(n): ctx.estimate_gradient_n(p, (r + 2) * chunk, chunk, v, g)
        st.synchronize(); dt = time.perf_counter() - t0
        v1, g1 = ctx.estimate_gradient(p, (n + 1) * chunk + chunk - 1)
        print("chunk", chunk, "us/est %.2f" % (dt / (n * chunk) * 1e6), "equal", bool((g == g1).all()), float(v) == float(v1))
    # the driver's protocol: 5 warm-up + 20 timed, one call of 20
    ctx.estimate_gradient_n(p, 0, 5, v, g); ctx.estimate_gradient_n(p, 5, 20, v, g); st.synchronize()
    ts = []
    for r in range(40):
        t0 = time.perf_counter(); ctx.estimate_gradient_n(p, 25 + 20 * r, 20, v, g); st.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
    print("in order:", ["%.1f" % t for t in ts[:8]])
    ts.sort(); print("20-step isolated calls us/est: median %.2f min %.2f max %.2f" % (ts[20], ts[0], ts[-1]))
