import numpy as np, torch, sys, time
sys.path.insert(0, '/root/repo')
import advancedvi_jl_amd as avi
from tests.helpers import SEED
d, M = 1024, 256
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
prob = avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32))
params, _ = avi.destructure(q)
st = torch.cuda.Stream()
def t_sync(f, n=50):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort(); return "median %.1f max %.1f us" % (ts[n // 2], ts[-1])
print("idle device sync before ctx:", t_sync(torch.cuda.synchronize))
with torch.cuda.stream(st):
    ctx = avi.MiviContext(np.float32, 1, d, M, 0, SEED); ctx.set_problem(prob)
    p = ctx.to_device(params); v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    ctx.estimate_gradient_n(p, 0, 20, v, g); ctx.estimate_gradient_n(p, 20, 20, v, g); st.synchronize()
    print("idle device sync with chains:", t_sync(torch.cuda.synchronize), " idle stream sync:", t_sync(st.synchronize))
    for mode in ("stream", "device", "stream+device", "device-before+stream+device"):
        ts = []
        for r in range(40):
            if mode.startswith("device-before"): torch.cuda.synchronize()
            t0 = time.perf_counter(); ctx.estimate_gradient_n(p, 40 + 20 * r, 20, v, g)
            if "stream" in mode: st.synchronize()
            if "device" in mode.replace("device-before", ""): torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 20 * 1e6)
        ts.sort(); print(mode, "median %.2f min %.2f max %.2f" % (ts[20], ts[0], ts[-1]))
