"""Developer: what one 20-estimate call + device-wide synchronize costs beyond its kernels, under the process's wait policy (env set by the caller)."""
import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
import advancedvi_jl_amd as avi
from tests.helpers import SEED
d, M = 1024, 256
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
ctx.set_problem(avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32)))
p = ctx.to_device(params)
v, g = ctx.empty(1), ctx.empty(ctx.params_len)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for i in range(200):
    ctx.estimate_gradient_n(p, i * n, n, v, g)
torch.cuda.synchronize()
ts = []
for i in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.estimate_gradient_n(p, (300 + i) * n, n, v, g)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append(((t2 - t0) * 1e6, (t1 - t0) * 1e6))
ts.sort()
print(f"env {os.environ.get('TAGENV','default')}: call+sync median {ts[len(ts)//2][0]:.1f} us (min {ts[0][0]:.1f}), host time inside the call median {sorted(t[1] for t in ts)[len(ts)//2]:.1f} us")
t = ctx.profile_batch(p, n, 20)
print("   kernels alone (us):", {k: round(v_, 1) for k, v_ in t.items() if v_ > 0}, "sum", round(sum(t.values()), 1))
