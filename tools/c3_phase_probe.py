"""Developer: the C3 logistic regression's value-only (logits pass alone) and gradient calls, fused route vs two-kernel route
(run once per route: MIVI_LR_NO_FUSED=1 selects the two-kernel one)."""
import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
import advancedvi_jl_amd as avi
from tests.helpers import SEED
n, d, M = int(os.environ.get("N", 1_000_000)), 512, 128
rng = np.random.default_rng(3)
p = d - 1
X = np.empty((n, p), dtype=np.float32)
X[:, :p - 1] = rng.standard_normal((n, p - 1), dtype=np.float32) / np.sqrt(p - 1.0)
X[:, p - 1] = 1.0
y = (rng.random(n) < 0.5).astype(np.uint8)
q = avi.FullRankGaussian(np.zeros(d, np.float32), 0.6 * np.eye(d, dtype=np.float32))
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
ctx.set_problem(avi.LogRegProblem(X, y, "logsigma_normal", 1.0))
print("route", ctx.logreg_kernels(), flush=True)
pd = ctx.to_device(params)
v, g = ctx.empty(1), ctx.empty(ctx.params_len)
for name, fn in (("value only (logits pass)", lambda i: ctx.estimate_objective(pd, i, n_samples=M, entropy=0, value=v)),
                 ("value + gradient", lambda i: ctx.estimate_gradient(pd, i, v, g))):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        fn(10 + i)
    torch.cuda.synchronize()
    print("%-28s %.1f us per call" % (name, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
