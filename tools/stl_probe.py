"""Developer: per-call time of 100-estimate batches with the sticking-the-landing estimator on the batch engine (north-star shape)."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
import advancedvi_jl_amd as avi
from tests.helpers import SEED
d, M = 1024, 256
ent = int(sys.argv[1]) if len(sys.argv) > 1 else 3
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
ctx.set_problem(avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32)))
p = ctx.to_device(params)
v, g = ctx.empty(1), ctx.empty(ctx.params_len)
for rep in range(3):
    ts = []
    for r in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.estimate_gradient_n(p, 100 * (10 * rep + r), 100, v, g)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("ent", ent, "rep", rep, "ms per 100-estimate call:", " ".join("%.2f" % t for t in ts), flush=True)
print("batch_info", ctx.lib.mivi_batch_info(ctx.h, None, 0) if hasattr(ctx.lib, "mivi_batch_info") else None)
