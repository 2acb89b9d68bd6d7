"""gaussian_expectation_gradient_and_hessian (Stein branch) on the north-star shape: device time per call, stage by stage
from a rocprofv3 run if wanted (tools/profile_round.sh style), here simply hipEvent-free wall clock over many calls."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import advancedvi_jl_amd as avi

CASES = [(1024, 256, "diag"), (1024, 256, "dense"), (512, 128, "diag"), (2048, 256, "diag")]
if len(sys.argv) > 1 and sys.argv[1] == "ns":
    CASES = CASES[:1]
for d, M, kind in CASES:
    rng = np.random.default_rng(0)
    q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
    if kind == "diag":
        prob = avi.DiagNormalProblem(np.full(d, 5, np.float32), np.ones(d, np.float32))
    else:
        prob = avi.DenseNormalProblem(np.full(d, 5, np.float32), np.tril(np.eye(d) + 1.0 / (2 * d)).astype(np.float32))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, 1)
    ctx.set_problem(prob)
    p = ctx.to_device(params)
    g = ctx.empty(d); H = ctx.empty(d * d)
    for i in range(20):
        ctx.gauss_expected_grad_hess(p, i, 0, g, H)
    torch.cuda.synchronize()
    n = 500
    t0 = time.perf_counter()
    for i in range(n):
        ctx.gauss_expected_grad_hess(p, i, 0, g, H)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    v = ctx.empty(1); gg = ctx.empty(ctx.params_len)
    for i in range(n):
        ctx.estimate_gradient(p, i, v, gg)
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / n
    print(f"d={d} M={M} {kind}: grad+hess {dt * 1e6:8.1f} us/call   (ELBO gradient estimate on the same context: {dt2 * 1e6:.1f} us)", flush=True)
    ctx.close()
