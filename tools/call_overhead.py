#!/usr/bin/env python
"""Developer: where an ISOLATED batched call's time goes (the driver's protocol is one 20-estimate call + a device-wide synchronize): host time
until mivi_estimate_gradient_n returns, time until the stream has drained, for n estimates per call."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi
d, M = 1024, 256
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
p0, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, 1, d, M, 0, 1)
ctx.set_problem(avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32)))
p = ctx.to_device(p0)
v, g = ctx.empty(1), ctx.empty(ctx.params_len)
for n in [int(x) for x in sys.argv[1:]] or [20, 40, 80]:
    for _ in range(30):
        ctx.estimate_gradient_n(p, 0, n, v, g)
    torch.cuda.synchronize()
    th, tt = [], []
    for r in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.estimate_gradient_n(p, (r + 1) * n, n, v, g)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        th.append(t1 - t0); tt.append(t2 - t0)
    th, tt = np.array(th) * 1e6, np.array(tt) * 1e6
    print(f"n={n}: host call {np.median(th):.1f} us, call + synchronize {np.median(tt):.1f} us (min {tt.min():.1f}) = {np.median(tt)/n:.2f} us per estimate", flush=True)
ctx.close()
