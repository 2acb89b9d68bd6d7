"""Developer: per-call time of 100-estimate batches with the sticking-the-landing estimator on the batch engine (north-star shape).
argv: entropy code [pre]   -- `pre`: first run and close an iso and a dense context the way bench.py's earlier legs do."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
import advancedvi_jl_amd as avi
from tests.helpers import SEED
d, M = 1024, 256
ent = int(sys.argv[1]) if len(sys.argv) > 1 else 3
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
params, _ = avi.destructure(q)


def batch_times(ctx, reps=2, label=""):
    p = ctx.to_device(params)
    v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    for rep in range(reps):
        ts = []
        for r in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.estimate_gradient_n(p, 100 * (10 * rep + r), 100, v, g)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(label, "rep", rep, "ms per 100-estimate call:", " ".join("%.2f" % t for t in ts), flush=True)


if "pre" in sys.argv:
    for kind in ("iso", "dense"):
        c0 = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
        if kind == "iso":
            c0.set_problem(avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32)))
        else:
            c0.set_problem(avi.DenseNormalProblem(np.full(d, 5.0, np.float32), np.tril(np.eye(d) + np.ones((d, d)) / (2 * d)).astype(np.float32)))
        batch_times(c0, 1, "pre-" + kind)
        if "keep" not in sys.argv:
            c0.close()
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
ctx.set_problem(avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32)))
batch_times(ctx, 3, "ent %d" % ent)
