"""Developer: per-step shader-clock stamps of the STL chain kernel (MIVI_STL_STAMPS=1), eager estimates at the north-star shape."""
import os, sys
os.environ["MIVI_STL_STAMPS"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi
d, M = 1024, 256
rng = np.random.default_rng(0)
mu = rng.normal(size=d).astype(np.float32)
Cm = np.tril(rng.normal(size=(d, d)) * (0.3 / np.sqrt(d))).astype(np.float32)
Cm[np.diag_indices(d)] = rng.uniform(0.5, 1.5, d)
q = avi.FullRankGaussian(mu, Cm)
prob = avi.DiagNormalProblem(np.zeros(d, np.float32), np.ones(d, np.float32))
p_h, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 3, 7)
ctx.set_problem(prob)
p = ctx.to_device(p_h)
for i in range(4):
    ctx.estimate_gradient(p, i)
