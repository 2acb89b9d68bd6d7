import sys, numpy as np
sys.path.insert(0, "/root/repo")
import advancedvi_jl_amd as avi
from tests.helpers import SEED
d, M = 256, 256
mu = np.linspace(1.0, 2.0, d).astype(np.float32)
q = avi.FullRankGaussian(mu, (1e-6 * np.eye(d)).astype(np.float32))
params, _ = avi.destructure(q)
prob = avi.DiagNormalProblem(np.zeros(d, np.float32), np.ones(d, np.float32))
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED); ctx.set_problem(prob)
p = ctx.to_device(params)
vals, grads = ctx.estimate_gradient_each(p, 40, 2)
ctx.synchronize()
g = grads.cpu().numpy()[0]
v1, g1 = ctx.estimate_gradient(p, 40)
g1 = g1.cpu().numpy()
np.set_printoptions(precision=4, linewidth=220)
print("ratio engine/single of dmu (expect 1):")
print((g[:d] / g1[:d]).reshape(-1, 32))
