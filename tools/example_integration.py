import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, advancedvi_jl_amd as avi
rng = np.random.default_rng(0)
n, p = 2000, 8
X = rng.normal(size=(n, p)); beta = rng.normal(size=p)
y = (rng.uniform(size=n) < 1 / (1 + np.exp(-X @ beta))).astype(np.uint8)
big = avi.LogRegProblem(X, y)
sub = avi.ReshufflingBatchSubsampling(np.arange(len(y)), 32)
alg2 = avi.KLMinRepGradProxDescent(avi.AutoMIVI(), n_samples=8, optimizer=avi.DoWG(), subsampling=sub)
q2, info2, _ = avi.optimize(avi.PhiloxRNG(2), alg2, 5 * len(sub), big, avi.MeanFieldGaussian(np.zeros(p + 1), np.ones(p + 1)))
print("elbo first/last", info2[0]["elbo"], info2[-1]["elbo"], "epoch", info2[-1]["epoch"])
print("beta err", np.linalg.norm(q2.location[:p] - beta) / np.linalg.norm(beta))
d = 16
q0 = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
prob = avi.DiagNormalProblem(np.full(d, 5, np.float32), np.ones(d, np.float32))
alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=256, optimizer=avi.Adam(1e-1), operator=avi.ClipScale())
q, info, state = avi.optimize(avi.PhiloxRNG(1), alg, 300, prob, q0)
print("mean", q.location[:3], "elbo", info[-1]["elbo"])
# the Stein / Price estimator of E_q[grad log pi], E_q[hess log pi] (inner estimator of the measure-space algorithms)
S = np.array([[2.0, -0.1], [-0.1, 2.0]])
quad = avi.DenseNormalProblem(np.zeros(2), np.linalg.cholesky(np.linalg.inv(S)))
lp, g, H = avi.gaussian_expectation_gradient_and_hessian_(avi.PhiloxRNG(3), avi.FullRankGaussian(np.ones(2), 0.1 * np.eye(2)),
                                                          10**6, None, None, quad)
print("E grad", g.cpu().numpy(), "(-S mu =", -S @ np.ones(2), ")  E hess", H.cpu().numpy().round(2).tolist())
