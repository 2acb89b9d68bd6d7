"""avi.optimize wall time: device-resident loop (default) vs host-driven `step` loop, reference-default algorithm
(DoWG + PolynomialAveraging + ClipScale) on the README-sized LogReg (n=1000, d=33, mean-field, 16 samples) and C2."""
import sys, time, warnings
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import advancedvi_jl_amd as avi
rng = np.random.default_rng(0)
X = rng.normal(size=(1000, 32)); y = (rng.uniform(size=1000) < 0.5).astype(np.uint8)
cases = [("README LogReg n=1000 d=33 mean-field M=16 (C1)", avi.LogRegProblem(X, y), avi.MeanFieldGaussian(np.zeros(33), np.ones(33)), 16),
         ("C2 diag-Gaussian d=1024 mean-field M=256", avi.DiagNormalProblem(np.full(1024, 5.0, np.float32), np.ones(1024, np.float32)),
          avi.MeanFieldGaussian(np.zeros(1024, np.float32), np.ones(1024, np.float32)), 256),
         ("NS diag-Gaussian d=1024 full-rank M=256", avi.DiagNormalProblem(np.full(1024, 5.0, np.float32), np.ones(1024, np.float32)),
          avi.FullRankGaussian(np.zeros(1024, np.float32), np.eye(1024, dtype=np.float32)), 256)]
for name, prob, q0, M in cases:
    alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=M, optimizer=avi.DoWG(), operator=avi.ClipScale())
    T = 1000
    for dev in (True, False):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            avi.optimize(avi.PhiloxRNG(1), alg, 20, prob, q0, device_loop=dev)   # warm
            t0 = time.perf_counter()
            q, info, _ = avi.optimize(avi.PhiloxRNG(1), alg, T, prob, q0, device_loop=dev)
            dt = time.perf_counter() - t0
        print(f"{name}: {'device loop' if dev else 'host loop  '} {dt / T * 1e6:8.1f} us/iteration  (elbo {info[-1]['elbo']:.4g})", flush=True)
