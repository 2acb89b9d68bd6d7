"""us per iteration of avi.optimize (device loop) over small/medium everyday configurations -- a pathology finder:
anything far above (number of kernels) x 5 us deserves a profile."""
import sys, time, warnings
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import advancedvi_jl_amd as avi
rng = np.random.default_rng(0)
def logreg(n, p, dt):
    X = (rng.normal(size=(n, p)) / np.sqrt(p)).astype(dt); y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    return avi.LogRegProblem(X, y)
rows = []
for dt in (np.float32, np.float64):
    for fam in ("mf", "fr"):
        for d, tgt in ((8, "diag"), (64, "diag"), (256, "diag"), (256, "dense"), (33, "logreg1k"), (129, "logreg20k"), (64, "funnel")):
            if tgt == "diag":
                prob = avi.DiagNormalProblem(np.full(d, 2.0, dt), np.ones(d, dt))
            elif tgt == "dense":
                prob = avi.DenseNormalProblem(np.full(d, 2.0, dt), np.tril(np.eye(d) + 0.5 / d).astype(dt))
            elif tgt == "funnel":
                prob = avi.FunnelProblem(d, 1.5)
            else:
                prob = logreg(1000 if tgt == "logreg1k" else 20000, d - 1, dt)
            q0 = avi.MeanFieldGaussian(np.zeros(d, dt), np.ones(d, dt)) if fam == "mf" else avi.FullRankGaussian(np.zeros(d, dt), np.eye(d, dtype=dt))
            for algname in ("dowg+clip+poly", "adam+clip"):
                for ent in ("cf", "stl"):
                    e = avi.ClosedFormEntropy() if ent == "cf" else avi.StickingTheLandingEntropy()
                    if algname.startswith("dowg"):
                        alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=16, entropy=e, optimizer=avi.DoWG(), operator=avi.ClipScale())
                    else:
                        alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=16, entropy=e, optimizer=avi.Adam(1e-2), operator=avi.ClipScale(), averager=avi.NoAveraging())
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        try:
                            avi.optimize(avi.PhiloxRNG(1), alg, 16, prob, q0)
                            T = 256
                            t0 = time.perf_counter()
                            avi.optimize(avi.PhiloxRNG(1), alg, T, prob, q0)
                            us = (time.perf_counter() - t0) / T * 1e6
                        except Exception as ex:   # noqa: BLE001
                            us = float("nan"); print("ERR", type(ex).__name__, ex)
                    rows.append((us, np.dtype(dt).name, fam, d, tgt, algname, ent))
rows.sort(reverse=True)
for r in rows[:25]:
    print(f"{r[0]:9.1f} us/iter  {r[1]} {r[2]} d={r[3]} {r[4]} {r[5]} {r[6]}")
print("...")
for r in rows[-6:]:
    print(f"{r[0]:9.1f} us/iter  {r[1]} {r[2]} d={r[3]} {r[4]} {r[5]} {r[6]}")
