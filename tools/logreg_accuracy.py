import sys, os, subprocess, numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    import advancedvi_jl_amd as avi
    rng = np.random.default_rng(0)
    n, p, M = 200000, 511, 128
    d = p + 1
    X = (rng.normal(size=(n, p)) / np.sqrt(p)).astype(np.float32)
    beta = rng.normal(size=p)
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-X @ beta))).astype(np.uint8)
    q = avi.FullRankGaussian(np.zeros(d, np.float32), 0.6 * np.eye(d, dtype=np.float32))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, 1)
    ctx.set_problem(avi.LogRegProblem(X, y))
    v, g = ctx.estimate_gradient(params, 3)
    np.save(sys.argv[1], np.concatenate([[float(v.item())], g.cpu().numpy().astype(np.float64)]))
else:
    for name, env in (("bf16x3", {}), ("f32", {"MIVI_LR_F32_LOGITS": "1"}), ("generic", {"MIVI_LOGREG_GENERIC": "1"})):
        subprocess.run([sys.executable, __file__, f"/tmp/acc_{name}.npy"], env={**os.environ, **env}, check=True, stderr=subprocess.DEVNULL)
    a, b, c = (np.load(f"/tmp/acc_{n}.npy") for n in ("bf16x3", "f32", "generic"))
    for nm, x in (("bf16x3 vs f32-mfma", (a, b)), ("bf16x3 vs generic", (a, c)), ("f32-mfma vs generic", (b, c))):
        u, w = x
        print(f"{nm:22s} value rel {abs(u[0]-w[0])/abs(w[0]):.2e}  grad rel-L2 {np.linalg.norm(u[1:]-w[1:])/np.linalg.norm(w[1:]):.2e}")
