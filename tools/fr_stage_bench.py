"""Full-rank stage A/B: second-generation (LDS-staged, split-K) kernels vs the first generation (MIVI_FR_GEN1=1).
For each shape: parity of value / gradient against the fp64 oracle on the device's own eps, then hipEvent-timed stages
(mivi_profile_kernel).  Run once per generation (the route is chosen per process):  python tools/fr_stage_bench.py [gen1]"""
import os, sys
if len(sys.argv) > 1 and sys.argv[1] == "gen1":
    os.environ["MIVI_FR_GEN1"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi
from oracle import oracle as O

gen = "gen1" if os.environ.get("MIVI_FR_GEN1") else "gen2"
shapes = [(256, 64), (1024, 256), (2048, 256), (4096, 1024)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ["SHAPES"].split(",")]
for target in ("diag", "dense"):
    for d, M in shapes:
        rng = np.random.default_rng(d + M)
        mu = rng.normal(size=d).astype(np.float32)
        Cm = np.tril(rng.normal(size=(d, d)) * (0.3 / np.sqrt(d))).astype(np.float32)
        Cm[np.diag_indices(d)] = rng.uniform(0.5, 1.5, d)
        q = avi.FullRankGaussian(mu, Cm)
        tm = rng.normal(size=d).astype(np.float32) + 5
        if target == "diag":
            ts = rng.uniform(0.5, 2.0, d).astype(np.float32)
            prob, otgt = avi.DiagNormalProblem(tm, ts), O.DiagNormalTarget(tm, ts)
        else:
            if d > 2048:
                continue
            L = np.tril(rng.normal(size=(d, d)) * (0.2 / np.sqrt(d))).astype(np.float32)
            L[np.diag_indices(d)] = rng.uniform(0.7, 1.3, d)
            prob, otgt = avi.DenseNormalProblem(tm, L), O.DenseNormalTarget(tm, L)
        p_h, _ = avi.destructure(q)
        for ent in (0, 3):
            ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, 7)
            ctx.set_problem(prob)
            p = ctx.to_device(p_h)
            if d <= 1024:
                _, eps = ctx.sample(p_h, 3)
                v, g = ctx.estimate_gradient(p, 3)
                v2, g2 = ctx.estimate_gradient(p, 4)   # speculative eps route
                v3, g3 = ctx.estimate_gradient(p, 3)
                ref = O.estimate_gradient(p_h.astype(np.float64), d, avi.FULLRANK, otgt, eps.cpu().numpy().astype(np.float64), ent)
                gg = g.cpu().numpy().astype(np.float64)
                ev = abs(float(v.item()) - ref["value"]) / abs(ref["value"])
                eg = np.linalg.norm(gg - ref["grad"]) / np.linalg.norm(ref["grad"])
                emu = np.linalg.norm(gg[:d] - ref["grad"][:d]) / np.linalg.norm(ref["grad"][:d])
                up = np.triu(gg[d:].reshape(d, d).T, 1)
                same = bool((g3 == g).all().item() and (v3 == v).all().item())
                print(f"[{gen}] {target} d={d} M={M} ent={ent}: value rel {ev:.2e}  grad rel-L2 {eg:.2e} (mu part {emu:.2e})  "
                      f"upper zeros {bool((up == 0).all())}  repeat bitwise {same}", flush=True)
            if ent == 0:
                t = {}
                stages = [("estimate", 0), ("sample", 2), ("vjp", 3)] + ([("dense", 4)] if target == "dense" else [])
                if gen == "gen2":
                    stages += [("gemmS", 6), ("reduce", 7)]
                for name, which in stages:
                    ctx.profile_kernel(which, p, 5)
                    t[name] = min(ctx.profile_kernel(which, p, 40) for _ in range(3)) * 1e3
                import torch, time
                vv, gg = ctx.empty(1), ctx.empty(ctx.params_len)
                for r in range(3):
                    ctx.estimate_gradient_n(p, 100 * r, 100, vv, gg)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for r in range(10):
                    ctx.estimate_gradient_n(p, 100 * (r + 3), 100, vv, gg)
                torch.cuda.synchronize()
                t["chain"] = (time.perf_counter() - t0) / 1000 * 1e6
                fl = float(d) * d * M
                print(f"[{gen}] {target} d={d} M={M}: " + "  ".join(f"{k} {v:8.2f} us" for k, v in t.items()) +
                      f"  | sample {fl / t['sample'] / 1e6:5.1f} TF  vjp {fl / t['vjp'] / 1e6:5.1f} TF", flush=True)
            ctx.close()
