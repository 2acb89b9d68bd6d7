#!/usr/bin/env python
"""Developer check of the third-generation batch engine (csrc/kernels_fullrank_batch.hip): every estimate of a batch against the single
calls (bitwise) and the timing of batched calls at the north-star shape.  `python tools/fb_check.py [parity] [time]`."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import advancedvi_jl_amd as avi  # noqa: E402
from tests.helpers import SEED, make_family, make_problem  # noqa: E402


def parity(shapes=((128, 128), (256, 128), (384, 256), (1024, 256)), counts=(1, 2, 3, 7, 20, 33, 52), ents=(0, 2), kind="diag"):
    bad = 0
    for d, M in shapes:
        for ent in ents:
            rng = np.random.default_rng(100 + d + M)
            q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
            prob, _ = make_problem(rng, kind, d, np.float32)
            params, _ = avi.destructure(q)
            ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
            ctx.set_problem(prob)
            ref = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
            ref.set_problem(prob)
            p, pr = ctx.to_device(params), ref.to_device(params)
            idx = 5
            for n in counts:
                vals, grads = ctx.estimate_gradient_each(p, idx, n)
                ctx.synchronize()
                vals, grads = vals.cpu().numpy(), grads.cpu().numpy()
                for i in range(n):
                    v1, g1 = ref.estimate_gradient(pr, idx + i)
                    v1, g1 = float(v1.item()), g1.cpu().numpy()
                    okv = float(vals[i]) == v1
                    okg = np.array_equal(grads[i], g1)
                    if not (okv and okg):
                        bad += 1
                        dm = np.abs(grads[i][:d] - g1[:d]).max()
                        G, G1 = grads[i][d:].reshape(d, d), g1[d:].reshape(d, d)   # [col][row]
                        dc = np.abs(G - G1)
                        nz = np.argwhere(dc > 0)
                        print(f"MISMATCH d={d} M={M} ent={ent} n={n} i={i}: value {vals[i]!r} vs {v1!r}; dmu max {dm:.3e}; dC max {dc.max():.3e} "
                              f"rel {dc.max() / max(np.abs(G1).max(), 1e-30):.3e}, {len(nz)} entries, first (col,row) {nz[:4].tolist()}; "
                              f"upper nonzero {int((np.triu(G.T, 1) != 0).sum())}")
                        if bad > 12:
                            return bad
                # the _n entry: last estimate only
                v, g = ctx.empty(1), ctx.empty(ctx.params_len)
                g.fill_(float("nan"))
                ctx.estimate_gradient_n(p, idx, n, v, g)
                ctx.synchronize()
                v1, g1 = ref.estimate_gradient(pr, idx + n - 1)
                if not (float(v.item()) == float(v1.item()) and np.array_equal(g.cpu().numpy(), g1.cpu().numpy())):
                    bad += 1
                    print(f"MISMATCH (_n) d={d} M={M} ent={ent} n={n}: {float(v.item())!r} vs {float(v1.item())!r}, "
                          f"grad max diff {np.nanmax(np.abs(g.cpu().numpy() - g1.cpu().numpy())):.3e}, nan {int(np.isnan(g.cpu().numpy()).sum())}")
                idx += n + 1
            ctx.close()
            ref.close()
            print(f"parity {kind} d={d} M={M} ent={ent}: done, mismatches so far {bad}", flush=True)
    return bad


def timing(d=1024, M=256, counts=(20, 100), reps=200, kind="diag"):
    import torch
    rng = np.random.default_rng(1)
    q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
    prob = avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32))
    if kind == "dense":
        prob, _ = make_problem(rng, "dense", d, np.float32)
    params, _ = avi.destructure(q)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
        ctx.set_problem(prob)
        p = ctx.to_device(params)
        v, g = ctx.empty(1), ctx.empty(ctx.params_len)
        for n in counts:
            idx = 0
            for _ in range(20):
                ctx.estimate_gradient_n(p, idx, n, v, g)
                idx += n
            stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.estimate_gradient_n(p, idx, n, v, g)
                idx += n
            stream.synchronize()
            dt = time.perf_counter() - t0
            # isolated calls (the driver's protocol: one call, device-wide synchronize)
            iso = []
            for _ in range(30):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ctx.estimate_gradient_n(p, idx, n, v, g)
                torch.cuda.synchronize()
                iso.append(time.perf_counter() - t1)
                idx += n
            iso.sort()
            print(f"timing {kind} n={n}: back-to-back {dt / reps / n * 1e6:.3f} us/estimate ({n * reps / dt:.0f} est/s); isolated median "
                  f"{iso[len(iso) // 2] / n * 1e6:.3f} us/estimate, min {iso[0] / n * 1e6:.3f}", flush=True)
        ctx.close()


if __name__ == "__main__":
    what = sys.argv[1:] or ["parity", "time"]
    if "parity" in what:
        b = parity()
        print("PARITY", "OK" if b == 0 else f"FAILED ({b})")
    if "time" in what:
        timing()
    if "parity_dense" in what:
        b = parity(kind="dense", counts=(1, 2, 7, 20, 33), ents=(0, 2))
        print("PARITY dense", "OK" if b == 0 else f"FAILED ({b})")
    if "time_dense" in what:
        timing(kind="dense")
    if "time100" in what:
        timing(counts=(100,), reps=20)


def stl_check(shapes=((256, 128), (512, 256), (1024, 256), (2048, 128)), n=6):
    """The sticking-the-landing estimators on the engine (W += C^-T eps through the explicit inverse, formed once per call) against the
    single calls (which solve C^T X = eps) and against the fp64 oracle: relative l2 distances of the gradients."""
    from oracle import oracle as O
    for d, M in shapes:
        for ent in (3, 4):
            for kind in ("diag", "dense"):
                if kind == "dense" and d > 1024:
                    continue
                rng = np.random.default_rng(7 + d + M)
                q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
                prob, tgt = make_problem(rng, kind, d, np.float32)
                params, _ = avi.destructure(q)
                ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
                ctx.set_problem(prob)
                ref = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
                ref.set_problem(prob)
                p, pr = ctx.to_device(params), ref.to_device(params)
                vals, grads = ctx.estimate_gradient_each(p, 11, n)
                ctx.synchronize()
                vals, grads = vals.cpu().numpy(), grads.cpu().numpy()
                worst = [0.0, 0.0, 0.0, 0.0]
                for i in (0, n - 1):
                    v1, g1 = ref.estimate_gradient(pr, 11 + i)
                    g1 = g1.cpu().numpy().astype(np.float64)
                    _, eps = ref.sample(pr, 11 + i)
                    o = O.estimate_gradient(params.astype(np.float64), d, avi.FULLRANK, tgt, eps.cpu().numpy().astype(np.float64), ent)
                    gb = grads[i].astype(np.float64)
                    nrm = max(1.0, np.linalg.norm(o["grad"]))
                    worst[0] = max(worst[0], np.linalg.norm(gb - g1) / nrm)
                    worst[1] = max(worst[1], np.linalg.norm(gb - o["grad"]) / nrm)
                    worst[2] = max(worst[2], np.linalg.norm(g1 - o["grad"]) / nrm)
                    worst[3] = max(worst[3], abs(float(vals[i]) - o["value"]) / abs(o["value"]))
                print(f"stl d={d} M={M} ent={ent} {kind}: batch-vs-single {worst[0]:.2e}  batch-vs-oracle {worst[1]:.2e}  single-vs-oracle {worst[2]:.2e}  "
                      f"value-vs-oracle {worst[3]:.2e}  |grad| {nrm:.3e}", flush=True)
                ctx.close()
                ref.close()


if __name__ == "__main__" and "stl" in sys.argv[1:]:
    stl_check()
    import torch  # noqa: F401
    rng = np.random.default_rng(1)
    d, M = 1024, 256
    q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
    prob, _ = make_problem(rng, "diag", d, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 3, SEED)
    ctx.set_problem(prob)
    p = ctx.to_device(params)
    v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    for n in (20, 100):
        idx = 0
        for _ in range(5):
            ctx.estimate_gradient_n(p, idx, n, v, g)
            idx += n
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            ctx.estimate_gradient_n(p, idx, n, v, g)
            idx += n
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print(f"timing stl n={n}: {dt / 50 / n * 1e6:.3f} us/estimate ({50 * n / dt:.0f} est/s)")
    print(ctx.profile_batch(p, 20, 20), ctx.profile_batch(p, 100, 10))
