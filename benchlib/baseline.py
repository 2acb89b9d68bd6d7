"""What bench.py checks and compares against, outside every timed region: the synthetic problems, parity against the fp64 oracle on
identical eps, and the CPU baseline (the C port of the oracle on the GPU box's host cores)."""
import json
import os
import sys
import time

import numpy as np

from .config import ROOT, SEED, PEAK_F32_MFMA_TF   # noqa: F401


def make_problem(avi, w):
    d = w["d"]
    q = (avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32)) if w["family"] == 0
         else avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32)))
    if w["target"] == "logreg":
        rng = np.random.default_rng(3)
        n, p = w["n"], d - 1
        X = np.empty((n, p), dtype=np.float32)
        X[:, :p - 1] = rng.standard_normal((n, p - 1), dtype=np.float32) / np.sqrt(p - 1.0)
        X[:, p - 1] = 1.0
        beta = rng.standard_normal(p, dtype=np.float32)
        y = (rng.random(n) < 1 / (1 + np.exp(-(X @ beta)))).astype(np.uint8)
        q = avi.FullRankGaussian(np.zeros(d, np.float32), 0.6 * np.eye(d, dtype=np.float32))
        return q, avi.LogRegProblem(X, y, "logsigma_normal", 1.0)
    if w["target"] == "funnel":
        return q, avi.FunnelProblem(d, 1.5)
    if w["target"] == "iso":
        prob = avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32))
    else:
        L = np.tril(np.eye(d) + np.ones((d, d)) / (2.0 * d)).astype(np.float32)
        prob = avi.DenseNormalProblem(np.full(d, 5.0, np.float32), L)
    return q, prob


def parity_vs_oracle(cx, p_dev, p_host, w, idx=11, batch=20):
    """Value and gradient of ONE estimate of workload `w` at its own shape against the fp64 numpy oracle on identical eps (read back from
    the device).  Test infrastructure, outside every timed region.  None for workloads the oracle cannot finish in seconds (C3)."""
    from oracle import oracle as O
    d = w["d"]
    if w["target"] == "iso":
        tgt = O.DiagNormalTarget(np.full(d, 5.0), np.ones(d))
    elif w["target"] == "dense":
        tgt = O.DenseNormalTarget(np.full(d, 5.0), np.tril(np.eye(d) + np.ones((d, d)) / (2.0 * d)).astype(np.float32).astype(np.float64))
    elif w["target"] == "funnel":
        tgt = O.FunnelStackedTarget(d, 1.5)
    elif w["target"] == "logreg":
        # BASELINE configs[2] at its FULL size: the oracle's batched evaluation (row-chunked f64 matrix products over the f32 data, the
        # per-column restatement to rounding: tests/test_oracle_pinning.py) takes seconds; ONE estimate, as the timed region issues them
        prob = getattr(cx, "problem", None)
        if prob is None or not hasattr(prob, "X") or not isinstance(prob.X, np.ndarray):
            return None
        tgt = O.LogRegTarget(prob.X, prob.y, prob.variant, prob.likeadj, keep_storage=True)
        _, eps = cx.sample(p_dev, idx)
        v, g = cx.estimate_gradient(p_dev, idx)
        cx.synchronize()
        ref = O.estimate_gradient(np.asarray(p_host, dtype=np.float64), d, w["family"], tgt, eps.cpu().numpy().astype(np.float64), w["entropy"], batch_target=True)
        return dict(value_rel=abs(float(v.item()) - ref["value"]) / abs(ref["value"]),
                    grad_rel_l2=float(np.linalg.norm(g.cpu().numpy().astype(np.float64) - ref["grad"]) / np.linalg.norm(ref["grad"])),
                    estimate_idx=idx, batch=1, note="one estimate at the full data size (n = %d) against the fp64 oracle on identical eps" % prob.X.shape[0])
    else:
        return None
    # a batch as the timed region issues it (mivi_estimate_gradient_each: the same kernels as mivi_estimate_gradient_n, every estimate kept):
    # EVERY value against the oracle, first / middle / last gradient
    n = int(batch)
    vals, grads = cx.estimate_gradient_each(p_dev, idx, n)
    cx.synchronize()
    vals, grads = vals.cpu().numpy().astype(np.float64), grads.cpu().numpy()
    p64 = np.asarray(p_host, dtype=np.float64)
    vrel, grel = 0.0, 0.0
    for i in range(n):
        _, eps = cx.sample(p_dev, idx + i)
        ref = O.estimate_gradient(p64, d, w["family"], tgt, eps.cpu().numpy().astype(np.float64), w["entropy"])
        vrel = max(vrel, abs(vals[i] - ref["value"]) / abs(ref["value"]))
        if i in (0, n // 2, n - 1):
            grel = max(grel, float(np.linalg.norm(grads[i].astype(np.float64) - ref["grad"]) / np.linalg.norm(ref["grad"])))
    return dict(value_rel=vrel, grad_rel_l2=grel, estimate_idx=idx, batch=n,
                note="max over EVERY estimate of a %d-estimate batch issued like the timed ones (values); gradients of its first, middle and last estimate" % n)


def cpu_baseline_blas(lib, CO, w, params, tm, ts, budget_s=8.0):
    """A second CPU leg for the full-rank family: the same estimate with its two contractions on the BLAS numpy links (OpenBLAS in this
    image), every core -- what the reference's `scale * eps` (src/families/location_scale.jl:76: a BLAS call, bench/benchmarks.jl:15 sets
    the BLAS threads) and the AD pull-back's products cost at best.  Timed twice: eps drawn inside the timed call with numpy's ziggurat
    generator (a whole estimate, like every other figure of this bench), and eps PRE-DRAWN outside it (the contractions + elementwise
    work alone).  Everything else (target, entropy term, tril, scaling) in numpy.  GFLOP/s: `gflops_executed` counts the two full
    d x d x n_mc GEMMs the BLAS runs (2 * 2 d^2 n_mc), `gflops_algorithmic` the triangular halves the estimate needs (2 d^2 n_mc)."""
    d, M = w["d"], w["n_mc"]
    mu = np.ascontiguousarray(params[:d], dtype=np.float32)
    Cm = np.asfortranarray(np.tril(np.asarray(params[d:], dtype=np.float32).reshape(d, d, order="F")))
    istd = (1.0 / ts).astype(np.float32)

    rng_np = np.random.default_rng(SEED & 0xFFFFFFFF)
    tril_mask = np.tril(np.ones((d, d), dtype=np.float32))
    pool = [np.asfortranarray(rng_np.standard_normal((d, M), dtype=np.float32)) for _ in range(8)]

    def one(eps=None):
        if eps is None:
            # numpy's ziggurat normals (what `rand(rng, Normal, d, M)` costs the reference, ~5 ns each)
            eps = np.asfortranarray(rng_np.standard_normal((d, M), dtype=np.float32))
        Z = Cm @ eps
        Z += mu[:, None]
        U = (Z - tm[:, None]) * istd[:, None]
        ell = -0.5 * float(np.sum(U * U, dtype=np.float64))
        W = -U * istd[:, None]
        G = W @ eps.T
        G *= tril_mask
        G *= -1.0 / M
        G[np.diag_indices(d)] -= 1.0 / np.diag(Cm)
        gmu = -W.sum(axis=1) / M
        return ell, gmu, G

    def leg(predrawn, budget):
        one(pool[0] if predrawn else None)
        t0 = time.perf_counter()
        one(pool[1] if predrawn else None)
        t1 = time.perf_counter() - t0
        reps = int(max(5, min(200, budget / max(t1, 1e-6))))
        ts_ = []
        for i in range(reps):
            t0 = time.perf_counter()
            one(pool[i % len(pool)] if predrawn else None)
            ts_.append(time.perf_counter() - t0)
        ts_.sort()
        return ts_[len(ts_) // 2], reps

    med, reps = leg(False, budget_s / 2)
    med_pre, reps_pre = leg(True, budget_s / 2)
    try:
        import numpy.__config__ as npc
        blas_name = str(npc.CONFIG["Build Dependencies"]["blas"]["name"])
    except Exception:   # noqa: BLE001
        blas_name = "numpy's BLAS"
    fl = 2.0 * d * d * M
    return dict(estimates_per_s=1.0 / med, median_s=med, reps=reps, blas=blas_name,
                gflops_algorithmic=fl / med / 1e9, gflops_executed=2 * fl / med / 1e9,
                eps_predrawn=dict(estimates_per_s=1.0 / med_pre, median_s=med_pre, reps=reps_pre,
                                  gflops_algorithmic=fl / med_pre / 1e9, gflops_executed=2 * fl / med_pre / 1e9,
                                  note="eps taken from a pool drawn before the timed region: the two GEMMs + the numpy elementwise work alone"),
                note="two GEMMs (d x d x n_mc each, f32) on the BLAS + numpy elementwise work, all cores; eps drawn with numpy's ziggurat generator (included)")


def cpu_baseline(w, params, budget_s=24.0):
    """The oracle's C leg (oracle/mivi_oracle.c: a port of the reference semantics with the closed-form VJP,
    cheaper than the reference's AD path) timed on this box's host cores.  Protocol (the reference's
    bench/benchmarks.jl:15 runs with the BLAS threads of the box): team sizes 1, 2, 4, ... up to every CPU this
    process may use -- each >= 20 repetitions of one whole estimate incl. eps generation, MEDIAN reported with the eps
    generation's share and the contractions' GFLOP/s (2 d^2 n_mc algorithmic flops per full-rank estimate), the whole leg
    bounded by `budget_s` seconds of wall time (the repetition count shrinks, never below 5, if the box is slow)."""
    from oracle import c_oracle as CO
    if not os.path.exists(CO.PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    # the port compiled for THIS box (-march=native) when gcc is here; the shipped x86-64-v3 build otherwise
    build = "-O3 -march=x86-64-v3 -fopenmp (shipped)"
    lib = None
    try:
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = CO.load(CO.NATIVE_PATH)
        build = "-O3 -march=native -fopenmp (built on this box)"
    except Exception:   # noqa: BLE001
        lib = CO.load()
    d, M, fam = w["d"], w["n_mc"], w["family"]
    try:
        avail = len(os.sched_getaffinity(0))     # CPUs this process may run on
    except AttributeError:
        avail = os.cpu_count() or 1
    try:                                          # cgroup v2 CPU quota, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            avail = max(1, min(avail, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    tm, ts = np.full(d, 5.0, np.float32), np.ones(d, np.float32)
    work = np.empty(2 * d * M, dtype=np.float32)
    grad = np.empty_like(np.ascontiguousarray(params, dtype=np.float32))
    eps_buf = np.empty((d, M), dtype=np.float32, order="F")
    fl = 2.0 * d * d * M if fam == 1 else 6.0 * d * M

    def one(i):
        t0 = time.perf_counter()
        eps = CO.fill_eps(lib, np.float32, SEED, i, d, M, out=eps_buf)
        t1 = time.perf_counter()
        CO.estimate_gradient(lib, np.float32, fam, d, M, params, eps, tm, ts, w["entropy"], work, grad)
        return t1 - t0, time.perf_counter() - t1

    legs = {}
    t_leg0 = time.perf_counter()
    teams = sorted({1, avail} | {t for t in (2, 4, 8, 16, 32) if t < avail})
    for nt in teams:
        lib.mo32_set_threads(nt)
        one(0)                                   # warm (thread team start-up, page faults)
        t1 = sum(one(1))
        share = budget_s / len(teams)
        reps = int(max(5, min(100, share / max(t1, 1e-6))))
        reps = max(reps, 20) if 20 * t1 <= share else reps
        ts_, te_ = [], []
        for i in range(reps):
            a, b = one(i + 2)
            ts_.append(a + b)
            te_.append((a, b))
        ts_.sort()
        med = ts_[len(ts_) // 2]
        med_eps = sorted(a for a, _ in te_)[len(te_) // 2]
        med_est = sorted(b for _, b in te_)[len(te_) // 2]
        legs[nt] = dict(threads=nt, reps=reps, median_s=med, min_s=ts_[0], max_s=ts_[-1], estimates_per_s=1.0 / med,
                        eps_generation_s=med_eps, estimate_s=med_est, gflops_estimate=fl / med_est / 1e9)
    wall = time.perf_counter() - t_leg0
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    best = max(legs.values(), key=lambda l: l["estimates_per_s"])
    blas = None
    if fam == 1:
        blas = cpu_baseline_blas(lib, CO, w, params, tm, ts, budget_s=8.0)
    sample = (f"median of {best['reps']} whole estimates of the same (d={d}, n_mc={M}) workload incl. eps generation, f32, OpenMP "
              f"{best['threads']} threads on '{model}' ({avail} CPUs available), {best['gflops_estimate']:.0f} GFLOP/s in the estimate "
              f"(2 d^2 n_mc flops, eps generation {best['eps_generation_s'] * 1e3:.2f} ms of {best['median_s'] * 1e3:.2f} ms), leg wall time {wall:.1f} s")
    value, cores, leg = best["estimates_per_s"], best["threads"], "c_port"
    if blas and blas["estimates_per_s"] > value:   # the CPU's best foot forward: whichever leg is faster is the reported baseline
        value, cores, leg = blas["estimates_per_s"], avail, "blas"
        sample = (f"median of {blas['reps']} whole estimates of the same (d={d}, n_mc={M}) workload, f32: both contractions on {blas['blas']} "
                  f"(all {avail} CPUs of '{model}', {blas['gflops_executed']:.0f} GFLOP/s executed), numpy ziggurat normals + numpy elementwise "
                  f"work included ({blas['eps_predrawn']['estimates_per_s']:.0f} estimates/s with eps pre-drawn); the C port's legs are in thread_scaling")
    return dict(value=value, unit="ELBO-grad-estimates/s", cores=cores, kind="port", leg=leg, build=build, blas=blas, sample=sample, cpu=model, cpus_available=avail,
                gflops=(best["gflops_estimate"] if leg == "c_port" else blas["gflops_algorithmic"]),
                one_thread=legs.get(1), all_cores=legs.get(avail), thread_scaling=[legs[t] for t in teams], threads=lib.mo32_max_threads())


