#!/usr/bin/env python
"""bench.py -- ELBO-gradient estimates / second of the RepGradELBO hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by
torch.distributed.run with one rank per GPU.  A *step* is one full `estimate_gradient!` (objective value
+ complete gradient vector) over one synthetic batch with every input resident in HBM.  W untimed warm-up
steps, then exactly K timed steps bracketed by barrier + synchronize, max over ranks, ONE JSON line on rank 0.

Workloads (BASELINE.json `configs` / north star):
  ns  (default) north-star: d=1024 full-rank Gaussian family, n_mc=256 per GPU, target MvNormal(5*1, I)
                (the reference's own bench target, bench/benchmarks.jl:43-47), ClosedFormEntropy, f32
  c2            BASELINE configs[1]: d=1024 mean-field, n_mc=256, same target, f32
  ns_dense      north-star family with the dense-Gaussian target N(m, L L') of SURVEY.md 8d
N > 1: weak scaling -- every rank draws its own n_mc-sample shard of ONE estimate of n_mc*N samples
(shard-invariant Philox stream); the partial vectors are summed by the peer-to-peer exchange kernel written for xGMI
(csrc/kernels_p2p.hip; RCCL when the areas cannot be mapped), the exchange of estimate t overlapped with the kernels of
estimate t+1; `value` counts n_mc-sample estimate units processed by all ranks per second; `dist` holds the route taken and the
per-stage times {partials, exchange, serial, pipelined}.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x38BEF07CF9CC549D
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured-achievable)
PEAK_F32_MFMA_TF = 157.3     # MI355X_MICROARCH.md: dense f32 MFMA peak (155 TF measured)
PEAK_BF16_MFMA_TF = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak
PLANE_BYTES = 4                 # bytes per operand-plane element of the batch engine (f16 hi + lo planes; ctx.plane_bytes())
PEAK_VALU_GINST = 1024 * 2.4 / 4   # wave64 vector instructions per ns: 256 CUs x 4 SIMDs, one per 4 cycles, 2.4 GHz (MI355X_MICROARCH.md)

# environment variables that do NOT change which kernels run: bench-harness controls and the RCCL library location
BENCH_ENV_OK = {"MIVI_FORCE_DIST", "MIVI_DIST_MODE", "MIVI_DIST_EAGER", "MIVI_BENCH_SKIP_C3", "MIVI_RCCL_LIB", "MIVI_DIST_PIPELINE"}

WORKLOADS = {
    "ns": dict(family=1, d=1024, n_mc=256, target="iso", entropy=0,
               name="north-star: d=1024 full-rank Gaussian family, n_mc=256, target MvNormal(5*1, I), ClosedFormEntropy"),
    "c2": dict(family=0, d=1024, n_mc=256, target="iso", entropy=0,
               name="configs[1]: d=1024 mean-field MvLocationScale, n_mc=256, target MvNormal(5*1, I), ClosedFormEntropy"),
    "ns_dense": dict(family=1, d=1024, n_mc=256, target="dense", entropy=0,
                     name="north-star family, dense-Gaussian target N(5*1, L L'), L = tril(I + 11'/(2d))"),
    "c3": dict(family=1, d=512, n_mc=128, target="logreg", entropy=0, n=1_000_000,
               name="configs[2]: hierarchical LogReg n=1e6, D=512 (511 coefficients + log sigma), full-rank q0=(0, 0.6 I), n_mc=128"),
    "c5": dict(family=0, d=2048, n_mc=64, target="funnel", entropy=3,
               name="configs[4] per-GPU shard: funnel d=2048 + Stacked bijector, mean-field, STL, 64 samples per GPU"),
    "ns_stl": dict(family=1, d=1024, n_mc=256, target="iso", entropy=3,
                   name="north-star family, StickingTheLandingEntropy (adds the C^-T eps solve)"),
}


def algorithmic_cost(w):
    """SURVEY.md 8(d) per-estimate figures (s = 4 bytes): bytes and flops of one estimate."""
    d, M, s = w["d"], w["n_mc"], 4
    if w["family"] == 0:
        return dict(bytes=4 * d * M * s + 4 * d * s, flops=6 * d * M)
    return dict(bytes=(d * (d + 1) // 2) * s + d * d * s + 4 * d * M * s + 2 * d * s, flops=2 * d * d * M)


for _k, _v in WORKLOADS.items():
    _v["key"] = _k


def pmc_traffic(kernel_substr, lanes=None):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (profiles/pmc_traffic.json, produced by tools/pmc_traffic.py: separate passes, counters in KiB, the gfx950 FETCH_SIZE
    correction of MI355X_MICROARCH.md already applied there per kernel according to the width of its loads -- the file records
    the factor it used and the calibration run it came from).  Batch-engine kernels are kept per lane count there (`by_lanes`, the lanes
    derived from every dispatch's grid): `lanes` picks that entry, or the nearest one (the caller scales per lane and says so)."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return None
    for k, v in tab.get("kernels", {}).items():
        if kernel_substr in k:
            if lanes and v.get("by_lanes"):
                key = min(v["by_lanes"], key=lambda x: abs(int(x) - lanes))
                v = v["by_lanes"][key]
            elif v.get("by_lanes") or "k_fb_" in k:   # (no lane count asked for / a round-4 file whose lane count was assumed, not derived)
                return None
            out = dict(bytes_per_launch=(v["fetch_kib"] + v["write_kib"]) * 1024.0, fetch_bytes=v["fetch_kib"] * 1024.0,
                       write_bytes=v["write_kib"] * 1024.0, source=tab.get("source"), profile="profiles/pmc_traffic.json")
            if "lanes_per_launch" in v:   # the batch engine: the profiled launches carried this many estimates
                out["lanes_per_launch"] = v["lanes_per_launch"]
            return out
    return None


def pmc_valu(kernel_substr):
    """Wave-level VALU instruction count per launch of a kernel (SQ_INSTS_VALU, its own rocprofv3 --pmc pass: tools/pmc_valu.sh ->
    profiles/pmc_valu.json)."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json")))
    except (OSError, ValueError):
        return None
    for k, v in tab.get("kernels", {}).items():
        if kernel_substr in k and "SQ_INSTS_VALU" in v:
            return dict(insts_valu=v["SQ_INSTS_VALU"], insts_salu=v.get("SQ_INSTS_SALU"), waves=v.get("SQ_WAVES"),
                        active_inst_valu_cycles=v.get("SQ_ACTIVE_INST_VALU"), busy_cycles=v.get("SQ_BUSY_CYCLES"),
                        avg_ns_under_the_profiler=v.get("avg_ns"), source=tab.get("source"))
    return None


def rocprof_avg(kernel_substr, workload="ns", lanes=None):
    """Average duration (us) of a kernel in the newest committed `rocprofv3 --kernel-trace --stats` summary of this workload's bench command
    (profiles/<tag>_<workload>_kernel_stats.md, written by tools/profile_round.sh): the in-chain figure next to the stand-alone one this
    process measures.  With `lanes`: the row of the summary's per-grid table whose launches carried exactly that many estimates (the lane
    count is derived from the grid there), preferring the summary taken at the driver's own --steps 20 (`<tag>_<workload>20_...`)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_kernel_stats.md" % workload)) +
                   glob.glob(os.path.join(ROOT, "profiles", "r*_%s20_kernel_stats.md" % workload)),
                   key=lambda f: os.path.basename(f).split("_")[:2])
    for f in reversed(files):
        try:
            hit = None
            for line in open(f):
                if kernel_substr not in line or not line.startswith("|"):
                    continue
                c = [x.strip() for x in line.strip().strip("|").split("|")]
                if len(c) == 7 and "x" in c[1] and not c[1].isdigit():        # per-grid table: kernel | workgroups | lanes | calls | avg_ns | min | max
                    if lanes and c[2].isdigit() and int(c[2]) == lanes:
                        return dict(avg_us=float(c[4]) / 1e3, calls=int(c[3]), lanes=lanes, source=os.path.relpath(f, ROOT))
                elif hit is None and len(c) >= 4 and c[1].isdigit():
                    hit = dict(avg_us=float(c[3]) / 1e3, calls=int(c[1]), lanes=None, source=os.path.relpath(f, ROOT))
            if hit and not lanes:
                return hit
        except (OSError, ValueError, IndexError):
            continue
    return None


def mf_roofline(ctx, params, cost, kernel_names=("k_mf_main<float>", "k_mf_sgd_loop<float> (100 estimates per launch)")):
    """Mean-field roofline leg.  Batched estimates (estimate_gradient_n, what the bench line times) run 100 estimates per
    launch of the launch-free loop kernel; a single call is one launch of the fused main kernel -- both are reported."""
    try:
        ms1 = ctx.profile_kernel(2, params, 300)
    except Exception:   # noqa: BLE001  -- no stand-alone stage hook for this target (fused funnel): whole single estimate, eager
        ms1 = ctx.profile_kernel(0, params, 300)
    single = dict(kernel=kernel_names[0], avg_launch_us=ms1 * 1e3, achieved=cost["bytes"] / (ms1 * 1e-3) / 1e9,
                  frac=cost["bytes"] / (ms1 * 1e-3) / 1e9 / PEAK_HBM_GBS, traffic=pmc_traffic("k_mf_main"))
    try:
        msl = ctx.profile_kernel(5, params, 30)
    except Exception:   # noqa: BLE001  -- loop not applicable (other target / MIVI_NO_FUSED_LOOP semantics unchanged)
        msl = None
    if msl is None:
        return dict(bound="hbm", kernel=single["kernel"], achieved=single["achieved"], peak=PEAK_HBM_GBS, unit="GB/s",
                    frac=single["frac"], traffic=single["traffic"], algorithmic_bytes_per_launch=cost["bytes"],
                    avg_launch_us=single["avg_launch_us"]), {"mf_fused_main": ms1}
    ach = 100 * cost["bytes"] / (msl * 1e-3) / 1e9
    loop_kernel = "k_mf_funnel_loop" if "funnel" in kernel_names[1] else "k_mf_sgd_loop"
    hbm_eq = dict(achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS, algorithmic_bytes_per_launch=100 * cost["bytes"],
                  note=("HBM-EQUIVALENT of SURVEY 8d's algorithmic bytes: Z and G never exist in memory (the launch moves `traffic.bytes_per_launch`, "
                        "about 1 % of them), so this fraction is not bounded by 1"))
    valu = pmc_valu(loop_kernel)
    times = {"mf_fused_main": ms1, "mf_loop_per_estimate": msl / 100}
    if valu is None:
        return dict(bound="valu", kernel=kernel_names[1], achieved=None, peak=PEAK_VALU_GINST, unit="G wave-instructions/s", frac=None,
                    traffic=pmc_traffic(loop_kernel), estimates_per_launch=100, avg_launch_us=msl * 1e3, single_call=single, hbm_equivalent=hbm_eq,
                    note="vector-ALU bound (Philox + Box-Muller + wave reductions); no profiles/pmc_valu.json with this kernel's SQ_INSTS_VALU: "
                         "tools/pmc_valu.sh collects it"), times
    # the kernel is vector-ALU bound: wave-level VALU instructions per launch (SQ_INSTS_VALU, own rocprofv3 pass) over the live launch time,
    # against 1024 SIMDs x one wave64 instruction per 4 cycles x 2.4 GHz
    g = valu["insts_valu"] / (msl * 1e-3) / 1e9
    return dict(bound="valu", kernel=kernel_names[1], achieved=g, peak=PEAK_VALU_GINST, unit="G wave-instructions/s", frac=g / PEAK_VALU_GINST,
                traffic=pmc_traffic(loop_kernel), estimates_per_launch=100, avg_launch_us=msl * 1e3, single_call=single,
                valu_insts_per_launch=valu["insts_valu"], valu_insts_per_estimate=valu["insts_valu"] / 100.0,
                valu_bound_us_per_estimate=valu["insts_valu"] / 100.0 / PEAK_VALU_GINST / 1e3, measured_us_per_estimate=msl * 1e3 / 100,
                pmc=valu, hbm_equivalent=hbm_eq,
                note=("vector-ALU roofline: SQ_INSTS_VALU per launch (profiles/pmc_valu.json; wave-level, every instruction priced at 4 cycles -- "
                      "transcendentals and f64 cost more, so the true bound is tighter) / live launch time vs 1024 SIMDs / 4 cycles x 2.4 GHz; "
                      "the HBM-equivalent of SURVEY 8d's algorithmic bytes is under hbm_equivalent")), times


def fr_roofline(ctx, params, cost, w, reps=300, lanes=0):
    """Full-rank roofline leg: graph-replayed launches of each stage (mivi_profile_kernel), the slower of the two
    contractions is the dominant kernel.  Both carry d^2*M algorithmic flops (lower triangle only).  On the second-generation
    route the products run on the bf16 matrix cores with the exact three-way operand split (six bf16 MFMAs per product
    block): `frac` stays f32-equivalent flops / the f32-MFMA peak the north star is priced against, and `bf16_pipe` says what
    the matrix pipe actually executes."""
    gen, bf3 = ctx.fullrank_route()
    stages = {"eps": ctx.profile_kernel(1, params, reps), "sample": ctx.profile_kernel(2, params, reps),
              "vjp": ctx.profile_kernel(3, params, reps)}
    if w["target"] == "dense":
        stages["dense_target"] = ctx.profile_kernel(4, params, reps)
    dom = "vjp" if stages["vjp"] >= stages["sample"] else "sample"
    names = {0: {"vjp": "k_fr_tile_mfma<MODE_VJP,4>", "sample": "k_fr_tile_mfma<MODE_SAMPLE,8>"},
             1: {"vjp": "k_fr_vjp32", "sample": "k_fr_prod32<SAMPLE> (product + fused target)"},
             2: {"vjp": "k_fr_vjp32", "sample": "k_fr_gemm<SAMPLE> + k_fr_reduce (split-K)"},
             3: {"vjp": "k_fr_vjp32 / k_fr_vjp64", "sample": "k_fr_prod64<SAMPLE> (product + fused target)"}}[gen]
    fl = cost["flops"] / 2
    ach = fl / (stages[dom] * 1e-3) / 1e12
    roof = dict(bound="mfma", kernel=names[dom], achieved=ach, peak=PEAK_F32_MFMA_TF, unit="TFLOP/s", frac=ach / PEAK_F32_MFMA_TF,
                traffic=pmc_traffic({"vjp": "k_fr_vjp32", "sample": "k_fr_prod32"}[dom] if gen else ("mfmaILi1" if dom == "vjp" else "mfmaILi0")),
                algorithmic_flops_per_launch=fl, avg_launch_us=stages[dom] * 1e3,
                timing="hipGraph replay of %d launches, hipEvents on the launch stream" % reps,
                other_contraction=dict(kernel=names["sample" if dom == "vjp" else "vjp"],
                                       avg_launch_us=stages["sample" if dom == "vjp" else "vjp"] * 1e3,
                                       achieved=fl / (stages["sample" if dom == "vjp" else "vjp"] * 1e-3) / 1e12))
    # Batches of estimates at the BASELINE sizes (what the bench line times) are LANE-BATCHED: one product launch and one VJP launch serve
    # FOUR estimates (k_fr_prod32q / k_fr_vjp32s).  Those are the launches of the timed region: the dominant one becomes the headline of the
    # block (4 x the algorithmic flops per launch), the one-estimate kernels stay in `single_launch` (an optimisation loop runs those).
    if gen == 1 and bf3 and w["target"] == "iso":
        try:
            t4 = {"sample": ctx.profile_kernel(10, params, reps), "vjp": ctx.profile_kernel(11, params, reps)}
        except Exception:   # noqa: BLE001  -- configuration outside the lane-batched route
            t4 = None
        if t4:
            stages["sample_4_lanes"], stages["vjp_4_lanes"] = t4["sample"], t4["vjp"]
            d4 = "vjp" if t4["vjp"] >= t4["sample"] else "sample"
            o4 = "sample" if d4 == "vjp" else "vjp"
            n4 = {"sample": "k_fr_prod32q (four estimates' products + fused target per launch, with the next eps draws riding)",
                  "vjp": "k_fr_vjp32s (four estimates' VJP per launch, strips of tiles)"}
            a4 = 4 * fl / (t4[d4] * 1e-3) / 1e12
            single = {k: roof[k] for k in ("kernel", "achieved", "frac", "traffic", "algorithmic_flops_per_launch", "avg_launch_us", "other_contraction")}
            roof.update(kernel=n4[d4], achieved=a4, frac=a4 / PEAK_F32_MFMA_TF, algorithmic_flops_per_launch=4 * fl, estimates_per_launch=4,
                        avg_launch_us=t4[d4] * 1e3, traffic=pmc_traffic({"vjp": "k_fr_vjp32s", "sample": "k_fr_prod32q"}[d4]),
                        other_contraction=dict(kernel=n4[o4], avg_launch_us=t4[o4] * 1e3, achieved=4 * fl / (t4[o4] * 1e-3) / 1e12,
                                               rocprof_in_chain=rocprof_avg({"vjp": "k_fr_vjp32s", "sample": "k_fr_prod32q"}[o4])),
                        rocprof_in_chain=rocprof_avg({"vjp": "k_fr_vjp32s", "sample": "k_fr_prod32q"}[d4]),
                        timing="hipGraph replay of %d launches of the four-lane kernel alone, hipEvents on the launch stream (rocprof_in_chain: "
                               "the same kernel inside the timed batches, the other branch's kernels beside it)" % reps,
                        single_launch=single)
            ach = a4
    # Batches on the BATCH ENGINE (full-rank family, diagonal-Gaussian target; csrc/kernels_fullrank_batch.hip): a step of the timed region is
    # three launches -- draws, product + target, VJP + values -- that cover ALL the lanes of the step (`lanes` = estimates per call of the
    # timed loop).  The dominant launch is the headline of the block: lanes x the algorithmic flops per launch / its duration, measured live
    # (hipEvents around back-to-back launches on the launch stream: nothing runs beside these kernels in the timed region either, so the
    # stand-alone figure IS the in-chain one; `rocprof_in_chain` quotes the committed rocprofv3 summary of the bench command next to it).
    if lanes and w["target"] in ("iso", "dense"):
        try:
            lanes = ctx.batch_lanes(lanes) or lanes      # a call of `lanes` estimates runs as equal steps of this many lanes
            tb = ctx.profile_batch(params, lanes, max(5, reps // 10))
        except Exception:   # noqa: BLE001  -- configuration outside the batch engine
            tb = None
        if tb:
            for k, v in tb.items():
                if v > 0.0:
                    stages["batch_%s_%d_lanes" % (k, lanes)] = v * 1e-3
            nm = {"product": "k_fb_prod<FB_DIAG> (tril(C) [eps_1 .. eps_L] + fused target for all lanes of a step, operands as f16 hi/lo planes in MFMA-fragment order)",
                  "vjp": "k_fb_vjp (tril(W_l eps_l') for all lanes of a step + their values)",
                  "dense_product": "k_fb_prod<FB_DENSE_G> (the dense target's -P (Z_l - m) for all lanes of a step)",
                  "stl_product": "k_fb_prod<FB_STL_U> (W_l += C^-T eps_l for all lanes of a step, C^-T formed once per call)"}
            if w["target"] == "dense":
                nm["product"] = "k_fb_prod<FB_DENSE_R> (tril(C) [eps_1 .. eps_L] -> R = Z - m as operand planes)"
            sub = {"product": "k_fb_prodILi1ELi%dE" % (1 if w["target"] == "dense" else 0), "vjp": "k_fb_vjp",
                   "dense_product": "k_fb_prodILi1ELi2E", "stl_product": "k_fb_prodILi1ELi3E"}   # (mangled template arguments: WJ, MODE; then the ring depth)
            kfl = {"product": fl, "vjp": fl, "dense_product": 2 * fl, "stl_product": fl}   # algorithmic flops per estimate of each launch
            live = [k for k in ("product", "vjp", "dense_product", "stl_product") if tb.get(k, 0.0) > 0.0]
            dk = max(live, key=lambda k: tb[k])
            aL = lanes * kfl[dk] / (tb[dk] * 1e-6) / 1e12

            def traffic_of(k):
                # HBM-side bytes of one launch at THIS lane count: tools/pmc_traffic.py keeps the batch-engine kernels per grid size (lanes
                # derived from the dispatch's grid, not assumed); another lane count's entry is scaled per lane and says so
                t = pmc_traffic(sub[k], lanes)
                if t and t.get("lanes_per_launch") and t["lanes_per_launch"] != lanes:
                    f = lanes / float(t["lanes_per_launch"])
                    t = dict(t, bytes_per_launch=t["bytes_per_launch"] * f, fetch_bytes=t["fetch_bytes"] * f, write_bytes=t["write_bytes"] * f,
                             scaled_from_lanes=t["lanes_per_launch"], lanes_per_launch=lanes)
                if t:
                    d_, M_ = w["d"], w["n_mc"]
                    alg = {"product": d_ * (d_ + 1) // 2 * 4 + lanes * 2 * d_ * M_ * 4, "vjp": lanes * (2 * d_ * M_ * 4 + d_ * d_ * 4),
                           "dense_product": d_ * d_ * 4 + lanes * 2 * d_ * M_ * 4, "stl_product": d_ * (d_ + 1) // 2 * 4 + lanes * 3 * d_ * M_ * 4}[k]
                    t["algorithmic_bytes_per_launch"] = alg
                    t["over_algorithmic"] = t["bytes_per_launch"] / alg
                    t["GBs"] = t["bytes_per_launch"] / (tb[k] * 1e-6) / 1e9          # the counters' bytes over THIS run's launch time
                    t["frac_of_8TBs"] = t["GBs"] / PEAK_HBM_GBS
                return t
            keep = {k: roof[k] for k in ("kernel", "achieved", "frac", "traffic", "algorithmic_flops_per_launch", "avg_launch_us", "other_contraction")}
            if "single_launch" in roof:
                keep = roof["single_launch"]
            # ONE basis for every kernel of the block.  On two f16 planes a product block costs three matrix-pipe products, so SURVEY 8d's
            # algorithmic flops no longer bound these kernels (the 16-bit pipe would finish them in a third of the time the f32-MFMA peak
            # allows): the binding roof is the MEMORY side, as the north star's own target says ("% of HBM roofline").  `achieved` = SURVEY 8d's
            # algorithmic BYTES of the launch / its duration, `peak` = 8 TB/s; beside it, always, the same launch's f32-accurate flops over
            # the f32-MFMA peak (`frac_f32_mfma`: may exceed 1) and the executed 16-bit-pipe flops over 2.5 PF (`frac_16bit_pipe`, <= 1).
            nprod = ctx.split_products()
            d_, M_ = w["d"], w["n_mc"]
            alg_bytes = {"product": d_ * (d_ + 1) // 2 * 4 + lanes * 2 * d_ * M_ * 4, "vjp": lanes * (2 * d_ * M_ * 4 + d_ * d_ * 4),
                         "dense_product": d_ * d_ * 4 + lanes * 2 * d_ * M_ * 4, "stl_product": d_ * (d_ + 1) // 2 * 4 + lanes * 3 * d_ * M_ * 4}
            # ... and the bytes the VJP launch MOVES as mivi_estimate_gradient_n lays it out (round 5's verdict: SURVEY 8d charges every lane a
            # dense d^2 gradient write, but only the caller's lane writes the zeros above the diagonal -- the scratch lanes write the 128 x 128
            # tiles of the lower triangle): W and eps planes read once, T (T + 1) / 2 tiles per scratch lane, d^2 for the caller's, d/dmu.
            # `roofline.achieved` / `frac` are on THESE bytes (they agree with the PMC counters to a few percent: `traffic`); the SURVEY 8d
            # figure stays beside them as `frac_survey_8d`.
            T_ = d_ // 128
            moved_vjp = lanes * (2 * d_ * M_ * 4 + d_ * 4) + (lanes - 1) * (T_ * (T_ + 1) // 2) * 128 * 128 * 4 + d_ * d_ * 4
            survey_bytes = dict(alg_bytes)
            alg_bytes["vjp"] = moved_vjp

            def tf(k):
                return lanes * kfl[k] / (tb[k] * 1e-6) / 1e12

            def gbs(k):
                return alg_bytes[k] / (tb[k] * 1e-6) / 1e9
            others = [dict(kernel=nm[k], avg_launch_us=tb[k], achieved=gbs(k), frac=gbs(k) / PEAK_HBM_GBS, frac_f32_mfma=tf(k) / PEAK_F32_MFMA_TF,
                           frac_16bit_pipe=nprod * tf(k) / PEAK_BF16_MFMA_TF, rocprof_in_chain=rocprof_avg(sub[k], w.get("key", "ns"), lanes))
                      for k in live if k != dk]
            roof.update(bound="hbm", kernel=nm[dk], achieved=gbs(dk), peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs(dk) / PEAK_HBM_GBS,
                        algorithmic_bytes_per_launch=alg_bytes[dk], survey_8d_bytes_per_launch=survey_bytes[dk],
                        achieved_survey_8d=survey_bytes[dk] / (tb[dk] * 1e-6) / 1e9, frac_survey_8d=survey_bytes[dk] / (tb[dk] * 1e-6) / 1e9 / PEAK_HBM_GBS,
                        algorithmic_flops_per_launch=lanes * kfl[dk], estimates_per_launch=lanes,
                        avg_launch_us=tb[dk], traffic=traffic_of(dk), rocprof_in_chain=rocprof_avg(sub[dk], w.get("key", "ns"), lanes),
                        other_contraction=others[0] if len(others) == 1 else others,
                        draws=dict(kernel="k_fb_eps (Philox + Box-Muller draws of all lanes as operand planes in both orientations)", avg_launch_us=tb["eps"],
                                   algorithmic_bytes_per_launch=lanes * 2 * PLANE_BYTES * w["d"] * w["n_mc"],
                                   achieved_GBs=lanes * 2 * PLANE_BYTES * w["d"] * w["n_mc"] / (tb["eps"] * 1e-6) / 1e9,
                                   note="%d bytes per element and orientation written once: bound by the memory side and the vector ALU" % PLANE_BYTES),
                        timing="%d back-to-back launches of each kernel for %d lanes, hipEvents on the launch stream" % (max(5, reps // 10), lanes),
                        basis="achieved = bytes the launch moves (vjp: W + eps planes read, lower-triangle tiles written per scratch lane, dense d^2 for the caller's lane) / launch time; peak = HBM 8 TB/s; frac_survey_8d = SURVEY 8d's lanes x (2 d n_mc + d^2) x 4 B instead; traffic = PMC counters; frac_f32_mfma = d^2 n_mc flops x lanes / time / 157.3 TF; frac_16bit_pipe = x%d executed / 2500 TF" % nprod,
                        f32_mfma=dict(achieved_TFLOPs=aL, peak=PEAK_F32_MFMA_TF, frac=aL / PEAK_F32_MFMA_TF),
                        pipe16=dict(products_per_block=nprod, executed_TFLOPs=nprod * aL, peak=PEAK_BF16_MFMA_TF, frac=nprod * aL / PEAK_BF16_MFMA_TF),
                        single_launch=keep)
            roof.pop("estimates_per_launch_note", None)
            return roof, stages
    if gen and bf3:
        roof["bf16_pipe"] = dict(mfma="v_mfma_f32_32x32x16_bf16 x6 per product block (exact 3-way f32 split)",
                                 executed_TFLOPs=6 * ach, peak=PEAK_BF16_MFMA_TF, frac=6 * ach / PEAK_BF16_MFMA_TF)
    return roof, stages


def other_roofline(cx, p, w, t_est):
    """Roofline block of the workloads whose dominant kernel is not one of the two full-rank contractions:
    C3 (logistic regression: the two data contractions), C5 / other mean-field targets (HBM)."""
    cost = algorithmic_cost(w)
    if w["target"] == "logreg":
        n, pdim, M2 = w["n"], w["d"] - 1, w["n_mc"]
        fl = 4.0 * n * pdim * M2                      # logits X beta and X^T R, 2 flops per MAC
        by = 2.0 * n * pdim * 4 + 2.0 * n * M2 * 4     # X read once per contraction, R written + read
        tl, tx = pmc_traffic("k_lr_logits_planes"), pmc_traffic("k_lr_xtr_planes")
        # the two data contractions stream X's operand planes once each (SURVEY 8d prices ONE fused pass: 2.05 GB); with three 16-bit products per block the
        # binding roof is the memory side: `achieved` = SURVEY 8d's algorithmic bytes / whole-estimate time against 8 TB/s, the flop fractions beside
        alg = float(n) * (pdim + 1) * 4 + float(n)                      # X (padded to D columns) once + y
        return dict(bound="hbm", kernel="k_lr_logits_planes + k_lr_xtr_planes", achieved=alg / t_est / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                    frac=alg / t_est / 1e9 / PEAK_HBM_GBS, algorithmic_bytes_per_launch=alg,
                    basis="achieved = SURVEY 8d algorithmic bytes (one pass over X, 2.05 GB) / whole-estimate time; peak = HBM 8 TB/s; the kernels read X's f16x2 planes twice (one orientation per contraction, built once per data set) + the residual planes once each way",
                    f32_mfma=dict(achieved_TFLOPs=fl / t_est / 1e12, peak=PEAK_F32_MFMA_TF, frac=fl / t_est / 1e12 / PEAK_F32_MFMA_TF),
                    pipe16=dict(products_per_block=3, executed_TFLOPs=3 * fl / t_est / 1e12, peak=PEAK_BF16_MFMA_TF, frac=3 * fl / t_est / 1e12 / PEAK_BF16_MFMA_TF),
                    hbm_executed=dict(achieved_GBs=by / t_est / 1e9, peak=PEAK_HBM_GBS, frac=by / t_est / 1e9 / PEAK_HBM_GBS, bytes_per_estimate=by),
                    traffic=(dict(bytes_per_launch=tl["bytes_per_launch"] + tx["bytes_per_launch"], logits=tl, xtr=tx) if tl and tx else None),
                    avg_launch_us=t_est * 1e6)
    try:
        roof, _ = mf_roofline(cx, p, cost, ("k_mf_main<float, funnel> + k_value_funnel (single call, eager)",
                                            "k_mf_funnel_loop<float> + k_mf_funnel_loop_value (100 estimates per launch pair)"))
        roof["note"] = ("HBM-equivalent of SURVEY 8d's algorithmic bytes; the kernel is VALU bound (two Philox blocks + exp per lane and "
                        "estimate), real traffic is the gradients and the per-estimate partials")
        return roof
    except Exception:   # noqa: BLE001  -- stage hook not applicable to this target: whole-estimate HBM equivalent
        return dict(bound="hbm", kernel="k_mf_main<float, funnel> (one launch per estimate; the previous estimate's value / row-0 finisher rides in it)",
                    achieved=cost["bytes"] / t_est / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", frac=cost["bytes"] / t_est / 1e9 / PEAK_HBM_GBS,
                    traffic=pmc_traffic("k_mf_mainIfLb1"), algorithmic_bytes_per_launch=cost["bytes"], avg_launch_us=t_est * 1e6,
                    note="whole estimate (hipGraph steady state); launch / latency bound: 0.27 MB of real traffic per estimate")


def stl_block(cx, p, w, reps=100):
    """The sticking-the-landing term of a full-rank workload: W += C^-T eps (two half-size chain solves + one update product),
    hipGraph-replayed alone (mivi_profile_kernel which = 8).  Algorithmic flops d^2 M (a triangular solve with M right-hand sides)."""
    try:
        ms = cx.profile_kernel(8, p, reps)
    except Exception:   # noqa: BLE001
        return None
    fl = float(w["d"]) ** 2 * w["n_mc"]
    sv, up = pmc_traffic("k_stl_solve64"), pmc_traffic("k_stl_update32")
    return dict(kernel="k_stl_solve64 (three half-size solves side by side: X2, Y1, F) + k_stl_update32 (X1 = Y1 - F^T X2)", avg_us=ms * 1e3,
                achieved_TFLOPs=fl / (ms * 1e-3) / 1e12, frac_of_f32_mfma_peak=fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TF,
                algorithmic_flops=fl, executed_flops=2.0 * fl,   # the parameter-only coupling solve has d/2 right-hand sides of its own
                bound="dependency chain: d/128 block steps per 16-column workgroup (2 M/16 + d/32 CUs busy), each pulling its half-triangle through one CU",
                traffic=(dict(solve=sv, update=up) if sv and up else None))


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


LINE_LIMIT = 4096   # bytes: the driver keeps an 8 KiB stdout tail and parses the LAST line (round 4's 25.7 KB line came back `parsed: null`)


def _num(x, sig=6):
    """A float rounded to `sig` significant digits (None / non-numbers pass through)."""
    if isinstance(x, bool) or not isinstance(x, (int, float)):
        return x
    if isinstance(x, int) or x == 0 or x != x or x in (float("inf"), float("-inf")):
        return x
    return float("%.*g" % (sig, x))


def _get(o, *path, default=None):
    for k in path:
        if not isinstance(o, dict) or k not in o or o[k] is None:
            return default
        o = o[k]
    return o


def compact_line(full):
    """The ONE stdout line the driver parses, built from the full result dict: <= LINE_LIMIT bytes, every contract key, `roofline` and
    `cpu_baseline` as flat objects, one number per `also` leg.  Everything else (per-kernel traffic blocks, thread scaling, stage times,
    notes) lives in the full file (`full`: gpurun_out/bench_full.json) and on stderr.  Pure function of its argument: tests/test_bench_line.py
    feeds it canned dicts on the CPU."""
    roof = full.get("roofline") or None
    r = None
    if roof:
        tr = roof.get("traffic") or None
        lanes = roof.get("estimates_per_launch", 1)
        ric = roof.get("rocprof_in_chain") or None
        r = {
            "bound": roof.get("bound"), "kernel": str(roof.get("kernel", ""))[:96],
            "achieved": _num(roof.get("achieved")), "peak": roof.get("peak"), "unit": roof.get("unit"), "frac": _num(roof.get("frac"), 4),
            # the same kernel on the pipe it executes on: split-operand products run on the 16-bit matrix pipe (2.5 PFLOP/s dense)
            "frac_survey_8d": _num(roof.get("frac_survey_8d"), 4),
            "frac_16bit_pipe": _num(_get(roof, "pipe16", "frac"), 4), "frac_f32_mfma": _num(_get(roof, "f32_mfma", "frac"), 4),
            "basis": str(roof.get("basis", ""))[:330] or None,
            "avg_launch_us": _num(roof.get("avg_launch_us"), 5), "lanes": lanes,
            "traffic": (None if not tr else {"bytes_per_launch": _num(tr.get("bytes_per_launch")), "lanes": tr.get("lanes_per_launch", lanes),
                                             "over_algorithmic": _num(tr.get("over_algorithmic"), 3), "GBs": _num(tr.get("GBs"), 5), "frac_of_8TBs": _num(tr.get("frac_of_8TBs"), 4),
                                             "src": str(tr.get("profile", ""))[:64] or None}),
            "rocprof_in_chain": (None if not ric else {"avg_us": _num(ric.get("avg_us"), 5), "lanes": ric.get("lanes"), "src": str(ric.get("source", ""))[:64]}),
        }
        oc = roof.get("other_contraction")
        if isinstance(oc, list):
            oc = oc[0] if oc else None
        if oc:
            r["other"] = {"kernel": str(oc.get("kernel", ""))[:48], "avg_launch_us": _num(oc.get("avg_launch_us"), 5), "frac": _num(oc.get("frac"), 4),
                          "frac_f32_mfma": _num(oc.get("frac_f32_mfma"), 4)}
        if roof.get("draws"):
            r["draws"] = {"avg_launch_us": _num(_get(roof, "draws", "avg_launch_us"), 5), "GBs": _num(_get(roof, "draws", "achieved_GBs"), 4)}
        we = full.get("whole_estimate") or {}
        r["whole_estimate"] = {"hbm_frac_of_8TBs": _num(we.get("hbm_equiv_frac_of_8TBs"), 4), "f32_mfma_TFs": _num(we.get("f32_mfma_TFs"), 4)}
    cb = full.get("cpu_baseline") or None
    c = None
    if cb:
        c = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "threads": cb.get("threads"),
             "kind": cb.get("kind"), "leg": cb.get("leg"), "cpu": str(cb.get("cpu", ""))[:48], "sample": str(cb.get("sample", ""))[:200],
             "one_thread": _num(_get(cb, "one_thread", "estimates_per_s"))}
    also = None
    if full.get("also"):
        also = {}
        for k, v in full["also"].items():
            if not isinstance(v, dict):
                continue
            if "error" in v:
                also[k] = None
            elif "value" in v:
                also[k] = _num(v["value"], 5)
                if k.endswith("_loop") and "us_per_step" in v:
                    also[k + "_us"] = _num(v["us_per_step"], 4)
            elif "us_per_call" in v:
                also[k] = _num(1e6 / v["us_per_call"], 5)
        if isinstance(full["also"].get("ns_f64"), dict) and "frac_f64_mfma" in full["also"]["ns_f64"]:
            also["ns_f64_frac_f64_mfma"] = _num(full["also"]["ns_f64"]["frac_f64_mfma"], 3)
        also["units"] = "estimates/s (c2 ns_dense ns_stl c5 c3 ns_host_boundary ns_f64), steps/s (*_loop, reference_benchmark_grid[_f64]), calls/s (stein), samples/s (ns_objective_1e5)"
    cfg = dict(full.get("config") or {})
    cfg["launch"] = str(cfg.get("launch", ""))[:200]
    cfg["workload"] = str(cfg.get("workload", ""))[:128]
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    line["value"] = _num(line["value"], 7)
    line["ms_per_step"] = _num(line["ms_per_step"], 6)
    line["config"] = {k: cfg.get(k) for k in ("workload", "d", "n_mc_per_gpu", "n_mc_total", "family", "launch")}
    line["roofline"] = r
    line["cpu_baseline"] = c
    line["elbo_rel_err_vs_cpu_fp64"] = _num(full.get("elbo_rel_err_vs_cpu_fp64"), 3)
    line["grad_rel_l2_vs_cpu_fp64"] = _num(_get(full, "parity_vs_fp64_oracle", "grad_rel_l2"), 3)
    line["steady_state_est_per_s"] = _num(_get(full, "steady_state", "estimates_per_s"), 6)
    line["repeat_ms_per_step"] = [_num(x, 4) for x in (full.get("repeat_ms_per_step") or [])][:5]
    line["also"] = also
    if full.get("dist"):
        d = full["dist"]
        line["dist"] = {"route": d.get("route"), "pipeline": str(d.get("pipeline", ""))[:64] or None,
                        "estimate_sharded_est_per_s": _num(_get(d, "estimate_sharded", "value"), 6),
                        "us_per_estimate": d.get("us_per_estimate"),
                        "p2p_verified": _get(d, "p2p_vs_allreduce", "verified"),
                        "also": ({k: (None if "error" in v else _num(v.get("value"), 5)) for k, v in d["also"].items()} if isinstance(d.get("also"), dict) else None)}
    line["full"] = full.get("full_path")
    s = json.dumps(line, separators=(",", ":"))
    # belt and braces: shed optional blocks, largest first, until the line fits
    for k in ("repeat_ms_per_step", "also", "dist", "steady_state_est_per_s"):
        if len(s) <= LINE_LIMIT:
            break
        line.pop(k, None)
        s = json.dumps(line, separators=(",", ":"))
    if len(s) > LINE_LIMIT:
        for blk, key in (("roofline", "kernel"), ("cpu_baseline", "sample"), ("config", "launch"), ("config", "workload")):
            if isinstance(line.get(blk), dict) and key in line[blk]:
                line[blk][key] = str(line[blk][key])[:40]
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) <= LINE_LIMIT, len(s)
    return s


def emit(full):
    """Full result -> gpurun_out/bench_full.json (+ stderr), compact line -> stdout (the last thing written there)."""
    path = os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        full["full_path"] = os.path.relpath(path, ROOT)
    except OSError:
        full["full_path"] = None
    sys.stderr.write("bench.py full result: " + json.dumps(full) + "\n")
    sys.stderr.flush()
    sys.stdout.write(compact_line(full) + "\n")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="ns", choices=sorted(WORKLOADS))
    ap.add_argument("--graph-chunk", type=int, default=100, help="estimates per hipGraph replay (N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the other BASELINE configurations (c2, ns_dense, ns_stl, c5, c3)")
    ap.add_argument("--concurrent", type=int, default=4,
                    help="extra (non-headline) leg: this many independent estimator contexts on separate HIP streams")
    args = ap.parse_args()

    # A/B switches select in-library REFERENCE routes: a bench line measured under one is not the product's number.  Refuse to run.
    ab = sorted(k for k in os.environ if k.startswith("MIVI_") and k not in BENCH_ENV_OK)
    if ab:
        raise SystemExit(f"bench.py: refusing to run with libmivi A/B switches set: {ab} (unset them; allowed: {sorted(BENCH_ENV_OK)})")

    import torch
    import advancedvi_jl_amd as avi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libmivi has no CPU fallback")
    torch.cuda.set_device(local_rank)
    import contextlib

    @contextlib.contextmanager
    def quiet_stdout():
        """RCCL prints a version banner on stdout at init; keep stdout for the one JSON line."""
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            yield
        finally:
            sys.stdout.flush()
            try:   # RCCL writes through C stdio: flush its buffer while fd 1 still points at stderr
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:   # noqa: BLE001
                pass
            os.dup2(saved, 1)
            os.close(saved)

    dist = None
    force_dist = os.environ.get("MIVI_FORCE_DIST", "0") == "1"   # exercise the N>1 code path on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # MIVI_FORCE_DIST=1 without a launcher: a one-rank group
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", "29533")
        with quiet_stdout():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.all_reduce(torch.zeros(1, device=f"cuda:{local_rank}"))   # communicator set-up (and its banner) up front

    w = WORKLOADS[args.workload]
    q, prob = make_problem(avi, w)
    params_h, _ = avi.destructure(q)
    K, W = args.steps, args.warmup
    stream = torch.cuda.Stream(device=local_rank)
    ent = [avi.ClosedFormEntropy(), avi.ClosedFormEntropyZeroGradient(), avi.MonteCarloEntropy(),
           avi.StickingTheLandingEntropy(), avi.StickingTheLandingEntropyZeroGradient()][w["entropy"]]

    with torch.cuda.stream(stream):
        single = (world == 1 and not force_dist)
        if single:
            ctx = avi.MiviContext(np.float32, w["family"], w["d"], w["n_mc"], ent.code, SEED, device=local_rank)
            ctx.set_problem(prob)
            params = ctx.to_device(params_h)
            value, grad = ctx.empty(1), ctx.empty(ctx.params_len)
            chunk = max(1, min(args.graph_chunk, K))
            use_graph = w["target"] != "logreg"
            if w["family"] == 1 and w["target"] in ("iso", "dense"):
                Lstep = ctx.batch_lanes(chunk) or chunk
                extra = (" | the dense target's product" if w["target"] == "dense" else "") + (" | the sticking-the-landing product (C^-T formed once per call)" if w["entropy"] in (3, 4) else "")
                launch_desc = (f"mivi_estimate_gradient_n x{chunk}: batch engine, {-(-chunk // Lstep)} step(s) of {Lstep} lanes, 3 launches per step on one stream "
                               f"(draws as operand planes -> product+target{' -> dense product' if w['target'] == 'dense' else ''}{' -> STL product' if w['entropy'] in (3, 4) else ''} -> VJP+values), no graph")
            elif use_graph:
                launch_desc = f"mivi_estimate_gradient_n x{chunk} (one hipGraph / launch-free kernel per call)"
            else:
                launch_desc = "eager single calls"

            def run(idx0, n):
                done = 0
                while use_graph and done + chunk <= n:
                    ctx.estimate_gradient_n(params, idx0 + done, chunk, value, grad)
                    done += chunk
                for i in range(done, n):
                    ctx.estimate_gradient(params, idx0 + i, value, grad)
        else:
            # N > 1 (or MIVI_FORCE_DIST=1 on one GPU): every rank draws its n_mc-sample shard of ONE estimate of n_mc * N samples; the
            # exchange runs behind the C ABI.  Route: the peer-to-peer kernel written for xGMI when the exchange areas can be mapped
            # (checked against the RCCL all-reduce route on the first estimate, every rank must agree), RCCL otherwise;
            # MIVI_DIST_MODE = auto | p2p | allreduce | rsag pins it.  Timed: the PIPELINED batch (mivi_estimate_gradient_dist_n: exchange of
            # estimate t under the kernels of t + 1 -- estimates at fixed parameters are independent, as in the N = 1 line); the
            # dependent-chain step and the per-stage times are reported beside it (`dist`).
            plan = avi.distributed.ShardPlan(w["n_mc"] * world, world)
            ctx = avi.MiviContext(np.float32, w["family"], w["d"], plan.count(rank), ent.code, SEED, device=local_rank,
                                  m_offset=plan.offset(rank), m_total=plan.n_samples)
            ctx.set_problem(prob)
            params = ctx.to_device(params_h)
            value, grad = ctx.empty(1), ctx.empty(ctx.params_len)
            dev = f"cuda:{local_rank}"
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8))
            if world > 1:
                dist.broadcast(idt, src=0)
            with quiet_stdout():
                ctx.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)

            def all_ok(flag):
                t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
                if dist:
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return int(t.item()) == 1

            want = os.environ.get("MIVI_DIST_MODE", "auto")
            dist_info = {"requested": want}
            p2p_ok = False
            if want in ("auto", "p2p"):
                try:
                    with quiet_stdout():
                        ctx.comm_enable_p2p()
                    p2p_ok = True
                except avi.MiviError as e:
                    dist_info["p2p_error"] = str(e)
                p2p_ok = all_ok(p2p_ok)
                if p2p_ok:   # self-check behind the ABI (mivi_p2p_selfcheck): the peer-to-peer estimate against the RCCL all-reduce estimate, every rank's verdict combined
                    try:
                        chk = ctx.p2p_selfcheck(params, 3)
                        dist_info["p2p_vs_allreduce"] = dict(value_rel=chk["value_rel"], grad_rel_l2=chk["grad_rel_l2"], verified=chk["verified"])
                        good = chk["verified"] or (world == 1 and chk["value_rel"] <= 1e-5 and chk["grad_rel_l2"] <= 1e-5)
                    except avi.MiviError as e:
                        good = False
                        dist_info["p2p_error"] = str(e)
                    p2p_ok = all_ok(good)
                if not p2p_ok:
                    try:
                        ctx.p2p_detach()
                    except avi.MiviError:
                        pass
            ctx.comm_set_route("p2p" if p2p_ok else (want if want in ("allreduce", "rsag") else "auto"))
            dist_info["route"] = ctx.comm_route()
            ctx.estimate_gradient_dist(params, 0, value, grad)          # allocate every work buffer before any capture
            ctx.synchronize()
            chunk = max(1, min(100, K))   # estimates per pipelined batch (each batch ends with the exchange of its last group: ~100 us of tail)
            pipelined = os.environ.get("MIVI_DIST_PIPELINE", "1") != "0"
            if pipelined and dist_info["route"] == "p2p":
                # the persistent exchange kernels must really run beside the compute chain on every rank: one warm batch, checked
                # (bounded waits: a device that serialises them reports an error instead of hanging); all ranks switch together
                try:
                    ctx.estimate_gradient_dist_n(params, 1, chunk, value, grad)
                    ctx.synchronize()
                    pipe_ok = True
                except avi.MiviError as e:
                    pipe_ok = False
                    dist_info["pipeline_error"] = str(e)
                if not all_ok(pipe_ok):
                    ctx.p2p_set_pipeline(False)
                    dist_info["pipeline"] = "off (exchange kernels did not run beside the compute chain): serial steps in one graph"
                else:
                    dist_info["pipeline"] = "persistent exchange kernel beside the lane-batched compute chain, groups of four estimates per epoch"

            def run(idx0, n):
                done = 0
                while pipelined and done + chunk <= n:
                    ctx.estimate_gradient_dist_n(params, idx0 + done, chunk, value, grad)
                    done += chunk
                for i in range(done, n):
                    ctx.estimate_gradient_dist(params, idx0 + i, value, grad)

        run(0, W)
        if single and W < chunk:   # make sure the hipGraph is captured + instantiated outside the timed region
            run(W, chunk)
        stream.synchronize()
        # pre-heat (untimed, NOT counted in `steps` / `warmup`): the same batched calls back to back for >= 300 ms, so that the timed region
        # -- 0.2 ms for the driver's --steps 20 -- does not sit on the clock ramp of a GPU that was idle a moment ago (measured in one
        # process, same call: 16-24 us per estimate in its first tens of milliseconds of GPU work, 11-13 us after a few hundred)
        t_heat, heat_calls = time.perf_counter(), 0
        idx_t = W + chunk          # the estimate index walks on, so the timed call continues the device-side counter (no counter-setting launch)
        while time.perf_counter() - t_heat < 0.3 or heat_calls < 3:
            run(idx_t, max(chunk, 1))
            idx_t += max(chunk, 1)
            heat_calls += 1
            if heat_calls % 8 == 0:
                stream.synchronize()
        for _ in range(3):   # ... ending on the timed region's own rhythm: one call, device-wide synchronize (the first isolated call behind a
            stream.synchronize()   # queue of back-to-back ones measured 1.2 us per estimate slower than the ones after it: `repeat_ms_per_step`)
            torch.cuda.synchronize()
            run(idx_t, max(chunk, 1))
            idx_t += max(chunk, 1)
        stream.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(idx_t, K)
        torch.cuda.synchronize()   # (device-wide: covers the launch stream and the interleaved chains' streams)
        if dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        repeats = []
        if single:   # the same timed region again, back to back (diagnostic: what a longer-running process sees for the same K steps)
            for r in range(5):
                torch.cuda.synchronize()
                tr = time.perf_counter()
                run(idx_t + (r + 1) * K, K)
                torch.cuda.synchronize()
                repeats.append((time.perf_counter() - tr) / K * 1e3)
            idx_t += 5 * K
        if dist:
            t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())

        if not single:
            # per-rank diagnostics of the timed region's exchange (route p2p): what the exchange kernel waited for -- the own compute chain,
            # the peers' pushes, the owners' reduced chunks -- and the bytes stored into every peer: one line tells a slow link from a slow rank
            try:
                ps = ctx.p2p_stats(reset=True)
                mine = torch.tensor([float(rank), ps["wait_handover_us"], ps["wait_pushes_us"], ps["wait_finals_us"], float(ps["groups"]),
                                     ps["bytes_per_peer_per_estimate"]], dtype=torch.float64, device=f"cuda:{local_rank}")
                allr = [torch.zeros_like(mine) for _ in range(world)]
                if dist and world > 1:
                    dist.all_gather(allr, mine)
                else:
                    allr = [mine]
                dist_info["per_rank"] = [dict(rank=int(t[0]), groups_of_estimates=int(t[4]),
                                              wait_us_per_group=dict(own_compute_chain=round(float(t[1]) / max(1.0, float(t[4])), 2),
                                                                     peers_pushes=round(float(t[2]) / max(1.0, float(t[4])), 2),
                                                                     owners_reduced_chunks=round(float(t[3]) / max(1.0, float(t[4])), 2)),
                                              bytes_to_each_peer_per_estimate=int(t[5]),
                                              link_GBs_if_exchange_bound=round(float(t[5]) * (K / max(dt, 1e-9)) / 1e9, 2))
                                         for t in (x.cpu() for x in allr)]
                dist_info["per_rank_note"] = ("exchange kernel workgroup 0, since the last reset (the pre-heat calls + the timed region); a rank whose "
                                              "peers_pushes wait stands out sits behind a slow sender or link, a large own_compute_chain wait means the exchange is not the bound")
            except Exception as e:   # noqa: BLE001  -- another route than p2p
                dist_info["per_rank_error"] = str(e)
            # Beside the sample-sharded line (`value`: what the north star names -- every estimate's samples over the ranks, the gradient exchanged):
            # the same K steps per rank with the ESTIMATES sharded instead -- every rank runs the one-GPU batch engine on estimate indices of
            # its own, no collective.  Estimates at fixed parameters are independent units; this is what a batch of them should do on N GPUs
            # (DESIGN.md 7), reported as an extra, never as `value`.
            try:
                ctx1 = avi.MiviContext(np.float32, w["family"], w["d"], w["n_mc"], ent.code, SEED, device=local_rank)
                ctx1.set_problem(prob)
                p1 = ctx1.to_device(params_h)
                v1, g1 = ctx1.empty(1), ctx1.empty(ctx1.params_len)
                ch1 = max(1, min(args.graph_chunk, K))
                base1 = (rank + 1) * (1 << 32)

                def run1(i0, n):
                    done = 0
                    while done + ch1 <= n:
                        ctx1.estimate_gradient_n(p1, base1 + i0 + done, ch1, v1, g1)
                        done += ch1
                    for i in range(done, n):
                        ctx1.estimate_gradient(p1, base1 + i0 + i, v1, g1)
                for r in range(3):
                    run1(r * K, K)
                torch.cuda.synchronize()
                if dist:
                    dist.barrier()
                t1s = time.perf_counter()
                run1(3 * K, K)
                torch.cuda.synchronize()
                if dist:
                    dist.barrier()
                dt1 = time.perf_counter() - t1s
                if dist:
                    tt = torch.tensor([dt1], dtype=torch.float64, device=f"cuda:{local_rank}")
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt1 = float(tt.item())
                dist_info["estimate_sharded"] = dict(value=K * world / dt1, unit="estimates/s", ms_per_step=dt1 / K * 1e3,
                                                     note="every rank: K estimates of n_mc samples on indices of its own (mivi_estimate_gradient_n, the batch engine), "
                                                          "no collective; barrier + max over ranks like the timed region")
                ctx1.close()
            except Exception as e:   # noqa: BLE001
                dist_info["estimate_sharded"] = dict(error=str(e))
            # per-stage times of the sharded step (hipEvents around hipGraph replays of 20 estimates, every rank collectively):
            # partial kernels | exchange + finalisation | the dependent-chain step | the pipelined step
            try:
                dist_info["us_per_estimate"] = {k: round(v, 3) for k, v in ctx.profile_dist(params, 20).items()}
            except avi.MiviError as e:
                dist_info["profile_error"] = str(e)
            try:
                ctx.synchronize()
            except avi.MiviError as e:
                dist_info["status_error"] = str(e)
            # BASELINE.json's own multi-GPU configurations and the sample count at which sharding the north-star estimate starts to pay, as
            # compact extras (`also`): configs[3] (C4: hierarchical LogReg, 128 samples per GPU, X replicated), configs[4] (C5: funnel, 64 per GPU), and
            # the north-star shape with 256 / 1024 / 4096 samples per GPU.  Each: one sharded estimate per step on the dependent chain
            # (mivi_estimate_gradient_dist: partial kernels -> exchange -> finalisation), RCCL route unless the peer-to-peer kernel was verified,
            # barrier + max over ranks; value = n_mc_per_gpu-sample estimate units of ALL ranks per second.
            if args.workload == "ns" and not args.no_also:
                dist_also = {}
                legs = [("c4", dict(WORKLOADS["c3"]), 12), ("c5", dict(WORKLOADS["c5"]), 200)]
                legs += [("ns_n_mc_%d" % m, dict(WORKLOADS["ns"], n_mc=m), 20) for m in (256, 1024, 4096)]
                for key, w2, k2 in legs:
                    if os.environ.get("MIVI_BENCH_SKIP_C3") and key == "c4":
                        continue
                    cx = None
                    try:
                        q2, prob2 = make_problem(avi, w2)
                        p2h, _ = avi.destructure(q2)
                        plan2 = avi.distributed.ShardPlan(w2["n_mc"] * world, world)
                        cx = avi.MiviContext(np.float32, w2["family"], w2["d"], plan2.count(rank), w2["entropy"], SEED, device=local_rank,
                                             m_offset=plan2.offset(rank), m_total=plan2.n_samples)
                        cx.set_problem(prob2)
                        p2 = cx.to_device(p2h)
                        v2, g2 = cx.empty(1), cx.empty(cx.params_len)
                        id2 = torch.zeros(128, dtype=torch.uint8, device=dev)
                        if rank == 0:
                            id2.copy_(torch.frombuffer(bytearray(cx.comm_unique_id()), dtype=torch.uint8))
                        if world > 1:
                            dist.broadcast(id2, src=0)
                        with quiet_stdout():
                            cx.comm_init(bytes(id2.cpu().numpy().tobytes()), rank, world)
                        for i in range(3):
                            cx.estimate_gradient_dist(p2, i, v2, g2)
                        cx.synchronize()
                        if dist:
                            dist.barrier()
                        torch.cuda.synchronize()
                        t2s = time.perf_counter()
                        for i in range(k2):
                            cx.estimate_gradient_dist(p2, 10 + i, v2, g2)
                        torch.cuda.synchronize()
                        if dist:
                            dist.barrier()
                        dt2 = time.perf_counter() - t2s
                        if dist:
                            tt = torch.tensor([dt2], dtype=torch.float64, device=dev)
                            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                            dt2 = float(tt.item())
                        dist_also[key] = dict(value=k2 * world / dt2, unit="estimates/s", us_per_step=dt2 / k2 * 1e6, n_mc_per_gpu=w2["n_mc"], route=cx.comm_route(),
                                              workload=w2["name"] if key in ("c4", "c5") else "north-star shape, %d samples per GPU" % w2["n_mc"])
                    except Exception as e:   # noqa: BLE001
                        dist_also[key] = dict(error=str(e))
                    finally:
                        if cx is not None:
                            try:
                                cx.close()
                            except Exception:   # noqa: BLE001
                                pass
                    del w2
                dist_info["also"] = dist_also
        out = None
        if rank == 0:
            cost = algorithmic_cost(w)
            est_per_s = K * world / dt
            # ---- roofline leg: hipEvent-timed launches of the dominant kernel on the launch stream ----------
            roof = None
            stages = {}
            if single and w["target"] in ("iso", "dense"):
                reps = 300
                if w["family"] == 0:
                    roof, stages = mf_roofline(ctx, params, cost)
                else:
                    roof, stages = fr_roofline(ctx, params, cost, w, lanes=chunk)
                    try:   # two EMPTY dependent launches with the grids / LDS footprints of the two contraction kernels (graph replay)
                        roof["latency_floor_us"] = ctx.profile_kernel(9, params, reps) * 1e3
                        roof["latency_floor_note"] = ("what the two-launch structure costs with no work in it; whole estimate minus this = "
                                                      "what kernel work can still win")
                    except Exception:   # noqa: BLE001
                        pass
                stages = {k: round(v * 1e3, 3) for k, v in stages.items()}   # us
                if w["family"] == 1 and w["entropy"] in (3, 4):
                    roof["stl_term"] = stl_block(ctx, params, w)
            elif single:
                roof = other_roofline(ctx, params, w, dt / K)
            elif w["family"] == 1 and w["target"] in ("iso", "dense"):
                # N > 1: the dominant kernel is the same per-rank product kernel on this rank's shard (no collective inside these launches)
                try:
                    roof, stages = fr_roofline(ctx, params, cost, w)
                    stages = {k: round(v * 1e3, 3) for k, v in stages.items()}
                    roof["note"] = "rank 0's shard; the exchange is reported in `dist`"
                except Exception as e:   # noqa: BLE001
                    roof = dict(error=str(e))
            whole = dict(hbm_equiv_GBs=cost["bytes"] * est_per_s / world / 1e9,
                         hbm_equiv_frac_of_8TBs=cost["bytes"] * est_per_s / world / 1e9 / PEAK_HBM_GBS,
                         f32_mfma_TFs=cost["flops"] * est_per_s / world / 1e12 if w["family"] == 1 else None)
            # ---- capacity leg (NOT the headline): S independent contexts on S streams -------------------------
            # `value` above: K estimates at FIXED parameters, issued as batched calls of `chunk` estimates (mivi_estimate_gradient_n; the north
            # star's family / target: the batch engine, three launches per step of up to 80 estimates -- config.launch).  What an optimiser
            # loop sees -- every estimate behind the previous update, one dependent chain -- is `also.ns_adam_loop`.
            conc = None
            if single and args.concurrent > 1 and w["target"] in ("iso", "dense"):
                S = args.concurrent
                streams = [torch.cuda.Stream(device=local_rank) for _ in range(S)]
                ctxs, bufs = [], []
                for si in range(S):
                    with torch.cuda.stream(streams[si]):
                        cx = avi.MiviContext(np.float32, w["family"], w["d"], w["n_mc"], ent.code, SEED + 1 + si, device=local_rank)
                        cx.set_problem(prob)
                        pv = cx.to_device(params_h)
                        vv, gg = cx.empty(1), cx.empty(cx.params_len)
                        cx.estimate_gradient_n(pv, 0, chunk, vv, gg)
                        ctxs.append(cx)
                        bufs.append((pv, vv, gg))
                torch.cuda.synchronize()
                reps = max(1, min(K // chunk, 10))
                tc0 = time.perf_counter()
                for r in range(reps):
                    for si in range(S):
                        with torch.cuda.stream(streams[si]):
                            ctxs[si].estimate_gradient_n(bufs[si][0], (r + 1) * chunk, chunk, bufs[si][1], bufs[si][2])
                torch.cuda.synchronize()
                tc = time.perf_counter() - tc0
                conc = dict(streams=S, estimates_per_s=S * reps * chunk / tc,
                            note="independent estimate chains on separate HIP streams; not the headline value")
                for cx in ctxs:
                    cx.close()
            # ---- BASELINE configs[1] (mean-field d=1024, n_mc=256) measured alongside the north-star workload ----
            # ---- the other BASELINE configurations, measured alongside the north-star line (N = 1) -----------------
            # each: a steady-state leg of >= 1000 estimates (hipGraph x100 where the target is graph-capturable) independent of
            # the driver's --steps, and its own roofline block
            also = None
            steady = None
            if single:
                n_ss = 1000
                run(idx_t + K, chunk)          # (graph already instantiated)
                t_h = time.perf_counter()      # the legs above left the GPU idle between their synchronisations: back to full clocks first
                while time.perf_counter() - t_h < 0.1:
                    run(idx_t + K, chunk)
                stream.synchronize()
                t0s = time.perf_counter()
                run(idx_t + K + chunk, n_ss)
                stream.synchronize()
                tss = time.perf_counter() - t0s
                steady = dict(estimates=n_ss, us_per_step=tss / n_ss * 1e6, estimates_per_s=n_ss / tss)
            if single and args.workload == "ns" and not args.no_also:
                also = {}
                for wn in ("c2", "ns_dense", "ns_stl", "c5") + (() if os.environ.get("MIVI_BENCH_SKIP_C3") else ("c3",)):
                    w2 = WORKLOADS[wn]
                    q2, prob2 = make_problem(avi, w2)
                    p2h, _ = avi.destructure(q2)
                    cx = avi.MiviContext(np.float32, w2["family"], w2["d"], w2["n_mc"], w2["entropy"], SEED, device=local_rank)
                    cx.set_problem(prob2)
                    p2 = cx.to_device(p2h)
                    v2, g2 = cx.empty(1), cx.empty(cx.params_len)
                    graphable = w2["target"] != "logreg"
                    n_est = 1000 if graphable else 40

                    def run2(i0, n):
                        if graphable:
                            for r in range(n // 100):
                                cx.estimate_gradient_n(p2, i0 + 100 * r, 100, v2, g2)
                        else:
                            for i in range(n):
                                cx.estimate_gradient(p2, i0 + i, v2, g2)
                    run2(0, 100 if graphable else 3)
                    stream.synchronize()
                    t20 = time.perf_counter()
                    run2(100, n_est)
                    stream.synchronize()
                    t2 = (time.perf_counter() - t20) / n_est
                    c2cost = algorithmic_cost(w2)
                    if w2["target"] == "logreg" or (w2["family"] == 0 and w2["target"] != "iso"):
                        roof2 = other_roofline(cx, p2, w2, t2)
                    elif w2["family"] == 0:
                        roof2, _ = mf_roofline(cx, p2, c2cost)
                    else:
                        roof2, _ = fr_roofline(cx, p2, c2cost, w2, reps=100, lanes=100)
                        if w2["entropy"] in (3, 4):
                            roof2["stl_term"] = stl_block(cx, p2, w2)
                    also[wn] = dict(workload=w2["name"], value=1.0 / t2, unit="estimates/s", us_per_step=t2 * 1e6, estimates=n_est,
                                    launch="hipGraph x100" if graphable else "eager", roofline=roof2,
                                    parity_vs_fp64_oracle=(None if args.no_cpu_baseline else parity_vs_oracle(cx, p2, p2h, w2, batch=4)))
                    cx.close()
                    del prob2, q2
                # every estimate of a batch delivered (mivi_estimate_gradient_each: values[n] and grads[n x len], dense with the zeros above the
                # diagonal written per estimate: 4.2 MB per lane instead of the 2.1 MB of the scratch lower triangles) -- timed once, full file only
                try:
                    n_e = 100
                    ve, ge = ctx.estimate_gradient_each(params, 90_000, n_e)
                    stream.synchronize()
                    t0s = time.perf_counter()
                    for r in range(5):
                        ctx.estimate_gradient_each(params, 90_000 + (r + 1) * n_e, n_e, values=ve, grads=ge)
                    stream.synchronize()
                    t_e = (time.perf_counter() - t0s) / (5 * n_e)
                    also["ns_each"] = dict(workload="north-star batches through mivi_estimate_gradient_each (every estimate's value and dense gradient kept), 5 x 100 estimates",
                                           value=1.0 / t_e, unit="estimates/s", us_per_step=t_e * 1e6)
                    del ve, ge
                except Exception as e:   # noqa: BLE001
                    also["ns_each"] = dict(error=str(e))
                # estimate_objective at a monitoring sample count (repgradelbo.jl:112-122): 10^5 samples of the north-star family per call
                try:
                    n_o = 100_000
                    vo = ctx.estimate_objective(params, 95_000, n_samples=n_o)
                    stream.synchronize()
                    t0s = time.perf_counter()
                    for r in range(10):
                        vo = ctx.estimate_objective(params, 95_001 + r, n_samples=n_o, value=vo)
                    stream.synchronize()
                    t_o = (time.perf_counter() - t0s) / 10
                    also["ns_objective_1e5"] = dict(workload="mivi_estimate_objective, 10^5 samples per call, north-star family and target (whole blocks of n_mc samples as lanes of the batch engine, values only)",
                                                    value=n_o / t_o, unit="samples/s", ms_per_call=t_o * 1e3)
                except Exception as e:   # noqa: BLE001
                    also["ns_objective_1e5"] = dict(error=str(e))
                # the Stein / Price estimator of E_q[grad], E_q[hess] on the north-star shape (gaussian_expectation_gradient_and_hessian!,
                # src/algorithms/gauss_expected_grad_hess.jl:32-60): the ELBO path's sampling + target kernels, eps G^T, one C^-T solve
                # with d right-hand sides; eager calls with consecutive indices, device-resident outputs
                g_s, H_s = ctx.empty(w["d"]), ctx.empty(w["d"] * w["d"])
                for i in range(20):
                    ctx.gauss_expected_grad_hess(params, 50_000 + i, 0, g_s, H_s)
                stream.synchronize()
                n_st = 300
                t0s = time.perf_counter()
                for i in range(n_st):
                    ctx.gauss_expected_grad_hess(params, 50_020 + i, 0, g_s, H_s)
                stream.synchronize()
                t_st = (time.perf_counter() - t0s) / n_st
                also["stein"] = dict(workload=f"mivi_gauss_expected_grad_hess, d={w['d']} full-rank, n={w['n_mc']}, the north-star target",
                                     us_per_call=t_st * 1e6, calls=n_st, launch="eager, consecutive indices",
                                     f32_mfma_TFs=(2.0 * w["d"] * w["d"] * w["n_mc"] * 1.5 + 2.0 * w["d"] ** 3 / 2) / t_st / 1e12,
                                     note="flops: triangular product + eps G^T (d^2 n each, the first half-counted) + the d-column solve (d^3 / 2 MACs)")
                del g_s, H_s
                # what a host that keeps its parameters in host memory sees (julia/MIVI.jl's estimate_gradient! without the device-resident
                # fast path): mivi_estimate_gradient_host = 4.2 MB of parameters up + 4.2 MB of gradient down over PCIe around the estimate
                import ctypes as C
                v_h, g_h = np.zeros(1, np.float32), np.zeros(ctx.params_len, np.float32)
                ph = np.ascontiguousarray(params_h, dtype=np.float32)
                def host_call(i):
                    st_ = ctx.lib.mivi_estimate_gradient_host(ctx.h, ph.ctypes.data_as(C.c_void_p), i, v_h.ctypes.data_as(C.c_void_p),
                                                              g_h.ctypes.data_as(C.c_void_p))
                    if st_ != 0:
                        raise RuntimeError(f"mivi_estimate_gradient_host status {st_}")
                for i in range(5):
                    host_call(70_000 + i)
                hb = []
                for i in range(50):
                    t0s = time.perf_counter()
                    host_call(70_005 + i)
                    hb.append(time.perf_counter() - t0s)
                hb.sort()
                t_hb = hb[len(hb) // 2]   # median: the staging copies of pageable host arrays vary by 10x with what else the host is doing
                also["ns_host_boundary"] = dict(workload="north-star estimate through mivi_estimate_gradient_host (host params in, host gradient out, synchronous)",
                                                us_per_step=t_hb * 1e6, us_min=hb[0] * 1e6, us_max=hb[-1] * 1e6, value=1.0 / t_hb, unit="estimates/s",
                                                pcie_bytes_per_step=2 * ctx.params_len * 4,
                                                note="PCIe-inclusive rate of the boundary a Julia host without device arrays uses; never the headline value")
                # the device-resident optimisation loop on the north-star problem (mivi_optimize_steps: estimate -> Adam + ClipScale fused into
                # the VJP epilogue, hipGraph of the whole chunk): what `optimize()` sustains, one dependent chain
                p_l = params.clone()
                st_l = ctx.empty(2 * p_l.numel()).zero_()
                T_l = 1000
                ctx.optimize_steps(p_l, st_l, 0, 0, T_l, 1, 1e-3, 1e-5)
                stream.synchronize()
                t0s = time.perf_counter()
                for r in range(3):
                    ctx.optimize_steps(p_l, st_l, (r + 1) * T_l, (r + 1) * T_l, T_l, 1, 1e-3, 1e-5)
                stream.synchronize()
                t_l = (time.perf_counter() - t0s) / (3 * T_l)
                also["ns_adam_loop"] = dict(workload="north-star problem, mivi_optimize_steps: Adam(1e-3) + ClipScale(1e-5), 3 x 1000 steps",
                                            value=1.0 / t_l, unit="steps/s", us_per_step=t_l * 1e6,
                                            note="the optimiser step rides in the VJP epilogue (k_fr_vjp32<FUSED>): 12.6 MB of parameter / moment traffic per step")
                del p_l, st_l
                # the same for BASELINE configs[4]'s shard (fused funnel, mean-field, STL): the launch-free loop with its per-step grid-wide
                # exchange (k_mf_funnel_sgd_loop)
                try:
                    w5 = WORKLOADS["c5"]
                    q5, prob5 = make_problem(avi, w5)
                    p5h, _ = avi.destructure(q5)
                    c5x = avi.MiviContext(np.float32, w5["family"], w5["d"], w5["n_mc"], w5["entropy"], SEED, device=local_rank)
                    c5x.set_problem(prob5)
                    p5 = c5x.to_device(p5h).clone()
                    st5 = c5x.empty(2 * p5.numel()).zero_()
                    c5x.optimize_steps(p5, st5, 0, 0, T_l, 1, 1e-3, 1e-5)
                    stream.synchronize()
                    t0s = time.perf_counter()
                    for r in range(3):
                        c5x.optimize_steps(p5, st5, (r + 1) * T_l, (r + 1) * T_l, T_l, 1, 1e-3, 1e-5)
                    stream.synchronize()
                    t_5 = (time.perf_counter() - t0s) / (3 * T_l)
                    also["c5_adam_loop"] = dict(workload="configs[4] shard (funnel d=2048 + Stacked, mean-field, STL, n_mc=64), mivi_optimize_steps: Adam(1e-3) + ClipScale(1e-5), 3 x 1000 steps",
                                                value=1.0 / t_5, unit="steps/s", us_per_step=t_5 * 1e6,
                                                note="one kernel for all steps; a step = one grid-wide exchange (row 0 couples every row): two cross-XCD hand-offs")
                    c5x.close()
                except Exception as e:   # noqa: BLE001
                    also["c5_adam_loop"] = dict(error=str(e))
                # the reference's OWN benchmark grid (bench/benchmarks.jl:43-94: optimize(alg, 10^4, normal(n_dims = 10), q), one sample per step,
                # Adam(1e-3), ClipScale; families x {ClosedFormEntropy, StickingTheLandingEntropy}): whole loops inside one kernel
                try:
                    rb = {}
                    for nm, fam_r, ent_r in (("meanfield", 0, 0), ("meanfield_stl", 0, 3), ("fullrank", 1, 0), ("fullrank_stl", 1, 3)):
                        d_r = 10
                        q_r = (avi.MeanFieldGaussian(np.zeros(d_r, np.float32), np.ones(d_r, np.float32)) if fam_r == 0
                               else avi.FullRankGaussian(np.zeros(d_r, np.float32), np.eye(d_r, dtype=np.float32)))
                        p_rh, _ = avi.destructure(q_r)
                        c_r = avi.MiviContext(np.float32, fam_r, d_r, 1, ent_r, SEED, device=local_rank)
                        c_r.set_problem(avi.DiagNormalProblem(np.full(d_r, 5.0, np.float32), np.ones(d_r, np.float32)))
                        p_r = c_r.to_device(p_rh).clone()
                        s_r = c_r.empty(2 * p_r.numel()).zero_()
                        c_r.optimize_steps(p_r, s_r, 0, 0, 1000, 1, 1e-3, 1e-5)
                        stream.synchronize()
                        t0s = time.perf_counter()
                        c_r.optimize_steps(p_r, s_r, 1000, 1000, 10_000, 1, 1e-3, 1e-5)
                        stream.synchronize()
                        t_r = (time.perf_counter() - t0s) / 10_000
                        rb[nm] = dict(steps_per_s=1.0 / t_r, us_per_step=t_r * 1e6, seconds_for_the_reference_benchmark_run=t_r * 1e4)
                        c_r.close()
                    also["reference_benchmark_grid"] = dict(workload="bench/benchmarks.jl: normal target d=10, n_samples=1, Adam(1e-3) + ClipScale, 10^4 iterations, f32",
                                                            value=rb["fullrank"]["steps_per_s"], unit="steps/s", grid=rb,
                                                            note="device-resident loops (mean-field: k_mf_sgd_loop; full-rank: k_fr_small_loop, one workgroup)")
                except Exception as e:   # noqa: BLE001
                    also["reference_benchmark_grid"] = dict(error=str(e))
                # ... and in the reference's OWN precision: bench/benchmarks.jl:59 sets T = Float64
                try:
                    rb = {}
                    for nm, fam_r, ent_r in (("meanfield", 0, 0), ("meanfield_stl", 0, 3), ("fullrank", 1, 0), ("fullrank_stl", 1, 3)):
                        d_r = 10
                        q_r = (avi.MeanFieldGaussian(np.zeros(d_r), np.ones(d_r)) if fam_r == 0 else avi.FullRankGaussian(np.zeros(d_r), np.eye(d_r)))
                        p_rh, _ = avi.destructure(q_r)
                        c_r = avi.MiviContext(np.float64, fam_r, d_r, 1, ent_r, SEED, device=local_rank)
                        c_r.set_problem(avi.DiagNormalProblem(np.full(d_r, 5.0), np.ones(d_r)))
                        p_r = c_r.to_device(p_rh).clone()
                        s_r = c_r.empty(2 * p_r.numel()).zero_()
                        c_r.optimize_steps(p_r, s_r, 0, 0, 1000, 1, 1e-3, 1e-5)
                        stream.synchronize()
                        t0s = time.perf_counter()
                        c_r.optimize_steps(p_r, s_r, 1000, 1000, 10_000, 1, 1e-3, 1e-5)
                        stream.synchronize()
                        t_r = (time.perf_counter() - t0s) / 10_000
                        rb[nm] = dict(steps_per_s=1.0 / t_r, us_per_step=t_r * 1e6, seconds_for_the_reference_benchmark_run=t_r * 1e4)
                        c_r.close()
                    also["reference_benchmark_grid_f64"] = dict(workload="bench/benchmarks.jl:59 as written: T = Float64, normal target d=10, n_samples=1, Adam(1e-3) + ClipScale, 10^4 iterations",
                                                                value=rb["fullrank"]["steps_per_s"], unit="steps/s", grid=rb)
                except Exception as e:   # noqa: BLE001
                    also["reference_benchmark_grid_f64"] = dict(error=str(e))
                # the north-star shape in Float64 (the reference's tests and benchmark run both precisions, klminrepgraddescent.jl:90-103): one
                # estimate per launch pair on the f64 MFMA tiles (kernels_fullrank.hip), 20 estimates per call like the headline
                try:
                    d_6, M_6 = (w["d"], w["n_mc"]) if w["family"] == 1 else (1024, 256)
                    q_6 = avi.FullRankGaussian(np.zeros(d_6), np.eye(d_6))
                    p_6h, _ = avi.destructure(q_6)
                    c_6 = avi.MiviContext(np.float64, 1, d_6, M_6, 0, SEED, device=local_rank)
                    c_6.set_problem(avi.DiagNormalProblem(np.full(d_6, 5.0), np.ones(d_6)))
                    p_6 = c_6.to_device(p_6h).clone()
                    v_6, g_6 = c_6.empty(1), c_6.empty(c_6.params_len)
                    c_6.estimate_gradient_n(p_6, 0, 20, v_6, g_6)
                    stream.synchronize()
                    t0s = time.perf_counter()
                    for r in range(10):
                        c_6.estimate_gradient_n(p_6, 20 * (r + 1), 20, v_6, g_6)
                    stream.synchronize()
                    t_6 = (time.perf_counter() - t0s) / 200
                    PEAK_F64_MFMA_TF = 78.6   # AMD's MI355X product figure for dense FP64 matrix; MI355X_MICROARCH.md lists no f64 peak
                    also["ns_f64"] = dict(workload=f"north-star shape in Float64: d={d_6} full-rank, n_mc={M_6}, MvNormal(5*1, I), 10 x mivi_estimate_gradient_n x20",
                                          value=1.0 / t_6, unit="estimates/s", us_per_estimate=t_6 * 1e6,
                                          f64_mfma_TFs=2.0 * d_6 * d_6 * M_6 / t_6 / 1e12,
                                          frac_f64_mfma=2.0 * d_6 * d_6 * M_6 / t_6 / 1e12 / PEAK_F64_MFMA_TF,
                                          hbm_frac_of_8TBs=(d_6 * (d_6 + 1) // 2 + d_6 * d_6 + 4 * d_6 * M_6 + 2 * d_6) * 8 / t_6 / 8e12,
                                          basis="SURVEY 8d: 2 d^2 n_mc flops (triangular product + tril VJP) and [d(d+1)/2 + d^2 + 4 d n_mc + 2 d] x 8 B per estimate; peak 78.6 TF dense f64 MFMA")
                    c_6.close()
                except Exception as e:   # noqa: BLE001
                    also["ns_f64"] = dict(error=str(e))
                # the north-star family with the FEW samples per step the reference's algorithms default to (n_samples = 1 .. 16): every row of
                # (mu, C) is independent under this target, one launch-free kernel runs the whole loop (k_fr_rows_loop)
                try:
                    fs = {}
                    for M_f in (1, 8, 16):
                        d_f = w["d"] if w["family"] == 1 else 1024
                        q_f = avi.FullRankGaussian(np.zeros(d_f, np.float32), np.eye(d_f, dtype=np.float32))
                        p_fh, _ = avi.destructure(q_f)
                        c_f = avi.MiviContext(np.float32, 1, d_f, M_f, 0, SEED, device=local_rank)
                        c_f.set_problem(avi.DiagNormalProblem(np.full(d_f, 5.0, np.float32), np.ones(d_f, np.float32)))
                        p_f = c_f.to_device(p_fh).clone()
                        s_f = c_f.empty(2 * p_f.numel()).zero_()
                        c_f.optimize_steps(p_f, s_f, 0, 0, 1000, 1, 1e-3, 1e-5)
                        stream.synchronize()
                        t0s = time.perf_counter()
                        for r in range(3):
                            c_f.optimize_steps(p_f, s_f, (r + 1) * 1000, (r + 1) * 1000, 1000, 1, 1e-3, 1e-5)
                        stream.synchronize()
                        t_f = (time.perf_counter() - t0s) / 3000
                        fs[f"n_mc={M_f}"] = dict(steps_per_s=1.0 / t_f, us_per_step=t_f * 1e6)
                        c_f.close()
                    also["ns_few_samples_adam_loop"] = dict(workload="north-star family and target (d=1024 full-rank, MvNormal(5*1, I)), n_mc = 1 / 8 / 16 per step, mivi_optimize_steps: Adam(1e-3) + ClipScale(1e-5), 3 x 1000 steps",
                                                            value=fs["n_mc=1"]["steps_per_s"], unit="steps/s", grid=fs,
                                                            note="row-separable launch-free loop (k_fr_rows_loop); the launch-per-step graph route at these shapes: 20 us per step (DESIGN.md 3)")
                except Exception as e:   # noqa: BLE001
                    also["ns_few_samples_adam_loop"] = dict(error=str(e))
                # the reference's DEFAULT algorithm settings (KLMinRepGradDescent: DoWG + PolynomialAveraging + ClipScale, n_samples small;
                # src/algorithms/constructors.jl:44-120) through mivi_optimize_loop: launch-free where the problem separates (round 4)
                try:
                    da = {}
                    for nm, fam_a, d_a, M_a in (("meanfield_d1024_m256", 0, 1024, 256), ("fullrank_d1024_m8", 1, 1024, 8), ("fullrank_d10_m1", 1, 10, 1)):
                        q_a = (avi.MeanFieldGaussian(np.zeros(d_a, np.float32), np.ones(d_a, np.float32)) if fam_a == 0
                               else avi.FullRankGaussian(np.zeros(d_a, np.float32), np.eye(d_a, dtype=np.float32)))
                        p_ah, _ = avi.destructure(q_a)
                        c_a = avi.MiviContext(np.float32, fam_a, d_a, M_a, 0, SEED, device=local_rank)
                        c_a.set_problem(avi.DiagNormalProblem(np.full(d_a, 5.0, np.float32), np.ones(d_a, np.float32)))
                        p_a = c_a.to_device(p_ah).clone()
                        s_a = c_a.dog_state()
                        c_a.dog_init(p_a, s_a, 1e-6)
                        avg_a = p_a.clone()
                        kw_a = dict(rule=3, op=1, averager=1, clip_epsilon=1e-5, opt_state=s_a, avg_params=avg_a)
                        c_a.optimize_loop(p_a, 500, 0, 0, **kw_a)
                        stream.synchronize()
                        t0s = time.perf_counter()
                        for r in range(3):
                            c_a.optimize_loop(p_a, 500, (r + 1) * 500, (r + 1) * 500, **kw_a)
                        stream.synchronize()
                        t_a = (time.perf_counter() - t0s) / 1500
                        da[nm] = dict(steps_per_s=1.0 / t_a, us_per_step=t_a * 1e6)
                        c_a.close()
                    try:   # ... and the reference README's own example (README.md:42-119: logistic regression on 208 rows x 60 features, one sample per step)
                        rng_l = np.random.default_rng(0)
                        X_l = rng_l.normal(size=(208, 60)).astype(np.float32)
                        y_l = (rng_l.uniform(size=208) < 0.5).astype(np.float32)
                        pr_l = avi.LogRegProblem(X_l, y_l, variant="lognormal_exp_bijector")
                        q_l = avi.MeanFieldGaussian(np.zeros(61, np.float32), np.full(61, 0.6, np.float32))
                        p_lh, _ = avi.destructure(q_l)
                        c_l = avi.MiviContext(np.float32, 0, 61, 1, 0, SEED, device=local_rank)
                        c_l.set_problem(pr_l)
                        p_l2 = c_l.to_device(p_lh).clone()
                        s_l2 = c_l.dog_state()
                        c_l.dog_init(p_l2, s_l2, 1e-6)
                        avg_l = p_l2.clone()
                        kw_l = dict(rule=3, op=1, averager=1, clip_epsilon=1e-5, opt_state=s_l2, avg_params=avg_l)
                        c_l.optimize_loop(p_l2, 300, 0, 0, **kw_l)
                        stream.synchronize()
                        t0s = time.perf_counter()
                        for r in range(3):
                            c_l.optimize_loop(p_l2, 300, (r + 1) * 300, (r + 1) * 300, **kw_l)
                        stream.synchronize()
                        t_l2 = (time.perf_counter() - t0s) / 900
                        da["meanfield_readme_logreg_n208_p60_m1"] = dict(steps_per_s=1.0 / t_l2, us_per_step=t_l2 * 1e6)
                        c_l.close()
                    except Exception as e:   # noqa: BLE001
                        da["meanfield_readme_logreg_n208_p60_m1"] = dict(error=str(e))
                    also["default_algorithm_loop"] = dict(workload="DoWG + PolynomialAveraging + ClipScale (the reference's default rule / averager / operator), diagonal-Gaussian target, mivi_optimize_loop, 3 x 500 steps",
                                                          value=da["meanfield_d1024_m256"]["steps_per_s"], unit="steps/s", grid=da,
                                                          note="launch-free: k_mf_gen_loop / k_fr_rows_loop (one exchange of two norm partials per step) / k_fr_small_loop; the hipGraph of launches: 8.9 / 30.4 / 11.5 us per step (DESIGN.md 3)")
                except Exception as e:   # noqa: BLE001
                    also["default_algorithm_loop"] = dict(error=str(e))
            # ---- parity + cpu_baseline leg (rank 0, N = 1 only) ---------------------------------------------
            rel = None
            parity_head = None
            cpub = None
            if single and not args.no_cpu_baseline:
                from oracle import c_oracle as CO
                if w["target"] == "iso":
                    cpub = cpu_baseline(w, params_h)
                    lib = CO.load()
                    _, eps = ctx.sample(params, 7)
                    v, _ = ctx.estimate_gradient(params, 7)
                    vref, _ = CO.estimate_gradient(lib, np.float64, w["family"], w["d"], w["n_mc"], params_h,
                                                   eps.cpu().numpy().astype(np.float64), np.full(w["d"], 5.0), np.ones(w["d"]),
                                                   w["entropy"])
                    rel = abs(float(v.item()) - vref) / abs(vref)
                    parity_head = parity_vs_oracle(ctx, params, params_h, w)
            out = {
                "metric": "ELBO-grad-estimates/sec", "value": est_per_s, "unit": "estimates/s",
                "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": w["name"], "d": w["d"], "n_mc_per_gpu": w["n_mc"], "n_mc_total": w["n_mc"] * world,
                           "family": "fullrank" if w["family"] else "meanfield", "seed": hex(SEED),
                           "launch": launch_desc if single else (f"mivi_estimate_gradient_dist_n x{chunk} (pipelined: exchange of estimate t under the kernels of t+1), route {dist_info['route']}" if pipelined else f"mivi_estimate_gradient_dist (dependent chain), route {dist_info['route']}"),
                           "fullrank_route": (list(ctx.fullrank_route()) if w["family"] == 1 else None)},
                "roofline": roof, "cpu_baseline": cpub,
                "repeat_ms_per_step": repeats, "elbo_rel_err_vs_cpu_fp64": rel, "parity_vs_fp64_oracle": parity_head, "stage_us": stages,
                "preheat": dict(calls=heat_calls, note="untimed batched calls for >= 300 ms before the timed region (GPU clock ramp), the last three each followed by a device-wide synchronize like the timed call; not counted in steps / warmup"), "whole_estimate": whole, "steady_state": steady, "concurrent": conc, "also": also,
                "dist": (None if single else dist_info),
            }
        if dist:
            dist.barrier()
            dist.destroy_process_group()
    if out is not None:
        emit(out)


if __name__ == "__main__":
    main()
