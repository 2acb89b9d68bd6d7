"""Roofline blocks of the bench line: the committed PMC / rocprofv3 summaries under profiles/ and the live per-kernel timings
(mivi_profile_kernel / mivi_profile_batch: hipEvents on the launch stream)."""
import json
import os

import numpy as np

from .config import (ROOT, PEAK_HBM_GBS, PEAK_F32_MFMA_TF, PEAK_BF16_MFMA_TF, PLANE_BYTES, PEAK_VALU_GINST, algorithmic_cost)   # noqa: F401


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
                        draws=dict(kernel="k_fb_eps (Philox + Box-Muller draws of all lanes as operand planes, ONE orientation since round 6)", avg_launch_us=tb["eps"],
                                   algorithmic_bytes_per_launch=lanes * PLANE_BYTES * w["d"] * w["n_mc"],
                                   achieved_GBs=lanes * PLANE_BYTES * w["d"] * w["n_mc"] / (tb["eps"] * 1e-6) / 1e9,
                                   note="%d bytes per element written once; bound by the generator's vector arithmetic (Philox's 32-bit multiplies + Box-Muller), not by the writes" % PLANE_BYTES),
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


