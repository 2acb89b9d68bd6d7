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

# the CPU legs (C port of the oracle under OpenMP, numpy's BLAS) must not keep spinning worker threads beside the GPU legs' host thread
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np   # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.config import (SEED, PEAK_HBM_GBS, PEAK_F32_MFMA_TF, PEAK_BF16_MFMA_TF, PLANE_BYTES, PEAK_VALU_GINST, BENCH_ENV_OK, WORKLOADS,   # noqa: E402,F401
                             algorithmic_cost)
from benchlib.rooflines import pmc_traffic, pmc_valu, rocprof_avg, mf_roofline, fr_roofline, other_roofline, stl_block   # noqa: E402,F401
from benchlib.baseline import make_problem, parity_vs_oracle, cpu_baseline_blas, cpu_baseline   # noqa: E402,F401
from benchlib import line as _line   # noqa: E402
from benchlib.line import LINE_LIMIT, compact_line, _num, _get   # noqa: E402,F401


def emit(full):
    """Full result -> gpurun_out/bench_full.json under this file's directory (+ stderr), compact line -> stdout (benchlib/line.py)."""
    return _line.emit(full, ROOT)


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
            # Round 6: on a batch-engine shape the sharded batches run on the ENGINE with ONE RCCL all-reduce per step over all of the step's
            # partial vectors (what the north star names: "an RCCL all-reduce over xGMI on the gradient"; per-rank compute 4.2 instead of
            # 13.8 us per estimate) -- the automatic choice here.  The peer-to-peer exchange kernel (its own four-estimate compute chain)
            # stays one MIVI_DIST_MODE=p2p away and remains the automatic route of the other shapes.
            engine_shape = (w["family"] == 1 and w["target"] in ("iso", "dense") and w["d"] % 128 == 0 and plan.count(rank) % 128 == 0 and
                            128 <= w["d"] <= 2048 and 128 <= plan.count(rank) <= 2048)
            if want == "auto" and engine_shape:
                want = "allreduce"
                dist_info["batch_kernels"] = "batch engine: draws / product / VJP per step of <= 24 lanes (80 on one rank), one ncclAllReduce per step over the lanes' partial vectors, one finalisation launch"
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
                    # best of three repeats (all kept in `repeat_us_per_step`): these legs run between CPU-heavy oracle / baseline work whose
                    # worker threads may still be spinning on the box's 16 CPUs -- a descheduled host thread starves the three-launch steps
                    # (seen as a 20x slower ns_stl / stein leg in two of five runs of round 6, the kernels' own durations unchanged)
                    reps2 = []
                    for rr in range(3 if graphable else 1):
                        t20 = time.perf_counter()
                        run2(100 + rr * n_est, n_est)
                        stream.synchronize()
                        reps2.append((time.perf_counter() - t20) / n_est)
                    t2 = min(reps2)
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
                                    repeat_us_per_step=[x * 1e6 for x in reps2],
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
                    t_e = float("inf")
                    for rr in range(3):
                        t0s = time.perf_counter()
                        for r in range(5):
                            ctx.estimate_gradient_each(params, 90_000 + (5 * rr + r + 1) * n_e, n_e, values=ve, grads=ge)
                        stream.synchronize()
                        t_e = min(t_e, (time.perf_counter() - t0s) / (5 * n_e))
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
                    t_o = float("inf")
                    for rr in range(3):   # (best of three, like the legs above)
                        t0s = time.perf_counter()
                        for r in range(10):
                            vo = ctx.estimate_objective(params, 95_001 + 10 * rr + r, n_samples=n_o, value=vo)
                        stream.synchronize()
                        t_o = min(t_o, (time.perf_counter() - t0s) / 10)
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
                reps_st = []
                for rr in range(3):   # (best of three: see the legs above)
                    t0s = time.perf_counter()
                    for i in range(n_st):
                        ctx.gauss_expected_grad_hess(params, 50_020 + rr * n_st + i, 0, g_s, H_s)
                    stream.synchronize()
                    reps_st.append((time.perf_counter() - t0s) / n_st)
                t_st = min(reps_st)
                also["stein"] = dict(workload=f"mivi_gauss_expected_grad_hess, d={w['d']} full-rank, n={w['n_mc']}, the north-star target",
                                     us_per_call=t_st * 1e6, calls=n_st, launch="eager, consecutive indices", repeat_us_per_call=[x * 1e6 for x in reps_st],
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

