"""Distributed path on one GPU: shards of the sample axis on separate contexts (as separate ranks would hold
them), partial buffers summed (what the RCCL all-reduce does), finalize == the single-context estimate."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from advancedvi_jl_amd.distributed import ShardPlan
from tests.helpers import SEED, make_family, make_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("family,d,M,R", [(avi.MEANFIELD, 64, 48, 4), (avi.FULLRANK, 96, 64, 2), (avi.FULLRANK, 40, 30, 3),
                                          (avi.FULLRANK, 256, 256, 2)])   # (last: second-generation kernels, 128 samples per shard)
@pytest.mark.parametrize("ent", [0, 2, 3])
def test_sharded_partials_sum_to_single_gpu_estimate(family, d, M, R, ent):
    rng = np.random.default_rng(2)
    q, _ = make_family(rng, d, family, np.float32)
    prob, _ = make_problem(rng, "diag", d, np.float32)
    params, _ = avi.destructure(q)
    full = avi.MiviContext(np.float32, family, d, M, ent, SEED)
    full.set_problem(prob)
    v_ref, g_ref = full.estimate_gradient(params, 17)
    v_ref, g_ref = float(v_ref.item()), g_ref.cpu().numpy().astype(np.float64)
    plan = ShardPlan(M, R)
    total = None
    shards = []
    for r in range(R):
        sh = avi.MiviContext(np.float32, family, d, plan.count(r), ent, SEED, m_offset=plan.offset(r), m_total=M)
        sh.set_problem(prob)
        part = sh.estimate_partials(params, 17).double()
        total = part if total is None else total + part
        shards.append(sh)
    v, g = shards[0].finalize(params, total.float())
    assert abs(float(v.item()) - v_ref) <= 2e-6 * abs(v_ref)          # differs only by fp32 summation order
    assert np.linalg.norm(g.cpu().numpy() - g_ref) <= 5e-6 * max(1.0, np.linalg.norm(g_ref))
    for sh in shards:
        sh.close()
    full.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("family,d,M,R", [(avi.MEANFIELD, 64, 48, 4), (avi.FULLRANK, 96, 64, 2), (avi.FULLRANK, 40, 30, 3),
                                           (avi.FULLRANK, 128, 256, 8)])
@pytest.mark.parametrize("ent", [0, 3])
def test_slice_finalisation_kernels(family, d, M, R, ent, dtype):
    """reduce-scatter -> per-rank slice finalise -> all-gather -> unpack, the collectives played by host slicing: the kernels
    mivi_finalize_slice / mivi_unpack_final against (i) the replicated finalize kernel and (ii) the numpy restatement
    oracle.finalize_slice / unpack_final that the gloo test runs."""
    from oracle import oracle as O
    import torch
    rng = np.random.default_rng(3)
    q, _ = make_family(rng, d, family, dtype)
    prob, _ = make_problem(rng, "diag", d, dtype)
    params, _ = avi.destructure(q)
    plan = ShardPlan(M, R)
    total, shards = None, []
    for r in range(R):
        sh = avi.MiviContext(dtype, family, d, plan.count(r), ent, SEED, m_offset=plan.offset(r), m_total=M)
        sh.set_problem(prob)
        part = sh.estimate_partials(params, 23).double()
        total = part if total is None else total + part
        shards.append(sh)
    c0 = shards[0]
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    L, n = c0.partials_len, c0.slice_len(R)
    padded = torch.zeros(n * R, dtype=tdt, device="cuda")
    padded[:L] = total.to(tdt)
    fin = torch.empty(n * R, dtype=tdt, device="cuda")
    p_dev = c0.to_device(params)
    for r in range(R):      # each rank's kernel on its slice; the all-gather is the concatenation
        shards[r].finalize_slice(p_dev, padded[r * n:(r + 1) * n].clone(), r, R, fin[r * n:(r + 1) * n])
    v, g = c0.unpack_final(fin)
    v0, g0 = c0.finalize(params, total.to(tdt))
    tol = 1e-6 if dtype == np.float32 else 1e-13
    assert abs(float(v.item()) - float(v0.item())) <= tol * abs(float(v0.item()))
    assert np.max(np.abs(g.cpu().numpy() - g0.cpu().numpy())) <= tol * max(1.0, float(g0.abs().max()))
    if family == avi.FULLRANK:
        assert np.all(np.triu(g.cpu().numpy()[d:].reshape(d, d, order="F"), 1) == 0.0)
    ref = np.concatenate([O.finalize_slice(padded[r * n:(r + 1) * n].cpu().numpy(), r * n, params.astype(np.float64), d, family, ent, M, L)
                          for r in range(R)])
    vr, gr = O.unpack_final(ref, d, family)
    assert abs(float(v.item()) - vr) <= (2e-6 if dtype == np.float32 else 1e-13) * abs(vr)
    assert np.max(np.abs(g.cpu().numpy() - gr)) <= (2e-6 if dtype == np.float32 else 1e-13) * max(1.0, np.max(np.abs(gr)))
    for sh in shards:
        sh.close()


@pytest.mark.parametrize("family,d,M", [(avi.MEANFIELD, 64, 48), (avi.FULLRANK, 128, 128), (avi.FULLRANK, 40, 30)])
def test_collective_behind_the_c_abi_on_one_gpu(family, d, M):
    """mivi_comm_init(world = 1, WITH a unique id) + mivi_estimate_gradient_dist through the RCCL that libmivi opens itself, both
    routes -- {partials, ncclAllReduce, finalise} (the default below 16 MB of partials) and {partials, ncclReduceScatter, slice
    finalise, ncclAllGather, unpack} -- against the plain estimate."""
    rng = np.random.default_rng(4)
    q, _ = make_family(rng, d, family, np.float32)
    prob, _ = make_problem(rng, "diag", d, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, family, d, M, 0, SEED)
    ctx.set_problem(prob)
    v0, g0 = ctx.estimate_gradient(params, 5)
    v0, g0 = float(v0.item()), g0.cpu().numpy().copy()
    ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    got = {}
    for route in ("allreduce", "rsag"):
        ctx.comm_set_route(route)
        assert ctx.comm_route() == route
        v1, g1 = ctx.estimate_gradient_dist(params, 5)
        ctx.synchronize()
        assert abs(float(v1.item()) - v0) <= 2e-6 * abs(v0)
        assert np.linalg.norm(g1.cpu().numpy() - g0) <= 5e-6 * max(1.0, np.linalg.norm(g0))
        got[route] = g1.cpu().numpy().copy()
    ctx.comm_set_route("auto")
    v1, g1 = ctx.estimate_gradient_dist(params, 5)       # default route for this size: one all-reduce
    assert np.array_equal(g1.cpu().numpy(), got["allreduce"])
    # and without a communicator (world 1): the slice kernels, no collective
    ctx.comm_init(None, 0, 1)
    v2, g2 = ctx.estimate_gradient_dist(params, 5)
    assert np.array_equal(g2.cpu().numpy(), got["rsag"])
    ctx.close()


@pytest.mark.parametrize("kind,ent,d,M,n", [("diag", 0, 256, 128, 7), ("diag", 2, 1024, 256, 20), ("dense", 0, 256, 256, 5), ("diag", 3, 256, 128, 6),
                                            ("diag", 0, 384, 128, 90), ("diag", 0, 128, 128, 200)])
def test_sharded_batches_on_the_batch_engine(kind, ent, d, M, n):
    """Round 6 (round 5's verdict, missing 3): mivi_estimate_gradient_dist_n on an engine shape runs draws / product / VJP for all estimates
    of a step as ONE launch each, the VJP leaving every lane's partial vector; one all-reduce per step; one finalisation launch
    (csrc/kernels_fullrank_batch.hip k_fb_vjp<PART>, k_fb_finalize_parts).
    (i) one rank that holds ALL samples: the batch's last estimate equals mivi_estimate_gradient_n's to rounding -- without a communicator
        and through the RCCL all-reduce libmivi opens itself (world 1);
    (ii) a SHARD (columns [M, 2M) of 2M samples per estimate): value and gradient equal the oracle's finalisation of that shard's partial
        vector alone with M_total = 2M (oracle.finalize_partials on oracle.estimate_gradient(...)["partials"], identical eps) -- the
        normalisation and the shard-invariant stream; 90 / 200 estimates: two / three steps of the engine (with a communicator the
        all-reduce + finalisation of a step run on a second stream under the next step's kernels, the partial vectors double-buffered)."""
    from oracle import oracle as O
    rng = np.random.default_rng(40 + d + n)
    q, q_o = make_family(rng, d, avi.FULLRANK, np.float32)
    prob, tgt = make_problem(rng, kind, d, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
    ctx.set_problem(prob)
    p = ctx.to_device(params)
    v0, g0 = ctx.empty(1), ctx.empty(ctx.params_len)
    ctx.estimate_gradient_n(p, 3, n, v0, g0)
    ctx.synchronize()
    v0, g0 = float(v0.item()), g0.cpu().numpy().astype(np.float64)
    for comm in (False, True):
        ctx.comm_init(ctx.comm_unique_id() if comm else None, 0, 1)
        if comm:
            ctx.comm_set_route("allreduce")
        v1, g1 = ctx.empty(1), ctx.empty(ctx.params_len)
        g1.fill_(float("nan"))
        ctx.estimate_gradient_dist_n(p, 3, n, v1, g1)
        ctx.synchronize()
        g1 = g1.cpu().numpy().astype(np.float64)
        assert abs(float(v1.item()) - v0) <= 1e-6 * abs(v0), (comm, float(v1.item()), v0)
        assert np.linalg.norm(g1 - g0) <= 2e-6 * max(1.0, np.linalg.norm(g0)), comm
        assert not np.any(np.triu(g1[d:].reshape(d, d, order="F"), 1)), comm        # exact zeros above the diagonal in the caller's buffer
    ctx.close()
    # (ii) the second shard of a two-rank job, alone
    sh = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED, m_offset=M, m_total=2 * M)
    sh.set_problem(prob)
    ps = sh.to_device(params)
    v2, g2 = sh.empty(1), sh.empty(sh.params_len)
    sh.estimate_gradient_dist_n(ps, 3, n, v2, g2)
    sh.synchronize()
    eps = O.philox_normal(SEED, 3 + n - 1, d, M, 2 * M)          # the LAST estimate's columns [M, 2M) of the one stream
    ref = O.estimate_gradient(params.astype(np.float64), d, O.FULLRANK, tgt, eps, ent)
    vr, gr = O.finalize_partials(ref["partials"], params.astype(np.float64), d, O.FULLRANK, ent, 2 * M)
    scale = abs(float(np.sum(ref["ell"])) / (2 * M)) + abs(ref["entropy"])      # (the value is a difference of these two: near zero at some draws)
    assert abs(float(v2.item()) - vr) <= 1e-5 * max(abs(vr), 0.1 * scale), (float(v2.item()), vr, scale)
    assert np.linalg.norm(g2.cpu().numpy().astype(np.float64) - gr) <= 2e-5 * max(1.0, np.linalg.norm(gr))
    sh.close()
