"""mivi_estimate_gradient_n on the second-generation full-rank route: a batch is dealt onto interleaved contexts (lane-batched launches:
four contexts' product kernels as one launch, likewise their VJP kernels; one or two graph branches) -- every estimate must still be
bitwise the single call's, for every batch length (partial last steps, batches shorter than the number of contexts, the switch from
four to eight contexts at 12) and for both Gaussian targets; the STL estimators keep one context per branch.  At the BASELINE sizes the
lane-batched launches are kernels of their own (k_fr_prod32q: two tiles x two lanes per workgroup; k_fr_vjp32s: strips of tiles): the
same bitwise requirement at the shapes that select them."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from tests.helpers import SEED, assert_batch_matches_single, engine_shape, make_family, make_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,ent", [("diag", 0), ("dense", 0), ("diag", 2), ("diag", 3)])
def test_every_batch_length_equals_single_calls(kind, ent):
    d, M = 128, 128
    rng = np.random.default_rng(21)
    q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
    prob, _ = make_problem(rng, kind, d, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
    ctx.set_problem(prob)
    ref = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
    ref.set_problem(prob)
    p, pr = ctx.to_device(params), ref.to_device(params)
    v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    idx = 3
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 11, 12, 13, 20, 49, 50, 51, 64, 101):
        g.fill_(float("nan"))
        ctx.estimate_gradient_n(p, idx, n, v, g)
        ctx.synchronize()
        v1, g1 = ref.estimate_gradient(pr, idx + n - 1)
        # the Monte Carlo entropy value (also the STL estimator's) holds sum(eps^2): a single call takes the f32 wave sums of k_eps' blocks, a batch
        # those of its own draw blocks (both then summed in f64) -- outside the engine the f32 VALUE may land one ulp apart (1 case in ~90)
        eng = engine_shape(d, M, kind=kind, n=n) and (ent not in (3, 4) or ctx.batch_takes_engine(p))
        assert_batch_matches_single(v.item(), v1.item(), g.cpu().numpy(), g1.cpu().numpy(), eng, n, ulps=1 if ent in (2, 3) else 0)
        idx += n + 2                                        # (a gap: the next call is NOT in order -- the counter is set again)
    # in-order calls continue the device-side estimate counter
    ctx.estimate_gradient_n(p, 1000, 20, v, g)
    ctx.estimate_gradient_n(p, 1020, 20, v, g)
    ctx.synchronize()
    v1, g1 = ref.estimate_gradient(pr, 1039)
    assert_batch_matches_single(v.item(), v1.item(), g.cpu().numpy(), g1.cpu().numpy(), engine_shape(d, M, kind=kind) and (ent not in (3, 4) or ctx.batch_takes_engine(p)),
                                ulps=1 if ent in (2, 3) else 0)
    ctx.close()
    ref.close()


def test_mixed_call_sequences_keep_every_result_exact():
    """One context through a sequence that changes everything the interleaved contexts cache: batch lengths on both sides of the
    four / eight-context switch, another target (the children borrow the target buffers), single calls in between, the sharded
    batches (peer-to-peer route at world 1: their lane-batched compute chain uses the same children with another index stride), and back."""
    d, M = 128, 128
    rng = np.random.default_rng(4)
    q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
    prob_a, _ = make_problem(rng, "diag", d, np.float32)
    prob_b, _ = make_problem(rng, "dense", d, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
    ref = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
    p, pr = ctx.to_device(params), ref.to_device(params)
    v, g = ctx.empty(1), ctx.empty(ctx.params_len)

    def check_n(prob_ref, idx, n, dist=False):
        ref.set_problem(prob_ref)
        if dist:
            ctx.estimate_gradient_dist_n(p, idx, n, v, g)
        else:
            ctx.estimate_gradient_n(p, idx, n, v, g)
        ctx.synchronize()
        v1, g1 = ref.estimate_gradient(pr, idx + n - 1)
        if dist:   # the sharded finalisation sums in another order than the one-GPU value kernel: rounding, not bits
            assert abs(float(v.item()) - float(v1.item())) <= 2e-6 * abs(float(v1.item()))
            assert np.linalg.norm(g.cpu().numpy() - g1.cpu().numpy()) <= 5e-6 * max(1.0, float(np.linalg.norm(g1.cpu().numpy())))
        else:
            assert_batch_matches_single(v.item(), v1.item(), g.cpu().numpy(), g1.cpu().numpy(), engine_shape(d, M, n=n), (idx, n))

    ctx.set_problem(prob_a)
    check_n(prob_a, 10, 20)
    check_n(prob_a, 30, 100)
    check_n(prob_a, 500, 7)
    ctx.set_problem(prob_b)                       # the children must pick the new target up
    check_n(prob_b, 40, 20)
    v1, g1 = ctx.estimate_gradient(p, 77)          # single calls between batches
    ref.set_problem(prob_b)
    v2, g2 = ref.estimate_gradient(pr, 77)
    assert float(v1.item()) == float(v2.item()) and np.array_equal(g1.cpu().numpy(), g2.cpu().numpy())
    check_n(prob_b, 78, 60)
    ctx.set_problem(prob_a)
    ctx.p2p_attach([ctx.p2p_export(0, 1)])
    ctx.comm_set_route("p2p")
    check_n(prob_a, 200, 20, dist=True)
    check_n(prob_a, 220, 9, dist=True)
    check_n(prob_a, 300, 20)                       # back to the one-GPU batches (another index stride for the same children)
    check_n(prob_a, 320, 3, dist=True)             # shorter than a group: the one-at-a-time chain
    check_n(prob_a, 330, 64)
    ctx.close()
    ref.close()


@pytest.mark.parametrize("d,M", [(1024, 256), (2048, 256), (1536, 256), (512, 256), (512, 512)])
@pytest.mark.parametrize("kind,ent", [("diag", 0), ("dense", 0), ("diag", 3)])
def test_lane_batched_kernels_equal_single_calls_at_the_baseline_sizes(d, M, kind, ent):
    """(1024, 256), (2048, 256) and (1536, 256: 48 block rows, 24 tile pairs, 21 super-blocks of strips): k_fr_prod32q + k_fr_vjp32s; (512, 256): k_fr_prod32m + k_fr_vjp32s; (512, 512): k_fr_prod32q + k_fr_vjp32m.
    The dense target keeps its second product on k_fr_prod32m; the STL estimator's first step carries the inversion riders (k_fr_prod32m),
    its later steps do not.  Batch lengths: one full step of four lanes, a partial last step, two branches of four lanes."""
    if kind == "dense" and d == 2048:
        pytest.skip("(the dense target's fixture at d = 2048 is a 16 MiB precision matrix: covered at 1024)")
    rng = np.random.default_rng(100 + d + M)
    q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
    prob, _ = make_problem(rng, kind, d, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
    ctx.set_problem(prob)
    ref = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
    ref.set_problem(prob)
    p, pr = ctx.to_device(params), ref.to_device(params)
    v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    idx = 2
    for n in (4, 7, 20, 52):
        g.fill_(float("nan"))
        ctx.estimate_gradient_n(p, idx, n, v, g)
        ctx.synchronize()
        v1, g1 = ref.estimate_gradient(pr, idx + n - 1)
        # the batch engine (f16 two-way operand splits; the STL term as a product with C^-T formed once per call) against the single calls
        # (exact bf16 three-way splits; the STL term solved): equal to rounding, both 1-3e-7 from the fp64 oracle (tests/test_gpu_each.py)
        eng = engine_shape(d, M, kind=kind, n=n) and (ent != 3 or ctx.batch_takes_engine(p))
        assert_batch_matches_single(v.item(), v1.item(), g.cpu().numpy(), g1.cpu().numpy(), eng, n, ulps=1 if ent == 3 else 0)
        idx += n
    ctx.close()
    ref.close()


def test_many_batches_of_random_lengths_match_single_calls():
    """The strip VJP kernel reads a tile's operand fragments while the previous tile's stores are still in flight (exact `vmcnt` accounting),
    the four-lane product shares one staged fragment between two estimates: sixty batches of random lengths at the north-star shape, every
    one bitwise the single call's."""
    d, M = 1024, 256
    rng = np.random.default_rng(77)
    q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
    prob, _ = make_problem(rng, "diag", d, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
    ctx.set_problem(prob)
    ref = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
    ref.set_problem(prob)
    p, pr = ctx.to_device(params), ref.to_device(params)
    v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    idx = 0
    for n in rng.integers(4, 70, size=60):
        n = int(n)
        ctx.estimate_gradient_n(p, idx, n, v, g)
        ctx.synchronize()
        v1, g1 = ref.estimate_gradient(pr, idx + n - 1)
        assert_batch_matches_single(v.item(), v1.item(), g.cpu().numpy(), g1.cpu().numpy(), True, (n, idx))
        idx += n
    ctx.close()
    ref.close()
