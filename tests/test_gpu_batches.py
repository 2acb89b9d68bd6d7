"""mivi_estimate_gradient_n on the second-generation full-rank route: a batch is dealt onto interleaved contexts (lane-batched launches:
four contexts' product kernels as one launch, likewise their VJP kernels; one or two graph branches) -- every estimate must still be
bitwise the single call's, for every batch length (partial last steps, batches shorter than the number of contexts, the switch from
four to eight contexts at 50) and for both Gaussian targets; the STL estimators keep one context per branch."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from tests.helpers import SEED, make_family, make_problem

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
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 13, 20, 49, 50, 51, 64, 101):
        g.fill_(float("nan"))
        ctx.estimate_gradient_n(p, idx, n, v, g)
        ctx.synchronize()
        v1, g1 = ref.estimate_gradient(pr, idx + n - 1)
        assert float(v.item()) == float(v1.item()), (n, float(v.item()), float(v1.item()))
        assert np.array_equal(g.cpu().numpy(), g1.cpu().numpy()), n
        idx += n + 2                                        # (a gap: the next call is NOT in order -- the counter is set again)
    # in-order calls continue the device-side estimate counter
    ctx.estimate_gradient_n(p, 1000, 20, v, g)
    ctx.estimate_gradient_n(p, 1020, 20, v, g)
    ctx.synchronize()
    v1, g1 = ref.estimate_gradient(pr, 1039)
    assert float(v.item()) == float(v1.item()) and np.array_equal(g.cpu().numpy(), g1.cpu().numpy())
    ctx.close()
    ref.close()
