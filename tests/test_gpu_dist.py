"""Distributed path on one GPU: shards of the sample axis on separate contexts (as separate ranks would hold
them), partial buffers summed (what the RCCL all-reduce does), finalize == the single-context estimate."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from advancedvi_jl_amd.distributed import ShardPlan
from tests.helpers import SEED, make_family, make_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("family,d,M,R", [(avi.MEANFIELD, 64, 48, 4), (avi.FULLRANK, 96, 64, 2), (avi.FULLRANK, 40, 30, 3)])
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
