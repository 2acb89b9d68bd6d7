"""mivi_estimate_objective at monitoring sample counts on the batch engine (full-rank f32 contexts of an engine shape: whole blocks of n_mc
samples run as lanes -- draws, product + target, value workgroups, no VJP): the same (estimate index, global sample column) stream as the
chunked route, so a context whose n_mc is NOT an engine shape (chunked route) must give the same number to rounding; and the closed form of
the Gaussian-target objective bounds both (repgradelbo.jl:112-122, the reference's estimate_objective(q = pi, n = 10^5) ~ 0 test)."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from tests.helpers import SEED, make_family, make_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,M,kind,cfg_ent", [(256, 128, "diag", 0), (512, 256, "dense", 0), (128, 128, "diag", 3), (1024, 256, "diag", 1), (384, 128, "dense", 2)])
def test_objective_on_the_engine_equals_the_chunked_route(d, M, kind, cfg_ent):
    rng = np.random.default_rng(d + M)
    q, _ = make_family(rng, d, avi.FULLRANK, np.float32, mu_scale=0.3)
    prob, _ = make_problem(rng, kind, d, np.float32)
    params, _ = avi.destructure(q)
    eng = avi.MiviContext(np.float32, avi.FULLRANK, d, M, cfg_ent, SEED)
    chk = avi.MiviContext(np.float32, avi.FULLRANK, d, 96, cfg_ent, SEED)      # (96 samples per estimate: not an engine shape)
    eng.set_problem(prob)
    chk.set_problem(prob)
    pe, pc = eng.to_device(params), chk.to_device(params)
    assert eng.batch_takes_engine(pe) or cfg_ent in (3, 4)
    for n in (8 * M, 8 * M + 37, 100 * M + 5, 20_000):
        for ent in (-1, 0, 1):
            ve = float(eng.estimate_objective(pe, 7, n_samples=n, entropy=ent).item())
            vc = float(chk.estimate_objective(pc, 7, n_samples=n, entropy=ent).item())
            assert abs(ve - vc) <= 2e-6 * max(abs(vc), 1.0), (n, ent, ve, vc)
    # fewer than eight blocks: the chunked route on both (bitwise the same kernels)
    assert float(eng.estimate_objective(pe, 3, n_samples=5 * M).item()) == pytest.approx(float(chk.estimate_objective(pc, 3, n_samples=5 * M).item()), rel=2e-6)
    eng.close()
    chk.close()


def test_objective_at_q_equal_to_the_target_is_about_zero():
    """test/algorithms/klminrepgraddescent.jl:36-37: estimate_objective(q = pi, n = 10^5) ~ 0 (atol 1e-2 there at d = 5; here d = 256 on the engine)."""
    d, M = 256, 128
    rng = np.random.default_rng(4)
    mu = rng.normal(size=d).astype(np.float32)
    sig = rng.uniform(0.5, 1.5, size=d).astype(np.float32)
    q = avi.FullRankGaussian(mu, np.diag(sig).astype(np.float32))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 1, SEED)
    ctx.set_problem(avi.DiagNormalProblem(mu, sig))
    v = float(ctx.estimate_objective(ctx.to_device(params), 11, n_samples=100_000).item())
    assert abs(v) <= 0.2, v      # MonteCarlo entropy at q = pi: every sample's log q - log pi is exactly 0 up to rounding of d terms
    ctx.close()
