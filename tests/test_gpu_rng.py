"""The eps stream on the device: Philox words bit-exact against the numpy restatement, Box-Muller
within a few ulp of a float64 evaluation of the same uniforms, shard invariance, moments."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("family", [avi.MEANFIELD, avi.FULLRANK], ids=["meanfield", "fullrank"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("d,M", [(8, 5), (37, 130), (256, 64)])
def test_device_eps_matches_restatement(family, dtype, d, M):
    scale = np.ones(d, dtype=dtype) if family == avi.MEANFIELD else np.eye(d, dtype=dtype)
    q = avi.MvLocationScale(np.zeros(d, dtype=dtype), scale)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(dtype, family, d, M, 0, SEED)
    Z, eps = ctx.sample(params, 42)
    eps = eps.cpu().numpy().astype(np.float64)
    ref = O.philox_normal(SEED, 42, d, 0, M, f64=(dtype == np.float64))
    tol = 4e-6 if dtype == np.float32 else 1e-13   # sqrt/log/sincospi differ by a few ulp at |eps| <= 6
    assert np.max(np.abs(eps - ref)) < tol
    assert np.max(np.abs(Z.cpu().numpy() - eps)) < (1e-6 if dtype == np.float32 else 1e-14)  # mu=0, C=I
    ctx.close()


def test_shard_invariance_of_the_stream():
    """GPU r owning global columns [r*M/R, (r+1)*M/R) regenerates exactly its slice."""
    d, M = 64, 96
    q = avi.MeanFieldGaussian(np.zeros(d, dtype=np.float32), np.ones(d, dtype=np.float32))
    params, _ = avi.destructure(q)
    full = avi.MiviContext(np.float32, avi.MEANFIELD, d, M, 0, SEED)
    _, eps_full = full.sample(params, 3)
    eps_full = eps_full.cpu().numpy()
    for r in range(3):
        sh = avi.MiviContext(np.float32, avi.MEANFIELD, d, M // 3, 0, SEED, m_offset=r * (M // 3), m_total=M)
        _, e = sh.sample(params, 3)
        assert np.array_equal(e.cpu().numpy(), eps_full[:, r * (M // 3):(r + 1) * (M // 3)])
        sh.close()
    full.close()


def test_sample_moments():
    """rand batch mean/var/cov at n = 10^6 within rtol 1e-2 (test/families/location_scale.jl:68-97)."""
    rng = np.random.default_rng(0)
    d = 10
    loc = rng.normal(size=d).astype(np.float32)
    L = np.tril(np.eye(d) + np.ones((d, d)) / 2).astype(np.float32)   # the reference test's scale, :13
    for q in (avi.FullRankGaussian(loc, L), avi.MeanFieldGaussian(loc, np.ones(d, dtype=np.float32))):
        Z = avi.rand(avi.PhiloxRNG(1), q, 10 ** 6).double()
        cov_true = (L @ L.T if q.family == avi.FULLRANK else np.eye(d)).astype(np.float64)
        assert np.allclose(Z.mean(dim=1).cpu().numpy(), loc, rtol=1e-2, atol=1e-2)
        cov = np.cov(Z.cpu().numpy())
        assert np.allclose(cov, cov_true, rtol=1e-2, atol=2e-2)


def test_f32_box_muller_accuracy_on_a_large_sample():
    """The f32 Box-Muller is written out for its arguments (csrc/philox.h: hardware log2 / sqrt, quarter-turn reduction, Cephes'
    polynomials): 2^18 draws against the float64 evaluation of the same uniforms -- absolute error below 1.5e-6 (measured 7e-7), and the
    tails are there (|eps| > 4 occurs)."""
    d, M = 1024, 256
    q = avi.MvLocationScale(np.zeros(d, dtype=np.float32), np.ones(d, dtype=np.float32))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.MEANFIELD, d, M, 0, SEED)
    _, eps = ctx.sample(params, 7)
    eps = eps.cpu().numpy().astype(np.float64)
    ref = O.philox_normal(SEED, 7, d, 0, M, f64=False)
    err = np.abs(eps - ref)
    assert err.max() < 1.5e-6, err.max()
    assert np.abs(eps).max() > 4.0
    rel = err / np.maximum(np.abs(ref), 1e-3)
    assert np.quantile(rel, 0.999) < 5e-7
    ctx.close()
