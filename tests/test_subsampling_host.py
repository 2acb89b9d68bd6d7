"""Host logic of the subsampling layer (no GPU): ReshufflingBatchSubsampling semantics (src/reshuffling.jl:13-60), the
`subsample` protocol (src/AdvancedVI.jl:313, docs/src/tutorials/subsampling.md:99-110) and, on the oracle, the identity the
reference's own test relies on: the average of likelihood-rescaled minibatch gradients under a shared sample batch is the
full-data gradient (test/general/subsampledobj.jl:62-89)."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from advancedvi_jl_amd import subsampling as S
from oracle import oracle as O


@pytest.mark.parametrize("n,bs", [(8, 1), (8, 3), (8, 4), (10, 10), (7, 20)])
def test_reshuffling_epochs_cover_the_dataset(n, bs):
    sub = avi.ReshufflingBatchSubsampling(np.arange(n), bs)
    assert len(sub) == -(-n // bs)                       # reshuffling.jl:23-25
    rng = avi.PhiloxRNG(7)
    st = S.init_subsampling(rng, sub)
    assert st.epoch == 1 and rng.counter == 1            # one shuffle = one index off the rng
    for epoch in (1, 2, 3):
        seen = []
        for k in range(len(sub)):
            batch, st, info = S.step_subsampling(rng, sub, st)
            seen.extend(batch.tolist())
            assert info["step"] == k + 1
            # the epoch counter moves when the LAST batch of an epoch is handed out (reshuffling.jl:46-56)
            assert info["epoch"] == (epoch + 1 if k == len(sub) - 1 else epoch)
            assert len(batch) == (bs if (k + 1) * bs <= n else n - k * bs)
        assert sorted(seen) == list(range(n))


def test_reshuffling_drops_a_short_trailing_batch_for_gradients():
    sub = avi.ReshufflingBatchSubsampling(np.arange(8), 3)     # batches of 3, 3, 2
    rng = avi.PhiloxRNG(11)
    st = S.init_subsampling(rng, sub)
    sizes, steps = [], []
    for _ in range(6):
        batch, st, info = S.step_subsampling(rng, sub, st, True)
        sizes.append(len(batch)); steps.append(info["step"])
    # the short batch is replaced by the first batch of the next epoch (step index 1), which is then not revisited
    assert sizes == [3, 3, 3, 3, 3, 3] and steps == [1, 2, 1, 2, 1, 2]


def test_reshuffling_is_deterministic_in_the_rng():
    sub = avi.ReshufflingBatchSubsampling(np.arange(20), 4)
    runs = []
    for _ in range(2):
        rng = avi.PhiloxRNG(0x38bef07cf9cc549d)
        st = S.init_subsampling(rng, sub)
        seq = []
        for _ in range(12):
            b, st, _ = S.step_subsampling(rng, sub, st, True)
            seq.append(b.tolist())
        runs.append(seq)
    assert runs[0] == runs[1]
    other = S.init_subsampling(avi.PhiloxRNG(1), sub)
    assert [b.tolist() for _, b in other.batches] != [b for b in runs[0][:5]]


def test_subsample_protocol():
    class Plain:
        pass
    m = Plain()
    assert avi.subsample(m, [0, 1]) is m                 # unspecialised models pass through (AdvancedVI.jl:313)
    q = avi.MeanFieldGaussian(np.zeros(3), np.ones(3))
    assert avi.subsample(q, [0]) is q
    X = np.arange(24.0).reshape(8, 3)
    prob = avi.LogRegProblem(X, np.zeros(8), likeadj=1.0)
    s1 = avi.subsample(prob, [5, 2, 7, 0])
    assert isinstance(s1, avi.LogRegSubset) and s1.parent is prob and s1.likeadj == 2.0 and s1.dimension() == 4
    s2 = avi.subsample(s1, [1, 3])                       # rows 2 and 0 of the parent
    assert s2.batch.tolist() == [2, 0] and s2.likeadj == 4.0
    with pytest.raises(ValueError):
        avi.subsample(prob, [])


@pytest.mark.parametrize("bs", [1, 2, 4])
def test_minibatch_gradients_average_to_the_full_gradient_on_the_oracle(bs):
    rng = np.random.default_rng(5)
    n, p, M = 8, 3, 6
    X = rng.normal(size=(n, p)); y = (rng.uniform(size=n) < 0.5).astype(float)
    tgt = O.LogRegTarget(X, y)
    d = p + 1
    q = O.MvLocationScale(rng.normal(size=d) * 0.1, np.tril(rng.normal(size=(d, d)) * 0.1) + np.eye(d))
    eps = rng.normal(size=(d, M))
    full = O.estimate_gradient(O.destructure(q), d, O.FULLRANK, tgt, eps, O.ENT_CLOSED_FORM)
    perm = rng.permutation(n)
    grads = [O.estimate_gradient(O.destructure(q), d, O.FULLRANK, tgt.subsample(perm[k:k + bs]), eps, O.ENT_CLOSED_FORM)
             for k in range(0, n, bs)]
    # prior + entropy terms appear once per minibatch estimate, the likelihood n/bs times a 1/(n/bs) share
    assert np.allclose(np.mean([g["grad"] for g in grads], axis=0), full["grad"], rtol=1e-12, atol=1e-12)
    assert np.isclose(np.mean([g["value"] for g in grads]), full["value"], rtol=1e-12)
