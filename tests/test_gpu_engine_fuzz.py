"""Randomised sweep over the batch engine's own shapes (d and n_mc multiples of 128 -- round 6: also of 32, padded --, d <= 2048): family full-rank f32, diagonal / dense target,
the five entropy estimators, batch lengths that cut into one or several steps of different widths -- every checked estimate against the
single call (to rounding) and against the fp64 oracle on the device's own eps.  Fixed seed: reproducible."""
import numpy as np
import pytest

import advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED, assert_batch_matches_single, make_family, make_problem

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        d = int(rng.choice([128, 256, 384, 512, 640, 896, 1024, 1280]))
        M = int(rng.choice([128, 256, 384, 512]))
        kind = ("diag", "dense")[int(rng.integers(0, 2))]
        ent = int(rng.integers(0, 5))
        count = int(rng.choice([2, 7, 17, 20, 26, 27, 40, 81, 97]))
        out.append((d, M, kind, ent, count))
    return out


def _ragged(n, seed):
    """round 6: d and n_mc multiples of 32 that are NOT multiples of 128 -- the geometry is padded to whole tiles, whole 32-blocks of padding carry
    zero planes and are neither summed nor stored (diagonal target; the dense target and the STL estimators fall to the single calls: bitwise)"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        d = int(rng.choice([160, 288, 416, 544, 992, 1056, 1504, 2016]))
        M = int(rng.choice([128, 160, 224, 288, 352, 480]))
        ent = int(rng.integers(0, 5))
        kind = "diag" if rng.integers(0, 5) else "dense"
        count = int(rng.choice([2, 5, 20, 27, 83]))
        out.append((d, M, kind, ent, count))
    return out + [(1024, 160, "diag", 0, 20), (160, 256, "diag", 2, 9), (2016, 2016, "diag", 0, 2)]


_WIDE = [(256, 2048, "diag", 0, 5), (128, 1024, "dense", 2, 9), (256, 1024, "diag", 3, 18), (128, 2048, "dense", 0, 3), (2048, 128, "dense", 1, 4)]   # the widest sample counts (16 re-basing boundaries in the VJP) and the largest d


@pytest.mark.parametrize("d,M,kind,ent,count", _cases(28, 20260929) + _WIDE + _ragged(16, 20260930))
def test_engine_fuzz(d, M, kind, ent, count):
    rng = np.random.default_rng(d * 31 + M * 7 + ent + count)
    q, q_o = make_family(rng, d, avi.FULLRANK, np.float32, mu_scale=0.5)
    prob, tgt = make_problem(rng, kind, d, np.float32)
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
    ref = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED)
    ctx.set_problem(prob)
    ref.set_problem(prob)
    p, pr = ctx.to_device(params), ref.to_device(params)
    eng = bool(ctx.batch_takes_engine(p))
    idx0 = int(rng.integers(0, 1 << 30))
    vals, grads = ctx.estimate_gradient_each(p, idx0, count)
    v_last, g_last = ctx.empty(1), ctx.empty(ctx.params_len)
    ctx.estimate_gradient_n(p, idx0, count, v_last, g_last)
    ctx.synchronize()
    vals, grads = vals.cpu().numpy(), grads.cpu().numpy()
    p64 = params.astype(np.float64)
    picks = sorted({0, count // 2, count - 1, int(rng.integers(0, count))})
    for i in picks:
        v1, g1 = ref.estimate_gradient(pr, idx0 + i)
        g1n = g1.cpu().numpy()
        if eng:
            assert_batch_matches_single(vals[i], v1.item(), grads[i], g1n, True, (i, count), ulps=1 if ent in (3, 4) else 0)
        else:
            assert float(vals[i]) == float(v1.item()) and np.array_equal(grads[i], g1n), i
        _, eps = ref.sample(pr, idx0 + i)
        o = O.estimate_gradient(p64, d, avi.FULLRANK, tgt, eps.cpu().numpy().astype(np.float64), ent)
        assert abs(float(vals[i]) - o["value"]) <= 2e-5 * max(abs(o["value"]), 1.0), (i, float(vals[i]), o["value"])
        assert np.linalg.norm(grads[i].astype(np.float64) - o["grad"]) <= 4e-5 * max(1.0, np.linalg.norm(o["grad"])), i
    # the batch entry's last estimate is the each entry's last row (same kernels, same lanes)
    assert float(v_last.item()) == float(vals[count - 1])
    assert np.array_equal(g_last.cpu().numpy(), grads[count - 1])
    # structural zeros above the diagonal
    gC = grads[count - 1][d:].reshape(d, d, order="F")
    assert np.all(np.triu(gC, 1) == 0.0)
    ctx.close()
    ref.close()
