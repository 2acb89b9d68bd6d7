#!/usr/bin/env python
"""Generates tests/golden/repgradelbo_cells.npz with the CPU oracle (oracle/oracle.py, float64).

The reference is pure Julia (no toolchain here) and commits no numeric vectors, so these
fixtures are produced by the restatement that tests/test_oracle_pinning.py pins against the reference's
known-answer tests.  One cell per {mean-field, full-rank} x {5 entropy estimators} x {diag, dense, logreg0,
logreg1, funnel} target: inputs (seed, estimate_idx, d, M, params, target parameters) and expected outputs
(eps of the Philox stream, Z, ell, G, entropy, value, grad; full-rank cells also the Stein gradient / Hessian estimate).  Re-run: `python tests/golden/gen_golden.py`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests.helpers import SEED, make_family, make_problem  # noqa: E402

D, M, IDX = 8, 6, 5
KINDS = ["diag", "dense", "logreg0", "logreg1", "funnel"]


def target_arrays(kind, tgt):
    if kind == "diag":
        return dict(t_mean=tgt.mean, t_std=tgt.std)
    if kind == "dense":
        return dict(t_mean=tgt.mean, t_L=tgt.L)
    if kind.startswith("logreg"):
        return dict(t_X=tgt.X, t_y=tgt.y, t_likeadj=np.array(tgt.likeadj))
    return dict(t_sigma_v=np.array(tgt.sigma_v))


def main():
    out = {"meta_seed": np.array(SEED, dtype=np.uint64), "meta_idx": np.array(IDX), "meta_d": np.array(D), "meta_M": np.array(M)}
    for family in (O.MEANFIELD, O.FULLRANK):
        for kind in KINDS:
            rng = np.random.default_rng(1000 * family + KINDS.index(kind))
            _, q = make_family(rng, D, family)
            _, tgt = make_problem(rng, kind, D)
            params = O.destructure(q)
            eps = O.philox_normal(SEED, IDX, D, 0, M, f64=True)   # the f64 stream (bit-defined by Philox + Box-Muller)
            key = f"f{family}_{kind}"
            out[key + "_params"] = params
            out[key + "_eps"] = eps
            for k, v in target_arrays(kind, tgt).items():
                out[key + "_" + k] = np.asarray(v)
            for ent in range(5):
                r = O.estimate_gradient(params, D, family, tgt, eps, ent)
                out[f"{key}_e{ent}_value"] = np.array(r["value"])
                out[f"{key}_e{ent}_grad"] = r["grad"]
                out[f"{key}_e{ent}_entropy"] = np.array(r["entropy"])
                if ent == 0:
                    out[key + "_Z"] = r["Z"]
                    out[key + "_ell"] = r["ell"]
                    out[key + "_G"] = r["G"]
            if family == O.FULLRANK:   # gaussian_expectation_gradient_and_hessian! (Stein branch) on the same inputs
                lp, g, H = O.gaussian_expectation_gradient_and_hessian(q, tgt, eps)
                out[key + "_stein_logpi"] = np.array(lp)
                out[key + "_stein_grad"] = g
                out[key + "_stein_hess"] = H
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "repgradelbo_cells.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
