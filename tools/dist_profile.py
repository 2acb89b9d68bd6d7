#!/usr/bin/env python
"""Developer: the sharded step on ONE GPU (world 1, peer-to-peer route forced) -- mivi_profile_dist's four legs at the north-star shape.
Run it with MIVI_P2P_DIRECT=0 for the ring-slot + push reference."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi  # noqa: E402
from tests.helpers import SEED  # noqa: E402

d, M = 1024, 256
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
params, _ = avi.destructure(q)
import torch  # noqa: E402
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED, stream=st.cuda_stream)
    ctx.set_problem(avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32)))
    ctx.p2p_attach([ctx.p2p_export(0, 1)])
    p = ctx.to_device(params)
    for reps in (20, 100):
        print(f"reps={reps}:", {k: round(v, 2) for k, v in ctx.profile_dist(p, reps).items()}, flush=True)
    ctx.close()
