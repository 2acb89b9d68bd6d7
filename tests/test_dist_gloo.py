"""World-size-2 CPU test (gloo) of the multi-GPU path: sample-axis sharding (ShardPlan), the shard-additive
partial buffer, the all-reduce wrapper the GPU path calls (advancedvi_jl_amd.distributed.allreduce_partials)
and the finalize arithmetic.  On CPU the per-shard partials are produced by the oracle (the HIP kernels need a
GPU); what is under test is the shard/collective logic and shard-invariance of the Philox stream."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, family, ent, q_out):
    sys.path.insert(0, ROOT)
    from advancedvi_jl_amd.distributed import ShardPlan, allreduce_partials, partials_len
    from oracle import oracle as O
    from tests.helpers import SEED, make_family, make_problem

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d, M, idx = 12, 11, 9           # 11 samples over 2 ranks: uneven shards (6 + 5)
        rng = np.random.default_rng(77)
        _, q = make_family(rng, d, family)
        _, tgt = make_problem(rng, "dense", d)
        params = O.destructure(q)
        plan = ShardPlan(M, world)
        lo, hi = plan.range(rank)
        eps_local = O.philox_normal(SEED, idx, d, lo, hi, f64=True)     # this rank's slice of the ONE stream
        part = O.estimate_gradient(params, d, family, tgt, eps_local, ent)["partials"]
        assert part.shape[0] == partials_len(d, family)
        t = torch.from_numpy(part.copy())
        allreduce_partials(t)                                           # product code: the collective
        value, grad = O.finalize_partials(t.numpy(), params, d, family, ent, M)
        # the sharded finalisation: reduce-scatter -> every rank finalises its 1/R slice -> all-gather -> unpack.  Product
        # code: slice_len / reduce_scatter_partials / allgather_final; the slice arithmetic is the host restatement of the
        # kernels (oracle.finalize_slice / unpack_final, compared with the kernels themselves in tests/test_gpu_dist.py).
        from advancedvi_jl_amd.distributed import allgather_final, reduce_scatter_partials, slice_len
        L = part.shape[0]
        n = slice_len(L, world)
        padded = torch.zeros(n * world, dtype=torch.float64)
        padded[:L] = torch.from_numpy(part)
        mine_sum = torch.zeros(n, dtype=torch.float64)
        reduce_scatter_partials(padded, mine_sum)
        fin = torch.zeros(n * world, dtype=torch.float64)
        mine = torch.from_numpy(O.finalize_slice(mine_sum.numpy(), rank * n, params, d, family, ent, M, L))
        allgather_final(mine, fin)
        value2, grad2 = O.unpack_final(fin.numpy(), d, family)
        assert abs(value2 - value) <= 1e-13 * abs(value) and np.max(np.abs(grad2 - grad)) <= 1e-13 * max(1.0, np.max(np.abs(grad)))
        if rank == 0:
            eps_full = O.philox_normal(SEED, idx, d, 0, M, f64=True)
            ref = O.estimate_gradient(params, d, family, tgt, eps_full, ent)
            q_out.put((abs(value - ref["value"]) / abs(ref["value"]), float(np.max(np.abs(grad - ref["grad"])))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("family", [0, 1])
@pytest.mark.parametrize("ent", [0, 3])
def test_sharded_estimate_equals_single(family, ent):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, family, ent, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    rel_v, max_g = q.get(timeout=10)
    assert rel_v < 1e-12 and max_g < 1e-11
