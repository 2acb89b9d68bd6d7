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


def _p2p_worker(rank, world, port, family, q_out):
    """The peer-to-peer exchange protocol (csrc/kernels_p2p.hip) played over gloo by two real processes: what a rank would store into
    its peers' staging / final areas travels by all_gather, the areas are double-buffered by epoch parity over three exchanges and
    every rank only ever touches ITS slice in the reduce phase (product geometry: advancedvi_jl_amd.distributed.p2p_geometry)."""
    sys.path.insert(0, ROOT)
    from advancedvi_jl_amd.distributed import ShardPlan, p2p_geometry, partials_len
    from oracle import oracle as O
    from tests.helpers import SEED, make_family, make_problem

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d, M, ent = 12, 11, 3
        rng = np.random.default_rng(78)
        _, q = make_family(rng, d, family)
        _, tgt = make_problem(rng, "dense", d)
        params = O.destructure(q)
        plan = ShardPlan(M, world)
        lo, hi = plan.range(rank)
        L = partials_len(d, family)
        n, cn, G, vs = p2p_geometry(L, world)
        assert (n, cn, G, vs) == O.p2p_geometry(L, world)
        stage = np.full((2, world, n), np.nan)          # my staging area: [parity][source rank][my slice]
        fin = np.full((2, world * n), np.nan)           # my final area
        worst = 0.0
        for epoch, idx in enumerate((9, 10, 11), start=1):
            p = epoch & 1
            part = O.estimate_gradient(params, d, family, tgt, O.philox_normal(SEED, idx, d, lo, hi, f64=True), ent)["partials"]
            padded = np.concatenate([part, np.zeros(world * n - L)])
            # phase 1: every rank "stores" slice s of its vector into rank s's staging area
            everyone = [torch.zeros(world * n, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(everyone, torch.from_numpy(padded.copy()))
            for src in range(world):
                stage[p, src] = everyone[src].numpy()[rank * n:(rank + 1) * n]
            # phase 2: reduce MY slice in rank order, finalise it (oracle.finalize_slice: the arithmetic of the kernel), "store" it everywhere
            mine = O.finalize_slice(stage[p].sum(axis=0) if world > 1 else stage[p, 0], rank * n, params, d, family, ent, M, L)
            slices = [torch.zeros(n, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(slices, torch.from_numpy(mine.copy()))
            fin[p] = np.concatenate([t.numpy() for t in slices])
            # phase 3: unpack
            value, grad = O.unpack_final(fin[p], d, family)
            ref = O.estimate_gradient(params, d, family, tgt, O.philox_normal(SEED, idx, d, 0, M, f64=True), ent)
            worst = max(worst, abs(value - ref["value"]) / abs(ref["value"]), float(np.max(np.abs(grad - ref["grad"]))))
            assert np.isnan(fin[p ^ 1]).all() or epoch > 1   # the other parity is untouched by this exchange
        q_out.put((rank, worst, vs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("family", [0, 1])
def test_p2p_exchange_protocol_over_gloo(family):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_p2p_worker, args=(r, 2, port, family, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for _ in range(2):
        rank, worst, vs = q.get(timeout=10)
        assert worst < 1e-11 and 0 <= vs < 2


def _engine_worker(rank, world, port, q_out):
    """Round 6: a sharded BATCH the way the batch engine runs it across ranks (csrc/api_batch.hip fb_batch, dist) -- every rank forms the
    partial vectors of ALL estimates of a step on its own sample columns, in the engine's tile-packed layout (oracle.engine_partials), ONE
    all-reduce sums the whole step's vectors, every rank finalises every lane (oracle.finalize_engine_partials = k_fb_finalize_parts)."""
    sys.path.insert(0, ROOT)
    from advancedvi_jl_amd.distributed import ShardPlan, allreduce_partials
    from oracle import oracle as O
    from tests.helpers import SEED, make_family, make_problem

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d, M, lanes, idx0, ent = 128, 24, 3, 5, 2         # 24 samples per estimate over 2 ranks, three estimates in one step
        rng = np.random.default_rng(79)
        _, q = make_family(rng, d, 1)
        _, tgt = make_problem(rng, "diag", d)
        params = O.destructure(q)
        lo, hi = ShardPlan(M, world).range(rank)
        step = np.concatenate([O.engine_partials(O.estimate_gradient(params, d, 1, tgt, O.philox_normal(SEED, idx0 + l, d, lo, hi, f64=True), ent)["partials"], d)
                               for l in range(lanes)])
        t = torch.from_numpy(step.copy())
        allreduce_partials(t)                            # ONE collective for the step
        n = t.numel() // lanes
        worst = 0.0
        for l in range(lanes):
            value, grad = O.finalize_engine_partials(t.numpy()[l * n:(l + 1) * n], params, d, ent, M)
            ref = O.estimate_gradient(params, d, 1, tgt, O.philox_normal(SEED, idx0 + l, d, 0, M, f64=True), ent)
            worst = max(worst, abs(value - ref["value"]) / abs(ref["value"]), float(np.max(np.abs(grad - ref["grad"]))))
            assert not np.any(np.triu(grad[d:].reshape(d, d, order="F"), 1))
        q_out.put((rank, worst))
    finally:
        dist.destroy_process_group()


def test_engine_sharded_step_over_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    for _ in range(2):
        rank, worst = q.get(timeout=10)
        assert worst < 1e-11
