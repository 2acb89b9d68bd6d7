"""Multi-GPU RepGradELBO: shard the MC batch, one RCCL all-reduce on the gradient partials.

The reference has no counterpart (single Julia task).  The estimator is a mean over M
i.i.d. samples plus parameter-only entropy terms, so rank r owns global sample columns
[r*M/R, (r+1)*M/R) -- regenerated from the same counter-based eps stream (shard-invariant Philox
indices) -- and produces the un-normalised partial buffer of include/mivi.h
    [sum_m W_im ; sum_m W (x) eps ; sum_m ell_m ; sum_m 0.5|eps_m|^2]
One `all_reduce(SUM)` (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests) and a finalize kernel
(scale by -1/M_total, add the closed-form entropy terms once) complete the estimate on every rank.
One process per GPU, launched by torch.distributed.run."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class ShardPlan:
    """Sample-axis partition for `world` ranks: contiguous, sizes differ by at most one."""

    n_samples: int
    world: int

    def __post_init__(self):
        if self.world < 1 or self.n_samples < self.world:
            raise ValueError("need at least one MC sample per rank")

    def count(self, rank: int) -> int:
        base, rem = divmod(self.n_samples, self.world)
        return base + (1 if rank < rem else 0)

    def offset(self, rank: int) -> int:
        base, rem = divmod(self.n_samples, self.world)
        return rank * base + min(rank, rem)

    def range(self, rank: int):
        o = self.offset(rank)
        return o, o + self.count(rank)


def partials_len(d: int, family: int) -> int:
    """Length of the shard-additive buffer (mivi_partials_len): [sum W; sum W (x) eps; sum ell; sum 0.5|eps|^2],
    the full-rank outer product packed to its lower triangle (d(d+1)/2 entries) so the all-reduce moves half the bytes."""
    return (2 * d if family == 0 else d + d * (d + 1) // 2) + 2


def allreduce_partials(partials, group=None, force=False):
    """In-place SUM all-reduce of the partial buffer over the process group (RCCL on GPUs).
    `force` issues the collective even for a single-rank group (used to exercise the path on one GPU)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        dist.all_reduce(partials, op=dist.ReduceOp.SUM, group=group)
    return partials


def slice_len(n_partials: int, world: int) -> int:
    """Elements per rank of the partial vector padded to a multiple of `world` (mivi_slice_len)."""
    return (n_partials + world - 1) // world


def p2p_geometry(n_partials: int, world: int):
    """(slice length, chunk length, chunk workgroups, value-owner rank) of the peer-to-peer exchange (mivi_p2p_geometry: host-only,
    identical on every rank)."""
    import ctypes as C

    from . import _lib
    out = (C.c_int64 * 4)()
    _lib.load().mivi_p2p_geometry(int(n_partials), int(world), out)
    return int(out[0]), int(out[1]), int(out[2]), int(out[3])


def reduce_scatter_partials(padded, out_slice, group=None):
    """SUM reduce-scatter of the padded partial vector: rank r receives elements [r n, (r+1) n).  RCCL does it in one
    collective; gloo (CPU tests) has no reduce-scatter, there it is an all-reduce followed by taking the slice."""
    import torch.distributed as dist

    if dist.get_backend(group) == "gloo":
        dist.all_reduce(padded, op=dist.ReduceOp.SUM, group=group)
        n = out_slice.numel()
        r = dist.get_rank(group)
        out_slice.copy_(padded[r * n:(r + 1) * n])
    else:
        dist.reduce_scatter_tensor(out_slice, padded, op=dist.ReduceOp.SUM, group=group)
    return out_slice


def allgather_final(final_slice, packed_final, group=None):
    """All-gather of the finalised slices into the packed final vector (rank order)."""
    import torch.distributed as dist

    dist.all_gather_into_tensor(packed_final, final_slice, group=group)
    return packed_final


class DistributedRepGradELBO:
    """RepGradELBO with the MC batch sharded over the ranks of a torch.distributed process group.

    estimate_gradient(params, idx) -> (value, grad) device tensors, identical on every rank and equal
    (up to fp32 summation order) to the single-GPU estimate with n_samples = plan.n_samples."""

    def __init__(self, q, prob, n_samples, entropy, seed, device=0, group=None, force_collective=False, mode="auto"):
        """mode "rsag": reduce-scatter -> every rank finalises its 1/R slice (mivi_finalize_slice) -> all-gather -> unpack
        (mivi_unpack_final); "allreduce": all-reduce + the whole finalisation on every rank (first-round structure, A/B);
        "mivi": the collective runs behind the C ABI (mivi_comm_init / mivi_estimate_gradient_dist, RCCL opened by libmivi;
        the unique id travels through the torch process group)."""
        import torch.distributed as dist

        from .context import MiviContext
        from .families import destructure

        self.group = group
        self.force_collective = force_collective
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        self.plan = ShardPlan(int(n_samples), self.world)
        self.ctx = MiviContext(q.eltype, q.family, len(q), self.plan.count(self.rank), entropy.code, seed, device=device,
                               m_offset=self.plan.offset(self.rank), m_total=self.plan.n_samples)
        self.ctx.set_problem(prob)
        L = self.ctx.partials_len
        if mode == "auto":   # two collectives cost one more launch + rendezvous (~10 us) than one all-reduce: they pay off only for
            mode = "rsag" if L * self.ctx.np_dtype.itemsize >= (16 << 20) else "allreduce"   # bandwidth-bound vectors
        self.mode = mode
        self.n_slice = slice_len(L, self.world)
        self.padded = self.ctx.empty(self.n_slice * self.world).zero_()   # partial vector + padding (stays zero)
        self.partials = self.padded[:L]
        self.slice_sum = self.ctx.empty(self.n_slice)
        self.final = self.ctx.empty(self.n_slice * self.world)
        self.value = self.ctx.empty(1)
        if mode == "mivi":
            import torch
            idt = torch.zeros(128, dtype=torch.uint8, device=self.ctx.tdevice)
            if self.rank == 0:
                idt.copy_(torch.frombuffer(bytearray(self.ctx.comm_unique_id()), dtype=torch.uint8))
            if self.world > 1:
                dist.broadcast(idt, src=0, group=group)
            self.ctx.comm_init(bytes(idt.cpu().numpy().tobytes()), self.rank, self.world)
        self.grad = self.ctx.empty(self.ctx.params_len)
        self._destructure = destructure

    def estimate_gradient(self, params, idx):
        p = self.ctx.to_device(params)
        if self.world == 1 and not self.force_collective and self.mode != "mivi":
            return self.ctx.estimate_gradient(p, idx, self.value, self.grad)
        if self.mode == "mivi":
            return self.ctx.estimate_gradient_dist(p, idx, self.value, self.grad)
        self.ctx.estimate_partials(p, idx, self.partials)
        if self.mode == "allreduce":
            allreduce_partials(self.partials, self.group, force=self.force_collective)
            return self.ctx.finalize(p, self.partials, self.value, self.grad)
        reduce_scatter_partials(self.padded, self.slice_sum, self.group)
        mine = self.final[self.rank * self.n_slice:(self.rank + 1) * self.n_slice]
        self.ctx.finalize_slice(p, self.slice_sum, self.rank, self.world, mine)
        allgather_final(mine, self.final, self.group)
        return self.ctx.unpack_final(self.final, self.value, self.grad)

    def close(self):
        self.ctx.close()
