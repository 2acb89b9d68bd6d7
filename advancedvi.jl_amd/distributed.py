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


class DistributedRepGradELBO:
    """RepGradELBO with the MC batch sharded over the ranks of a torch.distributed process group.

    estimate_gradient(params, idx) -> (value, grad) device tensors, identical on every rank and equal
    (up to fp32 summation order) to the single-GPU estimate with n_samples = plan.n_samples."""

    def __init__(self, q, prob, n_samples, entropy, seed, device=0, group=None, force_collective=False):
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
        self.partials = self.ctx.empty(self.ctx.partials_len)
        self.value = self.ctx.empty(1)
        self.grad = self.ctx.empty(self.ctx.params_len)
        self._destructure = destructure

    def estimate_gradient(self, params, idx):
        p = self.ctx.to_device(params)
        if self.world == 1 and not self.force_collective:
            return self.ctx.estimate_gradient(p, idx, self.value, self.grad)
        self.ctx.estimate_partials(p, idx, self.partials)
        allreduce_partials(self.partials, self.group, force=self.force_collective)
        return self.ctx.finalize(p, self.partials, self.value, self.grad)

    def close(self):
        self.ctx.close()
