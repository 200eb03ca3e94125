"""Sharding of independent Markov chains over ranks and the one collective of the path.

The reference parallelises by running one chain per MPI rank with a distinct seed and combining
statistics with scalar MPI_Allreduce calls (mpi/mpi_random.cc:5-29, mpi/mpi_wrapper.cc:187-202,
common/statistics.cc:29-95).  Here every rank (one process per GPU) owns a contiguous block of
global chain indices -- the chain index is a word of the Philox counter, so a chain's stream does
not depend on which GPU runs it -- and the per-chain moment sums produced by
mlmcpi_stats_accumulate are combined with ONE all-reduce of a packed fp64 buffer (RCCL over xGMI
with backend "nccl"; gloo on CPU for tests).  No collective sits inside the sweep itself.
"""
import math

import torch
import torch.distributed as dist

N_MOMENTS = 5  # [n, sum q, sum q^2, sum q^3, sum q^4] per chain


def distribute_n(n, rank, world):
    """mpi/mpi_wrapper.cc:187-202: split n samples/chains over ranks, remainder to the low ranks."""
    base = n // world
    return base + 1 if rank < n - base * world else base


def chain_block(n_chains, rank, world):
    """(first global chain index, number of chains) owned by `rank`."""
    first = sum(distribute_n(n_chains, r, world) for r in range(rank))
    return first, distribute_n(n_chains, rank, world)


def pack_moments(acc):
    """Sum the per-chain moment rows [B, 5] of this rank into one packed vector [5]."""
    return acc.reshape(-1, N_MOMENTS).sum(dim=0)


def allreduce_moments(packed, group=None):
    """The path's only collective: element-wise sum of the packed fp64 moments over all ranks."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    return packed


def summarise(packed, n_chains_total=None, chain_means=None):
    """Mean / variance (statistics.cc:30-36 without the autocorrelation window) from packed sums.
    If per-chain means are given, the error is the scatter of independent chain means, which is
    insensitive to autocorrelation inside a chain."""
    n, s1, s2 = float(packed[0]), float(packed[1]), float(packed[2])
    mean = s1 / n
    var = n / (n - 1.0) * (s2 / n - mean * mean) if n > 1 else float("nan")
    out = {"samples": int(n), "mean": mean, "variance": var, "naive_error": math.sqrt(var / n) if n > 1 else float("nan")}
    if chain_means is not None and chain_means.numel() > 1:
        out["error"] = float(chain_means.std(unbiased=True)) / math.sqrt(chain_means.numel())
    return out


# ---- level sharding of the multilevel estimator -------------------------------------------
# Levels are independent estimators (montecarlomultilevel.cc:27-45): level l runs on rank l % world.  Once per pass of
# the do-while of montecarlomultilevel.cc:115-165 the ranks exchange a table [n_level, 5] =
# (samples, mean, variance, tau_int, cost): every rank fills the rows of its levels, leaves the others zero, and one
# all-reduce(SUM) of the table (= the all-gather of disjoint rows) gives every rank the whole picture.  This is the
# same contract as LevelExchange::allreduce_sum in include/mlmcpi/multilevel.hh.
N_LEVEL_FIELDS = 5


def level_owner(level, world):
    return level % world


def owned_levels(n_level, rank, world):
    return [l for l in range(n_level) if level_owner(l, world) == rank]


def level_table(n_level, rows, device="cpu"):
    """rows: {level: (samples, mean, variance, tau_int, cost)} for the levels this rank owns."""
    t = torch.zeros((n_level, N_LEVEL_FIELDS), dtype=torch.float64, device=device)
    for level, row in rows.items():
        t[level] = torch.tensor(row, dtype=torch.float64, device=device)
    return t


def allreduce_level_table(table, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
    return table


def level_targets(table, epsilon):
    """montecarlomultilevel.cc:148-164: samples needed per level for a statistical error epsilon / sqrt(2);
    returns (targets [n_level], sufficient)."""
    n, var, tau, cost = table[:, 0], table[:, 2], table[:, 3], table[:, 4]
    total = torch.sqrt(var * cost).sum()
    targets = torch.ceil(2.0 / (epsilon * epsilon) * total * torch.sqrt(var / cost) * tau)
    return targets, bool((n >= targets).all())


def combine_levels(table):
    """montecarlomultilevel.cc:255-271: telescoping sum and its statistical error."""
    n, mean, var, tau = table[:, 0], table[:, 1], table[:, 2], table[:, 3]
    return float(mean.sum()), float(torch.sqrt((tau * var / n).sum()))
