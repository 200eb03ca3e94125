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


# ---- the level table of the multilevel estimator (montecarlomultilevel.cc:115-165) -----------------------------------------
# Once per pass of the do-while the ranks need, per level, (samples, mean, variance, tau_int, cost).  Rows of levels a
# rank does not hold are zero, and one all-reduce(SUM) gives every rank the whole table -- the contract of
# LevelExchange::allreduce_sum in include/mlmcpi/multilevel.hh.  Which rank holds what is decided by
# partition_instances below (equal-cost shares of (level, chain) pairs).
N_LEVEL_FIELDS = 5


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
    n, var, tau = table[:, 0], table[:, 2], table[:, 3]
    cost = torch.ceil(tau) * table[:, 4]   # montecarlomultilevel.cc:193-204: C_eff = ceil(tau_int) x cost per sample
    total = torch.sqrt(var * cost).sum()
    targets = torch.ceil(2.0 / (epsilon * epsilon) * total * torch.sqrt(var / cost) * tau)
    return targets, bool((n >= targets).all())


def combine_levels(table):
    """montecarlomultilevel.cc:255-271: telescoping sum and its statistical error."""
    n, mean, var, tau = table[:, 0], table[:, 1], table[:, 2], table[:, 3]
    return float(mean.sum()), float(torch.sqrt((tau * var / n).sum()))


# ---- (level, chain) instances cut into equal-cost shares (BASELINE configs[4]: "level instances sharded over 8 GPUs") ---
# Level cost is dominated by the finest two levels (cost ~ M_lat), so "level l on rank l % world" leaves most ranks idle
# and cannot use more ranks than levels.  The unit that shards is the level INSTANCE = (level, block of chains): the
# list of all (level, chain) pairs, level-major, is cut into `world` contiguous shares of equal cost; a rank owns, per
# level, one contiguous chain block (possibly empty).  Chain indices are global Philox words, so the estimator does not
# depend on the cut.  What the ranks exchange per pass is additive: per level
#   [n, sum y, sum y^2, chains, sum m_c, sum m_c^2, site-steps]     (m_c = mean of chain c)
# summed by ONE all-reduce of [n_level, 7] doubles; mean, variance, tau_int (scatter of the independent chain means
# against the naive error) and cost per sample follow identically on every rank (finish_level_sums).
N_LEVEL_SUMS = 7


def partition_instances(costs, n_chains, world):
    """costs[l] = cost of one sample of one chain of level l.  Returns shares[rank] = {level: (chain0, count)}; every
    (level, chain) pair belongs to the rank whose share of the cumulative cost contains the pair's midpoint."""
    total = float(sum(c * n_chains for c in costs))
    shares = [dict() for _ in range(world)]
    cum = 0.0
    for level, c in enumerate(costs):
        owner = [min(world - 1, int((cum + (b + 0.5) * c) * world / total)) for b in range(n_chains)]
        for r in sorted(set(owner)):
            first = owner.index(r)
            shares[r][level] = (first, owner.count(r))
        cum += c * n_chains
    return shares


def level_sums(n_level, rows, device="cpu"):
    """rows: {level: the 7 additive sums of this rank's chain block of that level}."""
    t = torch.zeros((n_level, N_LEVEL_SUMS), dtype=torch.float64, device=device)
    for level, row in rows.items():
        t[level] = torch.as_tensor(row, dtype=torch.float64, device=device)
    return t


def finish_level_sums(sums):
    """[n_level, 7] reduced sums -> the [n_level, 5] table (samples, mean, variance, tau_int, cost) of level_table."""
    s = sums.double().cpu()
    out = torch.zeros((s.shape[0], N_LEVEL_FIELDS), dtype=torch.float64)
    for l in range(s.shape[0]):
        n, s1, s2, nc, m1, m2, work = (float(v) for v in s[l])
        if n < 2:
            continue
        mean = s1 / n
        var = n / (n - 1.0) * max(s2 / n - mean * mean, 0.0)
        tau = 1.0
        if nc > 1 and var > 0:
            var_c = max(m2 - m1 * m1 / nc, 0.0) / (nc - 1.0)   # scatter of the chain means
            tau = max((var_c / nc) / (var / n), 1e-3)
        out[l] = torch.tensor([n, mean, var, tau, work / n], dtype=torch.float64)
    return out


# ---- Statistics across ranks, as the reference defines them (common/statistics.cc:4-95) ----------------------------------
# Python twin of include/mlmcpi/statistics.hh: the estimator's state is the packed buffer
#   [avg, avg_longterm, avg2_longterm, avg3_longterm, avg4_longterm, n, n_longterm, S_k[0 .. k_max)]
# and every estimator is a function of the rank-AVERAGE of the first five entries and of S_k and of the rank-SUM of
# the counts: one all_reduce(SUM) of the buffer per convergence check (plus one slot per rank for the termination test
# of montecarlosinglelevel.cc:84-86) replaces the reference's ~10 scalar MPI_Allreduce calls.
class Statistics:
    AVG, AVG_LT, AVG2_LT, AVG3_LT, AVG4_LT, N, N_LT, SK0 = range(8)

    def __init__(self, k_max, group=None):
        self.k_max, self.group = k_max, group
        self.hard_reset()

    def hard_reset(self):
        self.buf = [0.0] * (self.SK0 + self.k_max)
        self.window = []

    def reset(self):
        self.buf[self.N] = 0.0
        self.buf[self.AVG] = 0.0

    def record_sample(self, q):
        b = self.buf
        b[self.N] += 1.0
        b[self.N_LT] += 1.0
        n, nl = b[self.N], b[self.N_LT]
        self.window.insert(0, q)
        if len(self.window) > self.k_max:
            self.window.pop()
        b[self.AVG] = ((n - 1.0) * b[self.AVG] + q) / (1.0 * n)
        b[self.AVG_LT] = ((nl - 1.0) * b[self.AVG_LT] + q) / (1.0 * nl)
        b[self.AVG2_LT] = ((nl - 1.0) * b[self.AVG2_LT] + q * q) / (1.0 * nl)
        b[self.AVG3_LT] = ((nl - 1.0) * b[self.AVG3_LT] + q * q * q) / (1.0 * nl)
        b[self.AVG4_LT] = ((nl - 1.0) * b[self.AVG4_LT] + q * q * q * q) / (1.0 * nl)
        for k, qk in enumerate(self.window):
            n_k = nl - k
            b[self.SK0 + k] = ((n_k - 1.0) * b[self.SK0 + k] + self.window[0] * qk) / (1.0 * n_k)

    def local_samples(self):
        return int(self.buf[self.N])

    def reduce(self, local_value=0.0):
        """ONE all-reduce: returns the global view g (rank-averages / rank-sums) and the list of every rank's
        `local_value`."""
        on = dist.is_available() and dist.is_initialized()
        size = dist.get_world_size(self.group) if on else 1
        rank = dist.get_rank(self.group) if on else 0
        t = torch.tensor(self.buf + [0.0] * size, dtype=torch.float64)
        t[len(self.buf) + rank] = local_value
        if size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        g = t[:len(self.buf)].tolist()
        for s in (self.AVG, self.AVG_LT, self.AVG2_LT, self.AVG3_LT, self.AVG4_LT):
            g[s] /= size
        for k in range(self.k_max):
            g[self.SK0 + k] /= size
        return g, t[len(self.buf):].tolist()

    # estimators on a reduced view (statistics.cc:29-95)
    @classmethod
    def variance(cls, g):
        return 1.0 * g[cls.N_LT] / (g[cls.N_LT] - 1.0) * (g[cls.SK0] - g[cls.AVG_LT] * g[cls.AVG_LT])

    @classmethod
    def tau_int(cls, g):
        a2 = g[cls.AVG_LT] * g[cls.AVG_LT]
        t = 0.0
        for k in range(1, len(g) - cls.SK0):
            t += (1. - k / (1.0 * g[cls.N_LT])) * (g[cls.SK0 + k] - a2)
        return max(1.0, 1.0 + 2.0 * t / (g[cls.SK0] - a2))

    @classmethod
    def error(cls, g):
        return math.sqrt(cls.tau_int(g) * cls.variance(g) / (1.0 * g[cls.N]))

    @classmethod
    def variance_error(cls, g):
        a = g[cls.AVG_LT]
        return math.sqrt(1.0 / g[cls.N_LT] * (g[cls.AVG4_LT] - 4 * a * g[cls.AVG3_LT] + 8 * a * a * g[cls.AVG2_LT] -
                                              g[cls.AVG2_LT] * g[cls.AVG2_LT] - 4 * a * a * a * a))


def run_single_level(draw_qoi, k_max, n_min, epsilon, n_samples=0, n_burnin=0, group=None, max_passes=1000):
    """The do-while of MonteCarloSingleLevel::evaluate (montecarlosinglelevel.cc:23-87) on this rank's chain with ONE
    all-reduce per pass; draw_qoi() = sampler->draw + qoi->evaluate.  Returns (Statistics, reduced view, passes)."""
    on = dist.is_available() and dist.is_initialized()
    size = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    s = Statistics(k_max, group)
    for _ in range(n_burnin):
        s.record_sample(draw_qoi())
    s.reset()
    n_target = n_samples if n_samples > 0 else n_min
    n_local = distribute_n(n_target, rank, size)
    passes = 0
    while True:
        for _ in range(s.local_samples(), n_local):
            s.record_sample(draw_qoi())
        g, counts = s.reduce(float(s.local_samples()))
        if n_samples == 0:
            n_target = int(math.ceil(Statistics.tau_int(g) * 2.0 / (epsilon * epsilon) * Statistics.variance(g)))
        n_local = distribute_n(n_target, rank, size)
        passes += 1
        if all(counts[r] >= distribute_n(n_target, r, size) for r in range(size)) or passes >= max_passes:
            return s, g, passes
