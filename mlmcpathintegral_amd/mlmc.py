"""Batched multilevel Monte Carlo over device chains (1-D actions), level instances sharded over ranks.

The telescoping estimator of montecarlo/montecarlomultilevel.cc:71-204 with B independent chains per level:
    Q = sum_l E[Y_l],   Y_{L-1} = Q_{L-1}(x)                    (coarsest level: HMCSampler)
                        Y_l     = Q_l(theta) - Q_{l+1}(x_c)      (TwoLevelMetropolisStep fed by the sampler of level l+1)
A level INSTANCE is (level, block of chains); chains.partition_instances cuts the (level, chain) pairs into `world`
equal-cost shares, so a rank owns at most one chain block per level.  Per pass the ranks all-reduce the additive
[n_level, 7] sums of chains.py.  Everything that computes runs through the C ABI (ops.PathHMC, ops.PathTwoLevelStep,
mlmcpi_qoi_*, mlmcpi_stats_accumulate); this module is the host loop only -- the Python twin of MonteCarloMultiLevel
in include/mlmcpi/multilevel.hh, for tests and bench.py.
"""
import torch

from . import abi, chains, ops


class PathLevel:
    """One level instance: the sampler of level l+1 (or of the coarsest level) and the two-level step, chains
    [chain0, chain0 + B) of the level."""

    def __init__(self, acts, level, B, nt, dts, seed, chain0=0, n_sub=2, qoi=None):
        self.level, self.B, self.n_sub, self.chain0 = level, B, n_sub, chain0
        self.coarsest = level == len(acts) - 1
        self.qoi = qoi or ops.qoi_xsquared
        src = level if self.coarsest else level + 1           # level whose sampler feeds this estimator
        self.act_src = acts[src]
        # montecarlomultilevel.cc:27-45: every level owns its sampler; distinct Philox keys per level instance
        self.hmc = ops.PathHMC(acts[src], B, nt, dts[src], seed=seed + 7919 * (level + 1), chain0=chain0)
        self.x = ops.path_initialise(acts[src], B, seed + 7919 * (level + 1), chain0)
        self.step = None if self.coarsest else ops.PathTwoLevelStep(acts[level], acts[level + 1], B,
                                                                     seed=seed + 104729 * (level + 1), chain0=chain0)
        self.acc = torch.zeros((B, chains.N_MOMENTS), dtype=torch.float64, device=self.x.device)
        self.site_steps = 0   # leapfrog site-steps + two-level site passes issued (bench accounting)
        self.n_draws = 0
        self.step_accepted = torch.zeros(B, dtype=torch.int64, device=self.x.device)

    def thermalise(self, n, dt_fine=None):
        """Untimed burn-in (ops.hmc_thermalise) of the feeding sampler, and of the two-level step's own state theta by a
        direct HMC run on its level: theta must be an equilibrium sample of the fine action before the first measured step.
        (r02 started theta at zero and relied on 64 two-level draws to replace it: at the 2 ... 4 % acceptance of the
        coarse levels of config 5 a quarter of the chains were still at zero, and the level means came out too small.)"""
        ops.hmc_thermalise(self.hmc, self.x, n)
        if self.step is not None:
            fine = self.step.fine
            self.step.theta = ops.path_initialise(fine, self.B, self.hmc.seed + 2, self.chain0)
            direct = ops.PathHMC(fine, self.B, self.hmc.nt, dt_fine or 0.02, seed=self.hmc.seed + 2, chain0=self.chain0)
            ops.hmc_thermalise(direct, self.step.theta, n)
            for _ in range(16):
                self.hmc.draw(self.x, count_stats=False)
                self.step.draw(self.x)

    def sample(self):
        """One Y sample per chain (montecarlomultilevel.cc:118-146, fixed sub-sampling of the coarse chain)."""
        for _ in range(self.n_sub):
            self.hmc.draw(self.x)
        self.site_steps += self.n_sub * (self.hmc.nt + 1) * self.act_src.M * self.B
        if self.coarsest:
            y = self.qoi(self.x)
        else:
            self.step_accepted += self.step.draw(self.x)
            self.site_steps += self.step.fine.M * self.B
            y = self.qoi(self.step.theta) - self.qoi(self.x)
        ops.stats_accumulate(self.acc, y)
        self.n_draws += 1
        return y

    def sums(self):
        """The 7 additive sums of chains.level_sums for this chain block."""
        n = self.acc[:, 0]
        tot = chains.pack_moments(self.acc)
        m = self.acc[:, 1] / torch.clamp(n, min=1.0)
        return [float(tot[0]), float(tot[1]), float(tot[2]), float((n > 0).sum()), float(m.sum()), float((m * m).sum()),
                float(self.site_steps)]


class HierChain:
    """HierarchicalSampler (sampler/hierarchicalsampler.cc:55-81) for the action on level `top`, B chains: restrict the
    current sample down the levels (copy_from_fine), one HMC draw on the coarsest level, then one TwoLevelMetropolisStep
    per level up to `top`; a chain rejected on some level does not move on the finer ones (the reference's break)."""

    def __init__(self, acts, top, B, nt, dt, seed, chain0=0):
        L = len(acts)
        self.acts, self.top, self.L, self.B = acts, top, L, B
        self.state = {k: torch.zeros((B, acts[k].M), dtype=torch.float64, device="cuda") for k in range(top, L)}
        self.state[L - 1] = ops.path_initialise(acts[L - 1], B, seed, chain0)
        self.hmc = ops.PathHMC(acts[L - 1], B, nt, dt, seed=seed, chain0=chain0)
        self.steps = {k: ops.PathTwoLevelStep(acts[k], acts[k + 1], B, seed=seed + 31 * (k + 1), chain0=chain0)
                      for k in range(top, L - 1)}
        for k, st in self.steps.items():
            st.theta = self.state[k]  # the step's current state IS the sampler's state of that level (set_state without a copy)
        self.n_draws = 0
        self.n_accepted = torch.zeros(B, dtype=torch.int64, device="cuda")
        self.level_accepted = {k: torch.zeros(B, dtype=torch.int64, device="cuda") for k in range(top, L)}
        # site-steps per draw: HMC trajectory on the coarsest level + one pass per two-level step
        self.cost = (nt + 1) * acts[L - 1].M + sum(acts[k].M for k in range(top, L - 1))

    def thermalise(self, n_hmc, n_draws, dt_top=None):
        """The sampler's own level `top` by a direct HMC run from the reference's start (ops.hmc_thermalise; untimed, once),
        the coarser levels by restriction -- an equilibrium sample of the fine action, which is the state the hierarchical
        chain is supposed to be in --, then hierarchical draws."""
        if self.top == self.L - 1:
            ops.hmc_thermalise(self.hmc, self.state[self.top], n_hmc)
        else:
            self.state[self.top] = ops.path_initialise(self.acts[self.top], self.B, self.hmc.seed + 1, self.hmc.chain0)
            if self.top in self.steps:
                self.steps[self.top].theta = self.state[self.top]
            direct = ops.PathHMC(self.acts[self.top], self.B, self.hmc.nt, dt_top or 0.02, seed=self.hmc.seed + 1, chain0=self.hmc.chain0)
            ops.hmc_thermalise(direct, self.state[self.top], n_hmc)
            for k in range(self.top + 1, self.L):
                self.state[k].copy_(self.state[k - 1][:, ::2])
        for _ in range(n_draws):
            self.draw(count=False)

    def draw(self, count=True):
        for k in range(self.top + 1, self.L):  # hierarchicalsampler.cc:57-60
            self.state[k].copy_(self.state[k - 1][:, ::2])
        mask = self.hmc.draw(self.state[self.L - 1], count_stats=count)
        if count:
            self.level_accepted[self.L - 1] += mask
        for k in range(self.L - 2, self.top - 1, -1):
            mask = self.steps[k].draw(self.state[k + 1], mask=mask)
            if count:
                self.level_accepted[k] += mask
        if count:
            self.n_draws += 1
            self.n_accepted += mask
        return self.state[self.top]

    def p_accept(self):
        n = max(1, self.n_draws)
        return {k: float(v.double().mean()) / n for k, v in self.level_accepted.items()}


class HierPathLevel(PathLevel):
    """Level instance of MonteCarloMultiLevel with sampler = 'hierarchical' (montecarlomultilevel.cc:27-45,118-146,170-190):
    the coarse sampler of level l is a HierarchicalSampler on level l + 1 (HMC only on the coarsest level of the whole
    hierarchy), sub-sampled ceil(2 tau_int) draws apart, followed by the two-level step l + 1 -> l."""

    def __init__(self, acts, level, B, nt, dt_coarse, seed, chain0=0, window=20, qoi=None):
        self.level, self.B, self.chain0 = level, B, chain0
        L = len(acts)
        self.coarsest = level == L - 1
        self.qoi = qoi or ops.qoi_xsquared
        src = level if self.coarsest else level + 1
        self.act_src = acts[src]
        self.sampler = HierChain(acts, src, B, nt, dt_coarse, seed + 7919 * (level + 1), chain0)
        self.hmc = self.sampler.hmc
        self.step = None if self.coarsest else ops.PathTwoLevelStep(acts[level], acts[level + 1], B,
                                                                     seed=seed + 104729 * (level + 1), chain0=chain0)
        self.acc = torch.zeros((B, chains.N_MOMENTS), dtype=torch.float64, device="cuda")
        self.site_steps = 0
        self.n_draws = 0
        self.step_accepted = torch.zeros(B, dtype=torch.int64, device="cuda")
        self.window, self.n_sub = window, 1
        self.tau_coarse = 1.0
        # draw_coarse_sample (montecarlomultilevel.cc:170-190) re-reads ceil(2 tau_int) of the coarse sampler's QoI on every
        # coarse draw.  running = True (the default, r05): the same here -- every sampler draw is recorded in windowed
        # statistics on the device (mlmcpi_stats_window_record) and the number of draws up to the next coarse sample is
        # re-evaluated before each sample, with the autocovariances averaged over the chains of the batch (the reference
        # has one chain; a batch shares one count, so that its chains stay in step).  running = False: the count found in
        # thermalise() stays (r03 / r04 behaviour).
        self.running = True
        self.wstats = ops.stats_window_state(B, window)
        self.n_sub_sum = 0
        self._tau_ready = self._tau_queued = None

    def thermalise(self, n, dts=None):
        self.sampler.thermalise(n, 24, dt_top=None if dts is None else dts[self.sampler.top])
        # tau_int of the coarse sampler's QoI: the reference's windowed estimator (common/statistics.cc:38-61:
        # 1 + 2 sum_{k < window} (1 - k/n) C_k / C_0), with the autocovariances C_k averaged over all chains of the batch
        n_series = 8 * self.window   # length of the tau_int series; `n` stays the burn-in length the caller asked for
        qs = []
        for _ in range(n_series):
            qv = self.qoi(self.sampler.draw(count=False)).clone()
            ops.stats_window_record(self.wstats, qv)   # (the running estimate starts from this series)
            qs.append(qv)
        q = torch.stack(qs)   # [n_series, B]
        d = q - q.mean()
        c0 = float((d * d).mean())
        tau = 1.0
        if c0 > 0.0:
            for k in range(1, self.window):
                tau += 2.0 * (1.0 - k / n_series) * float((d[:-k] * d[k:]).mean()) / c0
        self.tau_coarse = max(1.0, tau)
        # ceil(2 tau_int), montecarlomultilevel.cc:173 (a stationary series cannot exceed 1 + 2 (window - 1))
        self.n_sub = min(max(1, int(-(-2.0 * self.tau_coarse // 1))), 2 * (1 + 2 * self.window))
        # experiment knob (bench.py --hier-sub-factor): more draws between coarse samples than the reference's ceil(2 tau_int)
        self.n_sub = max(1, int(round(self.n_sub * getattr(self, "sub_factor", 1.0))))
        if self.step is not None:
            # the level's own state: an equilibrium sample of its action by a direct HMC run (untimed, once)
            fine = self.step.fine
            self.step.theta = ops.path_initialise(fine, self.B, self.sampler.hmc.seed + 2, self.chain0)
            direct = ops.PathHMC(fine, self.B, self.hmc.nt, dts[self.level] if dts else 0.02,
                                 seed=self.sampler.hmc.seed + 2, chain0=self.chain0)
            self.n_burnin_theta = n   # (tests read it back)
            ops.hmc_thermalise(direct, self.step.theta, n)
            for _ in range(16):
                self.step.draw(self.sampler.draw(count=False))

    def _draws_to_next_sample(self):
        """ceil(2 tau_int) with the tau_int read behind the sample before the last one: the value travels to the host
        without stalling the stream (pinned buffer + event), so the levels of a pass keep overlapping; deterministic (the
        lag is exactly one sample; the estimate moves by O(1 / n) per draw)."""
        if not self.running:
            return self.n_sub
        ready, self._tau_ready = self._tau_ready, self._tau_queued
        if ready is not None:
            ev, buf = ready
            ev.synchronize()
            self.tau_coarse = max(1.0, float(buf[0]))
        n = max(1, int(-(-2.0 * self.tau_coarse // 1)))            # ceil(2 tau_int), montecarlomultilevel.cc:173
        n = min(n, 2 * (1 + 2 * self.window))                       # (what a stationary series can give with this window)
        return max(1, int(round(n * getattr(self, "sub_factor", 1.0))))

    def _queue_tau(self):
        buf = torch.empty(1, dtype=torch.float64, pin_memory=True)
        buf.copy_(ops.stats_window_tau_int(self.wstats, pooled=True).reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._tau_queued = (ev, buf)

    def sample(self):
        self.n_sub = self._draws_to_next_sample()
        for _ in range(self.n_sub):
            x = self.sampler.draw()
            if self.running:
                ops.stats_window_record(self.wstats, self.qoi(x))
        if self.running:
            self._queue_tau()
        self.n_sub_sum += self.n_sub
        self.site_steps += self.n_sub * self.sampler.cost * self.B
        if self.coarsest:
            y = self.qoi(x)
        else:
            self.step_accepted += self.step.draw(x)
            self.site_steps += self.step.fine.M * self.B
            y = self.qoi(self.step.theta) - self.qoi(x)
        ops.stats_accumulate(self.acc, y)
        self.n_draws += 1
        return y


def level_costs(acts, nt, n_sub):
    """site-steps per Y sample of one chain of each level: sampler trajectories of the feeding level + two-level pass"""
    L = len(acts)
    return [n_sub * (nt + 1) * acts[l if l == L - 1 else l + 1].M + (0 if l == L - 1 else acts[l].M) for l in range(L)]


class PathMLMC:
    def __init__(self, kind, M0, T_final, n_level, B, nt=20, dt0=0.05, seed=1, rank=0, world=1, n_sub=2, params=None,
                 hierarchical=False, dt_coarse=0.095):
        """hierarchical = True: sampler = 'hierarchical' of the reference (HierPathLevel); dt_coarse = the HMC step size on
        the coarsest level (0.095 is what the reference's auto-tuner finds at M_lat = 2048)"""
        p = dict(m0=1.0, mu2=1.0, lam=0.0, x0=0.0)
        p.update(params or {})
        self.acts = [abi.path_action(kind, M0 >> l, T_final, p["m0"], p["mu2"], p["lam"], p["x0"]) for l in range(n_level)]
        # stable leapfrog step ~ a^(1/2) for the kinetic term and acceptance ~ M dt^4: scale gently with the level
        self.dts = [dt0 * (2.0 ** (0.25 * l)) for l in range(n_level)]
        self.n_level, self.rank, self.world, self.B = n_level, rank, world, B
        self.concurrent_levels, self._streams = True, {}
        self.exchange = None  # object with allreduce_sum_host(list) -> list (comm.Comm); None: torch.distributed if initialised
        self.hierarchical = hierarchical
        if hierarchical:  # cost of a Y sample is dominated by the coarsest-level HMC on every level: about equal shares
            costs = [(nt + 1) * self.acts[-1].M + sum(a.M for a in self.acts[l:-1]) for l in range(n_level)]
            self.shares = chains.partition_instances(costs, B, world)
            self.levels = {l: HierPathLevel(self.acts, l, nb, nt, dt_coarse, seed, chain0=c0)
                           for l, (c0, nb) in self.shares[rank].items()}
        else:
            self.shares = chains.partition_instances(level_costs(self.acts, nt, n_sub), B, world)
            self.levels = {l: PathLevel(self.acts, l, nb, nt, self.dts, seed, chain0=c0, n_sub=n_sub)
                           for l, (c0, nb) in self.shares[rank].items()}

    def describe(self):
        return {str(l): {"chain0": lv.chain0, "chains": lv.B, "M_lat": lv.act_src.M} for l, lv in sorted(self.levels.items())}

    def state_entries(self):
        return sum(lv.B * lv.act_src.M for lv in self.levels.values())

    def thermalise(self, n, sub_factor=1.0):
        for lv in self.levels.values():
            if self.hierarchical:
                lv.sub_factor = sub_factor
                lv.thermalise(n, self.dts)
            else:
                lv.thermalise(n, self.dts[lv.level])

    def pass_(self, n_samples):
        """n_samples Y samples of every level instance this rank owns.  The instances are independent, and the coarse ones
        are too small to fill the GPU on their own (2048 sites x 512 chains): each level runs on a HIP stream of its own,
        forked from and joined to the caller's stream, so their kernels overlap (the library's work buffers are per
        stream).  Results do not depend on it."""
        if len(self.levels) < 2 or not self.concurrent_levels:
            for lv in self.levels.values():
                for _ in range(n_samples):
                    lv.sample()
            return
        main = torch.cuda.current_stream()
        if not self._streams:
            self._streams = {l: torch.cuda.Stream() for l in self.levels}
        fork = torch.cuda.Event()
        fork.record(main)
        for l, lv in self.levels.items():
            st = self._streams[l]
            st.wait_event(fork)
            with torch.cuda.stream(st):
                for _ in range(n_samples):
                    lv.sample()
            join = torch.cuda.Event()
            join.record(st)
            main.wait_event(join)

    def packed_finest(self):
        """per-chain moment rows of the finest level instance this rank owns (zeros if it owns none)"""
        if not self.levels:
            return torch.zeros((1, chains.N_MOMENTS), dtype=torch.float64, device="cuda")
        return self.levels[min(self.levels)].acc

    def table(self, device="cpu"):
        """[n_level, 5] (samples, mean, variance, tau_int, cost), identical on every rank: one all-reduce of the sums"""
        s = chains.level_sums(self.n_level, {l: lv.sums() for l, lv in self.levels.items()}, device=device)
        if self.exchange is not None and self.world > 1:  # the library communicator (RCCL through the C ABI)
            flat = self.exchange.allreduce_sum_host(s.reshape(-1).tolist())
            return chains.finish_level_sums(torch.tensor(flat, dtype=torch.float64).reshape(s.shape))
        return chains.finish_level_sums(chains.allreduce_level_table(s))

    def estimate(self, device="cpu"):
        t = self.table(device)
        q, e = chains.combine_levels(t)
        return q, e, t

    def p_accept(self):
        """{level: (HMC acceptance of its sampler, two-level acceptance or None)}"""
        return {l: (float(lv.hmc.n_accepted.double().mean()) / max(1, lv.hmc.n_total),
                    None if lv.step is None else float(lv.step_accepted.double().mean()) / max(1, lv.n_draws))
                for l, lv in self.levels.items()}
