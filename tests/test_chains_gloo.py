"""CPU, world_size 2 over gloo: chain sharding and the packed statistics all-reduce (the N>1 path)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_chains, q):
    sys.path.insert(0, ROOT)
    from mlmcpathintegral_amd import chains
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = chains.chain_block(n_chains, rank, world)
    # per-chain samples are a deterministic function of the GLOBAL chain index
    acc = torch.zeros((count, chains.N_MOMENTS), dtype=torch.float64)
    for b in range(count):
        g = first + b
        for k in range(10):
            v = float((g * 31 + k * 7) % 13) / 13.0
            acc[b] += torch.tensor([1.0, v, v * v, v ** 3, v ** 4], dtype=torch.float64)
    packed = chains.allreduce_moments(chains.pack_moments(acc))
    q.put((rank, first, count, packed.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_chains", [8, 7])
def test_sharded_statistics_equal_serial(n_chains):
    from mlmcpathintegral_amd import chains
    world, port = 2, 29517 + n_chains
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_chains, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # blocks tile [0, n_chains) without gaps or overlap
    assert res[0][1] == 0 and res[0][1] + res[0][2] == res[1][1] and res[1][1] + res[1][2] == n_chains
    serial = torch.zeros(chains.N_MOMENTS, dtype=torch.float64)
    for g in range(n_chains):
        for k in range(10):
            v = float((g * 31 + k * 7) % 13) / 13.0
            serial += torch.tensor([1.0, v, v * v, v ** 3, v ** 4], dtype=torch.float64)
    for r in res:
        assert torch.allclose(torch.tensor(r[3], dtype=torch.float64), serial, rtol=0, atol=1e-12)
    s = chains.summarise(serial)
    assert s["samples"] == 10 * n_chains


def test_distribute_n_matches_reference_rule():
    from mlmcpathintegral_amd import chains
    for n in (0, 1, 7, 8, 100):
        for world in (1, 2, 3, 8):
            parts = [chains.distribute_n(n, r, world) for r in range(world)]
            assert sum(parts) == n and max(parts) - min(parts) <= 1 and parts == sorted(parts, reverse=True)


def _level_worker(rank, world, port, n_level, q):
    sys.path.insert(0, ROOT)
    from mlmcpathintegral_amd import chains
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = {l: _level_row(l) for l in range(n_level) if l % world == rank}  # any disjoint cover of the levels will do
    table = chains.allreduce_level_table(chains.level_table(n_level, rows))
    targets, sufficient = chains.level_targets(table, 0.05)
    q.put((rank, sorted(rows), table.tolist(), targets.tolist(), sufficient, chains.combine_levels(table)))
    dist.barrier()
    dist.destroy_process_group()


def _level_row(level):
    # (samples, mean, variance, tau_int, cost): variance decays, cost grows towards the fine levels (level 0 = finest)
    return (1000.0 + 100 * level, 0.5 / (level + 1), 0.01 * 4.0 ** level, 1.0 + 0.5 * level, 64.0 / 2.0 ** level)


def test_level_sharded_table_exchange():
    """disjoint level rows per rank; one all-reduce of the [n_level, 5] table per pass."""
    from mlmcpathintegral_amd import chains
    world, port, n_level = 2, 29555, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_level_worker, args=(r, world, port, n_level, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    serial = chains.level_table(n_level, {l: _level_row(l) for l in range(n_level)})
    for r in res:
        assert torch.equal(torch.tensor(r[2], dtype=torch.float64), serial)
    t_serial, suff_serial = chains.level_targets(serial, 0.05)
    assert res[0][3] == res[1][3] == t_serial.tolist() and res[0][4] == res[1][4] == suff_serial
    q_serial, e_serial = chains.combine_levels(serial)
    assert res[0][5] == res[1][5] == (q_serial, e_serial)
    assert abs(q_serial - sum(0.5 / (l + 1) for l in range(n_level))) < 1e-15


# ---- chains sharded by global chain index: the union of the ranks' streams is the single-rank batch ----------------------
def _philox_worker(rank, world, port, B, q):
    """Each rank draws the initial states and one device-order sweep of ITS chains [rank B, (rank + 1) B) the way bench.py
    addresses them (chain0 = rank * B), with the oracle standing in for the kernels (same Philox counter contract)."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A = oracle.Action(oracle.SCHWINGER, Mt=8, Mx=8, beta=1.0)
    chain0 = rank * B
    rows = []
    for b in range(B):
        x = A.dev_initialise(11, chain0 + b)
        A.dev_sweep(x, True, 11, chain0 + b, 0)   # a heat-bath sweep: every link consumes its own Philox words
        rows.append(torch.from_numpy(np.asarray(x)))
    mine = torch.stack(rows)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if rank == 0:
        q.put(torch.cat(gathered).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_rank_chain_offsets_reproduce_the_single_rank_batch():
    """mpi/mpi_random.cc:5-29 gives every rank its own seed; here rank r owns the global chains [r B, (r + 1) B) and the
    chain index is a Philox counter word, so two ranks with B chains each must produce, bit for bit, the 2 B chains one
    rank would (first Philox words of every chain included: the initial state is one uniform per entry)."""
    import numpy as np
    import oracle
    world, port, B = 2, 29533, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_philox_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    A = oracle.Action(oracle.SCHWINGER, Mt=8, Mx=8, beta=1.0)
    for g in range(world * B):
        x = A.dev_initialise(11, g)
        A.dev_sweep(x, True, 11, g, 0)
        assert np.array_equal(got[g], np.asarray(x)), f"global chain {g}"
    assert len({tuple(row[:4]) for row in got}) == world * B   # and the chains differ from each other


# ---- (level, chain) instances in equal-cost shares, 8 ranks -----------------------------------------------------------------
def _instance_worker(rank, world, port, n_level, B, q):
    sys.path.insert(0, ROOT)
    from mlmcpathintegral_amd import chains
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    costs = [64.0 / 2.0 ** l for l in range(n_level)]
    mine = chains.partition_instances(costs, B, world)[rank]
    rows = {l: _instance_sums(l, c0, nb, costs[l]) for l, (c0, nb) in mine.items()}
    table = chains.finish_level_sums(chains.allreduce_level_table(chains.level_sums(n_level, rows)))
    q.put((rank, {l: v for l, v in mine.items()}, table.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _instance_sums(level, chain0, count, cost, n_samples=6):
    """additive sums of chains [chain0, chain0 + count) of a level; sample k of global chain g is a fixed function of (l, g, k)"""
    n = s1 = s2 = nc = m1 = m2 = 0.0
    for g in range(chain0, chain0 + count):
        ys = [((g * 17 + k * 5 + level * 3) % 11) / 11.0 / (level + 1) for k in range(n_samples)]
        n += len(ys); s1 += sum(ys); s2 += sum(y * y for y in ys)
        m = sum(ys) / len(ys)
        nc += 1; m1 += m; m2 += m * m
    return [n, s1, s2, nc, m1, m2, cost * n]


def test_level_instances_shard_over_eight_ranks():
    """BASELINE configs[4] on 8 GPUs: more ranks than levels.  The (level, chain) pairs are cut into 8 equal-cost
    shares; every rank owns something, the shares tile every level, and the reduced table equals the serial one."""
    from mlmcpathintegral_amd import chains
    world, port, n_level, B = 8, 29541, 5, 64
    costs = [64.0 / 2.0 ** l for l in range(n_level)]
    shares = chains.partition_instances(costs, B, world)
    loads = [sum(costs[l] * nb for l, (c0, nb) in s.items()) for s in shares]
    assert min(loads) > 0 and max(loads) / (sum(loads) / world) < 1.05
    for l in range(n_level):
        blocks = sorted(s[l] for s in shares if l in s)
        assert blocks[0][0] == 0 and sum(nb for _, nb in blocks) == B
        assert all(blocks[i][0] + blocks[i][1] == blocks[i + 1][0] for i in range(len(blocks) - 1))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_instance_worker, args=(r, world, port, n_level, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    serial = chains.finish_level_sums(chains.level_sums(n_level, {l: _instance_sums(l, 0, B, costs[l]) for l in range(n_level)}))
    for r in res:
        assert r[1] == shares[r[0]]
        assert torch.allclose(torch.tensor(r[2], dtype=torch.float64), serial, rtol=1e-13, atol=0)
    assert chains.combine_levels(serial)[0] == pytest.approx(float(serial[:, 1].sum()))


def test_hier_level_burns_its_own_state_in_for_the_length_asked(monkeypatch):
    """HierPathLevel.thermalise reassigned its argument to the tau_int series length (160), so the level's own
    state theta got 160 burn-in trajectories instead of the 400 the caller asked for.  Host logic only: the device calls
    are replaced by recorders."""
    import types
    from mlmcpathintegral_amd import mlmc
    calls = {"thermalise": [], "dt": []}

    class FakeSampler:
        top = 1
        hmc = types.SimpleNamespace(seed=5, nt=100)

        def thermalise(self, n_hmc, n_draws, dt_top=None):
            calls["sampler"] = (n_hmc, n_draws, dt_top)

        def draw(self, count=True):
            return torch.arange(4, dtype=torch.float64)

    class FakeStep:
        fine = object()
        theta = None

        def draw(self, x):
            return torch.zeros(4)

    monkeypatch.setattr(mlmc.ops, "path_initialise", lambda *a, **k: torch.zeros(4, dtype=torch.float64))
    monkeypatch.setattr(mlmc.ops, "PathHMC", lambda act, B, nt, dt, **k: calls["dt"].append(dt) or "hmc")
    monkeypatch.setattr(mlmc.ops, "hmc_thermalise", lambda hmc, x, n: calls["thermalise"].append(n))
    lv = object.__new__(mlmc.HierPathLevel)
    lv.level, lv.B, lv.chain0, lv.window, lv.coarsest = 0, 4, 0, 20, False
    lv.sampler, lv.hmc, lv.step = FakeSampler(), FakeSampler.hmc, FakeStep()
    lv.qoi = lambda x: x
    # r05: the series of the first tau_int estimate also feeds the windowed statistics of the running estimate
    lv.wstats = torch.zeros((4, 2 * 20 + 3), dtype=torch.float64)
    recorded = []
    monkeypatch.setattr(mlmc.ops, "stats_window_record", lambda state, q: recorded.append(q.clone()))
    lv.thermalise(400, [0.02, 0.03])
    assert len(recorded) == 8 * 20
    assert calls["sampler"] == (400, 24, 0.03)
    assert calls["thermalise"] == [400] and lv.n_burnin_theta == 400
    assert calls["dt"] == [0.02]
    assert lv.n_sub >= 1
