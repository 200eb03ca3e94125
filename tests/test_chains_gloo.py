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
