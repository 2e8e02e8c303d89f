"""N>1 path on CPU: two gloo ranks each own a contiguous shard of the groups (no data-path
collective), regenerate their slice of the synthetic stream from the counter-based RNG,
and the concatenation equals the single-process run.  Mirrors what bench.py does per GPU;
the shard engines here are the CPU oracle (no GPU in this container)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle_lib import oracle_engine
from parity import elect_all, synth_tick_host

G_PER_RANK, R, TICKS, SEED = 600, 3, 12, 0x6A6F736566696E65


def run_shard(rank, world, mode):
    e = oracle_engine(G_PER_RANK, R, seed=SEED, group_base=rank * G_PER_RANK)
    elect_all(e)
    sim = np.zeros((R, G_PER_RANK), dtype=np.uint64)
    for t in range(TICKS):
        e.step_dense_acks(synth_tick_host(e, mode, t, sim))
    return e


def _worker(rank, world, port, mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    e = run_shard(rank, world, mode)
    dist.barrier()
    # whole-job aggregates exactly as bench.py forms them: SUM of decisions, MAX of time
    dec = torch.tensor([float(e.counters()["decisions"])], dtype=torch.float64)
    dist.all_reduce(dec, op=dist.ReduceOp.SUM)
    tmax = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    np.savez(os.path.join(out_dir, f"shard{rank}.npz"), commit=e.read("commit"), head=e.read("head"),
             timeout=e.read("election_timeout"), match1=e.read("match", 1), dec=dec.numpy(), tmax=tmax.numpy())
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", [1])
def test_two_rank_shards_equal_single_process(tmp_path, mode):
    world = 2
    mp.spawn(_worker, args=(world, free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    shards = [np.load(tmp_path / f"shard{r}.npz") for r in range(world)]
    # single process over the union of the groups
    whole = oracle_engine(G_PER_RANK * world, R, seed=SEED, group_base=0)
    elect_all(whole)
    sim = np.zeros((R, G_PER_RANK * world), dtype=np.uint64)
    for t in range(TICKS):
        whole.step_dense_acks(synth_tick_host(whole, mode, t, sim))
    for name, col in (("commit", whole.read("commit")), ("head", whole.read("head")),
                      ("timeout", whole.read("election_timeout")), ("match1", whole.read("match", 1))):
        assert np.array_equal(np.concatenate([s[name] for s in shards]), col), name
    assert shards[0]["dec"][0] == whole.counters()["decisions"] == shards[1]["dec"][0]
    assert shards[0]["tmax"][0] == world  # MAX over ranks
