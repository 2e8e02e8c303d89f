"""N>1 path on CPU: two gloo ranks each own a contiguous shard of the groups (no data-path
collective), regenerate their slice of the synthetic stream from the counter-based RNG,
and the concatenation equals the single-process run.  Mirrors what bench.py does per GPU;
the shard engines are the CPU oracle here (no GPU in this container) and the HIP engine in the
`-m gpu` variant."""
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


def run_shard(rank, world, mode, device=False):
    ora = oracle_engine(G_PER_RANK, R, seed=SEED, group_base=rank * G_PER_RANK)
    engines = [ora]
    if device:  # the HIP engine; on a one-GPU box both ranks share device 0 (as bench.py does when devices < N).
        # The oracle shard beside it only generates the ack stream (the generator reads its state); what the
        # test compares are the DEVICE shard's columns.
        import torch as _t
        from josefine_amd import BatchedRaft
        engines.append(BatchedRaft(G_PER_RANK, R, seed=SEED, group_base=rank * G_PER_RANK,
                                   device_id=rank % max(1, _t.cuda.device_count())))
    for e in engines:
        elect_all(e)
    sim = np.zeros((R, G_PER_RANK), dtype=np.uint64)
    for t in range(TICKS):
        acks = synth_tick_host(ora, mode, t, sim)
        for e in engines:
            e.step_dense_acks(acks)
    return engines[-1]


def _worker(rank, world, port, mode, out_dir, device=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    e = run_shard(rank, world, mode, device)
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


@pytest.mark.gpu
def test_two_rank_device_shards_equal_the_single_process_oracle(tmp_path):
    """The same, with the shard engines on the device: each rank's HIP engine owns groups
    [rank*G, (rank+1)*G) (jg_config.group_base), and their concatenation is the oracle's whole run."""
    check_two_ranks(tmp_path, 1, device=True)


@pytest.mark.parametrize("mode", [1])
def test_two_rank_shards_equal_single_process(tmp_path, mode):
    check_two_ranks(tmp_path, mode, device=False)


def check_two_ranks(tmp_path, mode, device):
    world = 2
    mp.spawn(_worker, args=(world, free_port(), mode, str(tmp_path), device), nprocs=world, join=True)
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
