"""Closed loop: all R replicas of P partitions as instances of one engine, messages routed by
the host (tests/cluster_sim.py).  Elections with real vote grants (follower.rs:97-101,219-246),
heartbeats, replicate(), follower commit advance — every output row of one round is the next
round's input, so the two implementations must agree on everything to stay in lock-step."""
import numpy as np
import pytest

from josefine_amd import BatchedRaft, capi
from cluster_sim import Cluster
from oracle_lib import oracle_engine
from parity import compare_snapshots


def make_cluster(factory, P, R, seed, **chaos):
    e = factory(P * R, R, seed=seed, self_slots=Cluster.self_slots(P, R), flags=capi.CFG_SEPARATE_COMMIT_KEY)
    return Cluster(e, P, R, **chaos)


@pytest.mark.parametrize("R", [3, 5])
def test_cluster_converges_on_oracle(R):
    """CPU sanity of the simulation itself: partitions elect leaders, replicate and commit."""
    P = 40
    c = make_cluster(oracle_engine, P, R, seed=11)
    rng = np.random.default_rng(5)
    for _ in range(80):
        c.round(rng=rng)
    e = c.e
    roles = e.read("role").reshape(P, R)
    assert ((roles == capi.ROLE_LEADER).sum(axis=1) >= 1).mean() > 0.9  # nearly every partition has a leader
    commit = e.read("commit").reshape(P, R)
    assert commit.max() > 5
    # followers follow: some partition has every replica's commit advanced
    assert (commit.min(axis=1) > 0).any()
    # faults that do occur are the reference's own (Q3/Q8 after a second election), never engine limits
    assert (e.read("fault") < 128).all()


@pytest.mark.gpu
@pytest.mark.parametrize("R,P", [(3, 300), (5, 200)])
def test_cluster_lockstep_parity(R, P):
    dev = make_cluster(BatchedRaft, P, R, seed=21)
    ora = make_cluster(oracle_engine, P, R, seed=21)
    rng_d, rng_o = np.random.default_rng(9), np.random.default_rng(9)
    for rnd in range(70):
        md, fd, xd = dev.round(rng=rng_d)
        mo, fo, xo = ora.round(rng=rng_o)
        assert md.tobytes() == mo.tobytes(), f"round {rnd}: messages differ"
        assert fd.tobytes() == fo.tobytes(), f"round {rnd}: fsm rows differ"
        assert xd.tobytes() == xo.tobytes(), f"round {rnd}: fault rows differ"
        compare_snapshots(dev.e, ora.e, f"cluster R={R} round {rnd}")
    assert (ora.e.read("role") == capi.ROLE_LEADER).sum() > 0.8 * P
    assert int(ora.e.read("commit").max()) > 5


def test_chaos_cluster_on_oracle_reaches_the_reference_failure_modes():
    """CPU sanity of the chaotic simulation: lossy network + crashing processes drive the
    cluster into the reference's panic paths (Q3 leader step-down, Q8 append after restart …)."""
    P, R = 60, 3
    c = make_cluster(oracle_engine, P, R, seed=5, chaos_seed=77)
    rng = np.random.default_rng(6)
    seen = set()
    for _ in range(150):
        _, _, faults = c.round(rng=rng)
        seen |= set(int(x) for x in faults["code"])
    assert int(c.e.read("commit").max()) > 3
    assert seen and all(code < 128 for code in seen), seen


@pytest.mark.gpu
@pytest.mark.parametrize("R,P,seed", [(3, 300, 1), (5, 200, 2), (3, 300, 3)])
def test_chaos_cluster_lockstep_parity(R, P, seed):
    """Closed loop under message loss, duplication and process restarts: every round's output
    rows and every state column must agree, or the two clusters drift apart."""
    dev = make_cluster(BatchedRaft, P, R, seed=40 + seed, chaos_seed=1000 + seed)
    ora = make_cluster(oracle_engine, P, R, seed=40 + seed, chaos_seed=1000 + seed)
    rng_d, rng_o = np.random.default_rng(seed), np.random.default_rng(seed)
    seen = set()
    for rnd in range(120):
        md, fd, xd = dev.round(rng=rng_d)
        mo, fo, xo = ora.round(rng=rng_o)
        assert md.tobytes() == mo.tobytes(), f"round {rnd}: messages differ"
        assert fd.tobytes() == fo.tobytes(), f"round {rnd}: fsm rows differ"
        assert xd.tobytes() == xo.tobytes(), f"round {rnd}: fault rows differ"
        compare_snapshots(dev.e, ora.e, f"chaos cluster R={R} round {rnd}")
        seen |= set(int(x) for x in xo["code"])
    assert int(ora.e.read("commit").max()) > 3
    assert seen, "the chaotic run should have hit at least one of the reference's panic paths"
