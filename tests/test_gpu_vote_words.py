"""JG_CLUSTER_OPT_VOTE_WORDS (jg_dense_cluster_set_option, per cluster): the routed round with the election
vocabulary as mailbox words (josefine_amd/csrc/jg_votes.h) against the oracle clusters that move every message as a row.
The switch changes what TRAVELS, not what the nodes compute: every state column of every node after every round, the rows
left for the host, the faults and the applies are those of the row transport; the number of rows the transport moved is
smaller (that is the point, and how the test knows the switch was on).

The option is per cluster, so these run in the default `pytest -m gpu` beside the row transport's tests (which count
delivered rows against the row transport and create their clusters without it).  The per-partition logic behind the
option is also held to the oracle on the CPU (tests/test_vote_half.py, tests/test_vote_mail.py)."""
import os

import numpy as np
import pytest

from josefine_amd import BatchedRaft, capi
from oracle_lib import oracle_engine
from parity import compare_snapshots, elect_all

EMULATED = os.environ.get("JG_EMULATED_DEVICE") == "1"
pytestmark = pytest.mark.gpu


CASES = [(3, 3, (), 3000, 50), (5, 2, (), 3000, 50), (5, 2, (2,), 3000, 50), (3, 3, (2,), 3000, 50), (3, 25, (1, 2), 2000, 60), (5, 25, (1, 2, 3), 2000, 60),
         (4, 30, (1, 2), 2000, 60), (5, 1, (), 300000, 30),
         # (small enough for the emulated device of tests/test_host_device.py, which runs them on the CPU)
         (3, 4, (2,), 200, 40), (5, 3, (2,), 200, 30), (3, 25, (1, 2), 200, 40), (5, 25, (1, 2, 3), 160, 22)]


@pytest.mark.parametrize("R,percent,also,G,T", CASES, ids=[("small-" if c[3] <= 200 else "") + f"{c[0]}-{c[1]}-{len(c[2])}-{c[3]}" for c in CASES])
def test_routed_cluster_with_the_vote_mail(R, percent, also, G, T):
    from josefine_amd import DenseCluster as LibCluster
    from dense_node import RoutedCluster, cluster_failure_rows
    ora = RoutedCluster(oracle_engine, G, R, seed=5)
    nodes = [BatchedRaft(G, R, seed=5 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY) for r in range(R)]
    elect_all(nodes[0])
    nodes[0].drain_messages(), nodes[0].drain_applies()
    lib = LibCluster(nodes, vote_words=True)
    lib.set_appends(1)
    moved = moved_as_rows = 0
    for t in range(T):
        inj = cluster_failure_rows(99, t, G, R, percent, also=also) if t >= 3 else [None] * R
        if t == 20 and G <= 3000 and R >= 3:  # something the transport must leave alone: a client request at every replica of node 2
            inj[2] = dict(kind=np.full(G, capi.CMD_CLIENT_REQUEST, np.uint8), group=np.arange(G, dtype=np.uint32), id=np.arange(G, dtype=np.uint64) + 1000)
        up = [None if c is None else nodes[n].upload_rows(**c) for n, c in enumerate(inj)]
        st = lib.round_routed((t + 1) * 100, up)
        ora.round(np.ones(G, np.uint64), inject=inj)
        if G <= 3000 or t % 5 == 4 or t == T - 1:
            for n in range(R):
                compare_snapshots(nodes[n], ora.nodes[n], f"routed round {t} node {n}")
        want = [ora.pending(n) for n in range(R)]
        assert all(a <= b for a, b in zip(st["delivered"], want)), (t, st["delivered"], want)
        moved_as_rows += sum(st["delivered"])
        moved += sum(want)
        for rows in up:
            if rows is not None:
                rows.free()
    assert moved > 0 and moved_as_rows < moved // 2, (moved_as_rows, moved)  # (most of the mail is an election's: it went as words)
    for n in range(R):
        got, want = nodes[n].drain_messages(), ora.kept[n]
        assert got.tobytes() == want.tobytes(), (n, len(got), len(want))
        assert nodes[n].drain_faults().tobytes() == ora.nodes[n].drain_faults().tobytes()
        assert nodes[n].drain_applies().tobytes() == ora.nodes[n].drain_applies().tobytes()
        if not EMULATED:  # (tests/host_device.py: no reconvergence there - a wave reduction behind a branch may lose lanes)
            assert nodes[n].counters()["decisions"] == ora.nodes[n].counters()["decisions"]
    lib.close()


ANY_CASES = [(3, 3, 0, 3000, 45), (3, 3, 9, 3000, 45), (5, 2, 13, 3000, 45), (3, 4, 9, 200, 36), (5, 3, 13, 200, 30), (3, 25, 7, 160, 24)]


@pytest.mark.parametrize("R,percent,dual,G,T", ANY_CASES, ids=[("small-" if c[3] <= 200 else "") + f"{c[0]}-{c[1]}-{c[2]}-{c[3]}" for c in ANY_CASES])
def test_any_leader_cluster_with_the_vote_mail(R, percent, dual, G, T):
    """per-partition leadership (tests/test_any_leader.py::test_any_leader_cluster_device_parity under the switch): whole groups
    restart, the next replica's campaign is won through the mail, leadership moves and stays in the columns"""
    from josefine_amd import DenseCluster as LibCluster
    from dense_node import AnyLeaderCluster, any_failure_rows
    from test_any_leader import spread_leaders
    ora = AnyLeaderCluster(oracle_engine, G, R, seed=5)
    nodes = [BatchedRaft(G, R, seed=5 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY) for r in range(R)]
    spread_leaders(ora.nodes, G, R, dual_every=dual)
    spread_leaders(nodes, G, R, dual_every=dual)
    lib = LibCluster(nodes, lead=None, vote_words=True)
    lib.set_appends(1)
    leader_of = np.arange(G) % R
    failed = np.zeros(G, bool)
    moved = moved_as_rows = 0
    for t in range(T):
        inj, failing = any_failure_rows(99, t, G, R, percent, leader_of, whole_group=(R == 3), skip=failed) if t >= 3 else ([None] * R, [])
        failed[failing] = True
        ora.round(np.ones(G, np.uint64), inject=inj)
        up = [None if c is None else nodes[n].upload_rows(**c) for n, c in enumerate(inj)]
        st = lib.round_routed((t + 1) * 100, up)
        for n in range(R):
            compare_snapshots(nodes[n], ora.nodes[n], f"round {t} node {n}")
        want = [ora.pending(n) for n in range(R)]
        assert all(a <= b for a, b in zip(st["delivered"], want)), (t, st["delivered"], want)
        moved_as_rows += sum(st["delivered"])
        moved += sum(want)
        for rows in up:
            if rows is not None:
                rows.free()
    assert moved > 0 and moved_as_rows < moved, (moved_as_rows, moved)
    for n in range(R):
        got, want = nodes[n].drain_messages(), ora.kept[n]
        assert got.tobytes() == want.tobytes(), (n, len(got), len(want))
        assert nodes[n].drain_faults().tobytes() == ora.nodes[n].drain_faults().tobytes()
        assert nodes[n].drain_applies().tobytes() == ora.nodes[n].drain_applies().tobytes()
        if not EMULATED:  # (tests/host_device.py: no reconvergence there - a wave reduction behind a branch may lose lanes)
            assert nodes[n].counters()["decisions"] == ora.nodes[n].counters()["decisions"]
    lib.close()
