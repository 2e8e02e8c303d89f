"""Per-partition leadership inside the device-resident cluster (jg_dense_cluster_create with
JG_CLUSTER_ANY_LEADER; candidate.rs:101-113,216-238 -> leader.rs:124-174,234-245): every node leads the
partitions it was elected for and follows the others, over mailbox columns that belong to the CLUSTER.

  1. (CPU) the numpy statement of the round (tests/dense_node.py::AnyLeaderCluster, over the per-node dense
     entry points) behaves like a cluster: leaders on every node keep committing, two leaders of one group in
     one round are survived, elections that the transport carries are won and the winner replicates in
     column form; the same statement over the oracle and over tests/ref_py agrees column by column.
  2. (GPU) the library's cluster - claim, all leader halves, all follower halves, the device transport - is
     bit-identical to that statement over the oracle: every state column of every node after every round,
     the rows delivered per node and round, the rows left for the host; eager routed rounds and rounds
     replayed as a hipGraph.
"""
import numpy as np
import pytest

from josefine_amd import BatchedRaft, capi
from dense_node import AnyLeaderCluster, any_failure_rows
from node_step import elect_some
from oracle_lib import oracle_engine
from parity import compare_snapshots


def spread_leaders(nodes, G, R, dual_every=0):
    """Node g % R is elected leader of group g (synthetic votes); with dual_every: every such group ALSO on the next node."""
    g = np.arange(G)
    for n, e in enumerate(nodes):
        mask = g % R == n
        if dual_every:
            mask |= (g % dual_every == 0) & ((g + 1) % R == n)
        elect_some(e, mask)
        e.drain_messages(), e.drain_applies(), e.drain_faults()


def test_any_leader_closed_loop_on_the_oracle():
    G, R, T = 240, 3, 30
    cl = AnyLeaderCluster(oracle_engine, G, R, seed=5)
    spread_leaders(cl.nodes, G, R)
    for t in range(T):
        cl.round(np.ones(G, np.uint64))
    g = np.arange(G)
    for n, e in enumerate(cl.nodes):
        mine = g % R == n
        assert (e.read("role")[mine] == capi.ROLE_LEADER).all() and (e.read("role")[~mine] == capi.ROLE_FOLLOWER).all()
        assert (e.read("head")[mine] == T).all() and (e.read("commit")[mine] >= T - 3).all()
        assert (e.read("head")[~mine] >= T - 1).all() and (e.read("commit")[~mine] >= T - 5).all()
        assert not e.read("fault").any()
    assert sum(len(k) for k in cl.kept) == 0 and cl.delivered.sum() == 0  # nothing ever left the mailbox vocabulary


def run_trace(cl, G, R, T, percent, lib=None, nodes=None, seed=99, recreate=False):
    """A failure trace over a cluster (and, in lockstep, the library's cluster over `nodes`): groups fail once at most;
    with R = 3 the whole group restarts, so that the next replica's campaign is won through the transport.  recreate: the
    replicas come back on EMPTY stores (JG_CMD_RECREATE) and a group may fail any number of times."""
    leader_of = np.arange(G) % R
    failed = np.zeros(G, bool)
    for t in range(T):
        inj, failing = any_failure_rows(seed, t, G, R, percent, leader_of, whole_group=(R == 3), skip=None if recreate else failed,
                                        recreate=recreate) if t >= 3 else ([None] * R, [])
        failed[failing] = True
        cl.round(np.ones(G, np.uint64), inject=inj)
        if lib is not None:
            up = [None if c is None else nodes[n].upload_rows(**c) for n, c in enumerate(inj)]
            st = lib.round_routed((t + 1) * 100, up)
            for n in range(R):
                compare_snapshots(nodes[n], cl.nodes[n], f"round {t} node {n}")
            assert st["delivered"] == [cl.pending(n) for n in range(R)], t
            for rows in up:
                if rows is not None:
                    rows.free()
    return failed


def test_any_leader_elections_are_won_through_the_transport_on_the_oracle():
    """R = 3, whole groups restart: the designated candidate's VoteRequests travel as rows, the first grant is the quorum
    (candidate.rs:101-113), and from the next round on the WINNER's Tick is in the columns: leadership has moved to
    another node and stays dense (what the reference makes of such a leader - Q8: it cannot append - is the trace's)."""
    G, R, T = 300, 3, 40
    cl = AnyLeaderCluster(oracle_engine, G, R, seed=5)
    spread_leaders(cl.nodes, G, R, dual_every=11)
    failed = run_trace(cl, G, R, T, 3)
    assert failed.sum() > G // 4
    g = np.arange(G)
    won = 0
    for n, e in enumerate(cl.nodes):
        role = e.read("role")
        won += int((role[failed & ((g + 1) % R == n)] == capi.ROLE_LEADER).sum())
    assert won > failed.sum() // 2  # the campaigns were won, on every node
    assert cl.delivered.sum() > 0


def test_any_leader_recreated_groups_elect_and_append_again_on_the_oracle():
    """R = 3, whole groups come back on EMPTY stores (JG_CMD_RECREATE), again and again: every campaign is won through the
    transport - no synthetic vote anywhere - and the winner APPENDS (Q8 does not apply to a chain that starts over): the
    trace is stationary.  Oracle and the independent Python reading agree on every column of every node after every round."""
    G, R, T = 240, 3, 60
    a, b = AnyLeaderCluster(oracle_engine, G, R, seed=5), AnyLeaderCluster(ref_py_engine, G, R, seed=5)
    for cl in (a, b):
        spread_leaders(cl.nodes, G, R, dual_every=0)
    leader_of = np.arange(G) % R
    failed = np.zeros(G, bool)
    last = np.zeros(G, np.int64)
    for t in range(T):
        inj, failing = any_failure_rows(99, t, G, R, 4, leader_of, whole_group=True, recreate=True) if t >= 3 else ([None] * R, [])
        failed[failing] = True
        last[failing] = t
        for cl in (a, b):
            cl.round(np.ones(G, np.uint64), inject=inj)
        for n in range(R):
            compare_snapshots(b.nodes[n], a.nodes[n], f"round {t} node {n}")
    assert failed.sum() > G // 2 and (np.bincount(last[failed]) > 0).sum() > 10
    roles = np.stack([e.read("role") for e in a.nodes])
    heads = np.stack([e.read("head") for e in a.nodes])
    settled = failed & (last < T - 6)  # (an election takes three rounds)
    lead = (leader_of + 1) % R
    g = np.arange(G)
    # the designated candidate won (a group that fails again while its election is in flight can leave it Defeated by votes
    # that were meant for its previous campaign - it campaigns again at its next timeout: a few of a hundred) ...
    won = settled & (roles[lead, g] == capi.ROLE_LEADER)
    assert won.sum() > 0.9 * settled.sum() and settled.sum() > G // 3
    assert (heads[lead[won], g[won]] >= T - last[won] - 5).all()       # ... and has appended a block per round since
    assert (heads[lead[won], g[won]] <= T - last[won]).all()
    for e in a.nodes:
        assert not e.read("fault").any()
    assert a.kept[0].tobytes() == b.kept[0].tobytes() and sum(len(k) for k in a.kept) == 0


def ref_py_engine(*a, **kw):
    from ref_py.engine import RefEngine
    return RefEngine(*a, **kw)


@pytest.mark.parametrize("R", [3, 5])
def test_any_leader_cluster_oracle_vs_ref_py(R):
    """The round's statement over the C++ oracle and over the independent Python reading of the Rust: every column of
    every node after every round, with two leaders in some groups and leaders failing."""
    G, T = 60, 30
    a, b = AnyLeaderCluster(oracle_engine, G, R, seed=7), AnyLeaderCluster(ref_py_engine, G, R, seed=7)
    for cl in (a, b):
        spread_leaders(cl.nodes, G, R, dual_every=7)
    leader_of = np.arange(G) % R
    failed = np.zeros(G, bool)
    for t in range(T):
        inj, failing = any_failure_rows(3, t, G, R, 4, leader_of, whole_group=(R == 3), skip=failed) if t >= 3 else ([None] * R, [])
        failed[failing] = True
        for cl in (a, b):
            cl.round(np.ones(G, np.uint64), inject=inj)
        for n in range(R):
            compare_snapshots(b.nodes[n], a.nodes[n], f"round {t} node {n}")
    for n in range(R):
        assert a.kept[n].tobytes() == b.kept[n].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("R,percent,dual", [(3, 3, 0), (3, 3, 9), (5, 2, 13), (2, 0, 5), (3, -4, 0)])
def test_any_leader_cluster_device_parity(R, percent, dual):
    """(percent < 0: that many percent per round, the groups RE-CREATED - JG_CMD_RECREATE - and failing any number of times)"""
    from josefine_amd import DenseCluster as LibCluster
    G, T = 3000, 45
    recreate, percent = percent < 0, abs(percent)
    ora = AnyLeaderCluster(oracle_engine, G, R, seed=5)
    nodes = [BatchedRaft(G, R, seed=5 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY) for r in range(R)]
    spread_leaders(ora.nodes, G, R, dual_every=dual)
    spread_leaders(nodes, G, R, dual_every=dual)
    lib = LibCluster(nodes, lead=None)
    lib.set_appends(1)
    run_trace(ora, G, R, T, percent, lib=lib, nodes=nodes, recreate=recreate)
    assert sum(ora.delivered) > 0 or percent == 0
    for n in range(R):
        got, want = nodes[n].drain_messages(), ora.kept[n]
        assert got.tobytes() == want.tobytes(), (n, len(got), len(want))
        assert nodes[n].drain_faults().tobytes() == ora.nodes[n].drain_faults().tobytes()
        assert nodes[n].drain_applies().tobytes() == ora.nodes[n].drain_applies().tobytes()
    lib.close()


@pytest.mark.gpu
@pytest.mark.parametrize("R", [3, 5])
def test_any_leader_rounds_replayed_as_a_graph(R):
    """jg_dense_cluster_rounds with per-partition leadership (five launches per round, replayed as one hipGraph) == the
    statement's rounds: state, and every row the rounds queued (nothing is routed here)."""
    from josefine_amd import DenseCluster as LibCluster
    G, T = 2000, 24
    ora = AnyLeaderCluster(oracle_engine, G, R, seed=9)
    nodes = [BatchedRaft(G, R, seed=9 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY) for r in range(R)]
    spread_leaders(ora.nodes, G, R, dual_every=17)
    spread_leaders(nodes, G, R, dual_every=17)
    lib = LibCluster(nodes, lead=None)
    appends = (np.arange(G) % 3).astype(np.uint64)
    lib.set_appends(per_group=appends)
    lib.rounds(100, 100, 1)  # one eager round, then the replayed ones
    lib.rounds(200, 100, T - 1)
    for t in range(T):
        ora.dense_round(appends, 100)
    for n in range(R):
        compare_snapshots(nodes[n], ora.nodes[n], f"node {n}")
        want = np.concatenate([np.concatenate(rows[n]) for rows in ora.rows])  # (a node's two halves of a round: two steps)
        got = nodes[n].drain_messages()
        assert got.tobytes() == want.tobytes(), (n, len(got), len(want))
        assert nodes[n].drain_faults().tobytes() == ora.nodes[n].drain_faults().tobytes()
    lib.close()
