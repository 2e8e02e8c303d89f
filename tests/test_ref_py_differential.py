"""Differential test of the two CPU restatements of the reference: the C++ oracle
(oracle/raft_oracle.hpp — what the HIP engine is checked against) and tests/ref_py (a line-by-line
Python transliteration written from the Rust sources only).  They share no code and were read
from the reference independently; a common-mode misreading of progress.rs / election.rs /
leader.rs / follower.rs / candidate.rs / chain.rs would have to be made twice, identically, to
pass here.  More than 10^6 random commands over R = 1..8, every state column and every drained
row compared, zero disagreements allowed."""
import numpy as np
import pytest

from josefine_amd import capi
from dense_node import DenseCluster, random_follower_inbox, random_leader_inbox
from failures import failure_rows
from fuzz import assert_live, random_batch, random_batch_aware
from oracle_lib import oracle_engine
from parity import compare_drains, compare_snapshots, elect_all
from ref_py.engine import RefEngine

TOTAL = {"n": 0}


def pair(G, R, **kw):
    return RefEngine(G, R, **kw), oracle_engine(G, R, **kw)


@pytest.mark.parametrize("R", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_command_streams(R):
    """The sparse path: every role, every Command kind, forks / gaps / re-sent blocks, forged and
    foreign senders, restarts, the reference's panic paths — 128 k commands per R."""
    G, steps, rows = 192, 160, 800
    rng = np.random.default_rng(7000 + R)
    slots = rng.integers(0, R, G).astype(np.uint8)
    flags = capi.CFG_SEPARATE_COMMIT_KEY if R % 2 == 0 else 0
    ref, ora = pair(G, R, seed=R, self_slots=slots, flags=flags, election_timeout_ms=(300, 700))
    compare_snapshots(ref, ora, "RaftHandle::new")
    now = 0
    for s in range(steps):
        b = random_batch(rng, ora, rows, foreign_voters=True)
        now += int(rng.integers(0, 400))
        for e in (ref, ora):
            e.submit_columns(**b)
            e.step(now)
        TOTAL["n"] += rows
        compare_drains(ref, ora, f"R={R} step {s}")
        if s % 8 == 7 or s == steps - 1:
            compare_snapshots(ref, ora, f"R={R} step {s}")
    assert ref.counters()["decisions"] == ora.counters()["decisions"] > 0


@pytest.mark.parametrize("R", [1, 2, 3, 4, 5, 6, 7, 8])
def test_state_aware_command_streams(R):
    """The same comparison under the LIVE stream (tests/fuzz.py random_batch_aware: the command kind follows
    the group's role; faulted groups are restarted): most commands reach groups that are alive, a large share
    of the groups is led, commands turn into quorum decisions - asserted - so that the leader / majority paths
    of the two restatements are compared where the blind stream rarely gets (390 decisions per
    128 k blind commands at R = 5)."""
    G, steps, rows = 192, 100, 800
    rng = np.random.default_rng(9000 + R)
    slots = rng.integers(0, R, G).astype(np.uint8)
    flags = capi.CFG_SEPARATE_COMMIT_KEY if R % 2 == 1 else 0
    ref, ora = pair(G, R, seed=R, self_slots=slots, flags=flags, election_timeout_ms=(300, 700))
    stats = {}
    now = 0
    for s in range(steps):
        b = random_batch_aware(rng, ora, rows, stats)
        now += int(rng.integers(0, 200))
        for e in (ref, ora):
            e.submit_columns(**b)
            e.step(now)
        TOTAL["n"] += rows
        compare_drains(ref, ora, f"R={R} step {s}")
        if s % 8 == 7 or s == steps - 1:
            compare_snapshots(ref, ora, f"R={R} step {s}")
    assert ref.counters()["decisions"] == ora.counters()["decisions"]
    assert_live(stats, ora.counters()["decisions"], R)


@pytest.mark.parametrize("R", [1, 2, 3, 4, 5, 6, 7, 8])
def test_dense_ticks_elections_and_failures(R):
    """Leaders under dense ack blocks (drops, duplicates, acks above the head, bursts), with
    crashes + re-elections and random traffic mixed in."""
    G, ticks = 256, 60
    rng = np.random.default_rng(8000 + R)
    slots = rng.integers(0, R, G).astype(np.uint8)
    ref, ora = pair(G, R, seed=40 + R, self_slots=slots)
    for e in (ref, ora):
        elect_all(e)
    compare_snapshots(ref, ora, "elected")
    compare_drains(ref, ora, "elected")
    gi = np.arange(G)
    NO = np.uint64(capi.NO_ACK)
    for t in range(ticks):
        head = ora.read("head").astype(np.uint64)
        n_app = rng.integers(0, 3, G).astype(np.uint64)
        n_app = np.where(rng.random(G) < 0.01, 40, n_app).astype(np.uint64)
        acks = np.full((R, G), NO, dtype=np.uint64)
        for r in range(R):
            u = rng.random(G)
            a = np.where(u < 0.15, NO, head - np.minimum(rng.integers(0, 4, G).astype(np.uint64), head))
            a = np.where((u > 0.90) & (u < 0.95), head + n_app, a)                       # exactly the new head
            a = np.where(u > 0.993, head + n_app + np.uint64(1 + 5 * (t % 3)), a)        # above it: replay / panic
            acks[r] = a
        acks[slots, gi] = n_app
        if t == 31:
            acks[slots[:3], gi[:3]] = [capi.MAX_DENSE_APPENDS, NO, capi.MAX_DENSE_APPENDS - 1 if R == 2 else 3]
        for e in (ref, ora):
            e.step_dense_acks(acks)
        TOTAL["n"] += int((acks != NO).sum())
        if t % 5 == 4:
            rows, n = failure_rows(77, t, 0, G, R, ora.node_ids, slots, 4)
            b = random_batch(rng, ora, 200)
            for e in (ref, ora):
                if n:
                    e.submit_columns(**rows)
                    e.step(100 * t)
                e.submit_columns(**b)
                e.step(100 * t + 1)
            TOTAL["n"] += 200 + len(rows["kind"])
        compare_drains(ref, ora, f"R={R} tick {t}")
        if t % 6 == 5 or t == ticks - 1:
            compare_snapshots(ref, ora, f"R={R} tick {t}")
    assert ref.counters()["decisions"] == ora.counters()["decisions"] > G


@pytest.mark.parametrize("R,flags", [(3, 0), (5, capi.CFG_SEPARATE_COMMIT_KEY), (2, 0), (8, capi.CFG_SEPARATE_COMMIT_KEY)])
def test_dense_node_ticks(R, flags):
    """jg_step_dense_leader / jg_step_dense_follower as specified in the header ("equivalent to
    submitting those commands"): random mailboxes into leaders and followers, mixed roles."""
    G, ticks = 160, 40
    rng = np.random.default_rng(9000 + R)
    slots = rng.integers(0, R, G).astype(np.uint8)
    ref, ora = pair(G, R, seed=60 + R, self_slots=slots, flags=flags, election_timeout_ms=(300, 600))
    lead = np.arange(G) % 3 != 0  # two thirds of the groups are led here, the others follow
    for e in (ref, ora):
        e.submit_columns(np.full(int(lead.sum()), capi.CMD_TIMEOUT, np.uint8), np.nonzero(lead)[0].astype(np.uint32))
        e.step(0)
        ids = np.array(e.node_ids, np.uint32)
        for k in range(1, R // 2 + 1):
            g = np.nonzero(lead)[0].astype(np.uint32)
            e.submit_columns(np.full(len(g), capi.CMD_VOTE_RESPONSE, np.uint8), g,
                             from_=ids[(slots[g].astype(np.int64) + k) % R], term=np.ones(len(g), np.uint64),
                             flag=np.ones(len(g), np.uint8))
            e.step(0)
    compare_snapshots(ref, ora, "node set-up")
    compare_drains(ref, ora, "node set-up")
    now = 0
    self_ids = np.array(ora.node_ids, np.uint32)[slots]
    for t in range(ticks):
        now += int(rng.integers(40, 260))
        acks, hbr_has, hbr_commit = random_leader_inbox(rng, G, R, slots, ora.read("head").astype(np.uint64))
        outs = [e.step_dense_leader(now, acks, hbr_has, hbr_commit, tick=True) for e in (ref, ora)]
        for k in outs[0]:
            assert np.array_equal(outs[0][k], outs[1][k]), f"R={R} tick {t}: leader outbox {k}"
        fin = random_follower_inbox(rng, G, ora.node_ids, self_ids, ora.read("head"), ora.read("commit"), ora.read("term")) \
            if R > 1 else None
        if fin is not None:
            outs = [e.step_dense_follower(now, **fin, tick=True) for e in (ref, ora)]
            for k in outs[0]:
                assert np.array_equal(outs[0][k], outs[1][k]), f"R={R} tick {t}: follower outbox {k}"
        TOTAL["n"] += int((acks != capi.NO_ACK).sum()) + 2 * G
        compare_drains(ref, ora, f"R={R} node tick {t}")
        if t % 5 == 4 or t == ticks - 1:
            compare_snapshots(ref, ora, f"R={R} node tick {t}")


def test_closed_loop_clusters_agree():
    """R engines per side exchanging only mailbox columns (DenseCluster): 30 protocol rounds."""
    for R, lead in ((3, 0), (5, 2)):
        a, b = DenseCluster(RefEngine, 48, R, seed=5, lead=lead), DenseCluster(oracle_engine, 48, R, seed=5, lead=lead)
        rng = np.random.default_rng(R)
        for t in range(30):
            appends = rng.integers(0, 3, 48).astype(np.uint64)
            oa, ob = a.round(appends), b.round(appends)
            for r in range(R):
                for k in oa[r]:
                    assert np.array_equal(oa[r][k], ob[r][k]), (R, t, r, k)
            for ra, rb in zip(a.rows[-1], b.rows[-1]):
                assert ra.tobytes() == rb.tobytes()
        for r in range(R):
            compare_snapshots(a.nodes[r], b.nodes[r], f"cluster R={R} node {r}")
        assert int(a.nodes[lead].read("commit").min()) > 10


def test_zz_more_than_a_million_commands_were_compared():
    assert TOTAL["n"] >= 1_000_000, TOTAL


@pytest.mark.parametrize("R,also", [(3, ()), (3, (2,)), (5, ()), (5, (2, 3))])
def test_routed_cluster_ref_py_vs_oracle(R, also):
    """The configs[4] cluster trace (leader restarts, a restarted follower campaigns, the others answer
    through can_vote, every vote delivered a round later by the Python statement of the transport) on
    clusters of ref_py engines and of oracle engines: every state column of every node after every
    round, the rows delivered, the rows kept.  Restart, can_vote, Defeated and the re-campaign timers of
    R >= 2 elections are thereby held to the independent reading of the Rust, not only to the oracle."""
    from dense_node import RoutedCluster
    from josefine_amd.traces import cluster_failure_rows
    G, T = 48, 45
    ref = RoutedCluster(RefEngine, G, R, seed=11)
    ora = RoutedCluster(oracle_engine, G, R, seed=11)
    for t in range(T):
        inj = cluster_failure_rows(5, t, G, R, 6, also=also) if t >= 2 else None
        for cl in (ref, ora):
            cl.round(np.ones(G, np.uint64), inject=inj)
        for n in range(R):
            for name in capi.FIELD_NAMES:
                if name == "match":
                    for q in range(R):
                        assert np.array_equal(ref.nodes[n].read("match", q), ora.nodes[n].read("match", q)), (t, n, q)
                else:
                    assert np.array_equal(ref.nodes[n].read(name), ora.nodes[n].read(name)), (t, n, name)
            a = [p[1] for p in ref.inbound[n]]
            b = [p[1] for p in ora.inbound[n]]
            assert len(a) == len(b) and all(x.tobytes() == y.tobytes() for x, y in zip(a, b)), (t, n)
    assert sum(ref.delivered) == sum(ora.delivered) > 0
    for n in range(R):
        assert ref.kept[n].tobytes() == ora.kept[n].tobytes()
    if R == 3 and also:  # with a majority of fresh replicas the candidate wins: leadership really moved
        assert (ora.nodes[1].read("role") == capi.ROLE_LEADER).any()


@pytest.mark.parametrize("R,percent,also", [(3, 4, ()), (5, 3, ()), (5, 3, (2,)), (3, 4, (2,))])
def test_routed_failure_cluster_oracle_vs_ref_py(R, percent, also):
    """BASELINE configs[4] as the cluster runs it - R nodes, dense mailboxes for the steady-state traffic, every
    other message (the votes of the re-elections, a winner's Heartbeat) routed between the nodes as rows, leaders
    crashing and restarting (tests/dense_node.py::RoutedCluster, the host-side statement the device transport is
    held to) - once over the C++ oracle and once over the independent Python reading of the Rust: every state
    column of every node after every round, the rows delivered per node and the rows kept for the host.  The
    R >= 2 semantics of that trace are then pinned by two restatements that share no code."""
    from dense_node import RoutedCluster, cluster_failure_rows
    from parity import compare_snapshots
    G, T = 120, 36
    ora = RoutedCluster(oracle_engine, G, R, seed=5)
    ref = RoutedCluster(lambda *a, **kw: RefEngine(*a, **kw), G, R, seed=5)
    moved = 0
    for t in range(T):
        inj = cluster_failure_rows(99, t, G, R, percent, also=also) if t >= 3 else [None] * R
        ora.round(np.ones(G, np.uint64), inject=inj)
        ref.round(np.ones(G, np.uint64), inject=[None if c is None else dict(c) for c in inj])
        for n in range(R):
            compare_snapshots(ref.nodes[n], ora.nodes[n], f"routed round {t} node {n}")
            if n != ora.lead:
                moved = max(moved, int((ora.nodes[n].read("role") == capi.ROLE_LEADER).sum()))
        assert ora.delivered.tolist() == ref.delivered.tolist(), t
        assert [k.tobytes() for k in ora.kept] == [k.tobytes() for k in ref.kept], t
    assert ora.delivered.sum() > 10 * G // 10  # elections did run through the transport
    if also and R == 3:
        assert moved > 0  # ... and with the extra restart the candidate wins: a node other than the lead one leads


@pytest.mark.parametrize("R,P,seed,chaos", [(3, 30, 1, False), (5, 20, 2, False), (3, 30, 3, True), (5, 20, 4, True)])
def test_closed_loop_cluster_oracle_vs_ref_py(R, P, seed, chaos):
    """tests/cluster_sim.py - every replica of every partition an instance of ONE engine, every output row of a round
    the next round's input: elections with real vote grants, heartbeats, replicate(), follower commit advance, and
    (chaos) message loss, duplication and process restarts that drive the cluster into the reference's own panic
    paths - once over the C++ oracle and once over ref_py: each round's message, fsm and fault rows byte for byte
    and every state column, or the two clusters drift apart within a round or two."""
    from cluster_sim import Cluster

    def make(factory):
        e = factory(P * R, R, seed=40 + seed, self_slots=Cluster.self_slots(P, R), flags=capi.CFG_SEPARATE_COMMIT_KEY)
        return Cluster(e, P, R, **({"chaos_seed": 1000 + seed} if chaos else {}))

    ora, ref = make(oracle_engine), make(lambda *a, **kw: RefEngine(*a, **kw))
    rng_o, rng_r = np.random.default_rng(9), np.random.default_rng(9)
    codes = set()
    for rnd in range(90):
        mo, fo, xo = ora.round(rng=rng_o)
        mr, fr, xr = ref.round(rng=rng_r)
        assert mo.tobytes() == mr.tobytes(), f"round {rnd}: messages differ"
        assert fo.tobytes() == fr.tobytes(), f"round {rnd}: fsm rows differ"
        assert xo.tobytes() == xr.tobytes(), f"round {rnd}: fault rows differ"
        compare_snapshots(ref.e, ora.e, f"cluster R={R} round {rnd}")
        codes |= set(int(c) for c in xo["code"])
    assert (ora.e.read("role") == capi.ROLE_LEADER).sum() > (0.3 if chaos else 0.8) * P
    assert int(ora.e.read("commit").max()) > 3
    assert all(c < 128 for c in codes), codes  # only the reference's own failure modes, never an engine limit
