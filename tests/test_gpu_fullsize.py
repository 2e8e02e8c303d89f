"""Parity at BASELINE.json's full sizes.  The oracle cannot run 10^6 groups in seconds, so
(1) the counter-based trace lets it re-run *windows* of the full problem (any shard
regenerates its slice from the global group id) which are compared bit for bit, and
(2) size-independent properties are checked over every group: commit <= head, match <= head,
commit == element R/2 of the sorted match heads (the majority invariant), monotone commit,
no faults, and the closed form of the steady-state stream."""
import ctypes as C

import numpy as np
import pytest

from josefine_amd import BatchedRaft, capi
from oracle_lib import oracle_engine
from parity import DeviceSynth, elect_all, synth_tick_host

pytestmark = pytest.mark.gpu

SEED = 0x6A6F736566696E65


def run_device(G, R, mode, ticks, on_tick=None):
    dev = BatchedRaft(G, R, seed=SEED)
    elect_all(dev)
    dev.drain_messages(), dev.drain_applies()
    synth = DeviceSynth(dev)
    for t in range(ticks):
        synth.fill(mode, t)
        dev._check(dev.api.step_dense_acks_device(dev._h, synth.acks))
        if on_tick:
            on_tick(t, dev)
    synth.close()
    return dev


def window_oracle(base, W, R, mode, ticks):
    ora = oracle_engine(W, R, seed=SEED, group_base=base)
    elect_all(ora)
    sim = np.zeros((R, W), dtype=np.uint64)
    for t in range(ticks):
        ora.step_dense_acks(synth_tick_host(ora, mode, t, sim))
    return ora


def check_properties(dev, R):
    head, commit = dev.read("head"), dev.read("commit")
    m = np.stack([dev.read("match", r) for r in range(R)])
    assert not dev.read("fault").any()
    assert (commit <= head).all()
    assert (m <= head[None, :]).all()
    # majority invariant: once a leader has evaluated commit() after its last match change,
    # commit >= the R/2-th largest match head, and it never exceeds the largest
    kth = np.sort(m, axis=0)[::-1][R // 2]
    assert (commit >= kth).all() and (commit <= m.max(axis=0)).all()
    return head, commit


@pytest.mark.parametrize("G,R,mode,ticks", [
    (1_000_000, 5, 1, 24),   # configs[2] size, ragged stream (drops, stale duplicates)
    (1_250_000, 3, 1, 24),   # configs[3]: one GPU's shard of 10 M x 3
    (1_000_000, 5, 0, 30),   # configs[2]: the steady-state stream the bench times
])
def test_full_size_windows_and_properties(G, R, mode, ticks):
    prev = {"commit": None}

    def on_tick(t, dev):
        if t % 8 == 7:  # monotone commit, sampled
            c = dev.read("commit")
            if prev["commit"] is not None:
                assert (c >= prev["commit"]).all()
            prev["commit"] = c

    dev = run_device(G, R, mode, ticks, on_tick)
    head, commit = check_properties(dev, R)
    if mode == 0:  # closed form: 1 append per tick, acks lag one tick
        assert (head == ticks).all() and (commit == ticks - 1).all()
        assert (dev.read("repl_state") == (1 << R) - 1).all()
    W = 2048
    for base in (0, 4096 * 7, G // 2 - 1000, G - W):
        ora = window_oracle(base, W, R, mode, ticks)
        assert np.array_equal(commit[base:base + W], ora.read("commit")), base
        assert np.array_equal(head[base:base + W], ora.read("head")), base
        assert np.array_equal(dev.read("repl_state", 0, base, W), ora.read("repl_state")), base
        for r in range(R):
            assert np.array_equal(dev.read("match", r, base, W), ora.read("match", r)), (base, r)
    assert dev.counters()["dense_group_steps"] == G * ticks


def test_full_size_closed_loop_cluster():
    """BASELINE configs[2]-sized closed loop: 3 nodes x 1 M partitions as three engines exchanging
    dense mailbox columns; oracle clusters re-run windows of the partitions (appends are a
    function of the global partition id) and must agree on every column of every node."""
    from dense_node import DenseCluster
    from josefine_amd.traces import synth_hash

    G, R, T, W = 1_000_000, 3, 16, 1024

    def appends(base, n, t):
        return (synth_hash(SEED, t, np.arange(base, base + n, dtype=np.uint64), 5) % np.uint64(3)).astype(np.uint64)

    dc = DenseCluster(BatchedRaft, G, R, seed=9)
    for t in range(T):
        dc.round(appends(0, G, t))
    lead = dc.nodes[0]
    head, commit = lead.read("head"), lead.read("commit")
    assert not lead.read("fault").any() and (commit <= head).all() and int(commit.min()) > 0
    for r in (1, 2):
        f = dc.nodes[r]
        assert not f.read("fault").any() and (f.read("role") == capi.ROLE_FOLLOWER).all()
        assert (f.read("head") <= head).all() and (f.read("commit") <= commit).all()
    assert all(len(rows) == 0 for per_round in dc.rows for rows in per_round)
    for base in (0, 777_000, G - W):
        oc = DenseCluster(oracle_engine, W, R, seed=9, group_base=base)
        for t in range(T):
            oc.round(appends(base, W, t))
        for r in range(R):
            for name in ("commit", "head", "term", "voted_for", "election_timeout", "repl_state"):
                assert np.array_equal(dc.nodes[r].read(name, 0, base, W), oc.nodes[r].read(name)), (base, r, name)
            for q in range(R):
                assert np.array_equal(dc.nodes[r].read("match", q, base, W), oc.nodes[r].read("match", q)), (base, r, q)


def test_full_size_routed_cluster_failures():
    """BASELINE configs[4] as specified at full size: 5 nodes x 1 M partitions, per
    round 1 % of the partitions lose their leader (crash + restart), a restarted follower times out and
    campaigns, the other replicas answer through can_vote — every VoteRequest / VoteResponse routed
    between the nodes on the device (jg_dense_cluster_round_routed).  Oracle clusters with the Python
    statement of the transport re-run windows of the partitions and must agree on every state column
    of every node; the whole population is checked through what the reference's rules imply (§7.3 Q4/Q5:
    the failing partitions stay leaderless, the others keep committing)."""
    from josefine_amd import DenseCluster as LibCluster
    from josefine_amd.traces import cluster_failure_rows, elect_all
    from dense_node import RoutedCluster

    G, R, T, W, P = 1_000_000, 5, 30, 1024, 1
    nodes = [BatchedRaft(G, R, seed=9 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY)
             for r in range(R)]
    elect_all(nodes[0])
    nodes[0].drain_messages(), nodes[0].drain_applies()
    lib = LibCluster(nodes)
    lib.set_appends(1)
    failed = np.zeros(G, bool)
    delivered = 0
    for t in range(T):
        inj = cluster_failure_rows(SEED, t, G, R, P) if t >= 2 else [None] * R
        if inj[0] is not None:
            failed[inj[0]["group"]] = True
        up = [None if c is None else nodes[n].upload_rows(**c) for n, c in enumerate(inj)]
        st = lib.round_routed((t + 1) * 100, up)
        delivered += sum(st["delivered"])
        assert st["kept"] == 0 and st["fsm_rows"] == 0
        for rows in up:
            if rows is not None:
                rows.free()
    L = nodes[0]
    assert 0.2 * G < failed.sum() < 0.3 * G and delivered > failed.sum() * 2 * (R - 1)
    role = L.read("role")
    assert (role[failed] == capi.ROLE_FOLLOWER).all() and (role[~failed] == capi.ROLE_LEADER).all()
    assert (L.read("head")[~failed] == T).all() and (L.read("commit")[~failed] >= T - 3).all()
    for n in nodes:
        assert not n.read("fault").any() and (n.read("role")[failed] != capi.ROLE_LEADER).all()
        assert len(n.drain_messages()) == 0
    for base in (0, 555_000, G - W):
        oc = RoutedCluster(oracle_engine, W, R, seed=9, group_base=base)
        for t in range(T):
            oc.round(np.ones(W, np.uint64), inject=cluster_failure_rows(SEED, t, W, R, P, group_base=base) if t >= 2 else None)
        for r in range(R):
            for name in ("commit", "head", "term", "voted_for", "role", "leader_id", "election_timeout", "vote_seen",
                         "vote_granted", "repl_state", "fault"):
                assert np.array_equal(nodes[r].read(name, 0, base, W), oc.nodes[r].read(name)), (base, r, name)
    lib.close()


def test_full_size_stationary_trace_with_the_vote_mail():
    """BASELINE configs[4] as bench.py times it since round 5, at full size: 5 nodes x 1 M partitions, the STATIONARY trace
    (josefine_amd.traces.FailureRepairTrace: 1 %/round failures, the partition re-created 8 rounds later - JG_CMD_RECREATE at
    every replica, replica 0 campaigns and is ELECTED through the transport two rounds after: no synthetic vote) with the election's traffic as mailbox words (JG_CLUSTER_OPT_VOTE_WORDS),
    the client's proposals withdrawn and offered again on the device.  Oracle clusters that move every message as a row
    re-run windows of the partitions and must agree on every state column of every node; the whole population through
    what the trace implies (leadership exactly where it says, re-created partitions appending again, no fault, no row kept)."""
    from josefine_amd import DenseCluster as LibCluster
    from josefine_amd.traces import FailureRepairTrace, elect_all
    from dense_node import RoutedCluster

    G, R, T, W, P, D = 1_000_000, 5, 36, 1024, 1, 8
    nodes = [BatchedRaft(G, R, seed=9 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY)
             for r in range(R)]
    elect_all(nodes[0])
    nodes[0].drain_messages(), nodes[0].drain_applies()
    lib = LibCluster(nodes, vote_words=True)
    lib.set_appends(1)
    tr = FailureRepairTrace(SEED, G, R, P, D, node_ids=[nodes[r].node_ids[r] for r in range(R)])
    as_rows = 0
    for t in range(T):
        inj, failing, repaired = tr.rows(t)
        lists = [nodes[0].upload_u32(x) if len(x) else None for x in (failing, repaired)]
        if lists[0] is not None:
            lib.withdraw_appends(lists[0].ptr, len(failing))
        if lists[1] is not None:
            lib.offer_appends(lists[1].ptr, len(repaired), 1)
        up = [None if c is None else nodes[n].upload_rows(**c) for n, c in enumerate(inj)]
        st = lib.round_routed((t + 1) * 100, up)
        as_rows += sum(st["delivered"])
        assert st["kept"] == 0 and st["fsm_rows"] == 0, t
        for rows in up + lists:
            if rows is not None:
                rows.free()
    L = nodes[0]
    down = tr.leaderless()
    assert 0.06 < down.mean() < 0.12 and tr.ever_failed.mean() > 0.25  # (about p x (D + 2): a re-created partition's election takes two rounds)
    role = L.read("role")
    assert (role[down] != capi.ROLE_LEADER).all() and (role[~down] == capi.ROLE_LEADER).all()
    never = ~tr.ever_failed
    assert (L.read("head")[never] == T).all() and (L.read("commit")[never] >= T - 3).all()
    again = tr.ever_failed & ~down
    assert again.sum() > 0.15 * G and (L.read("head")[again] > 0).all() and (L.read("head")[again] < T).all()
    for n in nodes:
        assert not n.read("fault").any() and len(n.drain_messages()) == 0
    for n in nodes[1:]:
        assert (n.read("role") != capi.ROLE_LEADER).all()
    # the campaigns of the leaderless partitions travelled as words: fewer rows than campaigns' copies alone would be
    assert as_rows < 8 * (R - 1) * tr.ever_failed.sum()
    for base in (0, 555_000, G - W):
        oc = RoutedCluster(oracle_engine, W, R, seed=9, group_base=base)
        tw = FailureRepairTrace(SEED, W, R, P, D, group_base=base, node_ids=oc.member_ids)
        for t in range(T):
            inj = tw.rows(t)[0]
            oc.round(tw.appends(), inject=inj)
        assert np.array_equal(tw.leaderless(), down[base:base + W])
        for r in range(R):
            for name in ("commit", "head", "term", "voted_for", "role", "leader_id", "election_timeout", "vote_seen",
                         "vote_granted", "repl_state", "fault"):
                assert np.array_equal(nodes[r].read(name, 0, base, W), oc.nodes[r].read(name)), (base, r, name)
    lib.close()


@pytest.mark.parametrize("R,recreate", [(3, False), (3, True), (5, False), (5, True)], ids=["restart-3", "recreate-3", "restart-5", "recreate-5"])
def test_full_size_any_leader_cluster_elections_and_failures(R, recreate):
    """(R = 5, since round 6: a FIVE-node election is won through the device transport like a three-node one - it delivers every
    voter's first answer before anybody's second, so the candidate holds its quorum of grants before the refusals of its
    further copies overwrite them: candidate.rs:30-37, election.rs:33-35.  recreate: the failing groups come back on EMPTY stores, JG_CMD_RECREATE - any group, any number of times, the client
    never stops proposing: bench.py --cluster --any-leader --failures 1 --recreate, the stationary trace whose every vote is real.)
    Per-partition leadership at full size (jg_dense_cluster_create, JG_CLUSTER_ANY_LEADER): 3 nodes x 1 M partitions;
    every partition's leader is ELECTED through the device transport (Timeout at the designated candidate, VoteRequests
    routed, answered through can_vote, the first majority elects: candidate.rs:101-113), then every node leads a third
    of the partitions and follows the rest over the cluster's mailbox columns; from round 8 on 1 % of the partitions per
    round lose their whole group (restart), the next replica campaigns and wins through the transport, leadership moves.
    Windows of the partitions are re-run by the numpy statement of the round over oracle engines
    (tests/dense_node.py::AnyLeaderCluster) and must agree on every state column of every node; the whole population is
    checked through what the trace implies."""
    from josefine_amd import DenseCluster as LibCluster
    from josefine_amd.traces import any_failure_rows
    from dense_node import AnyLeaderCluster

    G, T, W, P, F0 = 1_000_000, 26, 1024, 1, 8
    nodes = [BatchedRaft(G, R, seed=9 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY)
             for r in range(R)]
    lib = LibCluster(nodes, lead=None)
    leader_of = np.arange(G) % R  # (interleaved: every wave serves both roles)

    def trace(t, n, base, failed):
        """round t's injected rows for the partitions [base, base + n) and what it offers: (columns per node, appends)"""
        if t == 0:  # the campaigns
            cols = []
            for r in range(R):
                mine = np.nonzero(leader_of[base:base + n] == r)[0].astype(np.uint32)
                cols.append(dict(kind=np.full(len(mine), capi.CMD_TIMEOUT, np.uint8), group=mine) if len(mine) else None)
            return cols, 0
        if t < F0:
            return [None] * R, int(t >= 4)
        cols, failing = any_failure_rows(SEED, t, n, R, P, leader_of[base:base + n], group_base=base, whole_group=True,
                                         skip=None if recreate else failed, recreate=recreate)
        failed[failing] = True
        return cols, 1

    failed = np.zeros(G, bool)
    offered = np.zeros(G, np.uint64)
    delivered = 0
    for t in range(T):
        cols, app = trace(t, G, 0, failed)
        offered = np.where(failed & (not recreate), np.uint64(0), np.uint64(app))  # the client withdraws from a partition that lost its leader (not from a re-created one)
        lib.set_appends(per_group=offered)
        up = [None if c is None else nodes[n].upload_rows(**c) for n, c in enumerate(cols)]
        st = lib.round_routed((t + 1) * 100, up)
        delivered += sum(st["delivered"])
        for rows in up:
            if rows is not None:
                rows.free()
    assert delivered >= 2 * (R - 1) * (R - 1) * G  # every election's VoteRequests and VoteResponses went through the transport: R - 1 copies to each of R - 1 peers, and as many answers
    assert 0.1 * G < failed.sum() < 0.25 * G
    healthy = ~failed
    rounds_with_appends = T - 4
    led = np.zeros(G, bool)
    for n, e in enumerate(nodes):
        role, head, commit, fault = e.read("role"), e.read("head"), e.read("commit"), e.read("fault")
        mine = healthy & (leader_of == n)
        assert (role[mine] == capi.ROLE_LEADER).all() and (head[mine] == rounds_with_appends).all() and (commit[mine] >= rounds_with_appends - 3).all()
        assert (role[healthy & (leader_of != n)] == capi.ROLE_FOLLOWER).all() and not fault[healthy].any()
        led |= (role == capi.ROLE_LEADER) & (fault == 0)
    assert led[healthy].all() and led[failed].mean() > 0.7  # the campaigns after the failures were won: leadership moved
    if recreate:  # ... and the winners of re-created groups APPEND (no Q8 for a chain that starts over)
        heads = np.stack([e.read("head") for e in nodes])
        roles = np.stack([e.read("role") for e in nodes])
        won = failed & (roles[(leader_of + 1) % R, np.arange(G)] == capi.ROLE_LEADER)
        assert won.sum() > 0.7 * failed.sum() and (heads[(leader_of + 1) % R, np.arange(G)][won] > 0).mean() > 0.8
    for base in (0, 333_333, G - W):
        oc = AnyLeaderCluster(oracle_engine, W, R, seed=9, group_base=base)
        f = np.zeros(W, bool)
        for t in range(T):
            cols, app = trace(t, W, base, f)
            oc.round(np.where(f & (not recreate), np.uint64(0), np.uint64(app)), inject=cols)
        assert np.array_equal(f, failed[base:base + W])
        for r in range(R):
            for name in ("commit", "head", "term", "voted_for", "role", "leader_id", "election_timeout", "vote_seen",
                         "vote_granted", "repl_state", "fault"):
                assert np.array_equal(nodes[r].read(name, 0, base, W), oc.nodes[r].read(name)), (base, r, name)
            for q in range(R):
                assert np.array_equal(nodes[r].read("match", q, base, W), oc.nodes[r].read("match", q)), (base, r, q)
    lib.close()
