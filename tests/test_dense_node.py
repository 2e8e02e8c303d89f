"""Dense node tick (jg_step_dense_leader / jg_step_dense_follower): the HIP engine against the
CPU oracle — outbox columns, every state column and every drained row, bit for bit — plus
the closed-loop cluster in which engines exchange nothing but mailbox columns."""
import numpy as np
import pytest

from josefine_amd import BatchedRaft, Command, capi
from oracle_lib import oracle_engine
from parity import compare_drains, compare_snapshots, elect_all
from dense_node import DenseCluster, random_follower_inbox, random_leader_inbox

NO = capi.NO_ACK


def _cmp_cols(a: dict, b: dict, what: str):
    for k in a:
        if not np.array_equal(a[k], b[k]):
            bad = np.argwhere(a[k] != b[k])[:6]
            raise AssertionError(f"{what}: outbox column {k} differs at {bad.tolist()}: "
                                 f"hip={[a[k][tuple(i)] for i in bad]} oracle={[b[k][tuple(i)] for i in bad]}")


# ---- oracle-only sanity (runs without a GPU): pins the mailbox specification itself ------------
def test_oracle_closed_loop_commits():
    """Steady state through real protocol rounds: 1 append per tick, every follower acks what
    the leader replicated — the commit index follows the head."""
    G, R, T = 64, 3, 40
    cl = DenseCluster(oracle_engine, G, R)
    for t in range(T):
        cl.round(np.ones(G, np.uint64))
    lead = cl.nodes[0]
    assert (lead.read("head") == T).all()
    commit = lead.read("commit")
    assert (commit >= T - 4).all() and (commit <= T).all()
    for r in (1, 2):
        f = cl.nodes[r]
        assert (f.read("role") == capi.ROLE_FOLLOWER).all() and (f.read("fault") == 0).all()
        assert (f.read("head") >= T - 2).all()
        assert (f.read("commit") >= T - 8).all()
        assert (f.read("voted_for") == 1).all() and (f.read("term") == 1).all()
    # nothing left the mailbox vocabulary
    assert all(len(rows) == 0 for per_round in cl.rows for rows in per_round)


def test_oracle_dense_follower_equals_commands():
    """The follower half is sugar for Heartbeat / AppendEntries / Tick commands."""
    G, R = 256, 3
    a, b = oracle_engine(G, R, seed=4), oracle_engine(G, R, seed=4)
    rng = np.random.default_rng(5)
    now = 0
    for t in range(25):
        now += 150
        inbox = random_follower_inbox(rng, G, a.node_ids, np.full(G, 1, np.uint32), a.read("head"), a.read("commit"),
                                      a.read("term"))
        out = a.step_dense_follower(now, **inbox, tick=True)
        for g in range(G):
            if inbox["hb_commit"][g] != NO:
                b.submit(g, Command.Heartbeat(int(inbox["term"][g]), int(inbox["hb_commit"][g]), int(inbox["leader"][g])))
            if inbox["ae_n"][g] != capi.AE_NONE:
                f = int(inbox["ae_from"][g])
                b.submit(g, Command.AppendEntries(int(inbox["term"][g]), int(inbox["leader"][g]),
                                                  [(f + 1 + k, f + k) for k in range(int(inbox["ae_n"][g]))]))
            if b.read("role", g0=g, n=1)[0] != capi.ROLE_LEADER:
                b.submit(g, Command.Tick())
        b.step(now)
        compare_snapshots(a, b, f"tick {t}")
        rows = b.drain_messages()
        b.drain_applies()
        ack = np.full(G, NO, np.uint64)
        for m in rows[rows["kind"] == capi.CMD_APPEND_RESPONSE]:
            ack[m["group"]] = m["id"]
        assert np.array_equal(ack, out["ack_head"])
        other = rows[(rows["kind"] != capi.CMD_APPEND_RESPONSE) & (rows["kind"] != capi.CMD_HEARTBEAT_RESPONSE)]
        mine = a.drain_messages()
        assert other.tobytes() == mine.tobytes()


# ---- parity proper ---------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("R,flags,layout", [(3, capi.CFG_SEPARATE_COMMIT_KEY, "slot0"), (5, capi.CFG_SEPARATE_COMMIT_KEY, "last"),
                                            (5, 0, "slot0"), (3, capi.CFG_SEPARATE_COMMIT_KEY, "mixed"),
                                            (1, 0, "slot0"), (2, capi.CFG_SEPARATE_COMMIT_KEY, "slot0")])
def test_dense_leader_tick_parity(R, flags, layout):
    G = 3000
    slots = {"slot0": None, "last": np.full(G, R - 1, np.uint8), "mixed": (np.arange(G) % R).astype(np.uint8)}[layout]
    dev = BatchedRaft(G, R, seed=21, flags=flags, self_slots=slots)
    ora = oracle_engine(G, R, seed=21, flags=flags, self_slots=slots)
    for e in (dev, ora):
        elect_all(e)
        e.drain_messages(), e.drain_applies()
    rng = np.random.default_rng(R * 7 + flags)
    sl = ora.read("self_slot")
    now = 0
    for t in range(40):
        now += 70
        acks, hbr_has, hbr_commit = random_leader_inbox(rng, G, R, sl, ora.read("head"))
        tick = t % 5 != 4
        oa = dev.step_dense_leader(now, acks, hbr_has, hbr_commit, tick=tick)
        ob = ora.step_dense_leader(now, acks, hbr_has, hbr_commit, tick=tick)
        if tick:
            _cmp_cols(oa, ob, f"R={R} tick {t}")
        compare_snapshots(dev, ora, f"R={R} leader tick {t}")
        compare_drains(dev, ora, f"R={R} leader tick {t}")
    assert dev.counters()["decisions"] == ora.counters()["decisions"]
    if R > 1 and flags:
        assert int(ora.read("commit").max()) > 0
    if R > 1 and not flags:  # Q9: the reference leader runs into the "commit" key
        assert (ora.read("fault") == capi.FAULT_RANGE_HIT_COMMIT_KEY).any()


@pytest.mark.gpu
def test_dense_leader_tick_restarted_leaders_send_columns_irregular_ones_rows():
    """The (from, n) columns can express a Tick exactly when the leader's chain is a run: the id set is [0, top], every
    parent id - 1.  A leader that was restarted with a commit and re-elected still has one (its head at its commit
    index, id_gen re-seeded: chain.rs:117-137; it can never append again, Q8, but it replicates): columns.  A leader
    whose chain has a gap does not: its Tick goes out as rows.  Exactly as the oracle says, both."""
    G, R = 600, 3
    dev = BatchedRaft(G, R, seed=2, flags=capi.CFG_SEPARATE_COMMIT_KEY)
    ora = oracle_engine(G, R, seed=2, flags=capi.CFG_SEPARATE_COMMIT_KEY)
    gap = np.arange(2, G, 4, dtype=np.uint32)       # a block far up the id space whose parent is genesis: {0, 9}
    restarted = np.arange(0, G, 4, dtype=np.uint32)
    for e in (dev, ora):
        n = len(gap)
        e.submit_columns(np.full(n, capi.CMD_APPEND_ENTRIES, np.uint8), gap, from_=np.full(n, 2, np.uint32), term=np.ones(n, np.uint64),
                         id=np.arange(n, dtype=np.uint64), aux=np.ones(n, np.uint64), blk_id=np.full(n, 9, np.uint64), blk_next=np.zeros(n, np.uint64))
        e.step(0)
        e.submit_columns(np.full(n, capi.CMD_RESTART, np.uint8), gap)  # (voted for node 2 by now: only a restart lets them campaign, Q4)
        e.step(0)
        elect_all(e)
        acks = np.full((R, G), NO, np.uint64)
        acks[0] = 2
        acks[0][gap] = 0
        e.step_dense_leader(100, acks, tick=False)
        acks[0], acks[1] = 0, 2
        acks[1][gap] = NO
        e.step_dense_leader(200, acks, tick=False)  # commit 2
        g = restarted
        e.submit_columns(np.full(len(g), capi.CMD_RESTART, np.uint8), g)
        e.step(300)
        e.submit_columns(np.full(len(g), capi.CMD_TIMEOUT, np.uint8), g)
        e.submit_columns(np.full(len(g), capi.CMD_VOTE_RESPONSE, np.uint8), g, from_=np.full(len(g), 2, np.uint32),
                         term=np.ones(len(g), np.uint64), flag=np.ones(len(g), np.uint8))
        e.step(300)
        e.drain_messages(), e.drain_applies(), e.drain_faults()
    role = ora.read("role")
    assert (role == capi.ROLE_LEADER).all() and int(ora.read("id_gen")[0]) == 2 and not ora.read("fault").any()
    acks[:] = NO
    acks[0] = 0
    for t in range(3):
        oa = dev.step_dense_leader(500 + 150 * t, acks, tick=True)
        ob = ora.step_dense_leader(500 + 150 * t, acks, tick=True)
        _cmp_cols(oa, ob, f"tick {t}")
        assert (ob["ae_n"][1][restarted] != capi.AE_NONE).all() and (ob["ae_n"][1][1::2] != capi.AE_NONE).all()  # columns
        assert (ob["ae_n"][1][gap] == capi.AE_NONE).all()                                                       # rows
        compare_snapshots(dev, ora, f"irregular leaders tick {t}")
        a, b = dev.drain_messages(), ora.drain_messages()
        assert a.tobytes() == b.tobytes() and len(b) >= len(gap) * (R - 1)
        assert set(b["group"].tolist()) <= set(gap.tolist())


def _restarted_leaders_below_their_top(G, R, seed, make=None):
    """Both engines with every group led by slot 0, chain [0, 5], commit index 2; every other group then restarted and
    re-elected: head = 2 (chain.rs:117-137), the run above it still in the store, progress heads 0 (Q10).
    `make`: the engine held against the oracle (default: the HIP engine)."""
    dev = (make or BatchedRaft)(G, R, seed=seed, flags=capi.CFG_SEPARATE_COMMIT_KEY)
    ora = oracle_engine(G, R, seed=seed, flags=capi.CFG_SEPARATE_COMMIT_KEY)
    g = np.arange(0, G, 2, dtype=np.uint32)
    for e in (dev, ora):
        elect_all(e)
        acks = np.full((R, G), NO, np.uint64)
        acks[0] = 2
        e.step_dense_leader(100, acks, tick=False)
        acks[0], acks[1] = 3, 2
        e.step_dense_leader(200, acks, tick=False)  # head 5, commit 2
        e.submit_columns(np.full(len(g), capi.CMD_RESTART, np.uint8), g)
        e.step(300)
        e.submit_columns(np.full(len(g), capi.CMD_TIMEOUT, np.uint8), g)
        e.submit_columns(np.full(len(g), capi.CMD_VOTE_RESPONSE, np.uint8), g, from_=np.full(len(g), 2, np.uint32),
                         term=np.ones(len(g), np.uint64), flag=np.ones(len(g), np.uint8))
        e.step(300)
        e.drain_messages(), e.drain_applies(), e.drain_faults()
    assert (ora.read("role") == capi.ROLE_LEADER).all() and not ora.read("fault").any()
    assert (ora.read("head")[g] == 2).all() and (ora.read("commit")[g] == 2).all() and (ora.read("head")[1::2] == 5).all()
    return dev, ora, g


def _restarted_leader_script(G, R, seed, make=None):
    dev, ora, g = _restarted_leaders_below_their_top(G, R, seed=seed, make=make)
    script = [({}, "quiet"), ({1: 4}, "one ack above the head"), ({2: 5}, "majority at 4"), ({1: 3}, "a stale ack"),
              ({1: 6}, "an ack above the top: recorded"), ({2: 5}, "again"), ({2: 7}, "the majority above the top: panic")]
    for t, (acked, what) in enumerate(script):
        acks = np.full((R, G), NO, np.uint64)
        acks[0] = 0
        for r, h in acked.items():
            acks[r][g] = h
            acks[r][1::2] = min(h, 5)
        oa = dev.step_dense_leader(500 + 150 * t, acks, tick=True)
        ob = ora.step_dense_leader(500 + 150 * t, acks, tick=True)
        _cmp_cols(oa, ob, f"tick {t} ({what})")
        compare_snapshots(dev, ora, f"restarted leaders, tick {t} ({what})")
        assert dev.drain_messages().tobytes() == ora.drain_messages().tobytes()
        assert dev.drain_faults().tobytes() == ora.drain_faults().tobytes()
        if t == 2:
            assert (ora.read("commit")[g] == 4).all() and (ora.read("head")[g] == 2).all()
        if t < 6:
            assert not ora.read("fault").any()
    assert (ora.read("fault")[g] == capi.FAULT_COMMIT_MISSING_BLOCK).all() and not ora.read("fault")[1::2].any()


def test_restarted_leader_counts_acks_above_its_head_oracle_vs_the_python_restatement():
    """CPU: the scenario of the next test on the C++ oracle against tests/ref_py (the independent transliteration of
    the Rust): what a restarted, re-elected leader makes of acknowledgements above its head is not the oracle's
    reading alone."""
    from ref_py.engine import RefEngine
    _restarted_leader_script(24, 3, seed=4, make=RefEngine)


@pytest.mark.gpu
def test_dense_leader_tick_restarted_leader_counts_acks_above_its_head():
    """A restarted, re-elected leader's head sits at its commit index, BELOW the top of the run its store kept: the
    acknowledgements it receives name blocks above its head, all of them blocks it holds (chain.rs:197-202 looks the
    key up, it does not compare with the head), so its commit index moves past its head; an acknowledgement above the
    TOP is recorded like any other (progress.rs:133-140) and panics only once the majority rests on it.  The node tick
    keeps such a leader's progress heads as lags below that top; state, columns and faults as the oracle's, tick by
    tick."""
    _restarted_leader_script(640, 3, seed=4)


@pytest.mark.gpu
def test_step_node_restarted_leader_counts_acks_above_its_head():
    """The same leaders through jg_step_node (rows in, fsm rows out): the Apply ranges of a commit index that moves
    above the head, and a ClientRequest at such a leader - the re-seeded id generator hands out an id the chain holds
    (Q8): the panic of chain.rs:166-176 as a fault, at that group only."""
    G, R = 640, 3
    dev, ora, g = _restarted_leaders_below_their_top(G, R, seed=5)
    all_g = np.arange(G, dtype=np.uint32)

    def feed(e, rows):
        for kind, groups, kw in rows:
            e.submit_columns(np.full(len(groups), kind, np.uint8), groups, **kw)

    def ar(slot, head, groups=all_g):
        n = len(groups)
        return (capi.CMD_APPEND_RESPONSE, groups, dict(from_=np.full(n, slot + 1, np.uint32), term=np.ones(n, np.uint64),
                                                       id=np.minimum(np.full(n, head, np.uint64), np.where(np.isin(groups, g), head, 5).astype(np.uint64)),
                                                       flag=np.ones(n, np.uint8)))
    script = [[], [ar(1, 4)], [ar(2, 5)], [ar(1, 3), ar(2, 5)],
              [(capi.CMD_CLIENT_REQUEST, all_g[::5], dict(id=np.arange(len(all_g[::5]), dtype=np.uint64)))], [ar(1, 5)]]
    for t, rows in enumerate(script):
        for e in (dev, ora):
            feed(e, rows)
        oa = dev.step_node(500 + 150 * t)
        ob = ora.step_node(500 + 150 * t)
        for k in ("beat_term", "beat_commit", "ae", "answer", "hb_commit"):
            assert np.array_equal(oa[k], ob[k]), f"tick {t}: outbox column {k}"
        compare_snapshots(dev, ora, f"restarted leaders through step_node, tick {t}")
        assert dev.drain_messages().tobytes() == ora.drain_messages().tobytes()
        assert dev.drain_applies().tobytes() == ora.drain_applies().tobytes()
        assert dev.drain_faults().tobytes() == ora.drain_faults().tobytes()
    f = ora.read("fault")
    assert f[g[np.isin(g, all_g[::5])]].all() and not f[1::2].any()


def _mixed_role_engines(G, R, seed):
    dev = BatchedRaft(G, R, seed=seed, election_timeout_ms=(300, 600))
    ora = oracle_engine(G, R, seed=seed, election_timeout_ms=(300, 600))
    for e in (dev, ora):
        g = np.arange(0, G, 7, dtype=np.uint32)  # every 7th group: a candidate
        e.submit_columns(np.full(len(g), capi.CMD_TIMEOUT, np.uint8), g)
        g = np.arange(3, G, 11, dtype=np.uint32)  # some leaders
        e.submit_columns(np.full(len(g), capi.CMD_TIMEOUT, np.uint8), g)
        e.submit_columns(np.full(len(g), capi.CMD_VOTE_RESPONSE, np.uint8), g, from_=np.full(len(g), 2, np.uint32),
                         term=np.ones(len(g), np.uint64), flag=np.ones(len(g), np.uint8))
        g = np.arange(5, G, 13, dtype=np.uint32)  # followers with queued client requests
        e.submit_columns(np.full(len(g), capi.CMD_CLIENT_REQUEST, np.uint8), g, id=np.arange(len(g), dtype=np.uint64))
        e.step(0)
        e.drain_messages(), e.drain_applies(), e.drain_faults()
    return dev, ora


@pytest.mark.gpu
@pytest.mark.parametrize("R", [3, 5])
def test_dense_follower_tick_parity(R):
    """Random leader traffic into followers, candidates, leaders and queued-request holders:
    in-order windows, re-sent windows (the head moves backwards), gaps (extend Err), stale
    leaders (assert), heartbeats ahead of the chain, election timers firing."""
    G = 4000
    dev, ora = _mixed_role_engines(G, R, seed=31)
    rng = np.random.default_rng(R)
    self_ids = np.full(G, 1, np.uint32)
    now = 0
    saw_candidate, n_rows = False, 0
    for t in range(50):
        now += int(rng.integers(50, 260))
        inbox = random_follower_inbox(rng, G, ora.node_ids, self_ids, ora.read("head"), ora.read("commit"),
                                      ora.read("term"))
        tick = t % 4 != 3
        oa = dev.step_dense_follower(now, **inbox, tick=tick)
        ob = ora.step_dense_follower(now, **inbox, tick=tick)
        _cmp_cols(oa, ob, f"R={R} follower tick {t}")
        compare_snapshots(dev, ora, f"R={R} follower tick {t}")
        saw_candidate |= bool((ora.read("role") == capi.ROLE_CANDIDATE).any())
        a, b = dev.drain_messages(), ora.drain_messages()
        assert a.tobytes() == b.tobytes(), f"R={R} follower tick {t}: exceptional rows differ"
        n_rows += len(b)
        compare_drains(dev, ora, f"R={R} follower tick {t}")
    assert dev.counters()["decisions"] == ora.counters()["decisions"]
    f = ora.read("fault")
    assert (f == capi.FAULT_EXTEND_MISSING_PARENT).any() and (f == capi.FAULT_FOLLOWER_STALE_LEADER).any()
    # the run exercised the slow path too: candidates, queue flushes / vote requests as rows
    assert saw_candidate and n_rows > 0


@pytest.mark.gpu
def test_dense_follower_uniform_leader_and_sparse_mix():
    """leader_id for every group instead of a column; sparse steps between dense ones."""
    G, R = 1500, 3
    dev, ora = BatchedRaft(G, R, seed=8), oracle_engine(G, R, seed=8)
    rng = np.random.default_rng(9)
    now = 0
    for t in range(30):
        now += 120
        inbox = random_follower_inbox(rng, G, ora.node_ids, np.full(G, 1, np.uint32), ora.read("head"),
                                      ora.read("commit"), ora.read("term"))
        inbox.pop("leader")
        oa = dev.step_dense_follower(now, **inbox, leader_id=2, tick=True)
        ob = ora.step_dense_follower(now, **inbox, leader_id=2, tick=True)
        _cmp_cols(oa, ob, f"tick {t}")
        if t % 6 == 5:  # a sparse step in between: vote requests and client requests
            g = rng.integers(0, G, 200).astype(np.uint32)
            for e in (dev, ora):
                e.submit_columns(np.full(200, capi.CMD_CLIENT_REQUEST, np.uint8), g, id=np.arange(200, dtype=np.uint64))
                e.submit_columns(np.full(200, capi.CMD_VOTE_REQUEST, np.uint8), g, from_=np.full(200, 3, np.uint32),
                                 term=np.full(200, 9, np.uint64), id=np.full(200, 10**6, np.uint64),
                                 aux=np.full(200, 9, np.uint64))
                e.step(now)
        compare_snapshots(dev, ora, f"tick {t}")
        compare_drains(dev, ora, f"tick {t}")


@pytest.mark.gpu
@pytest.mark.parametrize("R,lead", [(3, 0), (5, 2)])
def test_closed_loop_cluster_dense(R, lead):
    """R engines per side exchanging only mailbox columns: device cluster == oracle cluster,
    every column of every node after every round; commit follows the head."""
    G, T = 2000, 60
    dc = DenseCluster(BatchedRaft, G, R, lead=lead)
    oc = DenseCluster(oracle_engine, G, R, lead=lead)
    rng = np.random.default_rng(R)
    for t in range(T):
        appends = rng.integers(0, 3, G).astype(np.uint64)
        da, oa = dc.round(appends), oc.round(appends)
        for r in range(R):
            _cmp_cols(da[r], oa[r], f"round {t} node {r}")
            compare_snapshots(dc.nodes[r], oc.nodes[r], f"round {t} node {r}")
            assert dc.rows[-1][r].tobytes() == oc.rows[-1][r].tobytes()
    L = oc.nodes[lead]
    assert (L.read("fault") == 0).all()
    assert (L.read("commit") + 12 >= L.read("head")).all() and int(L.read("commit").min()) > T // 2
    for r in range(R):
        if r != lead:
            assert (oc.nodes[r].read("commit") > T // 2).all()
    assert dc.nodes[lead].counters()["decisions"] == L.counters()["decisions"]


@pytest.mark.gpu
@pytest.mark.parametrize("R,lead", [(3, 0), (5, 2)])
def test_library_driven_rounds_equal_eager_rounds(R, lead):
    """jg_dense_cluster_rounds drives the protocol round from inside the library and replays it as a
    hipGraph (logical time and step numbers from a device-resident clock).  Same mailboxes, same state,
    same exceptional rows as the call-by-call loop (dense_node.DenseCluster) and as the oracle."""
    import ctypes as C
    G, rounds = 4000, 24
    rng = np.random.default_rng(R)
    appends = rng.integers(0, 3, G).astype(np.uint64)
    eager = DenseCluster(BatchedRaft, G, R, seed=5, lead=lead)
    ora = DenseCluster(oracle_engine, G, R, seed=5, lead=lead)
    nodes = [BatchedRaft(G, R, seed=5 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY)
             for r in range(R)]
    elect_all(nodes[lead])
    nodes[lead].drain_messages(), nodes[lead].drain_applies()
    api = nodes[0].api
    arr = (C.c_void_p * R)(*[n._h for n in nodes])
    cl = C.c_void_p()
    nodes[0]._check(api.dense_cluster_create(arr, R, lead, C.byref(cl)))
    nodes[0]._check(api.dense_cluster_set_appends(cl, 0, appends.ctypes.data))
    # 10 rounds in one call (captured + replayed), 1 round (eager), 13 more (replayed again)
    for first, n in ((100, 10), (1100, 1), (1200, 13)):
        nodes[0]._check(api.dense_cluster_rounds(cl, first, 100, n))
    for _ in range(rounds):
        eager.round(appends)
        ora.round(appends)
    for r in range(R):
        compare_snapshots(nodes[r], eager.nodes[r], f"library-driven vs eager, node {r}")
        compare_snapshots(nodes[r], ora.nodes[r], f"library-driven vs oracle, node {r}")
        for fn in ("drain_messages", "drain_faults"):
            got = getattr(nodes[r], fn)()
            want = np.concatenate([rows[r] for rows in ora.rows]) if fn == "drain_messages" else getattr(ora.nodes[r], fn)()
            assert got.tobytes() == want.tobytes(), (r, fn, len(got), len(want))
    assert int(nodes[lead].read("commit").min()) >= 0 and int(nodes[lead].read("head").max()) == int(appends.max()) * rounds
    api.dense_cluster_destroy(cl)


def test_routed_cluster_elections_on_the_oracle():
    """CPU: the configs[4] cluster trace through the Python statement of the transport.  What the
    reference's rules make of it: the restarted leader is a follower, the
    designated candidate is refused by every replica that still remembers a vote — the failing
    groups stay leaderless, nothing faults, nothing leaves the transport's vocabulary."""
    from dense_node import RoutedCluster, cluster_failure_rows
    G, R, T = 600, 5, 40
    cl = RoutedCluster(oracle_engine, G, R, seed=5)
    failed = np.zeros(G, bool)
    for t in range(T):
        inj = cluster_failure_rows(77, t, G, R, 2) if t >= 3 else None
        if inj and inj[0] is not None:
            failed[inj[0]["group"]] = True
        cl.round(np.ones(G, np.uint64), inject=inj)
    L = cl.nodes[0]
    assert failed.sum() > G // 3
    assert (L.read("role")[failed] == capi.ROLE_FOLLOWER).all() and (L.read("role")[~failed] == capi.ROLE_LEADER).all()
    assert (L.read("commit")[~failed] >= T - 3).all()
    for n in cl.nodes:
        assert (n.read("fault") == 0).all()
        assert (n.read("role")[failed] != capi.ROLE_LEADER).all()
    assert sum(len(k) for k in cl.kept) == 0 and cl.delivered.sum() > 0
    # the voters that were not restarted still hold their vote for the old leader: that is what refuses the candidate
    assert (cl.nodes[2].read("voted_for")[failed] == L.node_ids[0]).all()


def test_stationary_failure_repair_trace_on_the_oracle():
    """CPU: configs[4] with the repair schedule (josefine_amd.traces.FailureRepairTrace: bench.py --cluster --failures) through
    the Python statement of the transport over oracle engines.  A failing partition is leaderless until its repair - it is
    re-created: every replica restarts on an empty store, replica 0 is seated and its Heartbeat brings the others in - and is
    led by node 0 again from the repair's round on, appending like a partition at tick 0; the leaderless fraction settles at
    p x D; nothing faults; no row leaves the transport's vocabulary."""
    from dense_node import RoutedCluster
    from josefine_amd.traces import FailureRepairTrace
    G, R, T, P, D = 800, 5, 70, 2, 6
    cl = RoutedCluster(oracle_engine, G, R, seed=5)
    tr = FailureRepairTrace(77, G, R, P, D, node_ids=cl.member_ids)
    frac, up_for = [], np.zeros(G, np.int64)  # rounds a partition has been appending since tick 0 / its re-creation
    for t in range(T):
        inj, failing, repaired = tr.rows(t)
        cl.round(tr.appends(), inject=inj)
        down = tr.leaderless()
        role = cl.nodes[0].read("role")
        assert (role[down] != capi.ROLE_LEADER).all() and (role[~down] == capi.ROLE_LEADER).all(), t
        frac.append(float(down.mean()))
        up_for[repaired] = 0
        up_for[~down] += 1
        for n in cl.nodes:
            assert (n.read("fault") == 0).all(), t
    assert tr.ever_failed.sum() > G // 2
    assert abs(np.mean(frac[20:45]) - np.mean(frac[45:])) < 0.02 and 0.06 < np.mean(frac[20:]) < 0.16  # flat, about p x D
    assert sum(len(k) for k in cl.kept) == 0 and cl.delivered.sum() > 0
    # a partition that is up appends one block per round since tick 0 or since it was re-created, and commits them
    up = ~tr.leaderless()
    assert (cl.nodes[0].read("head")[up] == up_for[up]).all()
    settled = up & (up_for > 4)
    assert settled.sum() > G // 2 and (cl.nodes[0].read("commit")[settled] >= up_for[settled] - 3).all()
    rep = settled & tr.ever_failed  # ... re-created ones too: every replica follows the seated leader
    assert rep.sum() > G // 4 and (cl.nodes[2].read("voted_for")[rep] == cl.member_ids[0]).all() and (cl.nodes[3].read("head")[rep] >= up_for[rep] - 2).all()


@pytest.mark.gpu
@pytest.mark.parametrize("R,words", [(5, False), (5, True), (3, True)])
def test_stationary_failure_repair_trace_device_parity(R, words):
    """the same trace through jg_dense_cluster_round_routed (rows, and with JG_CLUSTER_OPT_VOTE_WORDS): every state column of
    every node after every round == the oracle cluster's; nothing left for the host; the client's proposals withdrawn and
    offered again on the device (jg_dense_cluster_withdraw_appends / _offer_appends)"""
    from josefine_amd import DenseCluster as LibCluster
    from dense_node import RoutedCluster
    from josefine_amd.traces import FailureRepairTrace
    G, T, P, D = 3000, 60, 2, 6
    ora = RoutedCluster(oracle_engine, G, R, seed=5)
    nodes = [BatchedRaft(G, R, seed=5 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY) for r in range(R)]
    elect_all(nodes[0])
    nodes[0].drain_messages(), nodes[0].drain_applies()
    lib = LibCluster(nodes, vote_words=words)
    lib.set_appends(1)
    tr = FailureRepairTrace(99, G, R, P, D, node_ids=ora.member_ids)
    for t in range(T):
        inj, failing, repaired = tr.rows(t)
        lists = [nodes[0].upload_u32(x) if len(x) else None for x in (failing, repaired)]
        if lists[0] is not None:
            lib.withdraw_appends(lists[0].ptr, len(failing))
        if lists[1] is not None:
            lib.offer_appends(lists[1].ptr, len(repaired), 1)
        up = [None if c is None else nodes[n].upload_rows(**c) for n, c in enumerate(inj)]
        st = lib.round_routed((t + 1) * 100, up)
        ora.round(tr.appends(), inject=inj)
        for n in range(R):
            compare_snapshots(nodes[n], ora.nodes[n], f"round {t} node {n}")
        want = [ora.pending(n) for n in range(R)]
        assert (st["delivered"] == want) if not words else all(a <= b for a, b in zip(st["delivered"], want)), (t, st["delivered"], want)
        for rows in up + lists:
            if rows is not None:
                rows.free()
    for n in range(R):
        assert nodes[n].drain_messages().tobytes() == ora.kept[n].tobytes()
        assert nodes[n].drain_faults().tobytes() == ora.nodes[n].drain_faults().tobytes()
        assert nodes[n].drain_applies().tobytes() == ora.nodes[n].drain_applies().tobytes()
        assert nodes[n].counters()["decisions"] == ora.nodes[n].counters()["decisions"]
    lib.close()


@pytest.mark.gpu
@pytest.mark.parametrize("R,percent,also", [(3, 3, ()), (5, 2, ()), (5, 2, (2,)), (3, 3, (2,))])
def test_routed_cluster_device_transport_parity(R, percent, also):
    """jg_dense_cluster_round_routed (rows routed between the nodes on the device) == the Python-routed
    oracle cluster: every state column of every node after every round, the number of rows delivered
    per node and round, and the rows left for the host.  `also`: replicas that restart with the leader, so
    that the candidate wins and a node other than the lead one leads those groups (its Heartbeats travel
    as rows through the transport, its AppendEntries stay queued for the host)."""
    from josefine_amd import DenseCluster as LibCluster
    from dense_node import RoutedCluster, cluster_failure_rows
    G, T = 3000, 50
    ora = RoutedCluster(oracle_engine, G, R, seed=5)
    nodes = [BatchedRaft(G, R, seed=5 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY)
             for r in range(R)]
    elect_all(nodes[0])
    nodes[0].drain_messages(), nodes[0].drain_applies()
    lib = LibCluster(nodes)
    lib.set_appends(1)
    for t in range(T):
        inj = cluster_failure_rows(99, t, G, R, percent, also=also) if t >= 3 else [None] * R
        if t == 20:  # something the transport must leave alone: a client request at every replica of node 2
            inj[2] = dict(kind=np.full(G, capi.CMD_CLIENT_REQUEST, np.uint8), group=np.arange(G, dtype=np.uint32),
                          id=np.arange(G, dtype=np.uint64) + 1000)
        if t == 25:  # ... and a step that queues FSM rows but keeps no message: a Heartbeat as a row at node 2 (its
            # HeartbeatResponse is delivered, the Apply range stays for jg_drain_applies - without the response)
            g = np.arange(G, dtype=np.uint32)
            inj[2] = dict(kind=np.full(G, capi.CMD_HEARTBEAT, np.uint8), group=g, from_=np.full(G, nodes[0].node_ids[0], np.uint32),
                          term=ora.nodes[0].read("term").astype(np.uint64), id=ora.nodes[0].read("commit").astype(np.uint64))
        up = [None if c is None else nodes[n].upload_rows(**c) for n, c in enumerate(inj)]
        st = lib.round_routed((t + 1) * 100, up)
        ora.round(np.ones(G, np.uint64), inject=inj)
        for n in range(R):
            compare_snapshots(nodes[n], ora.nodes[n], f"routed round {t} node {n}")
        assert st["delivered"] == [ora.pending(n) for n in range(R)], t
        if t == 25:
            assert st["fsm_rows"] > 0
        for rows in up:
            if rows is not None:
                rows.free()
    assert sum(ora.delivered) > 0 and sum(len(k) for k in ora.kept) >= G
    for n in range(R):
        got, want = nodes[n].drain_messages(), ora.kept[n]
        assert got.tobytes() == want.tobytes(), (n, len(got), len(want))
        assert nodes[n].drain_faults().tobytes() == ora.nodes[n].drain_faults().tobytes()
        assert nodes[n].drain_applies().tobytes() == ora.nodes[n].drain_applies().tobytes()
    lib.close()


def _mailbox_range_case(make):
    """A follower whose head reaches 2^56 - 1 cannot put it into an answer word: engine fault, no answer."""
    G, R = 4, 3
    e = make(G, R, seed=3, flags=capi.CFG_SEPARATE_COMMIT_KEY)
    top = capi.MAILBOX_NONE - 1
    e.apply(1, Command.AppendEntries(1, 2, [(top, 0)]))  # a block far up the id space, parent = genesis
    assert e.handle(1).head == top and e.handle(1).fault == 0
    e.drain_messages()
    z = np.zeros(G, np.uint64)
    ae_n = np.full(G, capi.AE_NONE, np.uint8)
    ae_n[1] = 1
    out = e.step_dense_follower(100, term=z + 1, hb_commit=np.full(G, capi.NO_ACK, np.uint64), ae_from=z + top, ae_n=ae_n,
                                leader_id=2, tick=False)
    return e, out


def test_mailbox_word_range_is_loud_on_the_oracle():
    e, out = _mailbox_range_case(oracle_engine)
    assert e.handle(1).fault == capi.FAULT_ENGINE_MAILBOX_RANGE and e.handle(0).fault == 0
    assert (out["ack_head"] == capi.NO_ACK).all()


@pytest.mark.gpu
def test_mailbox_word_range_parity():
    (dev, od), (ora, oo) = _mailbox_range_case(BatchedRaft), _mailbox_range_case(oracle_engine)
    compare_snapshots(dev, ora, "mailbox range")
    _cmp_cols(od, oo, "mailbox range")
    assert dev.handle(1).fault == capi.FAULT_ENGINE_MAILBOX_RANGE
    assert dev.drain_faults().tobytes() == ora.drain_faults().tobytes()


@pytest.mark.gpu
def test_routed_transport_repeats_with_wide_keys():
    """The transport first builds its sort keys with a 12-bit emission index and repeats the (non-mutating)
    delivering pass with the wide field when a group emitted more rows than that in one step.  With the field
    narrowed to one bit (test hook, read once per process: hence the subprocess) every election triggers the repeat."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_dense_node.py"), "-q", "-m", "gpu", "-k",
                        "test_routed_cluster_device_transport_parity"], capture_output=True, text=True, timeout=600,
                       cwd=root, env={**os.environ, "JG_ROUTE_NARROW_BITS": "1"})
    assert r.returncode == 0 and "4 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_routed_round_argument_checks():
    """Misuse is refused with JG_EINVAL, not executed: injected rows that carry blocks, a node with a
    drain in transfer."""
    from josefine_amd import DenseCluster as LibCluster, EngineError
    G, R = 64, 3
    nodes = [BatchedRaft(G, R, seed=5 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY)
             for r in range(R)]
    elect_all(nodes[0])
    nodes[0].drain_messages(), nodes[0].drain_applies()
    lib = LibCluster(nodes)
    lib.set_appends(1)
    lib.round_routed(100, None)
    rows = nodes[1].upload_rows(kind=np.array([capi.CMD_APPEND_ENTRIES], np.uint8), group=np.array([3], np.uint32),
                                aux=np.array([1], np.uint64), blk_id=np.array([7], np.uint64), blk_next=np.array([6], np.uint64))
    with pytest.raises(EngineError, match="cannot carry blocks"):
        lib.round_routed(200, [None, rows, None])
    rows.free()
    nodes[2].apply(0, Command.Tick())
    nodes[2].drain_prefetch()
    with pytest.raises(EngineError, match="drain is in transfer"):
        lib.round_routed(300, None)
    nodes[2].drain_flush()
    lib.round_routed(300, None)  # and the cluster carries on
    assert (nodes[0].read("fault") == 0).all()
    lib.close()


@pytest.mark.gpu
def test_cluster_rounds_before_set_appends_append_nothing():
    """A cluster that is run before jg_dense_cluster_set_appends was ever called appends nothing: the lead
    node's own-slot column starts as JG_ANSWER(0, JG_HB_NONE) (zero appends), not as JG_NO_ACK - which is
    outside the own slot's domain and would put JG_FAULT_ENGINE_DENSE_APPENDS on every group it leads."""
    from josefine_amd import DenseCluster as LibCluster
    G, R = 2000, 3
    nodes = [BatchedRaft(G, R, seed=3 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY)
             for r in range(R)]
    elect_all(nodes[0])
    nodes[0].drain_messages(), nodes[0].drain_applies()
    lib = LibCluster(nodes)
    lib.rounds(100, 100, 4)
    lib.round_routed(500)
    for n in nodes:
        assert (n.read("fault") == 0).all()
    assert (nodes[0].read("head") == 0).all() and (nodes[0].read("role") == capi.ROLE_LEADER).all()
    lib.set_appends(1)
    lib.rounds(600, 100, 4)
    assert (nodes[0].read("head") == 4).all() and (nodes[0].read("fault") == 0).all()
    lib.close()
