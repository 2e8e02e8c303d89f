"""Parity tests proper: the HIP engine, through the C ABI, against the CPU oracle
on the same seeded inputs — bit-exact on every state column and output row."""
import numpy as np
import pytest

from josefine_amd import BatchedRaft, Command, capi
from oracle_lib import oracle_engine
from parity import compare_drains, compare_snapshots, elect_all, run_dense_ticks
from fuzz import assert_live, random_batch, random_batch_aware

pytestmark = pytest.mark.gpu


def pair(G, R, **kw):
    return BatchedRaft(G, R, **kw), oracle_engine(G, R, **kw)


@pytest.mark.parametrize("R", [1, 2, 3, 4, 5, 7, 8])
def test_election_setup_parity(R):
    dev, ora = pair(1000, R, seed=R)
    for e in (dev, ora):
        elect_all(e)
    compare_snapshots(dev, ora, f"R={R} election")
    compare_drains(dev, ora, f"R={R} election")
    assert (dev.read("role") == capi.ROLE_LEADER).all()


@pytest.fixture(params=["slot0", "last", "mixed"])
def slot_layout(request):
    """Own-slot layouts: the dense kernel skips the (implicit, SELF-SYNC) own match column by a
    wave-uniform branch when every group has the same own slot, per lane otherwise."""
    def make(G, R):
        if request.param == "slot0":
            return None
        if request.param == "last":
            return np.full(G, R - 1, np.uint8)
        return (np.arange(G) % R).astype(np.uint8)
    return make


@pytest.mark.parametrize("R,G,mode,ticks", [(3, 10_000, 1, 200), (5, 10_000, 1, 60), (5, 4096, 0, 40),
                                            (1, 1000, 1, 20), (2, 1000, 1, 30), (4, 1000, 1, 30),
                                            (8, 1000, 1, 30), (3, 1001, 1, 20)])
def test_dense_ack_stream_parity(R, G, mode, ticks, slot_layout):
    """BASELINE config #2 shape (10k x 3 ragged stream) and friends: compare after every tick."""
    dev, ora = pair(G, R, seed=0x6A6F7365 + R, self_slots=slot_layout(G, R))
    for e in (dev, ora):
        elect_all(e)
    run_dense_ticks(dev, ora, mode=mode, ticks=ticks, check_every=1)
    assert int(dev.read("commit").max()) > 0


@pytest.mark.parametrize("R", [3, 5])
def test_survey_config_2_at_full_length(R):
    """SURVEY.md 8(d) config #2 AS WRITTEN: G = 10^4, the ragged stream (0-2 appends, 5 % of the acks dropped, 5 % duplicated / stale,
    follower heads advancing by U{0..MAX_INFLIGHT}), T = 1 000 ticks, commit / match / repl_state / fault (and head, id_gen, role, term,
    every drained row) compared after EVERY tick - and R = 5 beside the survey's R = 3.  (test_dense_ack_stream_parity runs 200 ticks
    of it in three own-slot layouts; the stream's rarer states - BEHIND escapes after 1 021 / 65 533 appends, the wide commit column - want
    the full length.)"""
    G, T = 10_000, 1000
    dev, ora = pair(G, R, seed=0x6A6F7365 + 2)
    for e in (dev, ora):
        elect_all(e)
    run_dense_ticks(dev, ora, mode=1, ticks=T, check_every=1)
    assert int(dev.read("commit").min()) > T // 4 and dev.counters()["dense_group_steps"] == G * T


@pytest.mark.parametrize("R", [1, 3, 5])
def test_fuzz_command_stream_parity(R):
    """Random commands of every kind to every role, incl. the panic / Err paths."""
    G = 512
    dev, ora = pair(G, R, seed=99, flags=capi.CFG_SEPARATE_COMMIT_KEY if R == 5 else 0)
    rng = np.random.default_rng(1234 + R)
    now = 0
    budget = np.full(G, capi.CHAIN_WINDOW - 2)  # gaps / forks per group: stay inside the segment window
    for step in range(60):
        batch = random_batch(rng, ora, 1500, budget=budget, foreign_voters=True)
        now += int(rng.integers(0, 400))
        for e in (dev, ora):
            e.submit_columns(**batch)
            e.step(now)
        compare_snapshots(dev, ora, f"fuzz R={R} step {step}")
        compare_drains(dev, ora, f"fuzz R={R} step {step}")
    assert dev.counters()["decisions"] == ora.counters()["decisions"]
    # the run must actually have exercised leaders, candidates and faults
    roles = ora.read("role")
    assert len(np.unique(roles)) >= 2
    assert (ora.read("fault") != 0).any()


@pytest.mark.parametrize("R,flags", [(1, 0), (3, capi.CFG_SEPARATE_COMMIT_KEY), (5, capi.CFG_SEPARATE_COMMIT_KEY), (3, 0), (5, 0)])
def test_fuzz_state_aware_stream_parity(R, flags):
    """The LIVE fuzz: the command kind is chosen from what the group is (leaders get acks, requests and Ticks,
    candidates get votes, followers their leader's traffic and election timeouts, faulted groups restarts),
    with a few percent of anything at all mixed in - the general state machine (k_apply_rows) under traffic
    that keeps most groups alive and a large share of them LED.  Asserted: >= 30 % of the commands reach
    un-faulted groups, >= 10 % of the groups are led on average (R >= 3), >= 0.1 decisions per command."""
    G = 512
    dev, ora = pair(G, R, seed=99, flags=flags, election_timeout_ms=(300, 700))
    rng = np.random.default_rng(4321 + R + flags)
    stats = {}
    now = 0
    for step in range(60):
        batch = random_batch_aware(rng, ora, 1500, stats)
        now += int(rng.integers(0, 200))
        for e in (dev, ora):
            e.submit_columns(**batch)
            e.step(now)
        compare_snapshots(dev, ora, f"aware fuzz R={R} step {step}")
        compare_drains(dev, ora, f"aware fuzz R={R} step {step}")
    assert dev.counters()["decisions"] == ora.counters()["decisions"]
    assert_live(stats, ora.counters()["decisions"], R)


def test_dense_equals_sparse_path():
    """The dense tick is specified as sugar for ClientRequest + AppendResponse rows."""
    G, R = 2000, 3
    dense = BatchedRaft(G, R, seed=5)
    sparse = BatchedRaft(G, R, seed=5)
    ora = oracle_engine(G, R, seed=5)
    for e in (dense, sparse, ora):
        elect_all(e)
        e.drain_messages(), e.drain_applies()
    from parity import synth_tick_host
    sim = np.zeros((R, G), dtype=np.uint64)
    head_prev = dense.read("head").copy()
    commit_prev = dense.read("commit").copy()
    for t in range(25):
        acks = synth_tick_host(ora, 1, t, sim)
        dense.step_dense_acks(acks)
        # the same tick as explicit rows: appends first, then acks in slot order
        rows_g, rows_k, rows_from, rows_id = [], [], [], []
        n_app = acks[0]
        for k in range(int(n_app.max())):
            gs = np.nonzero(n_app > k)[0]
            rows_g.append(gs); rows_k.append(np.full(len(gs), capi.CMD_CLIENT_REQUEST)); rows_from.append(np.zeros(len(gs))); rows_id.append(np.zeros(len(gs)))
        for r in range(1, R):
            gs = np.nonzero(acks[r] != capi.NO_ACK)[0]
            rows_g.append(gs); rows_k.append(np.full(len(gs), capi.CMD_APPEND_RESPONSE)); rows_from.append(np.full(len(gs), dense.node_ids[r])); rows_id.append(acks[r][gs])
        sparse.submit_columns(np.concatenate(rows_k), np.concatenate(rows_g), from_=np.concatenate(rows_from),
                              id=np.concatenate(rows_id).astype(np.uint64))
        sparse.step(0)
        compare_snapshots(dense, sparse, f"dense vs sparse tick {t}",
                          ["commit", "head", "match", "repl_state", "fault", "id_gen"])
        # FSM instructions of the sparse path == the head / commit deltas of the dense path
        fsm = sparse.drain_applies()
        head, commit = dense.read("head"), dense.read("commit")
        notif = fsm[fsm["kind"] == capi.FSM_NOTIFY]
        assert len(notif) == int((head - head_prev).sum())
        for g in np.unique(fsm["group"][fsm["kind"] == capi.FSM_APPLY_LEADER]):
            rows = fsm[(fsm["group"] == g) & (fsm["kind"] == capi.FSM_APPLY_LEADER)]
            assert rows["a"][0] == commit_prev[g] and rows["b"][-1] == commit[g]
            assert (rows["a"][1:] == rows["b"][:-1]).all()  # consecutive ranges concatenate
        adv = np.nonzero(commit != commit_prev)[0]
        assert set(adv) == set(np.unique(fsm["group"][fsm["kind"] == capi.FSM_APPLY_LEADER]))
        head_prev, commit_prev = head.copy(), commit.copy()


@pytest.mark.parametrize("R,T", [(3, 2), (5, 4), (5, 7), (1, 3), (8, 5)])
def test_dense_fused_ticks_parity(R, T):
    """jg_step_dense_acks_device_n: T ticks per launch == T single ticks (state after each launch)."""
    from parity import synth_tick_host
    G = 5000
    dev, ora = pair(G, R, seed=31 + R)
    for e in (dev, ora):
        elect_all(e)
        e.drain_messages(), e.drain_applies()
    sim = np.zeros((R, G), dtype=np.uint64)
    t = 0
    for launch in range(6):
        block = np.stack([synth_tick_host(ora, 1, t + k, sim) for k in range(T)])
        t += T
        dev.step_dense_acks_n(block)
        ora.step_dense_acks_n(block)
        compare_snapshots(dev, ora, f"fused R={R} T={T} launch {launch}",
                          ["commit", "head", "match", "repl_state", "fault", "id_gen", "role"])
    assert dev.counters()["decisions"] == ora.counters()["decisions"]
    assert dev.counters()["dense_group_steps"] == G * t


def _dense_edge_case_engines():
    """Groups 0: plain leader; 1: follower; 2: leader that will get forged acks; 3: leader whose
    chain is not in FAST form (restarted with commit 2: id_gen == head, Q8); 4: faulted leader."""
    G, R = 6, 3
    dev, ora = pair(G, R, seed=3)
    for e in (dev, ora):
        for g in (0, 2, 4, 5):
            e.submit(g, Command.Timeout())
            e.submit(g, Command.VoteResponse(1, 2, True))
        e.submit(4, Command.AppendEntries(9, 2, []))           # leader + higher term -> fault (Q3)
        e.submit(3, Command.AppendEntries(1, 2, [(1, 0), (2, 1)]))
        e.submit(3, Command.Heartbeat(1, 2, 2))                 # commit 2
        e.submit(3, Command.Restart())                          # head = id_gen = commit = 2
        e.submit(3, Command.Timeout())
        e.submit(3, Command.VoteResponse(1, 2, True))
        e.step()
        e.drain_messages(), e.drain_applies(), e.drain_faults()
    assert list(ora.read("role")) == [2, 0, 2, 2, 2, 2]
    return dev, ora, G, R


@pytest.mark.parametrize("fused", [False, True])
def test_dense_faults_followers_and_irregular_chains(fused):
    dev, ora, G, R = _dense_edge_case_engines()
    NO = capi.NO_ACK
    T = 4
    acks = np.full((T, R, G), NO, dtype=np.uint64)
    acks[:, 0, :] = 0
    # tick 0: everyone healthy appends once, followers ack nothing yet
    acks[0, 0, [0, 2, 5]] = 1
    acks[0, 1, 3] = 2                      # an ack to the irregular leader (no append): fine, q = 0.. no commit change
    # tick 1: acks arrive; group 2 gets a forged head from one follower (no quorum yet: no fault)
    acks[1, 1, [0, 5]] = 1
    acks[1, 1, 2] = 10**6
    acks[1, 0, 5] = 3                      # three appends in one tick
    # tick 2: the second forged ack makes q = 10^6 > head -> chain.commit panics (chain.rs:197-202);
    #         the follower group is asked to append -> engine precondition fault; the irregular
    #         leader appends -> assert!(id > head) fails (chain.rs:163)
    acks[2, 2, 2] = 10**6
    acks[2, 0, 1] = 1
    acks[2, 0, 3] = 1
    acks[2, 2, [0, 5]] = 1
    # tick 3: the dead groups ignore everything; the living go on
    acks[3, 0, :] = 1
    acks[3, 1, [0, 5]] = 2
    if fused:
        dev.step_dense_acks_n(acks)
        ora.step_dense_acks_n(acks)
        compare_snapshots(dev, ora, "dense edge cases fused")
        compare_drains(dev, ora, "dense edge cases fused")
    else:
        for t in range(T):
            dev.step_dense_acks(acks[t])
            ora.step_dense_acks(acks[t])
            compare_snapshots(dev, ora, f"dense edge cases tick {t}")
            compare_drains(dev, ora, f"dense edge cases tick {t}")
    assert list(ora.read("fault")) == [0, capi.FAULT_ENGINE_DENSE_NONLEADER, capi.FAULT_COMMIT_MISSING_BLOCK,
                                       capi.FAULT_APPEND_ID_NOT_ABOVE_HEAD, capi.FAULT_LEADER_TERM_UNIMPLEMENTED, 0]
    assert int(ora.read("head")[5]) == 5 and int(ora.read("commit")[5]) == 2


@pytest.mark.parametrize("fused", [False, True])
def test_dense_own_match_head_not_in_sync(fused):
    """The leader's own progress head is implicit only while it equals the chain head.  An
    AppendResponse that names the leader's own NodeId (the reference does not check the sender,
    leader.rs:211-219) can push it above the head: the dense kernel must then go back to the
    stored column — and to the exact per-ack replay, since q may exceed the head."""
    G, R = 512, 3
    dev, ora = pair(G, R, seed=11)
    for e in (dev, ora):
        elect_all(e)
        e.drain_messages(), e.drain_applies()
    g = np.arange(0, G, 3, dtype=np.uint32)
    for e in (dev, ora):  # own NodeId is node_ids[0] == 1
        e.submit_columns(np.full(len(g), capi.CMD_APPEND_RESPONSE, np.uint8), g, from_=np.ones(len(g), np.uint32),
                         id=np.full(len(g), 7, np.uint64), flag=np.ones(len(g), np.uint8))
        e.step()
    compare_snapshots(dev, ora, "forged own ack")
    assert (ora.read("match", 0)[g] == 7).all()
    T = 12
    acks = np.full((T, R, G), capi.NO_ACK, dtype=np.uint64)
    acks[:, 0, :] = 1
    for t in range(1, T):
        acks[t, 1, :] = t      # follower 1 acks the previous head
        acks[t, 2, ::2] = t
    if fused:
        for t0 in range(0, T, 4):
            dev.step_dense_acks_n(acks[t0:t0 + 4])
            ora.step_dense_acks_n(acks[t0:t0 + 4])
            compare_snapshots(dev, ora, f"own head out of sync, ticks {t0}..")
            compare_drains(dev, ora, f"own head out of sync, ticks {t0}..")
    else:
        for t in range(T):
            dev.step_dense_acks(acks[t])
            ora.step_dense_acks(acks[t])
            compare_snapshots(dev, ora, f"own head out of sync, tick {t}")
            compare_drains(dev, ora, f"own head out of sync, tick {t}")
    # the forged head is overtaken at tick 8: from then on the own head tracks the chain head again
    assert (ora.read("match", 0) == ora.read("head")).all() and int(ora.read("head")[0]) == T


@pytest.mark.parametrize("R,per_tick", [(8, 10), (5, 150), (7, 20)])
def test_progress_lag_escape(R, per_tick):
    """A leader's progress heads and commit index are stored as lags below the chain head,
    64/(R+1) bits each; a value further behind than the field holds (126 blocks at R = 8, 254 at
    R = 7, 1022 at R = 5) switches to its wide column and comes back when it catches up —
    invisible in every result."""
    G = 256
    dev, ora = pair(G, R, seed=17)
    for e in (dev, ora):
        elect_all(e)
        e.drain_messages(), e.drain_applies()
    T = 30
    for t in range(T):
        acks = np.full((R, G), capi.NO_ACK, dtype=np.uint64)
        acks[0, :] = per_tick
        head_before = int(ora.read("head")[0])
        if R > 1:
            acks[1, ::2] = head_before            # slot 1 of the even groups keeps up ...
            if t == 20:
                acks[1, 1::2] = head_before       # ... of the odd groups only acks once, late
        if R > 2 and t % 7 == 3:
            acks[2, :] = max(head_before - 2 * per_tick, 0)   # slot 2 hovers around the field limit
        dev.step_dense_acks(acks)
        ora.step_dense_acks(acks)
        compare_snapshots(dev, ora, f"lag escape R={R} tick {t}")
    # through the general state machine as well (reads and rewrites the packed word)
    for e in (dev, ora):
        g = np.arange(G, dtype=np.uint32)
        e.submit_columns(np.full(G, capi.CMD_APPEND_RESPONSE, np.uint8), g, from_=np.full(G, e.node_ids[R - 1], np.uint32),
                         id=np.full(G, 5, np.uint64), flag=np.ones(G, np.uint8))
        e.submit_columns(np.full(G, capi.CMD_CLIENT_REQUEST, np.uint8), g)
        e.submit_columns(np.full(G, capi.CMD_TICK, np.uint8), g)
        e.step(10_000)
    compare_snapshots(dev, ora, f"lag escape R={R} sparse")
    compare_drains(dev, ora, f"lag escape R={R} sparse")
    assert int(ora.read("head")[0]) == T * per_tick + 1


def test_chain_window_overflow_is_loud():
    """More gaps / forks than JG_CHAIN_WINDOW segments: an engine fault, never a silent miss."""
    dev = BatchedRaft(2, 3)
    blocks = [(10 * i, 0) for i in range(1, capi.CHAIN_WINDOW + 2)]  # every block its own segment
    dev.apply(0, Command.AppendEntries(0, 2, blocks))
    assert dev.handle(0).fault == capi.FAULT_ENGINE_WINDOW_OVERFLOW
    assert [tuple(r) for r in dev.drain_faults()] == [(0, capi.FAULT_ENGINE_WINDOW_OVERFLOW)]
    assert dev.handle(1).fault == 0
    # a voter outside the membership is no fault: Election::vote counts whoever answers (election.rs:33-35)
    dev2, ora2 = pair(1, 3)
    for e in (dev2, ora2):
        e.apply(0, Command.Timeout())
        e.apply(0, Command.VoteResponse(1, 77, True))  # not a member
    assert dev2.handle(0).fault == 0
    compare_snapshots(dev2, ora2, "foreign voter")


@pytest.mark.parametrize("n_trees", [1, 300])
def test_chain_compact_parity(n_trees):
    """Random forests through Chain::compact: device walk vs oracle walk, bit for bit."""
    rng = np.random.default_rng(77 + n_trees)
    dev, ora = pair(1, 1)
    trees = []
    for _ in range(n_trees):
        n = int(rng.integers(0, 40))
        ids, blocks = [0], [(0, 0)]
        nxt = 1
        for _ in range(n):
            nxt += int(rng.integers(1, 3))                 # gaps in the id space
            parent = ids[-1] if rng.random() < 0.7 else int(rng.choice(ids))
            blocks.append((nxt, parent))
            ids.append(nxt)
        if n and rng.random() < 0.2:                        # an overwritten block (sled upsert: last wins)
            blocks.append((ids[len(ids) // 2], ids[0]))
        order = rng.permutation(len(blocks)) if rng.random() < 0.5 else np.arange(len(blocks))
        if len(blocks) > len(ids):                          # keep the overwrite after its original
            order = np.arange(len(blocks))
        commit = int(rng.choice(ids)) if rng.random() < 0.8 else nxt + 5
        trees.append(([blocks[i] for i in order], commit))
    a, b = dev.chain_compact(trees), ora.chain_compact(trees)
    for i, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x, y), (i, trees[i], x, y)
    assert sum(int(x.sum()) for x in a) > 0 or n_trees == 1


def test_device_resident_rows_path():
    """jg_step_device_rows (group-sorted batch already in HBM) == jg_submit + jg_step."""
    G, R = 3000, 3
    dev, ora = pair(G, R, seed=8)
    rng = np.random.default_rng(3)
    budget = np.full(G, capi.CHAIN_WINDOW - 2)
    for step in range(12):
        batch = random_batch(rng, ora, 4000, budget=budget)
        rows = dev.upload_rows(**batch)
        dev.step_device_rows(rows, now_ms=50 * step)
        ora.submit_columns(**batch)
        ora.step(50 * step)
        compare_snapshots(dev, ora, f"device rows step {step}")
        compare_drains(dev, ora, f"device rows step {step}")
        rows.free()


def test_device_rows_must_be_sorted():
    from josefine_amd import EngineError
    dev = BatchedRaft(8, 3)
    rows = dev.upload_rows([capi.CMD_NOOP] * 3, [1, 2, 3])
    # corrupt the order on the device: write group column [3, 2, 1]
    bad = np.array([3, 2, 1], dtype=np.uint32)
    dev._check(dev.api.device_upload(dev._h, rows.batch.group, bad.ctypes.data, bad.nbytes))
    dev.step_device_rows(rows)
    with pytest.raises(EngineError):
        dev.read("term")


def test_device_rows_block_range_is_checked_on_the_device():
    """A device-resident batch is not validated by the host.  An AppendEntries row whose (id, aux)
    leaves the block side arrays must not read past them: the row is not applied, and the next
    synchronising call reports JG_EINVAL - where the reference would have returned Err for a block it
    cannot read (chain.rs:180-185); host rows are rejected by jg_submit up front."""
    from josefine_amd import EngineError
    dev = BatchedRaft(8, 3)
    blocks = [(1, 0), (2, 1)]
    ok = dev.upload_rows([capi.CMD_APPEND_ENTRIES], [2], from_=[2], term=[1], id=[0], aux=[2],
                         blk_id=[b[0] for b in blocks], blk_next=[b[1] for b in blocks])
    dev.step_device_rows(ok)
    assert int(dev.read("head")[2]) == 2
    for forged_id, forged_aux in ((1, 2), (0, 3), (2**63, 2), (0, 2**64 - 1)):
        dev2 = BatchedRaft(8, 3)
        rows = dev2.upload_rows([capi.CMD_APPEND_ENTRIES, capi.CMD_TIMEOUT], [2, 5], from_=[2, 0], term=[1, 0], id=[forged_id, 0],
                                aux=[forged_aux, 0], blk_id=[1, 2], blk_next=[0, 1])
        dev2.step_device_rows(rows)
        with pytest.raises(EngineError, match="block range"):
            dev2.read("head")
    # no side arrays at all: only an empty AppendEntries is legal
    dev3 = BatchedRaft(8, 3)
    rows = dev3.upload_rows([capi.CMD_APPEND_ENTRIES], [1], from_=[2], term=[4], id=[0], aux=[0])
    dev3.step_device_rows(rows)
    assert int(dev3.read("term")[1]) == 4
    rows = dev3.upload_rows([capi.CMD_APPEND_ENTRIES], [1], from_=[2], term=[4], id=[0], aux=[1])
    dev3.step_device_rows(rows)
    with pytest.raises(EngineError, match="block range"):
        dev3._check(dev3.api.sync(dev3._h))


def test_abi_contract_on_device():
    """Status codes of the C ABI on the real library: capacity, ranges, validation, ordering."""
    import ctypes as C
    from josefine_amd import EngineError
    dev = BatchedRaft(4, 3)
    dev.apply(0, Command.Timeout())
    n = C.c_size_t(0)
    assert dev.api.drain_messages(dev._h, None, 0, C.byref(n)) == capi.OK and n.value == 2
    buf = np.zeros(1, dtype=capi.MSG_DTYPE)
    assert dev.api.drain_messages(dev._h, buf.ctypes.data, 1, C.byref(n)) == capi.ECAPACITY
    assert len(dev.drain_messages()) == 2
    out = np.zeros(8, np.uint64)
    assert dev.api.read_state(dev._h, capi.FIELD_TERM, 0, out.ctypes.data, 2, 3) == capi.EINVAL
    assert dev.api.read_state(dev._h, capi.FIELD_MATCH, 3, out.ctypes.data, 0, 1) == capi.EINVAL
    assert dev.api.read_state(dev._h, 99, 0, out.ctypes.data, 0, 1) == capi.EINVAL
    with pytest.raises(EngineError):
        dev.submit_columns([capi.CMD_TICK], [4])
    with pytest.raises(EngineError):
        dev.submit_columns([capi.CMD_APPEND_ENTRIES], [0], id=[0], aux=[2], blk_id=[1], blk_next=[0])
    with pytest.raises(EngineError):  # self slots are fixed after the first step
        dev._check(dev.api.set_self_slots(dev._h, np.zeros(4, np.uint8).ctypes.data))
    # queued commands must be stepped before a dense tick
    dev.submit_columns([capi.CMD_NOOP], [1])
    assert dev.api.step_dense_acks(dev._h, np.zeros((3, 4), np.uint64).ctypes.data) == capi.EINVAL
    dev.step()
    assert dev.api.abi_version() == capi.ABI_VERSION


def test_two_engines_are_independent():
    a, b = BatchedRaft(16, 3, seed=1), BatchedRaft(16, 3, seed=2)
    elect_all(a)
    assert (a.read("role") == capi.ROLE_LEADER).all() and (b.read("role") == capi.ROLE_FOLLOWER).all()
    assert not np.array_equal(a.read("election_timeout"), b.read("election_timeout"))
    a.close()
    b.apply(3, Command.Timeout())
    assert b.handle(3).is_candidate()


@pytest.mark.parametrize("R", [2, 3, 4, 5, 6, 8])
def test_dense_lag_space_boundaries(R, slot_layout):
    """The dense tick is evaluated in lag space (32-bit lags below the chain head in the packed
    progress word) wherever that is exact, and on the general path otherwise.  Random ack blocks
    aimed at the border between the two: acks exactly at / one above the head, forged acks far
    above it (-> replay, chain.commit panic), lags at ESC-1 / ESC / ESC+1 of the field, bursts
    of appends that push every lag out of its field, dropped and duplicated acks — every
    state column, decision counter and fault row against the oracle after every tick, in both
    the one-tick and the T-tick kernel."""
    G = 2048
    dev, ora = pair(G, R, seed=23 + R, self_slots=slot_layout(G, R))
    for e in (dev, ora):
        elect_all(e)
        e.drain_messages(), e.drain_applies()
    rng = np.random.default_rng(100 + R)
    esc = (1 << (64 // (R + 1))) - 1
    slots = ora.read("self_slot").astype(np.int64)
    gi = np.arange(G)
    NO = np.uint64(capi.NO_ACK)
    T = 24
    for t in range(T):
        head = ora.read("head").astype(np.uint64)
        acks = np.full((R, G), NO, dtype=np.uint64)
        n_app = rng.integers(0, 3, G).astype(np.uint64)
        n_app = np.where(rng.random(G) < 0.02, esc + 3 if esc < 70_000 else 5000, n_app)  # burst: every lag leaves its field
        if t == 9:   # outside the own slot's domain (JG_MAX_DENSE_APPENDS): an engine fault, nothing applied
            n_app = np.where(gi % 256 == 1, (1 << 20) + 1, n_app)
            n_app = np.where(gi % 256 == 2, 1 << 20, n_app)
            n_app = np.where(gi % 256 == 3, NO, n_app)
        if t == 10:  # the largest legal value (a million appends in one tick: once, on few groups)
            n_app = np.where(gi % 1024 == 5, (1 << 20) - 1, n_app)
        for r in range(R):
            kind = rng.integers(0, 14, G)
            lag = np.zeros(G, dtype=np.uint64)
            lag = np.where(kind == 1, 1, lag)
            lag = np.where(kind == 2, rng.integers(0, 7, G), lag)
            lag = np.where(kind == 3, esc - 1, lag)
            lag = np.where(kind == 4, esc, lag)
            lag = np.where(kind == 5, esc + 1, lag)
            lag = np.where(kind == 6, 2 * esc, lag)
            lag = np.where(kind == 7, head, lag).astype(np.uint64)                # an ack of block 0
            a = np.where(lag <= head, head - np.minimum(lag, head), 0).astype(np.uint64)
            a = np.where(kind == 8, head + np.uint64(1), a)                      # one above the head
            a = np.where((kind == 9) & (rng.random(G) < 0.05), head + np.uint64(10**9), a)  # forged
            a = np.where((kind == 10) | (kind == 11), NO, a)                     # dropped
            small = n_app < np.uint64(1 << 21)
            a = np.where((kind == 12) & small, head + n_app, a)                  # exactly the head the ack meets (appends first)
            a = np.where((kind == 13) & small, head + n_app + np.uint64(1), a)   # one above it -> replay
            acks[r] = a
        acks[slots, gi] = n_app
        if t % 6 == 5:  # the same three ticks through the T-tick kernel
            blk = np.stack([acks, acks, acks])
            dev.step_dense_acks_n(blk)
            ora.step_dense_acks_n(blk)
        else:
            dev.step_dense_acks(acks)
            ora.step_dense_acks(acks)
        compare_snapshots(dev, ora, f"lag-space boundaries R={R} tick {t}")
        compare_drains(dev, ora, f"lag-space boundaries R={R} tick {t}")
        assert dev.counters()["decisions"] == ora.counters()["decisions"]
    faults = np.bincount(ora.read("fault"), minlength=256)
    assert (R < 3 or faults[capi.FAULT_COMMIT_MISSING_BLOCK] > 0) and faults[0] > G // 4, faults[:8]
    assert faults[capi.FAULT_ENGINE_DENSE_APPENDS] > 0


@pytest.mark.parametrize("R", [3, 5])
def test_fused_ticks_nonleader_append_at_every_tick(R):
    """A follower asked to append (an engine precondition fault, JG_FAULT_ENGINE_DENSE_NONLEADER) in
    tick t of a T-tick launch, for every (T, t): the ack blocks of the later ticks reach the lane
    through the two-deep prefetch ring of k_leader_tick_dense_n."""
    G = 64
    for T in (1, 2, 3, 4, 7):
        for tick in range(T):
            dev, ora = pair(G, R, seed=31)
            acks = np.full((T, R, G), capi.NO_ACK, dtype=np.uint64)
            acks[:, 0, :] = 0
            acks[tick, 0, 3::5] = 1
            for e in (dev, ora):
                e.step_dense_acks_n(acks) if T > 1 else e.step_dense_acks(acks[0])
            compare_snapshots(dev, ora, f"non-leader append T={T} tick={tick}")
            compare_drains(dev, ora, f"non-leader append T={T} tick={tick}")
            assert (ora.read("fault")[3::5] == capi.FAULT_ENGINE_DENSE_NONLEADER).all()


@pytest.mark.parametrize("R,entry", [(5, "acks"), (5, "acks_n"), (5, "leader"), (8, "acks"), (8, "leader"), (4, "acks_n")])
def test_follower_down_stays_on_the_fast_path_and_exact(R, entry):
    """The common failure: one follower is down.  Its progress head falls behind the chain head by
    more than a lag field holds (BEHIND escape: the absolute head moves to the wide column once) and
    stays there; the dense kernels keep serving such groups in lag space.  Later a second and third
    follower go quiet too (quorum lost: the commit index falls BEHIND as well), then everybody comes
    back (acks for BEHIND slots: exact compare on the general path, fields return into range).
    State, decisions and (leader entry point) the Tick's outbox against the oracle all the way."""
    G = 256
    kw = dict(seed=41, flags=capi.CFG_SEPARATE_COMMIT_KEY)
    dev, ora = pair(G, R, **kw)
    for e in (dev, ora):
        elect_all(e)
        e.drain_messages(), e.drain_applies()
    esc = (1 << (64 // (R + 1))) - 1
    per = max(1, esc // 60)                       # appends per tick: ~70 ticks to leave a field
    T_down, T_quorum, T_back = 80, 160, 240
    rng = np.random.default_rng(7)
    now = 0
    t = 0
    while t < T_back + 6:
        n_t = 3 if entry == "acks_n" else 1
        blk = np.full((n_t, R, G), capi.NO_ACK, dtype=np.uint64)
        head = ora.read("head").astype(np.uint64)
        for k in range(n_t):
            blk[k, 0, :] = per + (rng.integers(0, 2, G) if t % 5 == 0 else 0)
            up = [r for r in range(1, R)]
            if t < T_back:
                up = [r for r in up if r != 1]                        # slot 1 is down from the start
            if T_quorum <= t < T_back:
                up = [r for r in up if r > R // 2 + 1]                # ... then more than a minority
            for r in up:
                blk[k, r, :] = head + np.uint64(per * k)              # acks the head it was sent
            if t >= T_back:
                blk[k, 1, ::2] = head[::2] // 2                       # the returning follower catches up in steps
        if entry == "leader":
            now += 100
            hbr_has = np.full((R, G), capi.HB_NONE, np.uint8)
            oa = dev.step_dense_leader(now, blk[0], hbr_has, np.zeros((R, G), np.uint64), tick=True)
            ob = ora.step_dense_leader(now, blk[0], hbr_has, np.zeros((R, G), np.uint64), tick=True)
            for kk in oa:
                assert np.array_equal(oa[kk], ob[kk]), f"tick {t}: outbox {kk}"
        elif entry == "acks_n":
            dev.step_dense_acks_n(blk)
            ora.step_dense_acks_n(blk)
        else:
            dev.step_dense_acks(blk[0])
            ora.step_dense_acks(blk[0])
        t += n_t
        if t % 16 < n_t or t in (T_down, T_quorum, T_back) or t > T_back - 4:
            compare_snapshots(dev, ora, f"follower down R={R} {entry} tick {t}")
            compare_drains(dev, ora, f"follower down R={R} {entry} tick {t}")
    assert dev.counters()["decisions"] == ora.counters()["decisions"]
    h, m1 = ora.read("head"), ora.read("match", 1)
    assert (h > 2 * esc).all() and not ora.read("fault").any() and (m1 > 0).all()


@pytest.mark.parametrize("which,tile", [("fuzz", "1"), ("fuzz", "small"),
                                        ("dense_equals_sparse or election_setup or device_r or chain_window or follower_down", "1")])
def test_run_per_lane_state_machine_kernel_holds_the_same_parity(which, tile):
    """jg_apply_runs_body - a RUN of a group's rows per lane, tiles of 1024 rows staged in LDS (256 for small batches:
    JG_APPLY_RUNS=small): what the cluster transport's delivered batches take - behind jg_step for every batch
    (JG_APPLY_RUNS, read once per process: hence the subprocess): the blind and the state-aware fuzz streams (runs of
    every length, runs that cross a tile's end, tiles inside one run) and the election / dense-equals-sparse suites
    against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", f"({which}) and not run_per_lane"],
                       capture_output=True, text=True, timeout=900, cwd=root, env={**os.environ, "JG_APPLY_RUNS": tile})
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
