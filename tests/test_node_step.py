"""jg_step_node — a node's whole tick from host rows, the dense kernels behind the Apply surface
(server::event_loop for many partitions: src/raft/server.rs:103-165).

Two links of one chain:
  1. (CPU) the oracle's restatement of the entry point (jo_step_node: every row through Raft::apply in
     arrival order, then the Tick; the classification only decides what is reported as columns) is
     indistinguishable from PLAIN jg_submit + jg_step over the same rows IN THE ORDER GIVEN - state,
     faults, and per partition the exact sequence of fsm rows (run-length decoded) and of messages,
     columns and rows alike; the plain path also run on tests/ref_py;
  2. (GPU) the HIP engine's jg_step_node is bit-identical to the oracle's: every state column, every
     drained row, every outbox word, tick after tick, on mixed leader / follower / candidate
     populations with traffic that exercises every reason to leave the column path - and, directly,
     to its OWN plain path (jg_submit + jg_step on a second device engine) in the order given.
"""
import numpy as np
import pytest

from josefine_amd import BatchedRaft, EngineError, capi
from node_step import (classify, columns_as_rows, compare_outboxes, elect_some, msgs_per_partition, node_traffic, plain_apply_equivalent,
                       rows_to_columns)
from oracle_lib import oracle_engine
from parity import compare_drains, compare_snapshots


def msg_tuples(rows, drop_from=True):
    out = []
    for r in rows:
        term = None if int(r["kind"]) == capi.CMD_APPEND_RESPONSE else int(r["term"])
        out.append((int(r["group"]), int(r["kind"]), int(r["to_kind"]), int(r["to_id"]), int(r["flag"]), term, int(r["id"]), int(r["aux"])))
    return out


def coalesce_fsm(rows):
    """[a,b] + [b,c] of one partition and kind -> [a,c] (what jg_step_node's run-length encoding does)."""
    out = []
    for r in rows:
        t = (int(r["group"]), int(r["kind"]), int(r["a"]), int(r["b"]))
        if out and t[1] != capi.FSM_NOTIFY and out[-1][0] == t[0] and out[-1][1] == t[1] and out[-1][3] == t[2]:
            out[-1] = (t[0], t[1], out[-1][2], t[3])
        else:
            out.append(t)
    return out


def mixed_pair(make_a, make_b, G, R, seed, lead_frac=0.6, **kw):
    a, b = make_a(G, R, seed=seed, **kw), make_b(G, R, seed=seed, **kw)
    rng = np.random.default_rng(seed)
    mask = rng.random(G) < lead_frac
    for e in (a, b):
        elect_some(e, mask)
        e.drain_messages(), e.drain_applies(), e.drain_faults()
    return a, b, rng


def ref_py_engine(*a, **kw):
    from ref_py.engine import RefEngine
    return RefEngine(*a, **kw)


def fsm_tuples(rows):
    return [tuple(int(x) for x in (r["group"], r["kind"], r["a"], r["b"])) for r in rows]


def expected_node_fsm(plain_rows, general):
    """The fsm rows jg_step_node queues for one step, from the plain path's: the general partitions' rows as they
    are (first: that launch comes first), then the column-form partitions' run-length encoded, partitions ascending."""
    t = fsm_tuples(plain_rows)
    gen = [x for x in t if general[x[0]]]
    per = {}
    for x in t:
        if not general[x[0]]:
            per.setdefault(x[0], []).append(x)
    return gen + [y for g in sorted(per) for y in coalesce_fsm_tuples(per[g])]


def node_messages_per_partition(node_rows, out, ids, slots, leader_of):
    """rpc_tx of one node step per partition, in emission order (include/josefine_gpu.h, jg_step_node "Outputs"): where
    an answer word is set - the queue the Heartbeat flushed (ClientRequest rows, follower.rs:190-197), the
    HeartbeatResponse, the AppendResponse, then whatever the Tick sent; a leader's Tick columns come after its rows."""
    rows = msgs_per_partition(msg_tuples(node_rows))
    answers = msgs_per_partition(columns_as_rows({"answer": out.get("answer"), "hb_commit": out.get("hb_commit")}, ids, slots, leader_of))
    ticks = msgs_per_partition(columns_as_rows({"beat_term": out.get("beat_term"), "beat_commit": out.get("beat_commit"), "ae": out.get("ae")},
                                               ids, slots, leader_of))
    out = {}
    for g in set(rows) | set(answers) | set(ticks):
        mine = rows.get(g, [])
        flushed = [r for r in mine if g in answers and r[1] == capi.CMD_CLIENT_REQUEST]
        out[g] = flushed + answers.get(g, []) + [r for r in mine if r not in flushed] + ticks.get(g, [])
    return out


def leader_of_rows(cols, general, G):
    leader_of = np.zeros(G, np.int64)
    for i in range(len(cols["kind"])):
        if cols["kind"][i] in (capi.CMD_HEARTBEAT, capi.CMD_APPEND_ENTRIES) and not general[cols["group"][i]]:
            leader_of[cols["group"][i]] = cols["from_"][i]
    return leader_of


@pytest.mark.parametrize("plain_backend", ["oracle", "ref_py"])
@pytest.mark.parametrize("R,flags", [(3, 0), (5, 0), (5, capi.CFG_SEPARATE_COMMIT_KEY), (4, 0), (1, 0)])
def test_oracle_node_step_is_plain_apply_in_arrival_order(R, flags, plain_backend):
    """jo_step_node == every row through plain Apply::apply IN THE ORDER GIVEN, then the Ticks (server.rs:120-161):
    state, faults, per partition the exact sequence of fsm_tx rows and of rpc_tx messages (columns and rows alike).
    With plain_backend = ref_py the plain path is the independent Python reading of the Rust (tests/ref_py):
    jg_step_node's semantics are then held to a restatement that shares no code with the oracle."""
    G, T = (400, 40) if plain_backend == "oracle" else (160, 25)
    arrival_order_differential(oracle_engine, oracle_engine if plain_backend == "oracle" else ref_py_engine, G, R, T, flags, seed=11 + R)


@pytest.mark.gpu
@pytest.mark.parametrize("R,flags,G", [(3, 0, 2500), (5, capi.CFG_SEPARATE_COMMIT_KEY, 2500), (5, 0, 600), (2, 0, 300), (1, 0, 200)])
def test_node_step_is_plain_apply_in_arrival_order_on_the_device(R, flags, G):
    """The round-3 review's differential, on the device alone: jg_step_node on one engine, jg_submit + jg_step over the
    same rows in the order given (+ a Tick row per partition) on another - no oracle in between."""
    arrival_order_differential(BatchedRaft, BatchedRaft, G, R, 40, flags, seed=51 + R)


def arrival_order_differential(make_node, make_plain, G, R, T, flags, seed):
    node, plain, rng = mixed_pair(make_node, make_plain, G, R, seed=seed, flags=flags, election_timeout_ms=(700, 1500))
    ids = np.array(node.node_ids)
    dense_rows = general_rows = 0
    for t in range(T):
        now = 100 * (t + 1)
        cols = node_traffic(rng, node, token0=1000 * t)
        slots = node.read("self_slot")
        node.submit_columns(**cols)
        out = node.step_node(now)
        general = plain_apply_equivalent(plain, cols, now)
        assert out["rows"] == len(cols["kind"]) and out["rows_general"] == int(general[cols["group"]].sum())
        dense_rows += out["rows"] - out["rows_general"]
        general_rows += out["rows_general"]
        compare_snapshots(node, plain, f"tick {t}")
        fa, fb = node.drain_faults(), plain.drain_faults()
        assert sorted(map(tuple, fa.tolist())) == sorted(map(tuple, fb.tolist())), t
        assert fsm_tuples(node.drain_applies()) == expected_node_fsm(plain.plain_fsm, general), t
        # messages: per partition, the node step's rows and what its columns stand for == the plain path's rows, in order
        got = node_messages_per_partition(node.drain_messages(), out, ids, slots, leader_of_rows(cols, general, G))
        want = msgs_per_partition(msg_tuples(plain.drain_messages()))
        assert got == want, (t, [(g, got.get(g), want.get(g)) for g in set(got) | set(want) if got.get(g) != want.get(g)][:3])
    assert dense_rows > 5 * general_rows > 0  # the traffic is mostly steady state, and the general path was exercised


def one_led_partition(make, R=3, head=0, **kw):
    """One partition led by this node, `head` blocks appended and committed by everybody."""
    e = make(1, R, seed=1, flags=capi.CFG_SEPARATE_COMMIT_KEY, election_timeout_ms=(700, 1500), **kw)
    elect_some(e, np.array([True]))
    ids = e.node_ids
    for h in range(1, head + 1):
        rows = [(capi.CMD_CLIENT_REQUEST, 0, 0, 0, h, 0, 0, None)] + [(capi.CMD_APPEND_RESPONSE, 0, ids[r], 1, h, 0, 1, None) for r in range(1, R)]
        e.submit_columns(**rows_to_columns(rows))
        e.step(10 * h)
    e.drain_messages(), e.drain_applies(), e.drain_faults()
    return e


ARRIVAL_KATS = {
    # round-3 review: acknowledgements of a block the leader has not appended yet, BEFORE the ClientRequest:
    # the second one completes a majority for head + 1 while the chain ends at head - chain.rs:197-202 panics
    "acks_above_the_head_before_the_append": (0, lambda ids: [(capi.CMD_APPEND_RESPONSE, 0, ids[1], 1, 1, 0, 1, None),
                                                               (capi.CMD_APPEND_RESPONSE, 0, ids[2], 1, 1, 0, 1, None),
                                                               (capi.CMD_CLIENT_REQUEST, 0, 0, 0, 77, 0, 0, None)]),
    # ... the same acknowledgements AFTER it are legitimate: commit 1
    "acks_of_the_new_head_after_the_append": (0, lambda ids: [(capi.CMD_CLIENT_REQUEST, 0, 0, 0, 77, 0, 0, None),
                                                               (capi.CMD_APPEND_RESPONSE, 0, ids[1], 1, 1, 0, 1, None),
                                                               (capi.CMD_APPEND_RESPONSE, 0, ids[2], 1, 1, 0, 1, None)]),
    # legitimate traffic, ack first: fsm_tx is Apply(0..=1) THEN Notify(2) (head 1 / commit 0 before the step)
    "ack_then_request": (-1, lambda ids: [(capi.CMD_APPEND_RESPONSE, 0, ids[1], 1, 1, 0, 1, None), (capi.CMD_CLIENT_REQUEST, 0, 0, 0, 78, 0, 0, None)]),
    "request_then_ack": (-1, lambda ids: [(capi.CMD_CLIENT_REQUEST, 0, 0, 0, 78, 0, 0, None), (capi.CMD_APPEND_RESPONSE, 0, ids[1], 1, 1, 0, 1, None)]),
    # a HeartbeatResponse without the commit makes the leader replicate() on the progress AS IT IS THEN (leader.rs:222-231)
    "ack_then_heartbeat_response": (2, lambda ids: [(capi.CMD_APPEND_RESPONSE, 0, ids[1], 1, 1, 0, 1, None),
                                                     (capi.CMD_HEARTBEAT_RESPONSE, 0, ids[1], 0, 1, 0, 0, None)]),
    "heartbeat_response_then_ack": (2, lambda ids: [(capi.CMD_HEARTBEAT_RESPONSE, 0, ids[1], 0, 1, 0, 0, None),
                                                     (capi.CMD_APPEND_RESPONSE, 0, ids[1], 1, 1, 0, 1, None)]),
    "request_then_heartbeat_response": (2, lambda ids: [(capi.CMD_CLIENT_REQUEST, 0, 0, 0, 79, 0, 0, None),
                                                         (capi.CMD_HEARTBEAT_RESPONSE, 0, ids[2], 0, 2, 0, 0, None)]),
}


def run_arrival_kat(make, name):
    head, rows_of = ARRIVAL_KATS[name]
    node, plain = one_led_partition(make, head=max(head, 0)), one_led_partition(oracle_engine, head=max(head, 0))
    if head < 0:  # head 1 / commit 0: one block appended, nobody has acknowledged it
        for e in (node, plain):
            e.submit_columns(**rows_to_columns([(capi.CMD_CLIENT_REQUEST, 0, 0, 0, 5, 0, 0, None)]))
            e.step(5)
            e.drain_messages(), e.drain_applies(), e.drain_faults()
    cols = rows_to_columns(rows_of(node.node_ids))
    node.submit_columns(**cols)
    out = node.step_node(1000)
    general = plain_apply_equivalent(plain, cols, 1000)
    compare_snapshots(node, plain, name)
    assert node.drain_faults().tolist() == plain.drain_faults().tolist(), name
    assert fsm_tuples(node.drain_applies()) == expected_node_fsm(plain.plain_fsm, general), name
    got = node_messages_per_partition(node.drain_messages(), out, np.array(node.node_ids), node.read("self_slot"), np.zeros(1, np.int64))
    assert got == msgs_per_partition(msg_tuples(plain.drain_messages())), name
    return node, out


@pytest.mark.parametrize("name", sorted(ARRIVAL_KATS))
def test_oracle_node_step_arrival_order_kats(name):
    node, out = run_arrival_kat(oracle_engine, name)
    if name == "acks_above_the_head_before_the_append":
        assert node.read("fault")[0] == capi.FAULT_COMMIT_MISSING_BLOCK and node.read("head")[0] == 0 and node.read("commit")[0] == 0
    if name == "acks_of_the_new_head_after_the_append":
        assert node.read("fault")[0] == 0 and node.read("head")[0] == 1 and node.read("commit")[0] == 1
    assert out["rows_general"] == 0  # every one of these partitions stays in column form


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ARRIVAL_KATS))
def test_node_step_arrival_order_kats(name):
    node, out = run_arrival_kat(BatchedRaft, name)
    assert out["rows_general"] == 0


def fsm_per_partition(rows):
    d = {}
    for t in fsm_tuples(rows):
        d.setdefault(t[0], []).append(t)
    return d


def fsm_equal_per_partition(a_rows, b_rows):
    """fsm rows of one step: per partition the same rows in the same order (b run-length decoded like a)."""
    def per(rows):
        d = {}
        for t in rows:
            d.setdefault(t[0], []).append(t)
        return d
    a = [tuple(int(x) for x in (r["group"], r["kind"], r["a"], r["b"])) for r in a_rows]
    pa, pb = per(a), per(coalesce_fsm(b_rows))
    return {g: coalesce_fsm_tuples(v) for g, v in pa.items()} == {g: coalesce_fsm_tuples(v) for g, v in pb.items()}


def coalesce_fsm_tuples(rows):
    out = []
    for t in rows:
        if out and t[1] != capi.FSM_NOTIFY and out[-1][1] == t[1] and out[-1][3] == t[2]:
            out[-1] = (t[0], t[1], out[-1][2], t[3])
        else:
            out.append(t)
    return out


@pytest.mark.parametrize("R", [3, 5])
def test_oracle_node_step_fsm_rows_decode_to_the_plain_paths(R):
    """The fsm_tx side of link 1, strictly: per partition and tick the node step's rows are the
    plain path's in the plain path's order, run-length encoded (an Apply range the acknowledgements that arrived
    before the ClientRequest completed, the Notify, the Apply range of the ones after it)."""
    G, T = 300, 60
    node, plain, rng = mixed_pair(oracle_engine, oracle_engine, G, R, seed=5 + R, flags=capi.CFG_SEPARATE_COMMIT_KEY,
                                  election_timeout_ms=(700, 1500))
    n_notify = n_apply = n_apply_first = 0
    for t in range(T):
        now = 100 * (t + 1)
        cols = node_traffic(rng, node, token0=1000 * t, p_noise=0.02)
        node.submit_columns(**cols)
        node.step_node(now)
        plain_apply_equivalent(plain, cols, now)
        a, b = node.drain_applies(), plain.plain_fsm
        assert fsm_equal_per_partition(a, b), t
        n_apply_first += sum(1 for g, rows in fsm_per_partition(a).items() if len(rows) > 1 and rows[0][1] != capi.FSM_NOTIFY and
                             any(x[1] == capi.FSM_NOTIFY for x in rows))
        n_notify += int((a["kind"] == capi.FSM_NOTIFY).sum())
        n_apply += int((a["kind"] != capi.FSM_NOTIFY).sum())
        node.drain_messages(), plain.drain_messages(), node.drain_faults(), plain.drain_faults()
    assert n_notify > G and n_apply > G  # commits did advance: the rows are not vacuous
    assert n_apply_first > G / 10        # ... and acknowledgements did complete a majority BEFORE the tick's ClientRequest


@pytest.mark.gpu
@pytest.mark.parametrize("R,flags,G", [(3, 0, 3000), (5, capi.CFG_SEPARATE_COMMIT_KEY, 3000), (5, 0, 700), (4, 0, 500), (2, 0, 300), (1, 0, 200)])
def test_node_step_parity(R, flags, G):
    T = 60
    dev, ora, rng = mixed_pair(BatchedRaft, oracle_engine, G, R, seed=21 + R, flags=flags, election_timeout_ms=(700, 1500))
    dense_rows = general_rows = 0
    for t in range(T):
        now = 100 * (t + 1)
        cols = node_traffic(rng, ora, token0=1000 * t)
        outs = []
        for e in (dev, ora):
            e.submit_columns(**cols)
            outs.append(e.step_node(now))
        compare_outboxes(outs[0], outs[1], f"tick {t}")
        compare_snapshots(dev, ora, f"tick {t}")
        compare_drains(dev, ora, f"tick {t}")
        dense_rows += outs[1]["rows"] - outs[1]["rows_general"]
        general_rows += outs[1]["rows_general"]
    assert dense_rows > 5 * general_rows > 0
    assert dev.counters()["decisions"] == ora.counters()["decisions"]
    assert outs[0]["bytes_h2d"] > 0 and outs[0]["bytes_d2h"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("R,flags,G", [(5, capi.CFG_SEPARATE_COMMIT_KEY, 70_001), (3, 0, 20_003), (2, 0, 257)])
def test_tiled_row_pass_equals_the_flat_one(R, flags, G):
    """jg_step_node's row passes TILED (the default since round 6: the rows binned by tile of 256 partitions, a tile's columns
    in LDS - jg_node.h k_node_bin_* / k_node_tile) against the FLAT passes (JG_CFG_FLAT_ROW_PASSES: k_node_prefill +
    k_node_classify + k_node_route, one thread per row scattering into the G-sized columns): the same traffic into two
    engines - partitions in the hundreds of tiles, a ragged last tile, rows of every kind incl. the general path's - and
    every outbox word, state column, drained row and counter equal after every tick; both are held to the oracle by the
    other tests of this file (the tiled one by default)."""
    T = 12
    tiled, flat, rng = mixed_pair(BatchedRaft, lambda *a, **kw: BatchedRaft(*a, **{**kw, "flags": kw.get("flags", 0) | capi.CFG_FLAT_ROW_PASSES}), G, R, seed=5 + R, flags=flags,
                                  election_timeout_ms=(700, 1500))
    general = 0
    for t in range(T):
        now = 100 * (t + 1)
        cols = node_traffic(rng, flat, token0=1000 * t)
        outs = []
        for e in (tiled, flat):
            e.submit_columns(**cols)
            outs.append(e.step_node(now))
        compare_outboxes(outs[0], outs[1], f"tick {t}")
        compare_snapshots(tiled, flat, f"tick {t}")
        compare_drains(tiled, flat, f"tick {t}")
        assert outs[0]["rows_general"] == outs[1]["rows_general"], t
        general += outs[1]["rows_general"]
    assert general > 0 and tiled.counters()["decisions"] == flat.counters()["decisions"]


@pytest.mark.gpu
@pytest.mark.parametrize("leader,follower,tick", [(True, False, True), (False, True, True), (True, True, False)])
def test_node_step_halves_and_no_tick(leader, follower, tick):
    G, R = 1500, 3
    dev, ora, rng = mixed_pair(BatchedRaft, oracle_engine, G, R, seed=77, election_timeout_ms=(700, 1500))
    for t in range(30):
        now = 100 * (t + 1)
        cols = node_traffic(rng, ora, token0=1000 * t)
        outs = []
        for e in (dev, ora):
            e.submit_columns(**cols)
            outs.append(e.step_node(now, leader=leader, follower=follower, tick=tick))
        compare_outboxes(outs[0], outs[1], f"tick {t}")
        compare_snapshots(dev, ora, f"tick {t}")
        compare_drains(dev, ora, f"tick {t}")


@pytest.mark.gpu
def test_node_step_mixed_with_the_other_entry_points_and_multi_device():
    """Node steps interleaved with plain steps and dense ack ticks; and the same through ONE handle
    over three shards (aliased onto the one GPU): rows bucketed by owner, columns concatenated."""
    G, R = 2000, 3
    ora = oracle_engine(G, R, seed=3, election_timeout_ms=(700, 1500))
    devs = [BatchedRaft(G, R, seed=3, election_timeout_ms=(700, 1500)),
            BatchedRaft(G, R, seed=3, election_timeout_ms=(700, 1500), device_ids=[0, 0, 0])]
    rng = np.random.default_rng(8)
    mask = rng.random(G) < 0.5
    for e in [ora] + devs:
        elect_some(e, mask)
        e.drain_messages(), e.drain_applies(), e.drain_faults()
    for t in range(24):
        now = 100 * (t + 1)
        cols = node_traffic(rng, ora, token0=1000 * t)
        want = None
        for e in [ora] + devs:
            e.submit_columns(**cols)
            out = e.step_node(now)
            if t % 4 == 3:  # a plain step in between
                e.apply_all(__import__("josefine_amd").Command.Tick(), now_ms=now + 1)
            if want is None:
                want = out
            else:
                compare_outboxes(out, want, f"tick {t}")
        for d in devs:
            compare_snapshots(d, ora, f"tick {t}")
        if t % 3 == 2:  # rows of several steps drained at once
            a = [ora.drain_messages(), ora.drain_applies(), ora.drain_faults()]
            for d in devs:
                b = [d.drain_messages(), d.drain_applies(), d.drain_faults()]
                for x, y in zip(a, b):
                    assert x.tobytes() == y.tobytes(), t


@pytest.mark.gpu
def test_absent_columns_and_in_place_submit():
    """Rows written in place into the engine's pinned columns (jg_submit_reserve / _commit) and rows
    submitted with absent optional columns are the same rows: pieces of one tick's traffic go in
    through jg_submit without term / aux, through jg_submit with every column, and in place with some
    optional columns named - against the oracle fed the whole batch at once."""
    import ctypes as C
    G, R = 2000, 3
    dev, ora, rng = mixed_pair(BatchedRaft, oracle_engine, G, R, seed=5, election_timeout_ms=(700, 1500))
    for t in range(20):
        now = 100 * (t + 1)
        cols = node_traffic(rng, ora, token0=1000 * t)
        n = len(cols["kind"])
        ora.submit_columns(**cols)
        # piece 1: leader-side rows only (term and aux are not needed: absent), plain jg_submit
        k = cols["kind"]
        lead = np.isin(k, (capi.CMD_APPEND_RESPONSE, capi.CMD_HEARTBEAT_RESPONSE, capi.CMD_CLIENT_REQUEST))
        cut = n // 2
        p1 = np.nonzero(lead & (np.arange(n) < cut))[0]
        p2 = np.nonzero(~lead & (np.arange(n) < cut))[0]
        p3 = np.arange(cut, n)
        # (the oracle applies a partition's rows in stream order; keep that order per partition on the device:
        #  p1, p2, p3 partition the first half by kind - only safe if no partition is split by it: use
        #  whole-partition pieces instead)
        first_half_groups = set(cols["group"][:cut].tolist())
        p3 = np.array([i for i in range(n) if i >= cut and int(cols["group"][i]) not in first_half_groups], dtype=np.int64)
        rest = np.array([i for i in range(n) if i >= cut and int(cols["group"][i]) in first_half_groups], dtype=np.int64)
        lead_groups = set(cols["group"][p1].tolist())
        p2_groups = set(cols["group"][p2].tolist())
        mixed = lead_groups & p2_groups
        order = []
        if len(p1):
            sel = np.array([i for i in p1 if int(cols["group"][i]) not in mixed], dtype=np.int64)
            if len(sel):
                dev.submit_columns(cols["kind"][sel], cols["group"][sel], from_=cols["from_"][sel], id=cols["id"][sel], flag=cols["flag"][sel])
                order.append(sel)
        sel = np.array(sorted([i for i in range(cut) if not (lead[i] and int(cols["group"][i]) not in mixed)] + rest.tolist()), dtype=np.int64)
        if len(sel):
            dev.submit_columns(cols["kind"][sel], cols["group"][sel], cols["from_"][sel], cols["term"][sel], cols["id"][sel], cols["aux"][sel],
                               cols["flag"][sel], cols["blk_id"], cols["blk_next"])
            order.append(sel)
        if len(p3):  # in place
            c = capi.CmdCols()
            nb = len(cols["blk_id"])
            dev._check(dev.api.submit_reserve(dev._h, len(p3), nb, C.byref(c)))

            def view(ptr, dt, m):
                return np.frombuffer((C.c_char * (m * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt)
            view(c.kind, np.uint8, len(p3))[:] = cols["kind"][p3]
            view(c.group, np.uint32, len(p3))[:] = cols["group"][p3]
            view(c.from_, np.uint32, len(p3))[:] = cols["from_"][p3]
            view(c.term, np.uint64, len(p3))[:] = cols["term"][p3]
            view(c.id, np.uint64, len(p3))[:] = cols["id"][p3]
            view(c.aux, np.uint64, len(p3))[:] = cols["aux"][p3]
            view(c.flag, np.uint8, len(p3))[:] = cols["flag"][p3]
            if nb:
                view(c.blk_id, np.uint64, nb)[:] = cols["blk_id"]
                view(c.blk_next, np.uint64, nb)[:] = cols["blk_next"]
            dev._check(dev.api.submit_commit(dev._h, len(p3), nb, capi.COL_FROM | capi.COL_TERM | capi.COL_AUX | capi.COL_FLAG))
        a, b = dev.step_node(now), ora.step_node(now)
        compare_outboxes(a, b, f"tick {t}")
        compare_snapshots(dev, ora, f"tick {t}")
        compare_drains(dev, ora, f"tick {t}")
    # a commit beyond the reservation, an unknown column bit, AppendEntries without the aux column
    c = capi.CmdCols()
    dev._check(dev.api.submit_reserve(dev._h, 4, 0, C.byref(c)))
    assert dev.api.submit_commit(dev._h, 1 << 40, 0, 0) == capi.EINVAL
    assert dev.api.submit_commit(dev._h, 1, 0, 64) == capi.EINVAL
    np.frombuffer((C.c_char * 1).from_address(c.kind), dtype=np.uint8)[0] = capi.CMD_APPEND_ENTRIES
    np.frombuffer((C.c_char * 4).from_address(c.group), dtype=np.uint32)[0] = 0
    assert dev.api.submit_commit(dev._h, 1, 0, capi.COL_FROM) == capi.EINVAL


def split_columns(cols, ora, col_slots):
    """Move the AppendResponse / HeartbeatResponse rows of `col_slots` out of a traffic batch into answer /
    hb_commit columns (one word per partition and slot; a second row for the same entry is dropped: a column
    has one)."""
    G = ora.G
    ids = list(ora.node_ids)
    answer = {r: np.full(G, capi.NO_ACK, np.uint64) for r in col_slots}
    hbc = {r: np.zeros(G, np.uint64) for r in col_slots}
    keep = np.ones(len(cols["kind"]), bool)
    for i in range(len(cols["kind"])):
        k, f = int(cols["kind"][i]), int(cols["from_"][i])
        if k not in (capi.CMD_APPEND_RESPONSE, capi.CMD_HEARTBEAT_RESPONSE) or f not in ids or ids.index(f) not in col_slots:
            continue
        r, g = ids.index(f), int(cols["group"][i])
        keep[i] = False
        w = int(answer[r][g])
        if k == capi.CMD_APPEND_RESPONSE:
            if int(cols["id"][i]) >= capi.MAILBOX_NONE or (w >> 8) != capi.MAILBOX_NONE:
                continue
            w = (int(cols["id"][i]) << 8) | (w & 0xff)
        else:
            if (w & 0xff) != capi.HB_NONE:
                continue
            has = 1 if cols["flag"][i] else 0
            w = (w & ~0xff) | has
            if not has:
                hbc[r][g] = cols["id"][i]
        answer[r][g] = w
    rest = {k: (v[keep] if k not in ("blk_id", "blk_next") else v) for k, v in cols.items()}
    return rest, answer, hbc


@pytest.mark.parametrize("R", [3, 5])
def test_oracle_column_inbound_is_the_row_inbound(R):
    """jo_node_inbox_columns: a peer's answers as ONE column are indistinguishable from the same answers as rows
    (for partitions in column form: with noise-free leader traffic every led partition is)."""
    G, T = 300, 40
    rows_e, cols_e, rng = mixed_pair(oracle_engine, oracle_engine, G, R, seed=31 + R, flags=capi.CFG_SEPARATE_COMMIT_KEY,
                                     election_timeout_ms=(700, 1500))
    col_slots = list(range(1, R))
    for t in range(T):
        now = 100 * (t + 1)
        cols = node_traffic(rng, rows_e, token0=1000 * t, p_noise=0.0)
        rest, answer, hbc = split_columns(cols, rows_e, col_slots)
        # the row engine gets exactly what the columns hold (duplicates dropped), as rows
        kept_rows = []
        for r in col_slots:
            ack = answer[r] >> np.uint64(8)
            for g in np.nonzero(answer[r] != np.uint64(capi.NO_ACK))[0]:
                w = int(answer[r][g])
                if (w & 0xff) != capi.HB_NONE:
                    kept_rows.append((capi.CMD_HEARTBEAT_RESPONSE, int(g), rows_e.node_ids[r], 0, int(hbc[r][g]), 0, w & 0xff, None))
                if (w >> 8) != capi.MAILBOX_NONE:
                    kept_rows.append((capi.CMD_APPEND_RESPONSE, int(g), rows_e.node_ids[r], 0, int(ack[g]), 0, 1, None))
        from node_step import rows_to_columns
        extra = rows_to_columns(kept_rows)
        rows_e.submit_columns(**rest)
        rows_e.submit_columns(**{k: v for k, v in extra.items()})
        for r in col_slots:
            cols_e.node_inbox_columns(r, answer[r], hbc[r])
        cols_e.submit_columns(**rest)
        a, b = rows_e.step_node(now), cols_e.step_node(now)
        for k in ("beat_term", "beat_commit", "ae", "answer"):
            assert np.array_equal(a[k], b[k]), (t, k)
        compare_snapshots(cols_e, rows_e, f"tick {t}")
        compare_drains(cols_e, rows_e, f"tick {t}")
    assert rows_e.counters()["decisions"] == cols_e.counters()["decisions"] > G


@pytest.mark.gpu
@pytest.mark.parametrize("R,with_hbc", [(3, True), (5, True), (5, False)])
def test_node_step_column_inbound_parity(R, with_hbc):
    """jg_node_inbox_columns on the device == the oracle's: answers of the other members as columns, everything
    else (client requests, the follower side, the noise) as rows; with and without the hb_commit column."""
    G, T = 2500, 40
    dev, ora, rng = mixed_pair(BatchedRaft, oracle_engine, G, R, seed=41 + R, election_timeout_ms=(700, 1500))
    col_slots = list(range(1, R))
    for t in range(T):
        now = 100 * (t + 1)
        cols = node_traffic(rng, ora, token0=1000 * t)
        rest, answer, hbc = split_columns(cols, ora, col_slots)
        outs = []
        for e in (dev, ora):
            for r in col_slots:
                e.node_inbox_columns(r, answer[r], hbc[r] if with_hbc else None)
            e.submit_columns(**rest)
            outs.append(e.step_node(now))
        compare_outboxes(outs[0], outs[1], f"tick {t}")
        compare_snapshots(dev, ora, f"tick {t}")
        compare_drains(dev, ora, f"tick {t}")
    assert dev.counters()["decisions"] == ora.counters()["decisions"]
    # rows AND a column from one sender in one step: refused, on both
    from josefine_amd import EngineError
    one = dict(kind=np.array([capi.CMD_APPEND_RESPONSE], np.uint8), group=np.array([0], np.uint32), from_=np.array([dev.node_ids[1]], np.uint32),
               id=np.array([0], np.uint64))
    for e in (ora, dev):
        e.node_inbox_columns(1, np.full(G, capi.NO_ACK, np.uint64))
        e.submit_columns(**one)
        with pytest.raises(EngineError):
            e.step_node(5000)
            e.read("term")
    with pytest.raises(EngineError):
        dev.node_inbox_columns(0, np.zeros(G, np.uint64))  # the own slot


@pytest.mark.gpu
def test_unchecked_rows_are_validated_on_the_device():
    """JG_COL_UNCHECKED: no validation pass on the host - the classification checks group and kind; a row out of range is
    not applied and the next synchronising call says so; the same rows without the bad one behave as ever."""
    import ctypes as C
    from josefine_amd import EngineError
    G, R = 500, 3
    dev, ora, rng = mixed_pair(BatchedRaft, oracle_engine, G, R, seed=9, election_timeout_ms=(700, 1500))
    cols = node_traffic(rng, ora, token0=0, p_noise=0.0)
    n = len(cols["kind"])

    def put(kind, group):
        c = capi.CmdCols()
        nb = len(cols["blk_id"])
        dev._check(dev.api.submit_reserve(dev._h, n, nb, C.byref(c)))

        def view(ptr, dt, m):
            return np.frombuffer((C.c_char * (m * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt)
        view(c.kind, np.uint8, n)[:] = kind
        view(c.group, np.uint32, n)[:] = group
        view(c.from_, np.uint32, n)[:] = cols["from_"]
        view(c.term, np.uint64, n)[:] = cols["term"]
        view(c.id, np.uint64, n)[:] = cols["id"]
        view(c.aux, np.uint64, n)[:] = cols["aux"]
        view(c.flag, np.uint8, n)[:] = cols["flag"]
        if nb:
            view(c.blk_id, np.uint64, nb)[:] = cols["blk_id"]
            view(c.blk_next, np.uint64, nb)[:] = cols["blk_next"]
        dev._check(dev.api.submit_commit(dev._h, n, nb, capi.COL_FROM | capi.COL_TERM | capi.COL_AUX | capi.COL_FLAG | capi.COL_UNCHECKED))

    put(cols["kind"], cols["group"])
    with pytest.raises(EngineError):
        dev.step(50)  # only jg_step_node takes an unchecked batch
    ora.submit_columns(**cols)
    a, b = dev.step_node(100), ora.step_node(100)
    compare_outboxes(a, b, "unchecked")
    compare_snapshots(dev, ora, "unchecked")
    compare_drains(dev, ora, "unchecked")
    bad_group = cols["group"].copy()
    bad_group[3] = G + 7
    put(cols["kind"], bad_group)
    with pytest.raises(EngineError, match="out of range"):
        dev.step_node(200)
        dev.read("term")


@pytest.mark.gpu
@pytest.mark.parametrize("R,flags,G", [(3, 0, 3000), (5, capi.CFG_SEPARATE_COMMIT_KEY, 3000), (1, 0, 200)])
def test_node_step_async_parity(R, flags, G):
    """JG_NODE_ASYNC: the step returns with nothing synchronised - its dense halves leave the general-path partitions
    alone, the engine comes back for them when the step is settled - and the NEXT tick's rows are submitted (into the
    second set of pinned columns) before this tick's outbox is asked for.  Every outbox word, state column and drained
    row equals the oracle's synchronous step."""
    T = 50
    dev, ora, rng = mixed_pair(BatchedRaft, oracle_engine, G, R, seed=31 + R, flags=flags, election_timeout_ms=(700, 1500))
    general_ticks = 0
    nxt = node_traffic(rng, ora, token0=0)
    dev.submit_columns(**nxt)
    for t in range(T):
        now = 100 * (t + 1)
        cols = nxt
        ora.submit_columns(**cols)
        b = ora.step_node(now)
        # (the next tick's traffic is generated from the oracle's state AFTER this tick, and submitted to the device while
        #  its asynchronous step is still in flight)
        nxt = node_traffic(rng, ora, token0=1000 * (t + 1), p_noise=0.03 if t % 3 else 0.0, p_reorder=0.08 if t % 3 else 0.0)
        a = dev.step_node(now, async_=True, between=lambda: dev.submit_columns(**nxt))
        compare_outboxes(a, b, f"tick {t}")
        compare_snapshots(dev, ora, f"tick {t}")
        compare_drains(dev, ora, f"tick {t}")
        general_ticks += b["rows_general"] > 0
    assert general_ticks > 5  # (the catch-up pass ran; steps without general rows: tests/cpp/test_event_loop_cluster.cpp, pipelined)
    assert dev.counters()["decisions"] == ora.counters()["decisions"]


def _same_rows(a, b, what):
    assert a.shape == b.shape and a.tobytes() == b.tobytes(), f"{what}: {len(a)} rows against {len(b)}"


@pytest.mark.gpu
@pytest.mark.parametrize("R,flags,G,peek,bus", [(3, 0, 3000, False, False), (5, capi.CFG_SEPARATE_COMMIT_KEY, 3000, True, False),
                                              (5, capi.CFG_SEPARATE_COMMIT_KEY, 2000, False, True), (1, 0, 200, False, False)])
def test_node_step_two_in_flight_parity(R, flags, G, peek, bus):
    """JG_NODE_KEEP: TWO steps in flight - tick t + 1 is begun before tick t's outbox has been viewed; the view then serves
    the OLDER step (its columns, and exactly its fsm_tx / rpc_tx rows and faults through the drains that follow) while the
    newer one runs.  Every outbox word and drained row of every tick equals the oracle's synchronous step of that tick,
    general-path ticks (the row count is looked at when the next step begins), exceptional rows and faults included;
    `peek`: the state columns are read while a step is outstanding (a read settles the newest step and consumes nothing).
    `bus`: with ABI v7's compact formats (the common AppendEntries word's late fetch of the rows reads the step's own copy)."""
    from josefine_amd import expand_fsm_rows
    T = 40
    import os
    seed = 77 + R + int(os.environ.get("JG_SOAK_SEED", "0"))  # (profiles/micro/r06_two_in_flight_soak.sh: other seeds, on the device)
    dev, ora, rng = mixed_pair(BatchedRaft, oracle_engine, G, R, seed=seed, flags=flags, election_timeout_ms=(700, 1500))
    own = int(ora.read("self_slot")[0])
    kw = dict(common_ae=True, fsm_fused=True) if bus else {}
    want = []  # per tick: the oracle's outbox and drains
    general_ticks = fsm_rows = msg_rows = 0

    def finish(t):
        nonlocal fsm_rows, msg_rows
        a = dev.node_outbox()
        b, drains = want[t]
        if bus:  # (WHICH partitions read AEC_INDIVIDUAL is a matter of representation: compare what the words stand for)
            a = _expand_common_ae(a, own, R)
        compare_outboxes(a, b, f"tick {t}")
        for fn, rows in drains.items():
            got = getattr(dev, fn)()
            _same_rows(expand_fsm_rows(got) if bus and fn == "drain_applies" else got, rows, f"tick {t}: {fn}")
        fsm_rows += len(drains["drain_applies"])
        msg_rows += len(drains["drain_messages"])

    for t in range(T):
        now = 100 * (t + 1)
        cols = node_traffic(rng, ora, token0=1000 * t, p_noise=0.03 if t % 3 else 0.0, p_reorder=0.08 if t % 3 else 0.0)
        ora.submit_columns(**cols)
        b = ora.step_node(now)  # (the plain formats: what the compact ones stand for)
        want.append((b, {fn: getattr(ora, fn)() for fn in ("drain_messages", "drain_applies", "drain_faults")}))
        general_ticks += b["rows_general"] > 0
        dev.submit_columns(**cols)
        dev.step_node_begin(now, async_=True, keep=True, **kw)
        if peek and t % 4 == 1:
            compare_snapshots(dev, ora, f"tick {t}, one step outstanding besides it")
        if t:
            finish(t - 1)
        if t % 7 == 3:
            with pytest.raises(EngineError, match="JG_NODE_KEEP"):
                dev.step(now)  # nothing else steps the engine while kept steps are outstanding
    finish(T - 1)
    compare_snapshots(dev, ora, "after the last tick")
    assert general_ticks > 5 and fsm_rows and msg_rows
    assert dev.counters()["decisions"] == ora.counters()["decisions"]
    # ... and the engine steps the plain way again afterwards
    cols = node_traffic(rng, ora, token0=999_000)
    ora.submit_columns(**cols), dev.submit_columns(**cols)
    compare_outboxes(dev.step_node(100 * (T + 1)), ora.step_node(100 * (T + 1)), "a plain step afterwards")
    compare_drains(dev, ora, "a plain step afterwards")
    compare_snapshots(dev, ora, "a plain step afterwards")


@pytest.mark.gpu
def test_node_step_keep_rules():
    """JG_NODE_KEEP's rules at the boundary: it goes with JG_NODE_ASYNC; at most two kept steps are outstanding; nothing else
    steps the engine meanwhile (a plain node step, jg_step, a drain prefetch); the drains deliver nothing that is not due; a
    view without an outstanding step shows the last one again; reads go through."""
    G, R = 300, 3
    dev, ora, rng = mixed_pair(BatchedRaft, oracle_engine, G, R, seed=5, election_timeout_ms=(700, 1500))
    with pytest.raises(EngineError, match="JG_NODE_ASYNC"):
        dev.step_node_begin(100, keep=True)
    outs = []
    for t in range(2):
        cols = node_traffic(rng, ora, token0=1000 * t)
        ora.submit_columns(**cols), dev.submit_columns(**cols)
        outs.append((ora.step_node(100 * (t + 1)), {fn: getattr(ora, fn)() for fn in ("drain_messages", "drain_applies", "drain_faults")}))
        dev.step_node_begin(100 * (t + 1), async_=True, keep=True)
        assert len(dev.drain_applies()) == 0 and len(dev.drain_messages()) == 0  # (nothing is due before its outbox has been viewed)
    with pytest.raises(EngineError, match="outstanding"):
        dev.step_node_begin(300, async_=True, keep=True)  # a third
    with pytest.raises(EngineError, match="outstanding"):
        dev.step_node_begin(300, async_=True)  # a step that would not keep its outputs
    with pytest.raises(EngineError, match="JG_NODE_KEEP"):
        dev.step(300)
    with pytest.raises(EngineError, match="JG_NODE_KEEP"):
        dev.drain_prefetch()
    compare_snapshots(dev, ora, "two steps outstanding: a read sees the engine after the newest")
    for t in range(2):
        a = dev.node_outbox()
        compare_outboxes(a, outs[t][0], f"view {t}")
        again = dev.node_outbox() if t == 1 else None  # (nothing outstanding any more: the same outbox again)
        if again is not None:
            compare_outboxes(again, outs[t][0], "the last view again")
        for fn, rows in outs[t][1].items():
            _same_rows(getattr(dev, fn)(), rows, f"view {t}: {fn}")
    # the views' rows handed over by VIEW drains as well (the queue's buffers change hands with the sets' landing buffers)
    for t in range(2, 8):
        cols = node_traffic(rng, ora, token0=1000 * t)
        ora.submit_columns(**cols), dev.submit_columns(**cols)
        want = (ora.step_node(100 * (t + 1)), ora.drain_applies())
        ora.drain_messages(), ora.drain_faults()
        dev.step_node_begin(100 * (t + 1), async_=True, keep=True)
        compare_outboxes(dev.node_outbox(), want[0], f"tick {t}")
        _same_rows(dev.drain_applies(), want[1], f"tick {t}: fsm rows")  # (the view form: josefine_amd.engine drains through jg_drain_applies_view)
        dev.drain_messages(), dev.drain_faults()
    compare_snapshots(dev, ora, "the end")


def _commit_in_place(dev, cols, extra_flags, lo=0, hi=None):
    """rows [lo, hi) of `cols` written straight into the engine's pinned columns (jg_submit_reserve / jg_submit_commit)"""
    import ctypes as C
    hi = len(cols["kind"]) if hi is None else hi
    n = hi - lo
    nb = len(cols["blk_id"]) if (lo == 0 and hi == len(cols["kind"])) else 0
    c = capi.CmdCols()
    dev._check(dev.api.submit_reserve(dev._h, n, nb, C.byref(c)))

    def view(ptr, dt, m):
        return np.frombuffer((C.c_char * (max(m, 1) * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt)[:m]
    view(c.kind, np.uint8, n)[:] = cols["kind"][lo:hi]
    view(c.group, np.uint32, n)[:] = cols["group"][lo:hi]
    view(c.from_, np.uint32, n)[:] = cols["from_"][lo:hi]
    view(c.term, np.uint64, n)[:] = cols["term"][lo:hi]
    view(c.id, np.uint64, n)[:] = cols["id"][lo:hi]
    view(c.aux, np.uint64, n)[:] = cols["aux"][lo:hi]
    view(c.flag, np.uint8, n)[:] = cols["flag"][lo:hi]
    if nb:
        view(c.blk_id, np.uint64, nb)[:] = cols["blk_id"]
        view(c.blk_next, np.uint64, nb)[:] = cols["blk_next"]
    dev._check(dev.api.submit_commit(dev._h, n, nb, capi.COL_FROM | capi.COL_TERM | capi.COL_AUX | capi.COL_FLAG | extra_flags))


@pytest.mark.gpu
@pytest.mark.parametrize("R,flags,G", [(3, 0, 3000), (5, capi.CFG_SEPARATE_COMMIT_KEY, 2000)])
def test_node_step_early_upload_parity(R, flags, G):
    """JG_COL_UPLOAD_NOW: a committed batch leaves for the device at once, on a copy stream of its own, into one of two
    device buffers by turns - while the previous (asynchronous) step is still in flight.  A batch that does not stay the
    step's whole input (rows committed behind it: every 5th tick here; a batch without AppendEntries blocks only - the
    blocks of a split batch go the plain way) is uploaded again by the step.  Every outbox word, state column and drained
    row equals the oracle's synchronous step."""
    T = 40
    dev, ora, rng = mixed_pair(BatchedRaft, oracle_engine, G, R, seed=77 + R, flags=flags, election_timeout_ms=(700, 1500))

    def commit(cols, t):
        n = len(cols["kind"])
        if t % 5 == 4 and n > 10 and not len(cols["blk_id"]):  # rows behind the early upload: the step uploads all of them itself
            _commit_in_place(dev, cols, capi.COL_UNCHECKED | capi.COL_UPLOAD_NOW, 0, n // 2)
            _commit_in_place(dev, cols, capi.COL_UNCHECKED, n // 2, n)
        else:
            _commit_in_place(dev, cols, capi.COL_UNCHECKED | capi.COL_UPLOAD_NOW)
    nxt = node_traffic(rng, ora, token0=0)
    commit(nxt, 0)
    for t in range(T):
        now = 100 * (t + 1)
        cols = nxt
        ora.submit_columns(**cols)
        b = ora.step_node(now)
        nxt = node_traffic(rng, ora, token0=1000 * (t + 1), p_noise=0.03 if t % 3 else 0.0, p_reorder=0.08 if t % 3 else 0.0)
        a = dev.step_node(now, async_=True, between=lambda: commit(nxt, t + 1))
        compare_outboxes(a, b, f"tick {t}")
        compare_snapshots(dev, ora, f"tick {t}")
        compare_drains(dev, ora, f"tick {t}")
    assert dev.counters()["decisions"] == ora.counters()["decisions"]


def _pack_kind(cols, ids, R):
    """JG_COL_PACKED_KIND: kind | sender slot << 4 | flag << 7 - and what the rows then SAY: a stranger (no slot: 7, R < 8)
    reads NodeId 0, a kind without a sender reads 0 whatever the bits"""
    kind, frm = cols["kind"], cols["from_"]
    slot = np.full(len(kind), 7, np.uint8)
    for r, i in enumerate(ids):
        slot[frm == i] = r
    carries = (kind >= capi.CMD_VOTE_REQUEST) & (kind <= capi.CMD_HEARTBEAT_RESPONSE)
    table = np.array(list(ids) + [0] * (8 - R), np.uint32)
    said = np.where(carries, table[slot], 0).astype(np.uint32)
    packed = (kind | (slot << 4) | ((cols["flag"] != 0).astype(np.uint8) << 7)).astype(np.uint8)
    return packed, said


def _expand_common_ae(a, own, R):
    """the node outbox of JG_NODE_COMMON_AE as the [R][G] block it stands for"""
    aec = a["aec"]
    ind = aec == np.uint64(capi.AEC_INDIVIDUAL)
    assert (a["ae"] is None) == (not ind.any())
    ae = np.repeat(aec[None, :], R, axis=0)
    ae[own] = np.uint64(NO_ACK)
    if ind.any():
        ae[:, ind] = a["ae"][:, ind]
    return dict(a, ae=ae)


NO_ACK = capi.NO_ACK


@pytest.mark.gpu
@pytest.mark.parametrize("R,flags,G,async_,id32", [(3, 0, 2500, True, False), (5, capi.CFG_SEPARATE_COMMIT_KEY, 2000, True, True), (5, 0, 600, False, True)])
def test_node_step_compact_bus_parity(R, flags, G, async_, id32):
    """The node step's three bus formats of ABI v7 together: rows committed with JG_COL_PACKED_KIND (sender slot and flag in
    the kind byte: no from / flag columns), the Tick's AppendEntries words as one word per partition (JG_NODE_COMMON_AE: the
    rows come down only in the ticks where some partition's words differ), a leader's fsm_tx rows of a step as one
    JG_FSM_LEADER_STEP row (JG_NODE_FSM_FUSED); id32: the id column as 32-bit values too (JG_COL_ID32).  What they stand for - every outbox word, state column and drained row -
    equals the oracle's step over the plain rows, tick after tick, on the mixed traffic of the other node-step tests."""
    from josefine_amd import expand_fsm_rows
    import os
    T = 40
    seed = 123 + R + int(os.environ.get("JG_SOAK_SEED", "0"))  # (profiles/micro/compact_bus_soak.sh: other seeds, on the device)
    dev, ora, rng = mixed_pair(BatchedRaft, oracle_engine, G, R, seed=seed, flags=flags, election_timeout_ms=(700, 1500))
    _, ora2, _ = mixed_pair(oracle_engine, oracle_engine, G, R, seed=seed, flags=flags, election_timeout_ms=(700, 1500))  # the oracle, asked for the formats itself
    ids = list(ora.node_ids)
    own = int(ora.read("self_slot")[0])
    seen = dict(individual=0, common_only=0, fused=0, plain_leader=0)

    def commit(cols, packed):
        import ctypes as C
        n, nb = len(cols["kind"]), len(cols["blk_id"])
        c = capi.CmdCols()
        dev._check(dev.api.submit_reserve(dev._h, n, nb, C.byref(c)))

        def view(ptr, dt, m):
            return np.frombuffer((C.c_char * (max(m, 1) * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt)[:m]
        view(c.kind, np.uint8, n)[:] = packed
        view(c.group, np.uint32, n)[:] = cols["group"]
        view(c.term, np.uint64, n)[:] = cols["term"]
        if id32:
            view(c.id, np.uint32, n)[:] = cols["id"].astype(np.uint32)
        else:
            view(c.id, np.uint64, n)[:] = cols["id"]
        view(c.aux, np.uint64, n)[:] = cols["aux"]
        if nb:
            view(c.blk_id, np.uint64, nb)[:] = cols["blk_id"]
            view(c.blk_next, np.uint64, nb)[:] = cols["blk_next"]
        dev._check(dev.api.submit_commit(dev._h, n, nb, capi.COL_TERM | capi.COL_AUX | capi.COL_UNCHECKED | capi.COL_PACKED_KIND |
                                         (capi.COL_ID32 if id32 else 0) | (capi.COL_UPLOAD_NOW if async_ else 0)))

    def traffic(t):
        cols = node_traffic(rng, ora, token0=1000 * t, p_noise=0.03 if t % 3 else 0.0, p_reorder=0.08 if t % 3 else 0.0)
        packed, said = _pack_kind(cols, ids, R)
        if id32:  # (the forged heads a mailbox word cannot hold do not fit 32 bits: what the rows then say, to both sides)
            cols = dict(cols, id=cols["id"] & np.uint64(0xFFFFFFFF))
        return dict(cols, from_=said, flag=(cols["flag"] != 0).astype(np.uint8)), packed
    nxt, nxt_packed = traffic(0)
    commit(nxt, nxt_packed)
    for t in range(T):
        now = 100 * (t + 1)
        ora.submit_columns(**nxt)
        ora2.submit_columns(**nxt)
        b = ora.step_node(now)
        b2 = ora2.step_node(now, common_ae=True, fsm_fused=True)
        nxt, nxt_packed = traffic(t + 1)
        a = dev.step_node(now, async_=async_, common_ae=True, fsm_fused=True, between=lambda: commit(nxt, nxt_packed))
        assert a["aec"] is not None
        seen["individual" if a["ae"] is not None else "common_only"] += 1
        # (the device says INDIVIDUAL also where a partition's rows took the general path: its words are in the rows then)
        common = a["aec"] != np.uint64(capi.AEC_INDIVIDUAL)
        assert np.array_equal(a["aec"][common], b2["aec"][common]), t
        a = _expand_common_ae(a, own, R)
        compare_outboxes(a, b, f"tick {t}")
        compare_outboxes(_expand_common_ae(b2, own, R), b, f"tick {t} (oracle's own formats)")
        compare_snapshots(dev, ora, f"tick {t}")
        for fn in ("drain_messages", "drain_faults"):
            x, y = getattr(dev, fn)(), getattr(ora, fn)()
            assert x.shape == y.shape and x.tobytes() == y.tobytes(), (t, fn)
        x, y, y2 = dev.drain_applies(), ora.drain_applies(), ora2.drain_applies()
        ora2.drain_messages(), ora2.drain_faults()
        seen["fused"] += int((x["kind"] == capi.FSM_LEADER_STEP).sum())
        seen["plain_leader"] += int((x["kind"] == capi.FSM_NOTIFY).sum())
        assert x.shape == y2.shape and x.tobytes() == y2.tobytes(), (t, "drain_applies: the fused rows themselves")
        x = expand_fsm_rows(x)
        assert x.shape == y.shape and x.tobytes() == y.tobytes(), (t, "drain_applies")
    # both forms of every format were exercised
    assert seen["individual"] and seen["fused"] > 10 * T, seen
    if os.environ.get("JG_EMULATED_DEVICE") != "1":  # (the stand-in's wave reductions behind divergent code: tests/host_device.py)
        assert dev.counters()["decisions"] == ora.counters()["decisions"]
    # the formats' rules: a step's commits agree on the kind column's format; jg_submit cannot follow a packed commit
    from josefine_amd import EngineError
    with pytest.raises(EngineError, match="JG_COL_PACKED_KIND"):  # (the last tick's `between` left a packed batch pending)
        dev.submit_columns(np.full(1, capi.CMD_TICK, np.uint8), np.zeros(1, np.uint32))


@pytest.mark.gpu
def test_bus_format_flags_do_not_outlive_an_empty_commit():
    """An EMPTY commit that names JG_COL_PACKED_KIND / JG_COL_ID32 says nothing about the rows that follow it: the step's
    first rows decide what the kind / id columns hold (plain rows through jg_submit here), tick after tick."""
    import ctypes as C
    G, R = 700, 3
    dev, ora, rng = mixed_pair(BatchedRaft, oracle_engine, G, R, seed=5, election_timeout_ms=(700, 1500))
    for t in range(6):
        cols = node_traffic(rng, ora, token0=1000 * t)
        c = capi.CmdCols()
        dev._check(dev.api.submit_reserve(dev._h, 0, 0, C.byref(c)))
        dev._check(dev.api.submit_commit(dev._h, 0, 0, capi.COL_UNCHECKED | capi.COL_PACKED_KIND | capi.COL_ID32))
        dev.submit_columns(**cols)
        ora.submit_columns(**cols)
        a, b = dev.step_node(100 * (t + 1)), ora.step_node(100 * (t + 1))
        compare_outboxes(a, b, f"tick {t}")
        compare_snapshots(dev, ora, f"tick {t}")
        compare_drains(dev, ora, f"tick {t}")


@pytest.mark.parametrize("R,flags", [(3, 0), (5, capi.CFG_SEPARATE_COMMIT_KEY), (2, 0)])
def test_oracle_bus_formats_stand_for_the_plain_step(R, flags):
    """CPU: the oracle's own restatement of JG_NODE_COMMON_AE / JG_NODE_FSM_FUSED (what the device's rows are held to byte for
    byte in test_node_step_compact_bus_parity) against the oracle's plain step: the common word expands to the [R][G] block,
    the fused rows expand to Apply / Notify / Apply - and both forms of each occur."""
    from josefine_amd import expand_fsm_rows
    G, T = 1500, 40
    a, b, rng = mixed_pair(oracle_engine, oracle_engine, G, R, seed=900 + R, flags=flags, election_timeout_ms=(700, 1500))
    own = int(a.read("self_slot")[0])
    seen = dict(fused=0, plain_notify=0, individual=0, common=0)
    for t in range(T):
        cols = node_traffic(rng, a, token0=1000 * t, p_noise=0.03 if t % 3 else 0.0)
        a.submit_columns(**cols), b.submit_columns(**cols)
        x, y = a.step_node(100 * (t + 1)), b.step_node(100 * (t + 1), common_ae=True, fsm_fused=True)
        seen["individual" if y["ae"] is not None else "common"] += 1
        compare_outboxes(_expand_common_ae(y, own, R), x, f"tick {t}")
        compare_snapshots(b, a, f"tick {t}")
        fx, fy = a.drain_applies(), b.drain_applies()
        seen["fused"] += int((fy["kind"] == capi.FSM_LEADER_STEP).sum())
        seen["plain_notify"] += int((fy["kind"] == capi.FSM_NOTIFY).sum())
        assert len(fy) <= len(fx)
        ey = expand_fsm_rows(fy)
        assert ey.shape == fx.shape and ey.tobytes() == fx.tobytes(), t
        for fn in ("drain_messages", "drain_faults"):
            p, q = getattr(a, fn)(), getattr(b, fn)()
            assert p.tobytes() == q.tobytes(), (t, fn)
    assert seen["fused"] > 5 * T and (seen["individual"] or R == 2), seen  # (one follower: its word IS the common word)
