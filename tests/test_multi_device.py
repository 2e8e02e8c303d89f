"""Multi-device engine behind ONE handle (SURVEY.md §8(b),(e); jg_config.n_devices / device_ids):
D shards aliased onto device 0 — which is all a 1-GPU box has, and exercises the same router, host
threads, per-shard streams and merge as D distinct devices — must be bit-identical, in every state
column and every drained row, to one engine over the same groups and to the CPU oracle.
The reference shape being kept: one caller owns the handle (event_loop, src/raft/server.rs:103-165)."""
import ctypes as C

import numpy as np
import pytest

from josefine_amd import BatchedRaft, Command, capi
from josefine_amd.engine import EngineError
from failures import failure_rows
from fuzz import random_batch
from oracle_lib import oracle_engine
from parity import DeviceSynth, compare_drains, compare_snapshots, elect_all, synth_tick_host

pytestmark = pytest.mark.gpu
SEED = 0x5EED


@pytest.mark.parametrize("D,G,R", [(2, 1001, 3), (3, 1000, 5), (4, 1023, 5), (4, 7, 3), (3, 2, 1)])
def test_sharded_engine_equals_single_engine_and_oracle(D, G, R):
    rng = np.random.default_rng(100 * D + R)
    slots = rng.integers(0, R, G).astype(np.uint8)
    kw = dict(seed=SEED, self_slots=slots, election_timeout_ms=(300, 700))
    multi = BatchedRaft(G, R, device_ids=[0] * D, **kw)
    one = BatchedRaft(G, R, **kw)
    ora = oracle_engine(G, R, **kw)
    S = -(-G // D)
    assert multi.n_shards == -(-G // S) and one.n_shards == 1
    los = [multi.shard(d).group_lo for d in range(multi.n_shards)]
    assert los == [d * S for d in range(multi.n_shards)]
    assert sum(multi.shard(d).G for d in range(multi.n_shards)) == G
    engines = (multi, one, ora)
    for e in engines:
        elect_all(e)
    compare_snapshots(multi, ora, "after election")
    compare_drains(multi, ora, "after election")
    one.drain_messages(), one.drain_applies(), one.drain_faults()
    sim = np.zeros((R, G), dtype=np.uint64)
    budget = np.full(G, 3)
    now = 0
    for it in range(36):
        now += 100
        what = it % 4
        if what in (0, 1):   # dense leader tick from a host [R][G] block (split per shard by the router)
            acks = synth_tick_host(ora, 1, it, sim)
            for e in engines:
                e.step_dense_acks(acks)
        elif what == 2:      # random commands: every role, every kind, forks, faults
            batch = random_batch(rng, ora, max(8, G // 2), budget=budget, foreign_voters=True)
            for e in engines:
                e.submit_columns(**batch)
                e.step(now)
        else:                # leader crashes + re-elections, and a second step before the drain
            rows, n = failure_rows(SEED, it, 0, G, R, ora.node_ids, slots, 7)
            batch = random_batch(rng, ora, 16, budget=budget)
            for e in engines:
                if n:
                    e.submit_columns(**rows)
                    e.step(now)
                e.submit_columns(**batch)
                e.step(now + 1)
        compare_snapshots(multi, ora, f"D={D} it {it}")
        ref = {fn: getattr(one, fn)() for fn in ("drain_messages", "drain_applies", "drain_faults")}
        for fn, y in ref.items():  # the single-device engine and the oracle agree ...
            x = getattr(ora, fn)()
            assert x.tobytes() == y.tobytes(), f"single vs oracle, it {it}: {fn}"
            z = getattr(multi, fn)()  # ... and so does the sharded one, row for row
            assert z.shape == y.shape and z.tobytes() == y.tobytes(), \
                f"D={D} it {it}: {fn} differs ({len(z)} vs {len(y)} rows)"
    assert multi.counters()["decisions"] == ora.counters()["decisions"] == one.counters()["decisions"]
    assert multi.counters()["commands"] == one.counters()["commands"]


def test_rows_of_several_steps_merge_in_step_order():
    """Drain only after several steps: per step groups ascending, steps in order — the merge must not
    emit shard 0's rows of all steps before shard 1's."""
    G, R, D = 10, 3, 3
    multi, one = BatchedRaft(G, R, seed=3, device_ids=[0] * D), BatchedRaft(G, R, seed=3)
    for e in (multi, one):
        for step, groups in enumerate(([9, 0, 4], [5, 1], [8, 0, 9])):
            kind = np.full(len(groups), capi.CMD_TIMEOUT if step != 1 else capi.CMD_CLIENT_REQUEST, np.uint8)
            e.submit_columns(kind, np.array(groups, np.uint32), id=np.arange(len(groups)) + 10 * step)
            e.step(step)
    m1, mD = one.drain_messages(), multi.drain_messages()
    assert mD.tobytes() == m1.tobytes()
    assert list(mD["group"][:6]) == [0, 0, 4, 4, 9, 9]  # step 0 first, ascending groups (2 VoteRequests each)


def test_per_shard_device_blocks_and_handles():
    """Device pointers are per shard: every shard generates its slice of the synthetic stream in
    its own memory (the counter hash is keyed by the global group id) and
    jg_step_dense_acks_shards launches them all; same state as one engine on the whole stream."""
    G, R, D, T = 5000, 5, 3, 12
    multi, one = BatchedRaft(G, R, seed=SEED, device_ids=[0] * D), BatchedRaft(G, R, seed=SEED)
    for e in (multi, one):
        elect_all(e)
        e.drain_messages(), e.drain_applies()
    shards = [multi.shard(d) for d in range(multi.n_shards)]
    synths = [DeviceSynth(s) for s in shards]
    s1 = DeviceSynth(one)
    for t in range(T):
        for sy in synths:
            sy.fill(1, t)
        s1.fill(1, t)
        whole = s1.download_acks()
        for s, sy in zip(shards, synths):  # the shard's stream is the slice of the global one
            assert np.array_equal(sy.download_acks(), whole[:, s.group_lo:s.group_lo + s.G])
        multi.step_dense_acks_shards([sy.acks.value for sy in synths])
        one._check(one.api.step_dense_acks_device(one._h, s1.acks))
    compare_snapshots(multi, one, "per-shard device blocks")
    assert multi.counters()["decisions"] == one.counters()["decisions"]
    # T ticks in one launch per shard
    bufs = []
    for s in shards:
        blk = np.concatenate([synth_tick_host_slice(s, t) for t in range(T, T + 4)])
        p = s.alloc(blk.nbytes)
        s.upload(p, blk)
        bufs.append(p)
    whole = np.stack([synth_whole(one, t) for t in range(T, T + 4)])
    multi.step_dense_acks_shards([p.value for p in bufs], n_ticks=4)
    one.step_dense_acks_n(whole)
    compare_snapshots(multi, one, "T ticks per launch per shard")
    # the parent refuses device pointers, with a message that says where they go
    with pytest.raises(EngineError, match="shard"):
        p = C.c_void_p()
        multi._check(multi.api.device_alloc(multi._h, 64, C.byref(p)))
    with pytest.raises(EngineError, match="shard"):
        multi._check(multi.api.step_dense_acks_device(multi._h, bufs[0]))


def synth_whole(e, t):
    """steady-state block of tick t for the whole engine (mode 0 closed form: 1 append, acks = t)."""
    a = np.full((e.R, e.G), t, np.uint64)
    a[0] = 1
    return a


def synth_tick_host_slice(s, t):
    return synth_whole(s, t)[None]


def test_read_state_ranges_cross_shard_boundaries():
    G, R, D = 103, 3, 4
    multi, one = BatchedRaft(G, R, seed=9, device_ids=[0] * D), BatchedRaft(G, R, seed=9)
    for e in (multi, one):
        elect_all(e)
    for name in ("term", "election_timeout", "role", "commit", "match"):
        for g0, n in ((0, G), (25, 3), (26, 0), (20, 60), (102, 1)):
            assert np.array_equal(multi.read(name, 1, g0, n), one.read(name, 1, g0, n)), (name, g0, n)
    with pytest.raises(EngineError):
        multi.read("term", 0, 100, 10)


def test_a_view_survives_the_other_queues_drains():
    """jg_drain_*_view rows stay valid until the next drain of the SAME queue: draining faults or the
    other queue, and stepping, in between must neither move nor overwrite them (the INTEGRATION.md
    adapter takes both views, then iterates)."""
    for kw in ({}, {"device_ids": [0, 0]}):
        G, R = 600, 3
        e = BatchedRaft(G, R, seed=1, **kw)
        elect_all(e)
        e.apply_all(Command.ClientRequest(7))
        view = e.drain_messages(copy=False)
        snap = view.copy()
        assert len(snap) >= G
        fsm_view = e.drain_applies(copy=False)
        fsm_snap = fsm_view.copy()
        e.drain_faults()
        e.apply_all(Command.Tick(), now_ms=500)  # a few thousand new rows behind the view
        e.drain_faults()
        e.drain_applies(copy=False)
        assert view.tobytes() == snap.tobytes(), kw
        again = e.drain_messages()  # releases the view; only the new rows come out
        assert len(again) == G * R and (again["kind"] != capi.CMD_VOTE_REQUEST).all()
        assert len(fsm_snap) == G  # one Notify per leader append
