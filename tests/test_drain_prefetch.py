"""jg_drain_prefetch / jg_drain_flush (pipelined drains): the compaction + transfer of a batch of
steps runs on a second stream, driven by the engine's drain thread, while the engine keeps
stepping; the drains deliver what has landed and never block.  Whatever the timing and the
batching, the concatenation of everything delivered must be the synchronous path's (the
oracle's) rows, bit for bit and in the same order."""
import numpy as np
import pytest

from josefine_amd import BatchedRaft, Command, capi
from dense_node import random_follower_inbox
from failures import failure_rows
from fuzz import random_batch
from oracle_lib import oracle_engine
from parity import compare_snapshots, elect_all, synth_tick_host

pytestmark = pytest.mark.gpu
FNS = ("drain_messages", "drain_applies", "drain_faults")


def drain_all(e):
    return {fn: [getattr(e, fn)()] for fn in FNS}


@pytest.mark.parametrize("R,D", [(3, 1), (5, 1), (3, 3)])
def test_pipelined_drains_deliver_the_same_rows(R, D):
    G = 3000
    rng = np.random.default_rng(R * 10 + D)
    kw = dict(seed=31, election_timeout_ms=(300, 700))
    pipe = BatchedRaft(G, R, **kw) if D == 1 else BatchedRaft(G, R, device_ids=[0] * D, **kw)
    ora = oracle_engine(G, R, **kw)
    got = {fn: [] for fn in FNS}
    want = {fn: [] for fn in FNS}
    for e in (pipe, ora):
        elect_all(e)
    slots = ora.read("self_slot")
    sim = np.zeros((R, G), np.uint64)
    budget = np.full(G, 3)
    now = 0
    for it in range(40):
        now += 90
        acks = synth_tick_host(ora, 1, it, sim)
        rows, n = failure_rows(5, it, 0, G, R, ora.node_ids, slots, 3)
        batch = random_batch(rng, ora, 500, budget=budget)
        fin = random_follower_inbox(rng, G, ora.node_ids, np.array(ora.node_ids, np.uint32)[slots], ora.read("head"),
                                    ora.read("commit"), ora.read("term"))
        for e in (pipe, ora):
            e.step_dense_acks(acks)
            if n:
                e.submit_columns(**rows)
                e.step(now)
            e.submit_columns(**batch)
            e.step(now + 1)
            if D == 1 and it % 3 == 0:   # exceptional rows of a dense node step, merged by step number
                e.step_dense_follower(now + 2, **fin, tick=True)
        for fn in FNS:
            want[fn].append(getattr(ora, fn)())
        if it % 5 == 4:                  # deliver whatever has landed, start the next batch (if the slot is free)
            for fn in FNS:
                got[fn].append(getattr(pipe, fn)())
            pipe.drain_prefetch()
        elif it % 5 == 2 and it > 5:     # drains between two prefetch points: never block, never lose or repeat rows
            for _ in range(2):
                for fn in FNS:
                    got[fn].append(getattr(pipe, fn)())
        elif it == 21:                   # a blocking flush in the middle
            pipe.drain_flush()
    pipe.drain_flush()
    for fn in FNS:
        got[fn].append(getattr(pipe, fn)())
        a, b = np.concatenate(got[fn]), np.concatenate(want[fn])
        assert a.shape == b.shape and a.tobytes() == b.tobytes(), (fn, len(a), len(b))
    compare_snapshots(pipe, ora, "after pipelined drains")
    assert pipe.counters()["decisions"] == ora.counters()["decisions"]


def test_prefetch_then_drain_is_the_synchronous_drain():
    G, R = 500, 3
    a, b = BatchedRaft(G, R, seed=2), BatchedRaft(G, R, seed=2)
    for e in (a, b):
        elect_all(e)
        e.apply_all(Command.ClientRequest(3))
        e.apply_all(Command.Tick(), now_ms=400)
    b.drain_flush()
    for fn in FNS:
        x, y = getattr(a, fn)(), getattr(b, fn)()
        assert x.tobytes() == y.tobytes() and (fn == "drain_faults" or len(x) > 0), fn
    # views of a pipelined engine stay put while later batches land behind them
    b.apply_all(Command.Tick(), now_ms=900)
    b.drain_flush()
    view = b.drain_messages(copy=False)
    snap = view.copy()
    assert len(snap) == G * R
    b.apply_all(Command.Tick(), now_ms=1400)
    b.drain_prefetch()
    b.drain_applies(), b.drain_faults()
    b.drain_flush()
    b.drain_applies(), b.drain_faults()
    assert view.tobytes() == snap.tobytes()
    assert len(b.drain_messages()) == G * R


@pytest.mark.parametrize("D", [1, 3])
def test_view_while_a_batch_is_in_transfer_repeats_nothing(D):
    """Take a non-empty view, start a large batch, drain the SAME queue's view again at once (the
    batch has most likely not landed: nothing new, the first view stays valid), flush, drain: every
    row is delivered exactly once, in order - on one engine and on a sharded one (the router used to
    hand the outstanding view's rows back to its queue and deliver them a second time)."""
    G, R = 60_000, 3
    kw = dict(seed=9)
    e = BatchedRaft(G, R, **kw) if D == 1 else BatchedRaft(G, R, device_ids=[0] * D, **kw)
    ora = oracle_engine(G, R, **kw)
    for x in (e, ora):
        elect_all(x)
        x.drain_messages(), x.drain_applies(), x.drain_faults()
    want_m, want_f = [], []
    got_m, got_f = [], []

    def step(now):
        for x in (e, ora):
            x.apply_all(Command.ClientRequest(7))
            x.apply_all(Command.Tick(), now_ms=now)
        want_m.append(ora.drain_messages())
        want_f.append(ora.drain_applies())

    step(400)
    e.drain_flush()                            # pipelined from here on; batch A has landed
    view_m = e.drain_messages(copy=False)
    view_f = e.drain_applies(copy=False)
    snap_m, snap_f = view_m.copy(), view_f.copy()
    assert len(snap_m) == G * R and len(snap_f) >= G
    got_m.append(snap_m), got_f.append(snap_f)
    for k in range(3):                         # a large batch B ...
        step(900 + 500 * k)
    e.drain_prefetch()                         # ... in transfer
    again_m = e.drain_messages(copy=False)     # the same queue, immediately
    again_f = e.drain_applies(copy=False)
    got_m.append(again_m.copy()), got_f.append(again_f.copy())
    if len(again_m) == 0:                      # (not landed yet: the first view must not have moved)
        assert view_m.tobytes() == snap_m.tobytes()
    if len(again_f) == 0:
        assert view_f.tobytes() == snap_f.tobytes()
    e.drain_flush()
    got_m.append(e.drain_messages()), got_f.append(e.drain_applies())
    got_m.append(e.drain_messages()), got_f.append(e.drain_applies())  # and nothing after that
    for got, want in ((got_m, want_m), (got_f, want_f)):
        a, b = np.concatenate(got), np.concatenate(want)
        assert a.shape == b.shape and a.tobytes() == b.tobytes(), (len(a), len(b))
