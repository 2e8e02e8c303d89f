"""Chain::compact (src/raft/chain.rs:239-253) on the engine's OWN chains (jg_chain_compact_resident):
the walk over the keys below the commit index, its quirk Q7 (the parent pointer of a removed block is
followed too, so only the top of a dead branch goes), the rows the host deletes from its store — and
that the id set afterwards behaves like the reference's (extends from removed parents fail, ranges and
later compactions see the holes).  Three implementations: tests/ref_py (line-by-line from the Rust),
the C++ oracle, the HIP engine."""
import numpy as np
import pytest

from josefine_amd import BatchedRaft, Command, capi
from fuzz import random_batch
from oracle_lib import oracle_engine
from parity import compare_snapshots
from ref_py.engine import RefEngine

BACKENDS = ["oracle", "ref_py", pytest.param("hip", marks=pytest.mark.gpu)]


def make(backend, G, R, **kw):
    return {"oracle": oracle_engine, "ref_py": RefEngine}.get(backend, BatchedRaft)(G, R, **kw)


@pytest.mark.parametrize("backend", BACKENDS)
def test_reference_vector_on_a_resident_chain(backend):  # chain.rs:327-343, through a follower's own chain
    e = make(backend, 1, 3)
    h = e.handle(0)
    h.apply(Command.AppendEntries(1, 2, [(1, 0), (2, 1), (3, 2), (4, 3), (5, 3), (6, 5)]))
    h.apply(Command.Heartbeat(1, 6, 2))
    assert h.commit == 6
    rows = e.chain_compact_resident()
    assert [(int(r["group"]), int(r["id"])) for r in rows] == [(0, 4)]      # assert!(!chain.has(4))
    h.apply(Command.AppendEntries(1, 2, [(9, 4)]))                            # its parent is gone now
    assert h.fault == capi.FAULT_EXTEND_MISSING_PARENT
    assert len(e.chain_compact_resident()) == 0                               # dead groups are left alone


@pytest.mark.parametrize("backend", BACKENDS)
def test_q7_only_the_top_of_a_dead_branch_goes(backend):
    """main 0<-1<-2<-5<-6, dead 2<-3<-4, commit 6: the walk removes 4 (it is not 5's parent), then
    follows 4's own parent pointer to 3 and keeps it.  A second pass sees 5 -> 2 again, removes 3."""
    e = make(backend, 2, 3)
    for g in (0, 1):
        e.submit(g, Command.AppendEntries(1, 2, [(1, 0), (2, 1), (3, 2), (4, 3), (5, 2), (6, 5)]))
        e.submit(g, Command.Heartbeat(1, 6, 2))
    e.step()
    first = e.chain_compact_resident()
    assert [(int(r["group"]), int(r["id"])) for r in first] == [(0, 4), (1, 4)]
    second = e.chain_compact_resident()
    assert [(int(r["group"]), int(r["id"])) for r in second] == [(0, 3), (1, 3)]
    assert len(e.chain_compact_resident()) == 0
    assert not e.read("fault").any()


@pytest.mark.parametrize("backend", ["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("R", [3, 5])
def test_compaction_in_the_middle_of_random_traffic(backend, R):
    """Random command streams (forks, gaps, re-sent blocks, elections, restarts) with a compaction
    every few steps: rows removed, every state column and every later output row against ref_py."""
    G = 160
    rng = np.random.default_rng(50 + R)
    dev, ref = make(backend, G, R, seed=4), RefEngine(G, R, seed=4)
    budget = np.full(G, capi.CHAIN_WINDOW - 2)
    now, removed = 0, 0
    for s in range(60):
        b = random_batch(rng, ref, 500, budget=budget, foreign_voters=True)
        now += int(rng.integers(0, 300))
        for e in (dev, ref):
            e.submit_columns(**b)
            e.step(now)
        for fn in ("drain_messages", "drain_applies", "drain_faults"):
            assert getattr(dev, fn)().tobytes() == getattr(ref, fn)().tobytes(), (s, fn)
        if s % 4 == 3:
            a, c = dev.chain_compact_resident(), ref.chain_compact_resident()
            assert a.tobytes() == c.tobytes(), f"step {s}: removed rows differ ({len(a)} vs {len(c)})"
            removed += len(a)
            compare_snapshots(dev, ref, f"after compaction at step {s}")
    compare_snapshots(dev, ref, "final")
    assert removed > 20


@pytest.mark.gpu
def test_compaction_on_a_sharded_engine():
    G, R = 90, 3
    multi, one = BatchedRaft(G, R, seed=1, device_ids=[0, 0, 0]), BatchedRaft(G, R, seed=1)
    for e in (multi, one):
        for g in range(0, G, 2):
            e.submit(g, Command.AppendEntries(1, 2, [(1, 0), (2, 1), (3, 2), (4, 3), (5, 3), (6, 5), (7, 5), (8, 7)]))
            e.submit(g, Command.Heartbeat(1, 8, 2))
        e.step()
    a, b = multi.chain_compact_resident(), one.chain_compact_resident()
    assert a.tobytes() == b.tobytes() and len(a) == 2 * (G // 2)
    compare_snapshots(multi, one, "sharded compaction")
