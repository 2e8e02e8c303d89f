"""BASELINE.json configs[4] at test size: dense ack stream + 1 %/tick leader failures with
re-elections through the sparse path (tests/failures.py), oracle vs HIP after every tick."""
import numpy as np
import pytest

from josefine_amd import BatchedRaft, capi
from failures import failure_rows, mix64, synth_hash
from oracle_lib import oracle_engine
from parity import compare_drains, compare_snapshots, elect_all, synth_tick_host

SEED = 0x6A6F736566696E65


def test_numpy_hash_matches_the_oracle_generator():
    """failures.py restates the counter-based hash in numpy; pin it to the C++ generator."""
    G, R = 64, 3
    ora = oracle_engine(G, R, seed=SEED)
    sim = np.zeros((R, G), dtype=np.uint64)
    acks = synth_tick_host(ora, 1, 5, sim)  # mode 1: n_append = hash(seed, tick, g, self) % 3
    assert np.array_equal(acks[0], synth_hash(SEED, 5, np.arange(G), 0) % np.uint64(3))
    assert int(mix64(np.uint64(0))) == 0xE220A8397B1DCDAF  # splitmix64 first output for state 0


def run_failure_trace(e, ticks, percent, check=None):
    elect_all(e)
    e.drain_messages(), e.drain_applies()
    gen = e if hasattr(e.api, "synth_fill_acks") else oracle_engine(e.G, e.R, seed=SEED)
    sim = np.zeros((e.R, e.G), dtype=np.uint64)
    slots = e.read("self_slot")
    n_fail = 0
    for t in range(ticks):
        e.step_dense_acks(synth_tick_host(gen, 0, t, sim))
        rows, n = failure_rows(SEED, t, 0, e.G, e.R, e.node_ids, slots, percent)
        n_fail += n
        if n:
            e.submit_columns(**rows)
            e.step(now_ms=100 * (t + 1))
        if check:
            check(t)
    return n_fail


def test_failure_trace_on_oracle_behaves_like_the_reference():
    """Re-elected leaders that had committed die on their first append (Q8); untouched groups go on."""
    G, R, ticks = 4000, 5, 40
    e = oracle_engine(G, R, seed=SEED)
    n_fail = run_failure_trace(e, ticks, percent=1)
    fault, head, commit = e.read("fault"), e.read("head"), e.read("commit")
    assert 0.5 * ticks * G / 100 < n_fail < 1.5 * ticks * G / 100
    healthy = fault == 0
    assert healthy.sum() > 0.6 * G
    never_failed = healthy & (head == ticks)
    assert (commit[never_failed] == ticks - 1).all()
    dead = fault[~healthy]
    assert set(np.unique(dead)) <= {capi.FAULT_APPEND_ID_NOT_ABOVE_HEAD, capi.FAULT_COMMIT_MISSING_BLOCK}
    assert (dead == capi.FAULT_APPEND_ID_NOT_ABOVE_HEAD).any()


@pytest.mark.gpu
@pytest.mark.parametrize("R,G,percent", [(5, 20_000, 1), (3, 20_000, 3)])
def test_failure_trace_parity(R, G, percent):
    dev = BatchedRaft(G, R, seed=SEED)
    ora = oracle_engine(G, R, seed=SEED)
    ticks = 40

    def lockstep():
        # drive both engines tick by tick with the same rows
        for e in (dev, ora):
            elect_all(e)
            e.drain_messages(), e.drain_applies()
        sim = np.zeros((R, G), dtype=np.uint64)
        slots = ora.read("self_slot")
        for t in range(ticks):
            acks = synth_tick_host(ora, 0, t, sim)
            rows, n = failure_rows(SEED, t, 0, G, R, ora.node_ids, slots, percent)
            for e in (dev, ora):
                e.step_dense_acks(acks)
                if n:
                    e.submit_columns(**rows)
                    e.step(now_ms=100 * (t + 1))
            compare_snapshots(dev, ora, f"failures R={R} tick {t}")
            compare_drains(dev, ora, f"failures R={R} tick {t}")

    lockstep()
    assert (ora.read("fault") == capi.FAULT_APPEND_ID_NOT_ABOVE_HEAD).any()
    assert dev.counters()["decisions"] == ora.counters()["decisions"]
