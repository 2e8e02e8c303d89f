"""CPU: properties of the oracle itself — the parity anchor has two ways of running the same
trace (the dense facade and explicit command rows) and a threaded leg for the CPU baseline; they
must agree with each other, and the state must keep the invariants the reference's code implies.
No GPU, no HIP library."""
import numpy as np
import pytest

from josefine_amd import capi
from josefine_amd.traces import elect_all
from oracle_lib import oracle_engine
from parity import compare_drains, compare_snapshots, synth_tick_host

FIELDS = ["commit", "head", "match", "repl_state", "fault", "id_gen", "role", "term"]


def _dense_as_rows(acks, node_ids, slots):
    """One dense tick as explicit rows: every group's appends first, then its acks in ascending
    slot order (the specification of jg_step_dense_acks, include/josefine_gpu.h)."""
    R, G = acks.shape
    gi = np.arange(G)
    kind, grp, frm, idc = [], [], [], []
    n_app = acks[slots, gi]
    for k in range(int(n_app.max()) if G else 0):
        gs = np.nonzero(n_app > k)[0]
        kind.append(np.full(len(gs), capi.CMD_CLIENT_REQUEST, np.uint8)), grp.append(gs)
        frm.append(np.zeros(len(gs), np.uint32)), idc.append(np.zeros(len(gs), np.uint64))
    for r in range(R):
        gs = np.nonzero((acks[r] != capi.NO_ACK) & (slots != r))[0]
        kind.append(np.full(len(gs), capi.CMD_APPEND_RESPONSE, np.uint8)), grp.append(gs)
        frm.append(np.full(len(gs), node_ids[r], np.uint32)), idc.append(acks[r][gs])
    return (np.concatenate(kind), np.concatenate(grp).astype(np.uint32), np.concatenate(frm),
            np.concatenate(idc).astype(np.uint64))


@pytest.mark.parametrize("R,mode", [(3, 1), (5, 1), (5, 0), (2, 1), (8, 1)])
def test_oracle_dense_facade_equals_rows(R, mode):
    G = 600
    slots = (np.arange(G) % R).astype(np.uint8)
    a = oracle_engine(G, R, seed=3, self_slots=slots)
    b = oracle_engine(G, R, seed=3, self_slots=slots)
    for e in (a, b):
        elect_all(e)
        e.drain_messages(), e.drain_applies()
    sim = np.zeros((R, G), dtype=np.uint64)
    sl = slots.astype(np.int64)
    prev_commit = a.read("commit").copy()
    prev_match = [a.read("match", r).copy() for r in range(R)]
    for t in range(40):
        acks = synth_tick_host(a, mode, t, sim)
        if t == 17:  # forged acks above the head in a majority of slots of a few groups: the panic path too
            for k in range(1, min(R // 2 + 1, R - 1) + 1):
                acks[(sl[:7] + k) % R, np.arange(7)] = 10**6
        a.step_dense_acks(acks)
        k, g, f, i = _dense_as_rows(acks, a.node_ids, sl)
        b.submit_columns(k, g, from_=f, id=i, flag=np.ones(len(k), np.uint8))
        b.step(0)
        compare_snapshots(a, b, f"R={R} tick {t}", FIELDS)
        assert a.counters()["decisions"] == b.counters()["decisions"]
        assert np.array_equal(a.drain_faults(), b.drain_faults())
        a.drain_applies(), b.drain_applies(), a.drain_messages(), b.drain_messages()
        # invariants of src/raft: the commit index never moves back and never passes the head of a
        # healthy leader; progress heads only grow (progress.rs:133-140)
        commit, head, fault = a.read("commit"), a.read("head"), a.read("fault")
        assert (commit >= prev_commit).all() and (commit[fault == 0] <= head[fault == 0]).all()
        for r in range(R):
            m = a.read("match", r)
            assert (m >= prev_match[r]).all()
            prev_match[r] = m
        prev_commit = commit
    if R - 1 >= R // 2 + 1:  # enough other slots for the majority element to be a forged one
        assert (a.read("fault")[:7] == capi.FAULT_COMMIT_MISSING_BLOCK).all()


def test_oracle_threaded_leg_equals_single_thread():
    """jo_set_threads (the all-cores leg of bench.py's cpu_baseline: threads over block
    partitions of the groups) changes nothing but the wall clock."""
    G, R = 5000, 5
    a, b = oracle_engine(G, R, seed=9), oracle_engine(G, R, seed=9)
    for e in (a, b):
        elect_all(e)
        e.drain_messages(), e.drain_applies()
    b.api.set_threads(b._h, 7)
    sa, sb = np.zeros((R, G), np.uint64), np.zeros((R, G), np.uint64)
    for t in range(30):
        a.step_dense_acks(synth_tick_host(a, 1, t, sa))
        b.step_dense_acks(synth_tick_host(b, 1, t, sb))
    compare_snapshots(a, b, "threads")
    compare_drains(a, b, "threads")
    assert a.counters() == b.counters()
