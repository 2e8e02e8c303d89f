"""Parity tests proper: the HIP engine, through the C ABI, against the CPU oracle
on the same seeded inputs — bit-exact on every state column and output row."""
import numpy as np
import pytest

from josefine_amd import BatchedRaft, Command, capi
from oracle_lib import oracle_engine
from parity import compare_drains, compare_snapshots, elect_all, run_dense_ticks
from fuzz import random_batch

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


@pytest.fixture(params=["1", "2"])
def dense_variant(request, monkeypatch):
    """Both dense kernels: one group per lane (8-B accesses) / two per lane (16-B accesses)."""
    monkeypatch.setenv("JG_DENSE_VARIANT", request.param)
    return request.param


@pytest.mark.parametrize("R,G,mode,ticks", [(3, 10_000, 1, 200), (5, 10_000, 1, 60), (5, 4096, 0, 40),
                                            (1, 1000, 1, 20), (2, 1000, 1, 30), (4, 1000, 1, 30),
                                            (8, 1000, 1, 30), (3, 1001, 1, 20)])
def test_dense_ack_stream_parity(R, G, mode, ticks, dense_variant):
    """BASELINE config #2 shape (10k x 3 ragged stream) and friends: compare after every tick."""
    dev, ora = pair(G, R, seed=0x6A6F7365 + R)
    for e in (dev, ora):
        elect_all(e)
    run_dense_ticks(dev, ora, mode=mode, ticks=ticks, check_every=1)
    assert int(dev.read("commit").max()) > 0


@pytest.mark.parametrize("R", [1, 3, 5])
def test_fuzz_command_stream_parity(R):
    """Random commands of every kind to every role, incl. the panic / Err paths."""
    G = 512
    dev, ora = pair(G, R, seed=99, flags=capi.CFG_SEPARATE_COMMIT_KEY if R == 5 else 0)
    rng = np.random.default_rng(1234 + R)
    now = 0
    budget = np.full(G, capi.CHAIN_WINDOW - 2)  # gaps / forks per group: stay inside the segment window
    for step in range(60):
        batch = random_batch(rng, ora, 1500, budget=budget)
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
