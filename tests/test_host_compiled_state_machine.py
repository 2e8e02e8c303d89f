"""The device's general state machine (jg_device.h: jg_load / jg_apply / jg_store) compiled for the HOST
(tests/host_compiled.py) and fuzzed against the oracle in the CPU suite: the blind and the live command streams of the
GPU parity suite - every role, every Command kind, forks / gaps / re-sent blocks, forged and foreign senders, restarts,
the reference's panic paths - every drained row byte for byte and every state column.  No GPU: what is checked is the
device SOURCE of the state machine, not its gfx950 code object (that is the GPU suite's)."""
import numpy as np
import pytest

from josefine_amd import capi
from fuzz import assert_live, random_batch, random_batch_aware
from host_compiled import HostCompiled
from oracle_lib import oracle_engine
from parity import compare_drains, compare_snapshots, elect_all


def pair(G, R, **kw):
    return HostCompiled(G, R, **kw), oracle_engine(G, R, **kw)


@pytest.mark.parametrize("R", [1, 2, 3, 5, 8])
def test_blind_command_streams(R):
    G, steps, rows = 192, 120, 800
    rng = np.random.default_rng(17000 + R)
    slots = rng.integers(0, R, G).astype(np.uint8)
    flags = capi.CFG_SEPARATE_COMMIT_KEY if R % 2 == 0 else 0
    dev, ora = pair(G, R, seed=R, self_slots=slots, flags=flags, election_timeout_ms=(300, 700))
    compare_snapshots(dev, ora, "RaftHandle::new")
    now = 0
    budget = np.full(G, capi.CHAIN_WINDOW - 2)  # gaps / forks per group: stay inside the engine's segment window (the oracle has none)
    for s in range(steps):
        b = random_batch(rng, ora, rows, budget=budget, foreign_voters=True)
        now += int(rng.integers(0, 400))
        for e in (dev, ora):
            e.submit_columns(**b)
            e.step(now)
        compare_drains(dev, ora, f"R={R} step {s}")
        if s % 8 == 7 or s == steps - 1:
            compare_snapshots(dev, ora, f"R={R} step {s}")
    assert dev.counters()["decisions"] == ora.counters()["decisions"] > 0


@pytest.mark.parametrize("R", [1, 3, 4, 5, 7])
def test_live_command_streams(R):
    G, steps, rows = 192, 100, 800
    rng = np.random.default_rng(19000 + R)
    slots = rng.integers(0, R, G).astype(np.uint8)
    flags = capi.CFG_SEPARATE_COMMIT_KEY if R % 2 == 1 else 0
    dev, ora = pair(G, R, seed=R, self_slots=slots, flags=flags, election_timeout_ms=(300, 700))
    stats = {}
    now = 0
    for s in range(steps):
        b = random_batch_aware(rng, ora, rows, stats)
        now += int(rng.integers(0, 200))
        for e in (dev, ora):
            e.submit_columns(**b)
            e.step(now)
        compare_drains(dev, ora, f"R={R} step {s}")
        if s % 8 == 7 or s == steps - 1:
            compare_snapshots(dev, ora, f"R={R} step {s}")
    assert dev.counters()["decisions"] == ora.counters()["decisions"]
    assert_live(stats, ora.counters()["decisions"], R)


def test_restarted_reelected_leader_keeps_its_lags_below_the_top_of_its_run():
    """jg_load / jg_store's lag base (jg_lane_base): a leader restarted with commit index 2 under a run [0, 5] and re-elected
    - head 2, the run above it still there, progress heads 0 (Q10) - then acknowledgements ABOVE its head through the
    general state machine: state after every command as the oracle's, the commit index moving past the head."""
    G, R = 64, 3
    dev, ora = pair(G, R, seed=3, flags=capi.CFG_SEPARATE_COMMIT_KEY)
    g = np.arange(G, dtype=np.uint32)
    for e in (dev, ora):
        elect_all(e)
        for k in range(5):
            e.submit_columns(np.full(G, capi.CMD_CLIENT_REQUEST, np.uint8), g, id=np.arange(G, dtype=np.uint64) + 100 * k)
        e.step(10)
        for r in (2,):
            e.submit_columns(np.full(G, capi.CMD_APPEND_RESPONSE, np.uint8), g, from_=np.full(G, r, np.uint32), term=np.ones(G, np.uint64),
                             id=np.full(G, 2, np.uint64), flag=np.ones(G, np.uint8))
        e.step(20)
    compare_snapshots(dev, ora, "leaders at head 5, commit 2")
    assert (ora.read("head") == 5).all() and (ora.read("commit") == 2).all()
    script = [(capi.CMD_RESTART, {}), (capi.CMD_TIMEOUT, {}),
              (capi.CMD_VOTE_RESPONSE, dict(from_=2, term=1, flag=1)),
              (capi.CMD_APPEND_RESPONSE, dict(from_=2, term=1, id=4, flag=1)),
              (capi.CMD_APPEND_RESPONSE, dict(from_=3, term=1, id=5, flag=1)),
              (capi.CMD_TICK, {}),
              (capi.CMD_APPEND_RESPONSE, dict(from_=2, term=1, id=6, flag=1)),   # above the top: recorded
              (capi.CMD_APPEND_RESPONSE, dict(from_=3, term=1, id=7, flag=1))]   # the majority above the top: the panic
    for t, (kind, kw) in enumerate(script):
        cols = {k: np.full(G, v, {"from_": np.uint32, "term": np.uint64, "id": np.uint64, "flag": np.uint8}[k]) for k, v in kw.items()}
        for e in (dev, ora):
            e.submit_columns(np.full(G, kind, np.uint8), g, **cols)
            e.step(300 + 150 * t)
        compare_drains(dev, ora, f"command {t}")
        compare_snapshots(dev, ora, f"command {t}")
        if t == 4:
            assert (ora.read("role") == capi.ROLE_LEADER).all() and (ora.read("head") == 2).all() and (ora.read("commit") == 4).all()
    assert (ora.read("fault") == capi.FAULT_COMMIT_MISSING_BLOCK).all()
