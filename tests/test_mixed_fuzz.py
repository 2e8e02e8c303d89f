"""All entry points interleaved on one engine: random sparse batches (every Command kind to
every role, incl. the panic / Err paths), dense ack ticks (single and fused), leader halves and
follower halves of the dense node tick — state columns, mailbox columns and drained rows
against the oracle after every call.  This is where the implicit columns (RUN / FAST chains, the
delta-packed progress + commit word) and the ordering of exceptional rows between sparse and
dense steps get exercised against each other.  JG_SOAK=<n> multiplies the iteration count."""
import os

import numpy as np
import pytest

from josefine_amd import BatchedRaft, capi
from oracle_lib import oracle_engine
from parity import compare_drains, compare_snapshots
from fuzz import random_batch
from dense_node import random_follower_inbox, random_leader_inbox

pytestmark = pytest.mark.gpu
SOAK = int(os.environ.get("JG_SOAK", "1"))


def _cmp_cols(a, b, what):
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{what}: outbox column {k} differs at {np.argwhere(a[k] != b[k])[:6].tolist()}"


@pytest.mark.parametrize("R,flags,seed", [(3, capi.CFG_SEPARATE_COMMIT_KEY, 1), (5, capi.CFG_SEPARATE_COMMIT_KEY, 2),
                                          (5, 0, 3), (2, capi.CFG_SEPARATE_COMMIT_KEY, 4), (8, capi.CFG_SEPARATE_COMMIT_KEY, 5)])
def test_interleaved_entry_points(R, flags, seed):
    G = 700
    slots = (np.arange(G) % R).astype(np.uint8) if seed % 2 else None
    kw = dict(seed=seed, flags=flags, self_slots=slots, election_timeout_ms=(300, 700))
    dev, ora = BatchedRaft(G, R, **kw), oracle_engine(G, R, **kw)
    rng = np.random.default_rng(1000 + seed)
    # forks / gaps per group over the whole run: a fork can cost two chain segments (isolate the
    # overwritten id, split what follows), so stay well inside the engine's JG_CHAIN_WINDOW
    budget = np.full(G, capi.CHAIN_WINDOW // 2 - 1)
    now = 0
    sl = ora.read("self_slot")
    self_ids = np.array(ora.node_ids, dtype=np.uint32)[sl]
    n_leader_its = 0
    for it in range(60 * SOAK):
        now += int(rng.integers(0, 250))
        op = rng.choice(["sparse", "acks", "acks_n", "leader", "follower", "elect"], p=[0.25, 0.2, 0.1, 0.2, 0.15, 0.1])
        what = f"R={R} seed={seed} it={it} {op}"
        if op == "elect":
            # a third of the groups: process restart, Timeout with voted_for == None (the only way to
            # campaign, Q4), granted votes from the next R/2 slots -> leaders (often with an id_gen
            # that no longer matches their head: irregular chains for the dense paths)
            g = np.nonzero(rng.random(G) < 0.33)[0].astype(np.uint32)
            n = len(g)
            ids = np.array(ora.node_ids, dtype=np.uint32)
            kind = [np.full(n, capi.CMD_RESTART, np.uint8), np.full(n, capi.CMD_TIMEOUT, np.uint8)]
            frm = [np.zeros(n, np.uint32), np.zeros(n, np.uint32)]
            for k in range(1, R // 2 + 1):
                kind.append(np.full(n, capi.CMD_VOTE_RESPONSE, np.uint8))
                frm.append(ids[(sl[g].astype(np.int64) + k) % R])
            kind, frm = np.concatenate(kind), np.concatenate(frm)
            grp = np.tile(g, len(kind) // max(n, 1)) if n else g
            for e in (dev, ora):
                e.submit_columns(kind, grp, from_=frm, term=np.ones(len(kind), np.uint64), flag=np.ones(len(kind), np.uint8))
                e.step(now)
        elif op == "sparse":
            batch = random_batch(rng, ora, 900, budget=budget, foreign_voters=True)
            for e in (dev, ora):
                e.submit_columns(**batch)
                e.step(now)
        elif op in ("acks", "acks_n"):
            T = 1 if op == "acks" else int(rng.integers(2, 5))
            acks = np.stack([random_leader_inbox(rng, G, R, sl, ora.read("head"))[0] for _ in range(T)])
            # only leaders may be asked to append on the dense path (else: engine fault, also compared)
            if rng.random() < 0.8:
                lead = ora.read("role") == capi.ROLE_LEADER
                for t in range(T):
                    acks[t][sl, np.arange(G)] = np.where(lead, acks[t][sl, np.arange(G)], 0)
            for e in (dev, ora):
                if T == 1:
                    e.step_dense_acks(acks[0])
                else:
                    e.step_dense_acks_n(acks)
        elif op == "leader":
            acks, hbr_has, hbr_commit = random_leader_inbox(rng, G, R, sl, ora.read("head"))
            lead = ora.read("role") == capi.ROLE_LEADER
            acks[sl, np.arange(G)] = np.where(lead, acks[sl, np.arange(G)], 0)
            tick = bool(rng.random() < 0.8)
            if rng.random() < 0.3:
                acks = None
            oa = dev.step_dense_leader(now, acks, hbr_has, hbr_commit, tick=tick)
            ob = ora.step_dense_leader(now, acks, hbr_has, hbr_commit, tick=tick)
            if tick:
                _cmp_cols(oa, ob, what)
        else:
            inbox = random_follower_inbox(rng, G, ora.node_ids, self_ids, ora.read("head"), ora.read("commit"),
                                          ora.read("term"))
            tick = bool(rng.random() < 0.8)
            oa = dev.step_dense_follower(now, **inbox, tick=tick)
            ob = ora.step_dense_follower(now, **inbox, tick=tick)
            _cmp_cols(oa, ob, what)
        compare_snapshots(dev, ora, what)
        n_leader_its += int((ora.read("role") == capi.ROLE_LEADER).sum() > G // 10)
        if rng.random() < 0.6:  # sometimes let rows of several calls pile up before draining
            compare_drains(dev, ora, what)
    compare_drains(dev, ora, "final")
    assert dev.counters()["decisions"] == ora.counters()["decisions"]
    assert n_leader_its > 5 * SOAK and (ora.read("fault") != 0).any()
