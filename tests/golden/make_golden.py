#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from tests/ref_py — the line-by-line Python transliteration
of the Rust reference, NOT from the C++ oracle the HIP engine is normally checked against.

The Rust reference cannot run here (no cargo/rustc), so these fixtures are not reference
outputs; they are the answers of a second, independent reading of the reference on fixed seeded
inputs.  tests/test_golden.py then holds the oracle, ref_py itself and the HIP engine to the
committed bytes: the oracle cannot drift silently between rounds, the device is compared against
something other than an oracle built in the same run, and the fixtures are no longer the
oracle's self-portrait.  The reference's own golden vectors (chain::compact tree, heartbeat /
vote-request expectations …) are transcribed directly in tests/test_reference_kats.py.
Inputs that are drawn "near the current state" (fuzz rows, follower mailboxes) are drawn from
the generating engine's state and stored in the fixture.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from fuzz import random_batch  # noqa: E402
from josefine_amd import capi  # noqa: E402
from josefine_amd.traces import elect_all, synth_fill_acks_host  # noqa: E402
from ref_py.engine import RefEngine as make_engine  # noqa: E402

SEED = 0x6A6F736566696E65


def digest(arr) -> str:
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def dense_fixture(G, R, mode, ticks, every):
    e = make_engine(G, R, seed=SEED + R)
    elect_all(e)
    sim = np.zeros((R, G), dtype=np.uint64)
    out = {"G": G, "R": R, "mode": mode, "ticks": ticks, "every": every, "seed": SEED + R}
    for t in range(ticks):
        e.step_dense_acks(synth_fill_acks_host(SEED + R, mode, t, 0, np.zeros(G, np.uint8), sim, R))
        if (t + 1) % every == 0:
            out[f"commit_{t+1}"] = e.read("commit")
            out[f"head_{t+1}"] = e.read("head")
            out[f"repl_{t+1}"] = e.read("repl_state")
            out[f"match_{t+1}"] = np.stack([e.read("match", r) for r in range(R)])
    out["decisions"] = e.counters()["decisions"]
    return out


def fuzz_fixture(G, R, steps, rows):
    e = make_engine(G, R, seed=99)
    rng = np.random.default_rng(4321 + R)
    budget = np.full(G, capi.CHAIN_WINDOW - 2)
    out = {"G": G, "R": R, "steps": steps}
    now = 0
    msg_h, fsm_h, flt_h = hashlib.sha256(), hashlib.sha256(), hashlib.sha256()
    for s in range(steps):
        b = random_batch(rng, e, rows, budget=budget)
        now += int(rng.integers(0, 400))
        for k, v in b.items():
            out[f"in{s}_{k}"] = v
        out[f"in{s}_now"] = now
        e.submit_columns(**b)
        e.step(now)
        msg_h.update(e.drain_messages().tobytes())
        fsm_h.update(e.drain_applies().tobytes())
        flt_h.update(e.drain_faults().tobytes())
    for name in ("term", "voted_for", "role", "commit", "head", "id_gen", "fault", "repl_state", "vote_seen",
                 "vote_granted", "election_timeout", "queued_reqs"):
        out[f"final_{name}"] = e.read(name)
    out["final_match"] = np.stack([e.read("match", r) for r in range(R)])
    out["digest_messages"] = msg_h.hexdigest()
    out["digest_applies"] = fsm_h.hexdigest()
    out["digest_faults"] = flt_h.hexdigest()
    return out


def node_fixture(G, R, rounds, ticks):
    """Dense node tick: a closed-loop cluster (digests of every mailbox column of every round +
    final state) and random leader traffic into one follower node (inputs stored, outputs digested)."""
    from dense_node import DenseCluster, random_follower_inbox

    out = {"G": G, "R": R, "rounds": rounds, "ticks": ticks}
    cl = DenseCluster(make_engine, G, R, seed=5)
    rng = np.random.default_rng(77)
    h = hashlib.sha256()
    for t in range(rounds):
        appends = rng.integers(0, 3, G).astype(np.uint64)
        out[f"appends_{t}"] = appends
        outs = cl.round(appends)
        for r in range(R):
            for k in sorted(outs[r]):
                h.update(np.ascontiguousarray(outs[r][k]).tobytes())
    out["cluster_digest"] = h.hexdigest()
    for r in range(R):
        for name in ("commit", "head", "term", "voted_for", "role", "fault"):
            out[f"cluster_{name}_{r}"] = cl.nodes[r].read(name)
    e = make_engine(G, R, seed=6, election_timeout_ms=(300, 600))
    rng = np.random.default_rng(78)
    h, now = hashlib.sha256(), 0
    for t in range(ticks):
        now += int(rng.integers(50, 260))
        inbox = random_follower_inbox(rng, G, e.node_ids, np.full(G, 1, np.uint32), e.read("head"), e.read("commit"),
                                      e.read("term"))
        for k, v in inbox.items():
            out[f"f{t}_{k}"] = v
        out[f"f{t}_now"] = now
        o = e.step_dense_follower(now, **inbox, tick=True)
        for k in sorted(o):
            h.update(np.ascontiguousarray(o[k]).tobytes())
        h.update(e.drain_messages().tobytes())
        h.update(e.drain_faults().tobytes())
    out["follower_digest"] = h.hexdigest()
    for name in ("commit", "head", "term", "voted_for", "leader_id", "role", "fault", "election_timeout", "id_gen"):
        out[f"follower_{name}"] = e.read(name)
    return out


def down_acks(R, G, t, head, per, T_quorum=120, T_back=200):
    """The ack block of tick t of the degraded-cluster trace (shared by the generator and the test):
    slot 1 is down from the start, more than a minority from T_quorum on, everybody returns at
    T_back (slot 1 of the even groups far behind, catching up in steps)."""
    acks = np.full((R, G), capi.NO_ACK, dtype=np.uint64)
    acks[0] = per
    up = [r for r in range(1, R)]
    if t < T_back:
        up = [r for r in up if r != 1]
    if T_quorum <= t < T_back:
        up = [r for r in up if r > R // 2 + 1]
    for r in up:
        acks[r] = head
    if t >= T_back:
        acks[1, ::2] = head[::2] // 2
    return acks


def down_fixture(G, R, ticks, every):
    """One follower down, then quorum lost, then recovery: the packed progress word's BEHIND escape
    and its way back."""
    e = make_engine(G, R, seed=SEED + 100 + R)
    elect_all(e)
    e.drain_messages(), e.drain_applies()
    esc = (1 << (64 // (R + 1))) - 1
    per = max(1, esc // 50)
    out = {"G": G, "R": R, "ticks": ticks, "every": every, "seed": SEED + 100 + R, "per": per}
    for t in range(ticks):
        e.step_dense_acks(down_acks(R, G, t, e.read("head").astype(np.uint64), per))
        if (t + 1) % every == 0:
            out[f"commit_{t+1}"] = e.read("commit")
            out[f"head_{t+1}"] = e.read("head")
            out[f"repl_{t+1}"] = e.read("repl_state")
            out[f"match_{t+1}"] = np.stack([e.read("match", r) for r in range(R)])
    out["decisions"] = e.counters()["decisions"]
    out["fault"] = e.read("fault")
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "down_r5.npz"), **down_fixture(192, 5, 210, 30))
    np.savez_compressed(os.path.join(HERE, "node_r3.npz"), **node_fixture(96, 3, 25, 25))
    np.savez_compressed(os.path.join(HERE, "dense_r3_ragged.npz"), **dense_fixture(512, 3, 1, 60, 20))
    np.savez_compressed(os.path.join(HERE, "dense_r5_steady.npz"), **dense_fixture(256, 5, 0, 30, 10))
    np.savez_compressed(os.path.join(HERE, "fuzz_r3.npz"), **fuzz_fixture(128, 3, 20, 400))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
