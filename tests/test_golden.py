"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from
tests/ref_py, the second restatement of the reference): the C++ oracle and ref_py must reproduce
them (CPU), and the HIP engine must match the committed bytes (-m gpu)."""
import hashlib
import os

import numpy as np
import pytest

from josefine_amd import BatchedRaft
from oracle_lib import oracle_engine
from josefine_amd.traces import synth_fill_acks_host
from parity import elect_all

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# "device source on the host": the device's state machine and dense kernels' per-group logic compiled by g++ (tests/host_compiled.py)
BACKENDS = ["oracle", "ref_py", "device source on the host", pytest.param("hip", marks=pytest.mark.gpu)]


def factory(backend):
    if backend == "ref_py":
        from ref_py.engine import RefEngine
        return RefEngine
    if backend == "device source on the host":
        from host_compiled import HostCompiled
        return HostCompiled
    return oracle_engine if backend == "oracle" else BatchedRaft


def make(backend, G, R, **kw):
    return factory(backend)(G, R, **kw)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", ["dense_r3_ragged.npz", "dense_r5_steady.npz"])
def test_dense_golden(backend, name):
    z = np.load(os.path.join(HERE, name))
    G, R, mode, ticks, every = (int(z[k]) for k in ("G", "R", "mode", "ticks", "every"))
    e = make(backend, G, R, seed=int(z["seed"]))
    elect_all(e)
    sim = np.zeros((R, G), dtype=np.uint64)
    for t in range(ticks):
        e.step_dense_acks(synth_fill_acks_host(int(z["seed"]), mode, t, 0, np.zeros(G, np.uint8), sim, R))
        if (t + 1) % every == 0:
            assert np.array_equal(e.read("commit"), z[f"commit_{t+1}"]), t
            assert np.array_equal(e.read("head"), z[f"head_{t+1}"]), t
            assert np.array_equal(e.read("repl_state"), z[f"repl_{t+1}"]), t
            for r in range(R):
                assert np.array_equal(e.read("match", r), z[f"match_{t+1}"][r]), (t, r)
    assert e.counters()["decisions"] == int(z["decisions"])


@pytest.mark.parametrize("backend", BACKENDS)
def test_degraded_cluster_golden(backend):
    """One follower down -> quorum lost -> everybody back (golden/down_r5.npz): progress heads and
    the commit index leave their lag fields (BEHIND escape) and return."""
    import sys
    sys.path.insert(0, HERE)
    from make_golden import down_acks

    z = np.load(os.path.join(HERE, "down_r5.npz"))
    G, R, ticks, every, per = (int(z[k]) for k in ("G", "R", "ticks", "every", "per"))
    e = make(backend, G, R, seed=int(z["seed"]))
    elect_all(e)
    e.drain_messages(), e.drain_applies()
    for t in range(ticks):
        e.step_dense_acks(down_acks(R, G, t, e.read("head").astype(np.uint64), per))
        if (t + 1) % every == 0:
            assert np.array_equal(e.read("commit"), z[f"commit_{t+1}"]), t
            assert np.array_equal(e.read("head"), z[f"head_{t+1}"]), t
            assert np.array_equal(e.read("repl_state"), z[f"repl_{t+1}"]), t
            for r in range(R):
                assert np.array_equal(e.read("match", r), z[f"match_{t+1}"][r]), (t, r)
    assert e.counters()["decisions"] == int(z["decisions"])
    assert np.array_equal(e.read("fault"), z["fault"]) and not z["fault"].any()
    esc = (1 << (64 // (R + 1))) - 1
    lag1 = z["head_180"].astype(np.int64) - z["match_180"][1].astype(np.int64)
    clag = z["head_180"].astype(np.int64) - z["commit_180"].astype(np.int64)
    assert (lag1 > esc).all() and (clag > esc).all()        # both left their fields on the way ...
    assert (z["head_210"] - z["commit_210"] < 4 * per).all()  # ... and the commit index came back


@pytest.mark.parametrize("backend", BACKENDS)
def test_fuzz_golden(backend):
    z = np.load(os.path.join(HERE, "fuzz_r3.npz"))
    G, R, steps = int(z["G"]), int(z["R"]), int(z["steps"])
    e = make(backend, G, R, seed=99)
    msg_h, fsm_h, flt_h = hashlib.sha256(), hashlib.sha256(), hashlib.sha256()
    for s in range(steps):
        cols = {k: z[f"in{s}_{k}"] for k in ("kind", "group", "from_", "term", "id", "aux", "flag", "blk_id", "blk_next")}
        e.submit_columns(**cols)
        e.step(int(z[f"in{s}_now"]))
        msg_h.update(e.drain_messages().tobytes())
        fsm_h.update(e.drain_applies().tobytes())
        flt_h.update(e.drain_faults().tobytes())
    for name in ("term", "voted_for", "role", "commit", "head", "id_gen", "fault", "repl_state", "vote_seen",
                 "vote_granted", "election_timeout", "queued_reqs"):
        assert np.array_equal(e.read(name), z[f"final_{name}"]), name
    for r in range(R):
        assert np.array_equal(e.read("match", r), z["final_match"][r])
    assert msg_h.hexdigest() == str(z["digest_messages"])
    assert fsm_h.hexdigest() == str(z["digest_applies"])
    assert flt_h.hexdigest() == str(z["digest_faults"])


@pytest.mark.parametrize("backend", BACKENDS)
def test_node_tick_golden(backend):
    """Dense node tick (jg_step_dense_leader / jg_step_dense_follower): closed-loop cluster and
    random follower traffic against the committed digests."""
    from dense_node import DenseCluster

    z = np.load(os.path.join(HERE, "node_r3.npz"))
    G, R, rounds, ticks = int(z["G"]), int(z["R"]), int(z["rounds"]), int(z["ticks"])
    cl = DenseCluster(factory(backend), G, R, seed=5)
    h = hashlib.sha256()
    for t in range(rounds):
        outs = cl.round(z[f"appends_{t}"])
        for r in range(R):
            for k in sorted(outs[r]):
                h.update(np.ascontiguousarray(outs[r][k]).tobytes())
    assert h.hexdigest() == str(z["cluster_digest"])
    for r in range(R):
        for name in ("commit", "head", "term", "voted_for", "role", "fault"):
            assert np.array_equal(cl.nodes[r].read(name), z[f"cluster_{name}_{r}"]), (r, name)
    e = make(backend, G, R, seed=6, election_timeout_ms=(300, 600))
    h = hashlib.sha256()
    for t in range(ticks):
        inbox = {k: z[f"f{t}_{k}"] for k in ("leader", "term", "hb_commit", "ae_from", "ae_n")}
        o = e.step_dense_follower(int(z[f"f{t}_now"]), **inbox, tick=True)
        for k in sorted(o):
            h.update(np.ascontiguousarray(o[k]).tobytes())
        h.update(e.drain_messages().tobytes())
        h.update(e.drain_faults().tobytes())
    assert h.hexdigest() == str(z["follower_digest"])
    for name in ("commit", "head", "term", "voted_for", "leader_id", "role", "fault", "election_timeout", "id_gen"):
        assert np.array_equal(e.read(name), z[f"follower_{name}"]), name
