"""A seed soak of the host-compiled device source against the oracle (tests/host_compiled.py): the blind / live command
streams, the dense halves over random mailboxes and the node step under its adversarial traffic, seed after seed, both
ways of serving the dense halves, for as many seconds as asked:  python tests/soak_host_compiled.py 600
Not collected by pytest (minutes, not seconds); round 4: 572 seeds x 2 modes in 600 s, no divergence."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from josefine_amd import capi
from fuzz import random_batch, random_batch_aware
from host_compiled import HostCompiled
from oracle_lib import oracle_engine
from parity import compare_drains, compare_snapshots, elect_all
from node_step import compare_outboxes, node_traffic
from test_node_step import mixed_pair
from dense_node import random_follower_inbox, random_leader_inbox

def blind(seed, R):
    G, steps, rows = 160, 80, 700
    rng = np.random.default_rng(seed)
    slots = rng.integers(0, R, G).astype(np.uint8)
    flags = capi.CFG_SEPARATE_COMMIT_KEY if seed % 2 else 0
    kw = dict(seed=seed, self_slots=slots, flags=flags, election_timeout_ms=(300, 700))
    dev, ora = HostCompiled(G, R, **kw), oracle_engine(G, R, **kw)
    now = 0; budget = np.full(G, capi.CHAIN_WINDOW - 2)
    for s in range(steps):
        b = random_batch(rng, ora, rows, budget=budget, foreign_voters=True) if seed % 3 else random_batch_aware(rng, ora, rows, {})
        now += int(rng.integers(0, 400))
        for e in (dev, ora): e.submit_columns(**b); e.step(now)
        compare_drains(dev, ora, f"seed {seed} R {R} step {s}")
        if s % 8 == 7: compare_snapshots(dev, ora, f"seed {seed} R {R} step {s}")

def node(seed, R):
    G = 400
    flags = capi.CFG_SEPARATE_COMMIT_KEY if seed % 2 else 0
    dev, ora, rng = mixed_pair(HostCompiled, oracle_engine, G, R, seed=seed, flags=flags, election_timeout_ms=(700, 1500))
    for t in range(40):
        cols = node_traffic(rng, ora, token0=1000 * t, p_noise=0.05, p_reorder=0.15)
        outs = []
        for e in (dev, ora):
            e.submit_columns(**cols); outs.append(e.step_node(100 * (t + 1)))
        compare_outboxes(outs[0], outs[1], f"seed {seed} tick {t}")
        compare_snapshots(dev, ora, f"seed {seed} tick {t}"); compare_drains(dev, ora, f"seed {seed} tick {t}")

def halves(seed, R):
    G, ticks = 160, 40
    rng = np.random.default_rng(seed)
    slots = np.full(G, int(rng.integers(0, R)), np.uint8)
    flags = capi.CFG_SEPARATE_COMMIT_KEY if seed % 2 else 0
    kw = dict(seed=seed, self_slots=slots, flags=flags, election_timeout_ms=(300, 600))
    dev, ora = HostCompiled(G, R, **kw), oracle_engine(G, R, **kw)
    lead = rng.random(G) < 0.65
    for e in (dev, ora):
        g = np.nonzero(lead)[0].astype(np.uint32)
        e.submit_columns(np.full(len(g), capi.CMD_TIMEOUT, np.uint8), g); e.step(0)
        ids = np.array(e.node_ids, np.uint32)
        for k in range(1, R // 2 + 1):
            e.submit_columns(np.full(len(g), capi.CMD_VOTE_RESPONSE, np.uint8), g, from_=ids[(slots[g].astype(np.int64) + k) % R], term=np.ones(len(g), np.uint64), flag=np.ones(len(g), np.uint8)); e.step(0)
    for e in (dev, ora): e.drain_messages(); e.drain_applies(); e.drain_faults()
    now = 0; self_ids = np.array(ora.node_ids, np.uint32)[slots]
    for t in range(ticks):
        now += int(rng.integers(40, 260))
        acks, hh, hc = random_leader_inbox(rng, G, R, slots, ora.read("head").astype(np.uint64))
        not_led = (ora.read("role") != capi.ROLE_LEADER) | (ora.read("fault") != 0)
        acks[slots, np.arange(G)] = np.where(not_led, 0, acks[slots, np.arange(G)])
        outs = [e.step_dense_leader(now, acks, hh, hc, tick=True) for e in (dev, ora)]
        for k in outs[0]: assert np.array_equal(outs[0][k], outs[1][k]), (seed, t, k)
        compare_drains(dev, ora, f"seed {seed} L {t}")
        if R > 1:
            fin = random_follower_inbox(rng, G, ora.node_ids, self_ids, ora.read("head"), ora.read("commit"), ora.read("term"))
            outs = [e.step_dense_follower(now, **fin, tick=True) for e in (dev, ora)]
            for k in outs[0]: assert np.array_equal(outs[0][k], outs[1][k]), (seed, t, k)
        compare_drains(dev, ora, f"seed {seed} F {t}")
        if t % 5 == 4: compare_snapshots(dev, ora, f"seed {seed} {t}")

t0 = time.time(); n = 0
budget = float(sys.argv[1])
seed = 1000
while time.time() - t0 < budget:
    R = [1, 2, 3, 3, 5, 5, 4, 7, 8][seed % 9]
    for fast in (True, False):
        HostCompiled.fast = fast
        blind(seed, R); halves(seed, R)
        if R <= 5: node(seed, R)
    seed += 1; n += 1
print("soak ok:", n, "seeds x 2 modes in", int(time.time() - t0), "s")
