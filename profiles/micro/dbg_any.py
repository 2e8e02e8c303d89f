import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from josefine_amd import BatchedRaft, capi, DenseCluster as LibCluster
from dense_node import AnyLeaderCluster, any_failure_rows
from test_any_leader import spread_leaders
from oracle_lib import oracle_engine
from parity import compare_snapshots
R, percent, dual = 3, 3, 9
G, T = 3000, 45
ora = AnyLeaderCluster(oracle_engine, G, R, seed=5)
nodes = [BatchedRaft(G, R, seed=5 + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY) for r in range(R)]
spread_leaders(ora.nodes, G, R, dual_every=dual); spread_leaders(nodes, G, R, dual_every=dual)
lib = LibCluster(nodes, lead=None); lib.set_appends(1)
leader_of = np.arange(G) % R
failed = np.zeros(G, bool)
bad = None
for t in range(T):
    inj, failing = any_failure_rows(99, t, G, R, percent, leader_of, whole_group=True, skip=failed) if t >= 3 else ([None] * R, [])
    failed[failing] = True
    ora.round(np.ones(G, np.uint64), inject=inj)
    up = [None if c is None else nodes[n].upload_rows(**c) for n, c in enumerate(inj)]
    st = lib.round_routed((t + 1) * 100, up)
    want = [sum(len(r) for _, r in ora.inbound[n]) for n in range(R)]
    print(t, st["delivered"], want, st["kept"])
    if bad is not None:
        for n in range(R):
            for f in ("role","term","head","commit","voted_for","fault","leader_id"):
                a, b = nodes[n].read(f), ora.nodes[n].read(f)
                d = np.nonzero(a != b)[0]
                if len(d): print("node", n, f, d[:10], a[d[:10]], b[d[:10]])
        for n in range(R):
            for src, rows in bad[n]:
                pass
        break
    if st["delivered"] != want:
        bad = [list(x) for x in ora.inbound]
        for n in range(R):
            got, wantk = nodes[n].drain_messages(), ora.kept[n]
            print("kept node", n, len(got), len(wantk), "dev kinds", dict(zip(*[x.tolist() for x in np.unique(got["kind"], return_counts=True)])),
                  "ora kinds", dict(zip(*[x.tolist() for x in np.unique(wantk["kind"], return_counts=True)])))
            a = set(map(tuple, got.tolist())); b = set(map(tuple, wantk.tolist()))
            print("  only dev", sorted(a - b)[:6]); print("  only ora", sorted(b - a)[:6])
        # the spec's inbound Heartbeat rows whose senders are slot 2 / 1: which groups
        for n in range(R):
            for src, rows in ora.inbound[n]:
                hb = rows[rows["kind"] == 6]
                print(" spec HB to", n, "from", src, "groups", hb["group"][:40].tolist())
        # which groups have inbound rows in the spec, by kind
        for n in range(R):
            for src, rows in ora.inbound[n]:
                ks, cs = np.unique(rows["kind"], return_counts=True)
                print(" to", n, "from", src, dict(zip(ks.tolist(), cs.tolist())))
        own = ora.owner
        roles = [e.read("role") for e in ora.nodes]
        lose = [(roles[n] == capi.ROLE_LEADER) & (own != n) for n in range(R)]
        print("losers per node", [int(x.sum()) for x in lose], "groups", [np.nonzero(x)[0][:8] for x in lose])
        for n in range(R):
            gs = np.nonzero(lose[n])[0][:5]
            for g in gs:
                print("  node", n, "g", g, "dev role/term/head/fault", nodes[n].read("role")[g], nodes[n].read("term")[g], nodes[n].read("head")[g], nodes[n].read("fault")[g],
                      "ora", ora.nodes[n].read("role")[g], ora.nodes[n].read("term")[g], ora.nodes[n].read("head")[g], ora.nodes[n].read("fault")[g], "failed", failed[g], "owner", own[g])
