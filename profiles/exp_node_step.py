"""jg_step_node throughput probe: a node that leads G partitions (R replicas), steady state - per tick and
partition one ClientRequest, an AppendResponse from every follower (the previous tick's head) and, every
other tick, a HeartbeatResponse from every follower; rows shuffled.  Prints ms per tick by phase.
usage: python profiles/exp_node_step.py [G] [R] [ticks]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from josefine_amd import BatchedRaft, capi  # noqa: E402
from josefine_amd.traces import elect_all  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
T = int(sys.argv[3]) if len(sys.argv) > 3 else 30
e = BatchedRaft(G, R, seed=1, flags=capi.CFG_SEPARATE_COMMIT_KEY)
elect_all(e)
e.drain_messages(), e.drain_applies(), e.drain_faults()
ids = np.array(e.node_ids, np.uint32)
rng = np.random.default_rng(0)
g = np.arange(G, dtype=np.uint32)


def batch(t):
    kind = [np.full(G, capi.CMD_CLIENT_REQUEST, np.uint8)]
    group = [g]
    frm = [np.zeros(G, np.uint32)]
    idc = [g.astype(np.uint64) + np.uint64(t * G)]
    flag = [np.zeros(G, np.uint8)]
    for r in range(1, R):
        kind.append(np.full(G, capi.CMD_APPEND_RESPONSE, np.uint8)), group.append(g), frm.append(np.full(G, ids[r], np.uint32))
        idc.append(np.full(G, t, np.uint64)), flag.append(np.ones(G, np.uint8))
        if t % 2 == 1:
            kind.append(np.full(G, capi.CMD_HEARTBEAT_RESPONSE, np.uint8)), group.append(g), frm.append(np.full(G, ids[r], np.uint32))
            idc.append(np.full(G, max(t - 1, 0), np.uint64)), flag.append(np.ones(G, np.uint8))
    kind, group, frm, idc, flag = map(np.concatenate, (kind, group, frm, idc, flag))
    p = rng.permutation(len(kind))
    return dict(kind=kind[p], group=group[p], from_=frm[p], id=idc[p], flag=flag[p])


batches = [batch(t) for t in range(T)]
tm = {"submit": 0.0, "step": 0.0, "outbox": 0.0, "drain": 0.0}
rows = fsm = 0
c0 = e.counters()["decisions"]
t_all = time.perf_counter()
for t in range(T):
    t0 = time.perf_counter()
    e.submit_columns(**batches[t])
    t1 = time.perf_counter()
    e._check(e.api.step_node(e._h, 100 * (t + 1), capi.NODE_LEADER_HALF | capi.NODE_TICK))
    t2 = time.perf_counter()
    o = capi.NodeOutbox()
    e._check(e.api.node_outbox_view(e._h, C.byref(o)))
    t3 = time.perf_counter()
    fsm += len(e.drain_applies(copy=False))
    e.drain_messages(copy=False), e.drain_faults()
    t4 = time.perf_counter()
    rows += int(o.rows)
    if t >= 5:
        tm["submit"] += t1 - t0
        tm["step"] += t2 - t1
        tm["outbox"] += t3 - t2
        tm["drain"] += t4 - t3
wall = time.perf_counter() - t_all
dec = e.counters()["decisions"] - c0
n = T - 5
print(f"G={G} R={R}: {rows / T:.0f} rows/tick, {fsm / T:.0f} fsm rows/tick, general {int(o.rows_general)}, h2d {int(o.bytes_h2d)/1e6:.1f} MB d2h {int(o.bytes_d2h)/1e6:.1f} MB per tick")
print("ms per tick:", {k: round(v / n * 1e3, 3) for k, v in tm.items()}, "total", round(sum(tm.values()) / n * 1e3, 3))
print(f"decisions/s through the surface: {dec / wall:.3e}  (head {int(e.read('head')[0])}, commit {int(e.read('commit')[0])})")
assert (e.read("head") == T).all() and not e.read("fault").any()
