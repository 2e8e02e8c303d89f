"""Round 6, review item 7: what ONE REAL NODE's dense halves cost by where the partitions it leads sit among its local slots.

The one-GPU cluster of `bench.py --cluster --any-leader` hosts all R nodes on one device over mailbox columns indexed by ONE
partition numbering, so "node g % R leads partition g" (interleaved) makes every node's led partitions a third of every
128-byte line of its columns: 129 us per round against 75 with leadership in contiguous blocks.  A real josefine node is one
engine of its own: it addresses a partition by a LOCAL slot, and its adapter (the event loop's partition table) is free to
number the slots led-first - another node numbers its own differently.  This script is that node: one engine, R = 3,
G local slots, leader of a third of them, follower of the rest, driven for T protocol rounds through the halves jg_step_node
runs underneath (jg_step_dense_leader + jg_step_dense_follower) with the steady-state traffic of its peers - once with the
led partitions FIRST among its slots, once with them interleaved (every third slot).  Run it under rocprofv3 --kernel-trace
--stats: the kernels' average durations are the result (the host arrays the Python wrappers upload per call are not timed).

    python profiles/micro/exp_node_layout.py [led_first|interleaved] [G] [T]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from josefine_amd import BatchedRaft, capi  # noqa: E402
from josefine_amd.traces import elect_where  # noqa: E402

layout = sys.argv[1] if len(sys.argv) > 1 else "led_first"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 24
R = 3
NO = capi.NO_ACK

e = BatchedRaft(G, R, seed=7, flags=capi.CFG_SEPARATE_COMMIT_KEY)
g = np.arange(G)
led = (g < G // R) if layout == "led_first" else (g % R == 0)
elect_where(e, led)
e.drain_messages(), e.drain_applies(), e.drain_faults()
ids = e.node_ids
other = ~led
head = 0
for t in range(T):
    now = 100 * (t + 1)
    # -- what the peers sent since the last round.  Where this node LEADS: both followers acknowledged the head it had (and, every
    # other round, answered its Heartbeat with the commit index), the client proposes one block.
    acks = np.full((R, G), NO, np.uint64)
    acks[0] = np.where(led, 1, 0)
    hbr_has = np.full((R, G), capi.HB_NONE, np.uint8)
    if t:
        acks[1:, led] = head
        if t % 2 == 0:
            hbr_has[1:, led] = 1
    e.step_dense_leader(now, acks, hbr_has, np.zeros((R, G), np.uint64), tick=True)
    # ... where it FOLLOWS (leader: node ids[1]): the block appended this round, a Heartbeat with the commit index every other round
    term = np.where(other, 1, 0).astype(np.uint64)
    hb_commit = np.where(other & (t % 2 == 0), max(head - 1, 0), NO).astype(np.uint64)
    ae_from = np.where(other, head, 0).astype(np.uint64)
    ae_n = np.where(other, 1, capi.AE_NONE).astype(np.uint8)
    e.step_dense_follower(now, term, hb_commit, ae_from, ae_n, leader_id=ids[1], tick=True)
    head += 1
role, h, c, fault = e.read("role"), e.read("head"), e.read("commit"), e.read("fault")
assert not fault.any()
assert (role[led] == capi.ROLE_LEADER).all() and (h[led] == T).all() and (c[led] >= T - 2).all(), "the partitions it leads"
assert (role[other] == capi.ROLE_FOLLOWER).all() and (h[other] == T).all() and (c[other] >= T - 4).all(), "the partitions it follows"
assert len(e.drain_messages()) == 0, "rows left the mailbox vocabulary"
print(f"{layout}: one node, {G} local slots x {R} replicas, leads {int(led.sum())}, {T} rounds: heads {T}, commits >= {int(c.min())}, no row outside the columns")
