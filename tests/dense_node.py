"""Drivers for the dense node tick tests: random mailbox generators and a closed-loop cluster
of R engines (one per node) that exchange nothing but dense mailbox columns.  Pure array
plumbing — no Raft logic here."""
import numpy as np

from josefine_amd import capi

NO = capi.NO_ACK


def random_leader_inbox(rng, G, R, slots, heads, p_hbr=0.3, p_missing=0.05):
    """acks as the ragged synthetic stream would give + HeartbeatResponses, a few of them
    without the commit (leader.rs:222-231 -> replicate() again)."""
    acks = np.full((R, G), NO, dtype=np.uint64)
    hbr_has = np.full((R, G), capi.HB_NONE, dtype=np.uint8)
    hbr_commit = np.zeros((R, G), dtype=np.uint64)
    g = np.arange(G)
    for r in range(R):
        own = slots == r
        u = rng.random(G)
        # follower acks: somewhere at or below the leader head, sometimes nothing, rarely above it
        a = np.where(u < 0.2, NO, np.maximum(heads.astype(np.int64) - rng.integers(0, 4, G), 0).astype(np.uint64))
        a = np.where(u > 0.995, heads + np.uint64(10), a)
        acks[r] = np.where(own, rng.integers(0, 3, G).astype(np.uint64), a)
        h = rng.random(G)
        has = np.where(h < p_hbr, 1, capi.HB_NONE)
        has = np.where(h < p_hbr * p_missing, 0, has)
        hbr_has[r] = np.where(own, capi.HB_NONE, has).astype(np.uint8)
        hbr_commit[r] = np.where(hbr_has[r] == 0, rng.integers(0, 3, G), 0).astype(np.uint64)
    return acks, hbr_has, hbr_commit


def random_follower_inbox(rng, G, node_ids, self_ids, heads, commits, terms):
    """One tick of leader traffic for followers: mostly in-order windows and heartbeats at or
    below the head, with re-sent windows, gaps, stale terms and a second 'leader' mixed in."""
    ids = np.array(node_ids, dtype=np.uint32)
    # a leader that is not the node itself; mostly the next node id, sometimes another one
    pick = rng.integers(1, len(ids), G)
    self_idx = np.searchsorted(ids, self_ids)
    lead_main = ids[(self_idx + 1) % len(ids)]
    lead_other = ids[(self_idx + pick) % len(ids)]
    leader = np.where(rng.random(G) < 0.97, lead_main, lead_other).astype(np.uint32)
    term = np.where(rng.random(G) < 0.9, np.maximum(terms, 1), terms + rng.integers(0, 3, G).astype(np.uint64))
    term = np.where(rng.random(G) < 0.03, np.maximum(terms.astype(np.int64) - 1, 0).astype(np.uint64), term)
    u = rng.random(G)
    hb = np.where(u < 0.5, np.minimum(heads, commits + rng.integers(0, 3, G).astype(np.uint64)), NO)
    hb = np.where(u < 0.03, heads + np.uint64(2), hb)  # the leader is ahead: has_committed = false
    v = rng.random(G)
    ae_n = np.where(v < 0.75, rng.integers(0, capi.MAX_INFLIGHT + 1, G), capi.AE_NONE).astype(np.uint8)
    back = rng.integers(0, 3, G).astype(np.uint64)
    ae_from = np.where(v < 0.70, heads, np.where(v < 0.745, heads - np.minimum(back, heads), heads + np.uint64(1)))
    return dict(leader=leader, term=term.astype(np.uint64), hb_commit=hb.astype(np.uint64),
                ae_from=ae_from.astype(np.uint64), ae_n=ae_n)


class DenseCluster:
    """R engines, engine r hosts replica slot r of every one of the G groups; slot `lead` is
    made leader of all groups.  One round = leader half on the leader node, follower half on
    every other node, the outboxes of one being the inboxes of the others."""

    def __init__(self, factory, G, R, seed=3, lead=0, group_base=0):
        self.G, self.R, self.lead = G, R, lead
        self.nodes = [factory(G, R, seed=seed + r, self_slots=np.full(G, r, np.uint8), group_base=group_base,
                              flags=capi.CFG_SEPARATE_COMMIT_KEY) for r in range(R)]
        self.now = 0
        self.acks = np.full((R, G), NO, dtype=np.uint64)
        self.hbr_has = np.full((R, G), capi.HB_NONE, dtype=np.uint8)
        self.hbr_commit = np.zeros((R, G), dtype=np.uint64)
        from josefine_amd.traces import elect_all
        elect_all(self.nodes[lead])
        self.nodes[lead].drain_messages()
        self.nodes[lead].drain_applies()
        self.rows = []

    def round(self, appends, dt_ms=100):
        self.now += dt_ms
        L = self.nodes[self.lead]
        self.acks[self.lead] = appends
        out = L.step_dense_leader(self.now, self.acks, self.hbr_has, self.hbr_commit, tick=True)
        outs = {self.lead: out}
        lead_id = L.node_ids[self.lead]
        for r in range(self.R):
            if r == self.lead:
                continue
            fo = self.nodes[r].step_dense_follower(self.now, out["term"], out["hb_commit"], out["ae_from"][r],
                                                   out["ae_n"][r], leader_id=lead_id, tick=True)
            self.acks[r] = fo["ack_head"]
            self.hbr_has[r] = fo["hb_has"]
            self.hbr_commit[r] = fo["hb_commit"]
            outs[r] = fo
        self.rows.append([n.drain_messages() for n in self.nodes])
        return outs
