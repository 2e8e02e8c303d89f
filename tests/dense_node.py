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


# The phases of a routed round (josefine_amd/csrc/jg_route.h JG_ROUTE_PHASE_*): the steps every node takes in lockstep.
PHASE_DELIVERED, PHASE_INJECTED, PHASE_LEADER, PHASE_FOLLOWER = 1, 2, 3, 4


def emission_index(groups):
    """emission index of every row within its group, for the rows ONE step drained (group-major, a group's rows in
    emission order)"""
    g = np.asarray(groups, np.int64)
    n = len(g)
    if not n:
        return np.zeros(0, np.int64)
    first = np.r_[True, g[1:] != g[:-1]]
    start = np.maximum.accumulate(np.where(first, np.arange(n), 0))
    return np.arange(n) - start


def routable(rows, member_ids):
    """Which drained message rows the cluster transport delivers: everything addressed to peers
    that is a plain message.  AppendEntries rows (the payload lives in the sender's block store)
    and ClientRequest rows (instructions to the host adapter about its request mirror) stay in
    the queue for the host."""
    k = rows["kind"]
    plain = (k != capi.CMD_APPEND_ENTRIES) & (k != capi.CMD_CLIENT_REQUEST)
    to_all = rows["to_kind"] == capi.TO_PEERS
    to_one = (rows["to_kind"] == capi.TO_PEER) & np.isin(rows["to_id"], member_ids)
    return plain & (to_all | to_one)


class RoutedCluster(DenseCluster):
    """DenseCluster + a transport for the rows outside the mailbox vocabulary (votes, ...): what a
    node emits in round t is applied by its addressees at the start of round t+1, per group in
    the order (phase of the round it was emitted in, emission index within the sender's step, sender
    slot) - every sender's first row of a phase before anybody's second, each sender's stream in its
    own order (one connection per peer pair: tcp.rs) - followed, as a step of its own, by the rows
    injected for that round.  The phases are the steps the nodes take in lockstep: the delivered
    rows, the injected rows, the leader half, the follower half.  The host-side statement of
    jg_dense_cluster_round_routed, for any backend."""

    def __init__(self, factory, G, R, **kw):
        super().__init__(factory, G, R, **kw)
        self.member_ids = np.array([self.nodes[r].node_ids[r] for r in range(R)], dtype=np.uint32)
        self.inbound = [[] for _ in range(R)]  # per node: list of (src, structured rows, their phases, their emission indices)
        self.kept = [np.zeros(0, dtype=capi.MSG_DTYPE) for _ in range(R)]
        self.delivered = np.zeros(R, dtype=np.int64)

    def pending(self, n):
        """rows queued for node n's next round"""
        return sum(len(p[1]) for p in self.inbound[n])

    def inbound_order(self, n):
        """node n's mail in the transport's order: (rows, sender slot of every row, its ord = phase << 8 | emission index)"""
        parts = self.inbound[n]
        if not parts:
            return np.zeros(0, dtype=capi.MSG_DTYPE), np.zeros(0, np.int64), np.zeros(0, np.int64)
        rows = np.concatenate([p[1] for p in parts])
        src = np.concatenate([np.full(len(p[1]), p[0], np.int64) for p in parts])
        phase, k = np.concatenate([p[2] for p in parts]), np.concatenate([p[3] for p in parts])
        order = np.lexsort((src, k, phase, rows["group"]))  # group, then phase, then emission index, then sender
        return rows[order], src[order], (phase << 8 | k)[order]

    @staticmethod
    def columns_of(rows):
        return dict(kind=rows["kind"], group=rows["group"], from_=rows["from"], term=rows["term"], id=rows["id"], aux=rows["aux"], flag=rows["flag"])

    def _inbound_columns(self, n):
        rows = self.inbound_order(n)[0]
        self.inbound[n] = []
        return self.columns_of(rows) if len(rows) else None

    @staticmethod
    def _inject_columns(inject):
        if inject is None or not len(inject["kind"]):
            return None
        m = len(inject["kind"])
        z8, z4 = np.zeros(m, np.uint64), np.zeros(m, np.uint32)
        return dict(kind=inject["kind"], group=inject["group"], from_=inject.get("from_", z4), term=inject.get("term", z8), id=inject.get("id", z8),
                    aux=inject.get("aux", z8), flag=inject.get("flag", np.zeros(m, np.uint8)))

    def sparse_steps(self, n, now, inject, emitted):
        """node n's first two steps of a round: the delivered rows, then the injected ones; what they emit -> emitted"""
        for phase, cols in ((PHASE_DELIVERED, self._inbound_columns(n)), (PHASE_INJECTED, self._inject_columns(inject))):
            if cols is None:
                continue
            self.delivered[n] += len(cols["kind"])
            self.nodes[n].submit_columns(**cols)
            self.nodes[n].step(now)
            out = self.nodes[n].drain_messages()
            if len(out):
                emitted.append((out, np.full(len(out), phase, np.int64), emission_index(out["group"])))

    def transport(self, emitted):
        """emitted[s]: list of (rows, phase, emission index) of sender s, in the order of its steps"""
        for s in range(self.R):
            if not emitted[s]:
                continue
            rows = np.concatenate([e[0] for e in emitted[s]])
            phase, k = np.concatenate([e[1] for e in emitted[s]]), np.concatenate([e[2] for e in emitted[s]])
            ok = routable(rows, self.member_ids)
            self.kept[s] = np.concatenate([self.kept[s], rows[~ok]])
            rows, phase, k = rows[ok], phase[ok], k[ok]
            for n in range(self.R):
                if n == s:
                    continue
                to_n = (rows["to_kind"] == capi.TO_PEERS) | (rows["to_id"] == self.member_ids[n])
                if to_n.any():
                    self.inbound[n].append((s, rows[to_n], phase[to_n], k[to_n]))

    def round(self, appends, inject=None, dt_ms=100):
        now = self.now + dt_ms
        emitted = [[] for _ in range(self.R)]
        for n in range(self.R):
            self.sparse_steps(n, now, inject[n] if inject else None, emitted[n])
        outs = self.dense_round(appends, dt_ms)
        for s, parts in enumerate(self.dense_emitted()):
            emitted[s].extend(parts)
        self.transport(emitted)
        return outs

    def dense_emitted(self):
        """what the dense round just run left in self.rows, per node as (rows, phase, emission index) parts"""
        drained = self.rows.pop()
        return [[(rows, np.full(len(rows), PHASE_LEADER if s == self.lead else PHASE_FOLLOWER, np.int64), emission_index(rows["group"]))] if len(rows) else []
                for s, rows in enumerate(drained)]

    def dense_round(self, appends, dt_ms):
        # client requests are offered only where the lead node leads: at a leaderless replica the
        # reference queues them (follower.rs:258-270), which the dense append column cannot express
        leads = self.nodes[self.lead].read("role") == capi.ROLE_LEADER
        return DenseCluster.round(self, np.where(leads, np.asarray(appends, dtype=np.uint64), np.uint64(0)), dt_ms)


OWNER_NONE = 255


class AnyLeaderCluster(RoutedCluster):
    """PER-PARTITION LEADERSHIP (jg_dense_cluster_create with JG_CLUSTER_ANY_LEADER): every node leads the groups it was
    elected for and follows the others; the mailbox columns are the cluster's, indexed by group and slot.  The numpy
    statement of one round, over the per-node dense entry points of any backend:
      0. owner[g] = the lowest slot whose node is a healthy leader of g;
      1. every node's leader half: inbox and ClientRequests only where it owns the group; the owner's Tick goes into the
         cluster's columns, a leader's that is not the owner (two terms' leaders in one round) travels as ROWS;
      2. every node's follower half: mail (sender = the owner) only where somebody else owns the group.
    Rows are transported exactly as RoutedCluster does (every node takes both halves: phases 3 and 4)."""

    def __init__(self, factory, G, R, seed=3, group_base=0):
        self.G, self.R, self.lead = G, R, 0
        self.nodes = [factory(G, R, seed=seed + r, self_slots=np.full(G, r, np.uint8), group_base=group_base,
                              flags=capi.CFG_SEPARATE_COMMIT_KEY) for r in range(R)]
        self.now = 0
        self.acks = np.full((R, G), NO, dtype=np.uint64)
        self.hbr_has = np.full((R, G), capi.HB_NONE, dtype=np.uint8)
        self.hbr_commit = np.zeros((R, G), dtype=np.uint64)
        self.o_term = np.zeros(G, np.uint64)
        self.o_hbc = np.full(G, NO, np.uint64)
        self.o_from = np.zeros((R, G), np.uint64)
        self.o_n = np.full((R, G), capi.AE_NONE, np.uint8)
        self.rows = []
        self.member_ids = np.array([self.nodes[r].node_ids[r] for r in range(R)], dtype=np.uint32)
        self.inbound = [[] for _ in range(R)]
        self.kept = [np.zeros(0, dtype=capi.MSG_DTYPE) for _ in range(R)]
        self.delivered = np.zeros(R, dtype=np.int64)
        self.owner = np.full(G, OWNER_NONE, np.uint8)

    def tick_rows(self, n, out, groups):
        """The Tick of node n for `groups`, as the message rows its column words stand for (leader.rs:234-245)."""
        rows = []
        for g in groups:
            if int(out["hb_commit"][g]) != NO:
                rows.append((g, capi.CMD_HEARTBEAT, capi.TO_PEERS, 0, 0, 0, self.member_ids[n], int(out["term"][g]), int(out["hb_commit"][g]), 0))
            for r in range(self.R):
                if int(out["ae_n"][r][g]) != capi.AE_NONE:
                    rows.append((g, capi.CMD_APPEND_ENTRIES, capi.TO_PEER, 0, 0, self.member_ids[r], self.member_ids[n], int(out["term"][g]),
                                 int(out["ae_from"][r][g]), int(out["ae_n"][r][g])))
        return np.array(rows, dtype=capi.MSG_DTYPE) if rows else np.zeros(0, dtype=capi.MSG_DTYPE)

    def dense_round(self, appends, dt_ms):
        self.now += dt_ms
        G, R = self.G, self.R
        appends = np.broadcast_to(np.asarray(appends, dtype=np.uint64), (G,))
        leads = np.stack([(n.read("role") == capi.ROLE_LEADER) & (n.read("fault") == 0) for n in self.nodes])
        owner = np.full(G, OWNER_NONE, np.uint8)
        for r in reversed(range(R)):
            owner[leads[r]] = r
        # a group that passes from one owner straight to another: the old owner's row of the answers still holds what it said
        # when it last followed (it never writes its own row while it owns the group) - withdrawn, not replayed to the new owner
        handed = np.nonzero((owner != self.owner) & (owner != OWNER_NONE) & (self.owner != OWNER_NONE))[0]
        self.acks[self.owner[handed], handed] = NO
        self.hbr_has[self.owner[handed], handed] = capi.HB_NONE
        self.owner = owner
        drained = [None] * R
        outs = {}
        for n in range(R):  # 1. the leader halves
            mine = owner == n
            acks = np.where(mine[None, :], self.acks, np.uint64(NO))
            acks[n] = np.where(mine, appends, np.uint64(0))
            has = np.where(mine[None, :], self.hbr_has, np.uint8(capi.HB_NONE)).astype(np.uint8)
            out = self.nodes[n].step_dense_leader(self.now, acks, has, self.hbr_commit, tick=True)
            outs[n] = out
            self.o_term[mine], self.o_hbc[mine] = out["term"][mine], out["hb_commit"][mine]
            self.o_from[:, mine], self.o_n[:, mine] = out["ae_from"][:, mine], out["ae_n"][:, mine]
            lose = np.nonzero(leads[n] & ~mine)[0]
            rows = np.concatenate([self.nodes[n].drain_messages(), self.tick_rows(n, out, lose)])
            drained[n] = [rows[np.argsort(rows["group"], kind="stable")]]  # (the leader half's rows; the follower half's below)
        for n in range(R):  # 2. the follower halves
            mail = (owner != OWNER_NONE) & (owner != n)
            sender = self.member_ids[np.where(mail, owner, 0)].astype(np.uint32)
            fo = self.nodes[n].step_dense_follower(self.now, np.where(mail, self.o_term, np.uint64(0)), np.where(mail, self.o_hbc, np.uint64(NO)),
                                                   np.where(mail, self.o_from[n], np.uint64(0)),
                                                   np.where(mail, self.o_n[n], np.uint8(capi.AE_NONE)).astype(np.uint8), leader=sender, tick=True)
            keep = owner != n  # (the own slot's word of a group this node owns is nobody's: whoever owns a group reads `appends`)
            self.acks[n] = np.where(keep, fo["ack_head"], self.acks[n])
            self.hbr_has[n] = np.where(keep, fo["hb_has"], self.hbr_has[n])
            self.hbr_commit[n] = np.where(keep, fo["hb_commit"], self.hbr_commit[n])
            drained[n].append(self.nodes[n].drain_messages())
        self.rows.append(drained)
        return outs

    def dense_emitted(self):
        drained = self.rows.pop()
        return [[(rows, np.full(len(rows), phase, np.int64), emission_index(rows["group"])) for rows, phase in zip(halves, (PHASE_LEADER, PHASE_FOLLOWER)) if len(rows)]
                for halves in drained]


from josefine_amd.traces import any_failure_rows, cluster_failure_rows  # noqa: E402,F401  (shared with bench.py)
