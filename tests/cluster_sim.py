"""Closed-loop cluster simulation for the parity tests.

One engine hosts all R replicas of P partitions as G = P*R independent instances
(instance g = partition g // R, replica slot g % R, NodeId slot+1 — the
examples/multi-node topology repeated P times).  The host plays josefine's
transport (src/raft/tcp.rs) and event loop (src/raft/server.rs:103-165): every round
it ticks all instances, feeds client requests to the current leaders, drains the
outbound messages and routes them to their destination instances as the next round's
commands.  Nothing here decides anything about Raft; it only moves rows.
"""
import numpy as np

from josefine_amd import capi


class Cluster:
    def __init__(self, engine, P, R, chaos_seed=None, drop=0.05, dup=0.05, restart=0.002):
        self.e, self.P, self.R = engine, P, R
        self.now = 0
        self.inbox = None  # dict of columns for the next step
        self.next_token = 1
        # optional unreliable network + crashing processes (same seed => same perturbation for
        # every engine driven with the same messages): drops (tcp.rs:90-93 drops when a peer
        # queue is full), duplicates, and process restarts
        self.chaos = np.random.default_rng(chaos_seed) if chaos_seed is not None else None
        self.p_drop, self.p_dup, self.p_restart = drop, dup, restart

    @staticmethod
    def self_slots(P, R):
        return (np.arange(P * R) % R).astype(np.uint8)

    def route(self, msgs):
        """Message rows -> command columns for the destination instances."""
        R = self.R
        kind, group, frm, term, idc, aux, flag = [], [], [], [], [], [], []
        blk_id, blk_next = [], []
        # Wire order: the k-th message of every sender travels before anyone's (k+1)-th
        # (drained rows are group-major; real sockets interleave the senders).  This matters
        # for the reference's elections: the candidate broadcasts its VoteRequest once per
        # peer (candidate.rs:30-37) and a voter's later rejections overwrite its grant
        # (election.rs:34), so a quorum must be seen before the duplicates arrive.
        if len(msgs):
            g_arr = msgs["group"].astype(np.int64)
            first = np.r_[True, g_arr[1:] != g_arr[:-1]]
            start = np.maximum.accumulate(np.where(first, np.arange(len(msgs)), 0))
            k_in_sender = np.arange(len(msgs)) - start
            msgs = msgs[np.argsort(k_in_sender, kind="stable")]
            if self.chaos is not None:
                u = self.chaos.random(len(msgs))
                keep = u >= self.p_drop
                dup = keep & (u >= 1.0 - self.p_dup)
                idx = np.concatenate([np.nonzero(keep)[0], np.nonzero(dup)[0]])
                msgs = msgs[np.sort(idx, kind="stable")]
        for m in msgs:
            k = int(m["kind"])
            if k == capi.CMD_CLIENT_REQUEST or int(m["to_kind"]) in (capi.TO_QUEUE, capi.TO_CLIENT, capi.TO_LOCAL):
                continue  # client proxying / queue mirror rows: host-side bookkeeping only
            g = int(m["group"])
            p, s = divmod(g, R)
            if int(m["to_kind"]) == capi.TO_PEERS:
                dsts = [p * R + q for q in range(R) if q != s]
            else:
                slot = int(m["to_id"]) - 1
                if not (0 <= slot < R):
                    continue
                dsts = [p * R + slot]
            for d in dsts:
                kind.append(k); group.append(d); frm.append(int(m["from"])); term.append(int(m["term"]))
                flag.append(int(m["flag"]))
                if k == capi.CMD_APPEND_ENTRIES:
                    # leader.rs:124-174: the `aux` blocks after key `id`; the leaders of this
                    # simulation only ever append, so block i has next = i-1 (chain.rs:164-167)
                    start, n = int(m["id"]), int(m["aux"])
                    idc.append(len(blk_id)); aux.append(n)
                    for b in range(start + 1, start + 1 + n):
                        blk_id.append(b); blk_next.append(b - 1)
                else:
                    idc.append(int(m["id"])); aux.append(int(m["aux"]))
        return dict(kind=np.array(kind, np.uint8), group=np.array(group, np.uint32), from_=np.array(frm, np.uint32),
                    term=np.array(term, np.uint64), id=np.array(idc, np.uint64), aux=np.array(aux, np.uint64),
                    flag=np.array(flag, np.uint8), blk_id=np.array(blk_id, np.uint64),
                    blk_next=np.array(blk_next, np.uint64))

    def round(self, dt_ms=100, propose_prob=0.5, rng=None):
        """One event-loop turn: deliver last round's messages, tick everyone, propose on leaders."""
        e = self.e
        self.now += dt_ms
        if self.inbox is not None and len(self.inbox["kind"]):
            e.submit_columns(**self.inbox)
        G = self.P * self.R
        if self.chaos is not None:
            crash = np.nonzero(self.chaos.random(G) < self.p_restart)[0].astype(np.uint32)
            if len(crash):
                e.submit_columns(np.full(len(crash), capi.CMD_RESTART, np.uint8), crash)
        e.submit_columns(np.full(G, capi.CMD_TICK, np.uint8), np.arange(G, dtype=np.uint32))
        roles = e.read("role")
        leaders = np.nonzero((roles == capi.ROLE_LEADER) & (e.read("fault") == 0))[0]
        if rng is not None and len(leaders):
            pick = leaders[rng.random(len(leaders)) < propose_prob]
            if len(pick):
                toks = np.arange(self.next_token, self.next_token + len(pick), dtype=np.uint64)
                self.next_token += len(pick)
                e.submit_columns(np.full(len(pick), capi.CMD_CLIENT_REQUEST, np.uint8), pick.astype(np.uint32), id=toks)
        e.step(self.now)
        msgs = e.drain_messages()
        fsm = e.drain_applies()
        faults = e.drain_faults()
        self.inbox = self.route(msgs)
        return msgs, fsm, faults
