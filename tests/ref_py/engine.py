"""ref_py.engine — G independent `ref_py.raft.Raft` groups behind the interface the parity
helpers drive (`BatchedRaft`'s: submit_columns / step / step_dense_* / read / drain_* / counters),
speaking the row vocabulary of include/josefine_gpu.h.  TEST INFRASTRUCTURE.

Everything here is translation between the header's SoA / row encodings and the Rust-shaped
`Command` / `Message` / `Instruction` values of ref_py.raft — written from the header's text:
  - a panic / Err of the reference becomes the group's sticky fault code; the group then ignores
    commands until JG_CMD_RESTART (= Raft::new + Chain::new on the persisted tree);
  - AppendEntries rows are (range start key, number of blocks), Apply instructions are key
    ranges, the client queue is mirrored through CLIENT_REQUEST rows;
  - the dense entry points are specified as "equivalent to submitting those commands".
The run-length encodings are checked against the Rust-shaped values on the way (`_check_*`).
"""
from __future__ import annotations

import numpy as np

from josefine_amd import _capi as capi

from . import raft as rr

M64 = (1 << 64) - 1


def mix64(z: int) -> int:
    """splitmix64 finaliser (DESIGN.md "Logical time and randomness")."""
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


FAULT_OF = {  # where the reference dies -> JG_FAULT_* (include/josefine_gpu.h)
    "leader.rs:34": capi.FAULT_LEADER_TERM_UNIMPLEMENTED,
    "chain.rs:163": capi.FAULT_APPEND_ID_NOT_ABOVE_HEAD,
    "progress.rs:43": capi.FAULT_PROGRESS_UNKNOWN_NODE,
    "chain.rs:201": capi.FAULT_COMMIT_MISSING_BLOCK,
    "chain.rs:180-185": capi.FAULT_EXTEND_MISSING_PARENT,
    "follower.rs:149": capi.FAULT_FOLLOWER_STALE_LEADER,
    "candidate.rs:64": capi.FAULT_CANDIDATE_TICK_ELECTED,
    "chain.rs:219-226": capi.FAULT_RANGE_HIT_COMMIT_KEY,
}
KIND = {"VoteRequest": capi.CMD_VOTE_REQUEST, "VoteResponse": capi.CMD_VOTE_RESPONSE,
        "AppendEntries": capi.CMD_APPEND_ENTRIES, "AppendResponse": capi.CMD_APPEND_RESPONSE,
        "Heartbeat": capi.CMD_HEARTBEAT, "HeartbeatResponse": capi.CMD_HEARTBEAT_RESPONSE,
        "ClientRequest": capi.CMD_CLIENT_REQUEST, "ClientResponse": capi.CMD_CLIENT_RESPONSE}
TO = {"Peers": capi.TO_PEERS, "Peer": capi.TO_PEER, "Local": capi.TO_LOCAL, "Client": capi.TO_CLIENT}


class RefEngine:
    def __init__(self, n_groups, n_replicas=1, node_ids=None, self_slots=None, seed=0, device_id=0, group_base=0,
                 flags=0, heartbeat_timeout_ms=100, election_timeout_ms=(500, 1000), api=None):
        from josefine_amd.engine import EngineError
        self.G, self.R = int(n_groups), int(n_replicas)
        self.node_ids = [int(x) for x in (node_ids if node_ids is not None else range(1, self.R + 1))]
        # jg_engine_create's argument checks (RaftConfig::validate, config.rs:60-84; progress.rs:16)
        if not 1 <= self.R <= capi.MAX_REPLICAS:
            raise EngineError(capi.EINVAL, "n_replicas out of range")
        if 0 in self.node_ids[:self.R] or len(set(self.node_ids[:self.R])) != self.R:
            raise EngineError(capi.EINVAL, "id cannot be 0 / duplicate node id")
        if heartbeat_timeout_ms < 5:
            raise EngineError(capi.EINVAL, "heartbeat timeout is too low")
        if election_timeout_ms[1] <= election_timeout_ms[0] or self.G == 0:
            raise EngineError(capi.EINVAL, "election timeout range is empty / no groups")
        self.seed, self.group_base, self.flags = int(seed), int(group_base), int(flags)
        self.hb_timeout = int(heartbeat_timeout_ms)
        self.el_min, self.el_max = (int(x) for x in election_timeout_ms)
        self.slots = np.zeros(self.G, np.uint8) if self_slots is None else np.asarray(self_slots, np.uint8).copy()
        self.draws = [0] * self.G
        self.fault = [0] * self.G
        self.groups = [self._new_group(g, None, 0) for g in range(self.G)]
        self._pending = []     # (group, raft.Command or "Restart", blocks)
        self._msgs, self._fsm, self._faults = [], [], []
        self._n_cmds = 0
        self._decisions = 0
        self.api = None

    # ---- construction ---------------------------------------------------------------------------
    def _rand(self, g):
        def rand_range(lo, hi):  # replaces thread_rng().gen_range(lo..hi) (follower.rs:105)
            key = ((self.group_base + g) * 0xD1342543DE82EF95 + self.draws[g]) & M64
            self.draws[g] += 1
            return lo + mix64(self.seed ^ mix64(key)) % (hi - lo)
        return rand_range

    def _new_group(self, g, db, now):
        s = int(self.slots[g])
        others = [self.node_ids[r] for r in range(self.R) if r != s]  # config.nodes: the other slots, ascending
        return rr.Raft(self.node_ids[s], others, self.hb_timeout, self.el_min, self.el_max, self._rand(g), db=db,
                       now=now, separate_commit_key=bool(self.flags & capi.CFG_SEPARATE_COMMIT_KEY))

    # ---- input ----------------------------------------------------------------------------------
    def submit_columns(self, kind, group, from_=None, term=None, id=None, aux=None, flag=None, blk_id=None,
                       blk_next=None):
        n = len(kind)
        z = np.zeros(n, np.uint64)
        from_ = z if from_ is None else from_
        term = z if term is None else term
        id = z if id is None else id
        aux = z if aux is None else aux
        flag = z if flag is None else flag
        for i in range(n):
            k, g = int(kind[i]), int(group[i])
            assert 0 <= g < self.G and 0 <= k < capi.CMD_RECREATE + 1
            blocks = None
            if k == capi.CMD_APPEND_ENTRIES:
                a, b = int(id[i]), int(aux[i])
                blocks = [rr.Block(int(blk_id[j]), int(blk_next[j])) for j in range(a, a + b)]
            self._pending.append((g, self._command(k, int(from_[i]), int(term[i]), int(id[i]), int(aux[i]),
                                                   int(flag[i]), blocks)))

    @staticmethod
    def _command(k, frm, term, id_, aux, flag, blocks):
        """One row of the command table of include/josefine_gpu.h -> enum Command (mod.rs:160-227)."""
        C = rr.Command
        if k == capi.CMD_TICK:
            return C("Tick")
        if k == capi.CMD_PROPOSE:
            return C("Propose")
        if k == capi.CMD_VOTE_REQUEST:
            return C("VoteRequest", term=term, candidate_id=frm, last_term=aux, head=id_)
        if k == capi.CMD_VOTE_RESPONSE:
            return C("VoteResponse", term=term, from_=frm, granted=bool(flag))
        if k == capi.CMD_APPEND_ENTRIES:
            return C("AppendEntries", term=term, leader_id=frm, blocks=blocks)
        if k == capi.CMD_APPEND_RESPONSE:
            return C("AppendResponse", node_id=frm, term=term, head=id_, success=bool(flag))
        if k == capi.CMD_HEARTBEAT:
            return C("Heartbeat", term=term, commit=id_, leader_id=frm)
        if k == capi.CMD_HEARTBEAT_RESPONSE:
            return C("HeartbeatResponse", commit=id_, has_committed=bool(flag))
        if k == capi.CMD_TIMEOUT:
            return C("Timeout")
        if k == capi.CMD_NOOP:
            return C("Noop")
        if k == capi.CMD_CLIENT_REQUEST:
            return C("ClientRequest", req_id=id_)
        if k == capi.CMD_CLIENT_RESPONSE:
            return C("ClientResponse", req_id=id_)
        if k == capi.CMD_RECREATE:
            return C("Recreate")
        return C("Restart")

    def submit(self, group, cmd):
        """One josefine_amd.engine.Command (the reference's variant / field names) for one group."""
        blk = cmd.blocks if cmd.kind == capi.CMD_APPEND_ENTRIES else []
        self.submit_columns([cmd.kind], [group], [cmd.from_], [cmd.term], [0 if blk else cmd.id],
                            [len(blk) if cmd.kind == capi.CMD_APPEND_ENTRIES else cmd.aux], [cmd.flag],
                            [b[0] for b in blk], [b[1] for b in blk])

    def apply(self, group, cmd, now_ms=0):
        from josefine_amd.engine import RaftHandle
        self.submit(group, cmd)
        self.step(now_ms)
        return RaftHandle(self, group)

    def handle(self, group):
        from josefine_amd.engine import RaftHandle
        return RaftHandle(self, group)

    def chain_compact(self, trees):
        """Batched Chain::compact over explicit (id, next) trees (duplicate ids: sled upsert, the
        last entry wins) -> removed masks per entry."""
        out = []
        for blocks, commit in trees:
            ch = rr.Chain()
            for b_id, b_next in blocks:
                ch.db.insert(rr.block_key(b_id), rr.Block(int(b_id), int(b_next)))
            ch.commit = int(commit)
            ch.compact()
            out.append(np.array([0 if ch.db.contains_key(rr.block_key(b_id)) else 1 for b_id, _ in blocks], np.uint8))
        return out

    def chain_compact_resident(self):
        """Chain::compact (chain.rs:239-253) on every healthy group's own chain -> removed (group, id)
        rows, group ascending, ids in the order of the walk (descending)."""
        rows = []
        for g in range(self.G):
            if self.fault[g]:
                continue
            ch = self.groups[g].chain
            before = [k for k in ch.db.keys if len(k) == 8]
            ch.compact()
            gone = sorted((int.from_bytes(k, "big") for k in before if not ch.db.contains_key(k)), reverse=True)
            rows += [(g, 0, i) for i in gone]
        a = np.zeros(len(rows), dtype=capi.COMPACT_DTYPE)
        for i, r in enumerate(rows):
            a[i] = r
        return a

    def apply_all(self, cmd, now_ms=0):
        n = self.G
        self.submit_columns(np.full(n, cmd.kind, np.uint8), np.arange(n, dtype=np.uint32),
                            np.full(n, cmd.from_, np.uint32), np.full(n, cmd.term, np.uint64),
                            np.full(n, cmd.id, np.uint64), np.full(n, cmd.aux, np.uint64), np.full(n, cmd.flag, np.uint8))
        self.step(now_ms)

    # ---- one command on one group ---------------------------------------------------------------
    def _apply(self, g, cmd, now, rows=True, capture=None):
        """RaftHandle::apply on group g; a panic becomes the sticky fault.  Returns the row-encoded
        messages (or feeds `capture`) and appends fsm rows unless rows is False."""
        if cmd.kind in ("Restart", "Recreate"):  # process restart: Raft::new + Chain::new on the persisted tree - or (Recreate) on an empty one
            old = self.groups[g]
            self.fault[g] = 0
            try:
                self.groups[g] = self._new_group(g, old.chain.db if cmd.kind == "Restart" else rr.Sled(), now)
            except rr.Panic as p:  # (cannot happen: Chain::new only asserts on a fresh tree)
                self._raise(g, p)
            return []
        if self.fault[g]:
            return []
        r = self.groups[g]
        r.now = now
        r.rpc, r.fsm = [], []
        was_follower = r.role.name == "Follower"
        d0 = r.decisions
        try:
            r.apply(cmd)
        except rr.Panic as p:
            self._raise(g, p)
        self._decisions += r.decisions - d0
        out = self._encode_msgs(g, r, cmd, was_follower)
        if rows:
            self._encode_fsm(g, r)
        return out

    def _raise(self, g, p):
        code = next(c for w, c in FAULT_OF.items() if p.where.startswith(w))
        self.fault[g] = code
        self._step_faults.append((g, code))

    def _encode_msgs(self, g, r, cmd, was_follower):
        out = []
        flush = []  # queued ClientRequests a follower sends on a Heartbeat (follower.rs:190-197): ONE row
        for m in r.rpc:
            if isinstance(m, tuple):
                if m[0] == "queue_push":
                    out.append((g, capi.CMD_CLIENT_REQUEST, capi.TO_QUEUE, 0, 0, r.id, 0, m[1], 0))
                else:
                    out.append((g, capi.CMD_CLIENT_REQUEST, capi.TO_QUEUE, capi.QUEUE_DROP, 0, r.id, 0, 0, m[1]))
                continue
            c = m.command
            assert m.from_ == ("Peer", r.id)                                    # mod.rs:390-400
            to_kind, to_id = TO[m.to[0]], m.to[1]
            k = KIND[c.kind]
            if c.kind == "ClientRequest" and cmd.kind == "Heartbeat" and was_follower:
                flush.append((to_id, c.req_id))
                continue
            if flush:
                out.append(self._flush_row(g, r, flush))
                flush = []
            if c.kind == "VoteRequest":
                assert c.candidate_id == r.id
                row = (0, c.term, c.head, c.last_term)
            elif c.kind == "VoteResponse":
                assert c.from_ == r.id
                row = (int(c.granted), c.term, 0, 0)
            elif c.kind == "AppendEntries":
                assert c.leader_id == r.id
                self._check_append_entries(r, c)
                row = (0, c.term, c.range_start, len(c.blocks))
            elif c.kind == "AppendResponse":
                assert c.node_id == r.id
                row = (int(c.success), c.term, c.head, 0)
            elif c.kind == "Heartbeat":
                assert c.leader_id == r.id
                row = (0, c.term, c.commit, 0)
            elif c.kind == "HeartbeatResponse":
                row = (int(c.has_committed), 0, c.commit, 0)
            else:  # ClientRequest forward / ClientResponse
                row = (0, 0, c.req_id, 0)
            out.append((g, k, to_kind, row[0], to_id, r.id, row[1], row[2], row[3]))
        if flush:
            out.append(self._flush_row(g, r, flush))
        return out

    @staticmethod
    def _flush_row(g, r, flush):
        assert len({t for t, _ in flush}) == 1
        return (g, capi.CMD_CLIENT_REQUEST, capi.TO_PEER, capi.QUEUE_FLUSH, flush[0][0], r.id, 0, 0, len(flush))

    @staticmethod
    def _check_append_entries(r, c):
        """The row stands for "the next `aux` stored blocks after skipping the first item of
        range(id..)": re-expand it and compare with the Vec<Block> the Rust built."""
        got, skipped = [], False
        for _k, v in r.chain.db.range(rr.block_key(c.range_start), None, False):
            if not isinstance(v, rr.Block):
                break
            if not skipped:
                skipped = True
                continue
            if len(got) == len(c.blocks):
                break
            got.append((v.id, v.next))
        assert got == [(b.id, b.next) for b in c.blocks], (got, c.blocks)

    def _encode_fsm(self, g, r):
        """Instruction::Notify -> a NOTIFY row; the Apply instructions of one range -> ONE range row
        (checked: expanding the row over the stored keys gives exactly those instructions)."""
        i = 0
        f = r.fsm
        while i < len(f):
            e = f[i]
            if e[0] == "Notify":
                self._step_fsm.append((g, capi.FSM_NOTIFY, e[2], e[1]))
                i += 1
            elif e[0] == "Range":
                _, kind, a, b = e
                j = i + 1
                ids = []
                while j < len(f) and f[j][0] == "Apply":
                    ids.append(f[j][1])
                    j += 1
                keys = [int.from_bytes(k, "big") for k in r.chain.db.keys if len(k) == 8]
                if kind == "leader":   # range(a..=b).skip(1)
                    exp = [k for k in keys if a <= k <= b][1:]
                    self._step_fsm.append((g, capi.FSM_APPLY_LEADER, a, b))
                else:                  # range(a..b)
                    exp = [k for k in keys if a <= k < b]
                    self._step_fsm.append((g, capi.FSM_APPLY_FOLLOWER, a, b))
                assert ids == exp, (kind, a, b, ids, exp)
                i = j
            else:
                raise AssertionError(f"Apply instruction outside a range: {e}")

    # ---- steps ----------------------------------------------------------------------------------
    def _begin(self):
        self._step_msgs, self._step_fsm, self._step_faults = {}, [], []

    def _end(self, fsm_rows=True):
        for g in sorted(self._step_msgs):           # drained group-major within one step
            self._msgs.extend(self._step_msgs[g])
        if fsm_rows:
            self._fsm.extend(sorted(self._step_fsm, key=lambda r: r[0]))  # stable: per group in emission order
        self._faults.extend(sorted(self._step_faults, key=lambda r: r[0]))

    def step(self, now_ms=0):
        self._begin()
        pend, self._pending = self._pending, []
        self._n_cmds += len(pend)
        for g, cmd in pend:                          # per group in stream order (groups are independent)
            self._step_msgs.setdefault(g, []).extend(self._apply(g, cmd, int(now_ms)))
        self._end()

    def step_dense_acks(self, acks, now_ms=0):
        """jg_step_dense_acks: per group the appends (each with its self-ack), then the acks in
        ascending slot order; no rows are queued (the caller reads head / commit deltas)."""
        acks = np.asarray(acks, np.uint64).reshape(self.R, self.G)
        self._begin()
        for g in range(self.G):
            self._dense_acks_group(g, acks, now_ms)
        self._end(fsm_rows=False)

    def step_dense_acks_n(self, acks):
        for t in range(len(acks)):
            self.step_dense_acks(acks[t])

    def _dense_acks_group(self, g, acks, now, msgs=None):
        """Returns False if the group took no part (dead / non-leader / faulted on the way in)."""
        if self.fault[g]:
            return False
        r = self.groups[g]
        s = int(self.slots[g])
        n_app = int(acks[s, g]) if acks is not None else 0
        if r.role.name != "Leader":     # acks are ignored by followers / candidates (follower.rs:62, candidate.rs:194)
            if n_app:
                self.fault[g] = capi.FAULT_ENGINE_DENSE_NONLEADER
                self._step_faults.append((g, self.fault[g]))
            return False
        if n_app >= capi.MAX_DENSE_APPENDS:
            self.fault[g] = capi.FAULT_ENGINE_DENSE_APPENDS
            self._step_faults.append((g, self.fault[g]))
            return False
        if acks is None:
            return True
        for _ in range(n_app):
            if self.fault[g]:
                break
            self._apply(g, rr.Command("ClientRequest", req_id=0), now, rows=False)
        for q in range(self.R):
            if q == s or self.fault[g]:
                continue
            h = int(acks[q, g])
            if h == capi.NO_ACK:
                continue
            self._apply(g, rr.Command("AppendResponse", node_id=self.node_ids[q], term=0, head=h, success=True), now,
                        rows=False)
        return True

    @staticmethod
    def _run_form(r):
        """"the leader's chain is a run: the id set is [0, top] and every block's parent its predecessor" - what the
        (from, n) mailbox encoding of AppendEntries can express, whatever head and id_gen are."""
        ch = r.chain
        keys = [k for k in ch.db.keys if len(k) == 8]
        for i, k in enumerate(keys):
            b = ch.db.map[k]
            if b.id != i or b.next != (i - 1 if i else 0):
                return False
        return len(keys) > 0

    @staticmethod
    def _top(r):
        keys = [k for k in r.chain.db.keys if len(k) == 8]
        return r.chain.db.map[keys[-1]].id if keys else 0

    def step_dense_leader(self, now_ms=0, acks=None, hbr_has=None, hbr_commit=None, tick=True):
        G, R = self.G, self.R
        acks = None if acks is None else np.asarray(acks, np.uint64).reshape(R, G)
        hbr_has = None if hbr_has is None else np.asarray(hbr_has, np.uint8).reshape(R, G)
        hbr_commit = None if hbr_commit is None else np.asarray(hbr_commit, np.uint64).reshape(R, G)
        out = None
        if tick:
            out = {"term": np.zeros(G, np.uint64), "hb_commit": np.full(G, capi.NO_ACK, np.uint64),
                   "ae_from": np.zeros((R, G), np.uint64), "ae_n": np.full((R, G), capi.AE_NONE, np.uint8)}
        if acks is None and hbr_has is None and not tick:
            return out
        self._begin()
        for g in range(G):
            if self.fault[g]:
                continue
            r = self.groups[g]
            s = int(self.slots[g])
            rows = self._step_msgs.setdefault(g, [])
            if r.role.name != "Leader" or (acks is not None and int(acks[s, g]) >= capi.MAX_DENSE_APPENDS):
                self._dense_acks_group(g, acks, now_ms)   # (raises the precondition faults)
                continue
            if hbr_has is not None:                       # 1. HeartbeatResponses, ascending slot
                for q in range(R):
                    if q == s or self.fault[g]:
                        continue
                    has = int(hbr_has[q, g])
                    if has == capi.HB_NONE:
                        continue
                    c = rr.Command("HeartbeatResponse", commit=0 if has else int(hbr_commit[q, g]), has_committed=bool(has))
                    rows.extend(self._apply(g, c, now_ms, rows=False))
            if not self.fault[g]:                         # 2. appends + acks
                self._dense_acks_group(g, acks, now_ms)
            if tick and not self.fault[g]:                # 3. Command::Tick -> columns (or rows)
                columns = self._run_form(r)
                trows = self._apply(g, rr.Command("Tick"), now_ms, rows=False)
                if not columns:
                    rows.extend(trows)
                else:
                    out["term"][g] = r.state.current_term
                    for (_g, kind, _tk, _flag, to_id, _from, _term, id_, aux) in trows:
                        if kind == capi.CMD_HEARTBEAT:
                            out["hb_commit"][g] = id_
                        else:
                            q = self.node_ids.index(to_id)
                            out["ae_from"][q, g] = id_
                            out["ae_n"][q, g] = aux
        self._end(fsm_rows=False)
        return out

    def step_dense_follower(self, now_ms, term, hb_commit, ae_from, ae_n, leader=None, leader_id=0, tick=True):
        G = self.G
        term, hb_commit, ae_from = (np.asarray(x, np.uint64) for x in (term, hb_commit, ae_from))
        ae_n = np.asarray(ae_n, np.uint8)
        out = {"ack_head": np.full(G, capi.NO_ACK, np.uint64), "hb_commit": np.zeros(G, np.uint64),
               "hb_has": np.full(G, capi.HB_NONE, np.uint8)}
        self._begin()
        for g in range(G):
            if self.fault[g]:
                continue
            r = self.groups[g]
            # (the leader half ticks the groups that are leaders when the round begins: one Tick per group and round)
            was_leader = r.role.name == "Leader"
            lead = int(leader[g]) if leader is not None else int(leader_id)
            rows = []
            if int(hb_commit[g]) != capi.NO_ACK:
                rows += self._apply(g, rr.Command("Heartbeat", term=int(term[g]), commit=int(hb_commit[g]), leader_id=lead),
                                    now_ms, rows=False)
            if int(ae_n[g]) != capi.AE_NONE and not self.fault[g]:
                f, n = int(ae_from[g]), int(ae_n[g])
                blocks = [rr.Block(f + 1 + k, f + k) for k in range(n)]
                rows += self._apply(g, rr.Command("AppendEntries", term=int(term[g]), leader_id=lead, blocks=blocks),
                                    now_ms, rows=False)
            if tick and not self.fault[g] and not was_leader and self.groups[g].role.name != "Leader":
                rows += self._apply(g, rr.Command("Tick"), now_ms, rows=False)
            keep = self._step_msgs.setdefault(g, [])
            for row in rows:
                (_g, kind, _tk, flag, _to, _from, _t, id_, _aux) = row
                if kind == capi.CMD_APPEND_RESPONSE:
                    out["ack_head"][g] = id_
                elif kind == capi.CMD_HEARTBEAT_RESPONSE:
                    out["hb_commit"][g] = id_
                    out["hb_has"][g] = flag
                else:
                    keep.append(row)
        self._end(fsm_rows=False)
        return out

    # ---- output ---------------------------------------------------------------------------------
    def drain_messages(self, copy=True):
        rows, self._msgs = self._msgs, []
        a = np.zeros(len(rows), dtype=capi.MSG_DTYPE)
        for i, (g, kind, to_kind, flag, to_id, frm, term, id_, aux) in enumerate(rows):
            a[i] = (g, kind, to_kind, flag, 0, to_id, frm, term, id_, aux)
        return a

    def drain_applies(self, copy=True):
        rows, self._fsm = self._fsm, []
        a = np.zeros(len(rows), dtype=capi.FSM_DTYPE)
        for i, (g, kind, x, y) in enumerate(rows):
            a[i] = (g, kind, (0, 0, 0), x, y)
        return a

    def drain_faults(self):
        rows, self._faults = self._faults, []
        a = np.zeros(len(rows), dtype=capi.FAULT_DTYPE)
        for i, (g, code) in enumerate(rows):
            a[i] = (g, code)
        return a

    def counters(self):
        return {"commands": self._n_cmds, "decisions": self._decisions, "dense_group_steps": 0, "launches": 0}

    def read(self, name, replica=0, g0=0, n=None):
        """jg_read_state (the RaftHandle introspection of mod.rs:437-468, column-wise)."""
        fld = capi.FIELD_NAMES[name]
        n = self.G - g0 if n is None else n
        out = np.zeros(n, dtype=capi.FIELD_DTYPES[fld])
        for i in range(n):
            out[i] = self._field(name, g0 + i, replica)
        return out

    def _field(self, name, g, replica):
        r = self.groups[g]
        role = r.role.name
        st = r.state
        if name == "term":
            return st.current_term
        if name == "voted_for":
            return st.voted_for or 0
        if name == "has_voted":
            return int(st.voted_for is not None)
        if name == "role":
            return {"Follower": capi.ROLE_FOLLOWER, "Candidate": capi.ROLE_CANDIDATE, "Leader": capi.ROLE_LEADER}[role]
        if name == "commit":
            return r.chain.commit
        if name == "head":
            return r.chain.head
        if name == "id_gen":
            return r.chain.id_gen
        if name == "match":
            return r.role.progress.progress[self.node_ids[replica]].head if role == "Leader" else 0
        if name == "repl_state":
            if role != "Leader":
                return 0
            return sum(1 << q for q in range(self.R)
                       if r.role.progress.progress[self.node_ids[q]].state == "Replicate")
        if name in ("vote_seen", "vote_granted"):
            if role != "Candidate":
                return 0
            v = r.role.election.votes
            return sum(1 << q for q in range(self.R)
                       if self.node_ids[q] in v and (name == "vote_seen" or v[self.node_ids[q]]))
        if name == "fault":
            return self.fault[g]
        if name == "leader_id":
            return (r.role.leader_id or 0) if role == "Follower" else 0
        if name == "has_leader":
            return int(role == "Follower" and r.role.leader_id is not None)
        if name == "election_time":
            return st.election_time or 0
        if name == "election_timeout":
            return st.election_timeout or 0
        if name == "heartbeat_time":
            return r.role.heartbeat_time if role == "Leader" else 0
        if name == "queued_reqs":
            return len(r.role.queued_reqs) if role != "Leader" else 0
        if name == "self_slot":
            return int(self.slots[g])
        raise KeyError(name)
