"""ref_py.raft — a SECOND, independent restatement of josefine's `src/raft` hot path, written
line by line from the Rust sources and from nothing else (in particular not from
oracle/raft_oracle.hpp and not from the device code).  TEST INFRASTRUCTURE: only tests/ may
import it.  Its purpose is to pin the C++ oracle (tests/test_ref_py_differential.py) and to
generate the committed golden fixtures (tests/golden/make_golden.py), so that parity does not
rest on one reading of the reference.

Shapes are kept as in the Rust: `HashMap` -> dict, `Vec` -> list, `sort_by(|a, b| b.cmp(a))` ->
sorted(reverse=True), sled -> a dict walked in key order (8-byte big-endian block keys plus the
literal b"commit" key, so that unbounded ranges behave as they do on sled), role changes ->
new role objects built exactly as the `From` impls build them, panics / `Err` -> `Panic`.

What is virtualised (DESIGN.md "Logical time and randomness"): `Instant::now()` is the `now`
attribute (ms), `thread_rng().gen_range(min..max)` is the injected `rand_range(lo, hi)`.

All citations are /root/reference/src/raft/<file>:<lines>.
"""
from __future__ import annotations

import bisect
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

MAX_INFLIGHT = 5  # progress.rs:117


class Panic(Exception):
    """The reference process would be gone: a panic!/assert!/expect/unimplemented!, or an `Err`
    that propagates out of `apply` (server.rs:125-159 unwinds the event loop on it)."""

    def __init__(self, where: str):
        super().__init__(where)
        self.where = where


# ---------------------------------------------------------------------------------- chain.rs
def block_key(block_id: int) -> bytes:
    """BlockId::new (chain.rs:63-67): the id as 8 big-endian bytes — also the sled key."""
    return int(block_id).to_bytes(8, "big")


COMMIT_KEY = b"commit"  # chain.rs:120,198


@dataclass
class Block:  # chain.rs:86-91 (data stays with the host)
    id: int
    next: int


class Sled:
    """What the reference uses of `sled::Db`: an ordered byte-key map (insert = upsert,
    contains_key, get, remove, range in key order)."""

    def __init__(self):
        self.map: Dict[bytes, object] = {}
        self.keys: List[bytes] = []  # sorted

    def contains_key(self, k: bytes) -> bool:
        return k in self.map

    def get(self, k: bytes):
        return self.map.get(k)

    def insert(self, k: bytes, v) -> None:
        if k not in self.map:
            bisect.insort(self.keys, k)
        self.map[k] = v

    def remove(self, k: bytes) -> None:
        if k in self.map:
            del self.map[k]
            del self.keys[bisect.bisect_left(self.keys, k)]

    def range(self, lo: Optional[bytes], hi: Optional[bytes], hi_inclusive: bool):
        """Lazy, in key order; lo inclusive.  (Snapshot of the key list: callers here never
        mutate while iterating forward.)"""
        i = 0 if lo is None else bisect.bisect_left(self.keys, lo)
        while i < len(self.keys):
            k = self.keys[i]
            if hi is not None and (k > hi or (k == hi and not hi_inclusive)):
                return
            yield k, self.map[k]
            i += 1


class Chain:
    """chain.rs:99-254 over `Sled`."""

    def __init__(self, db: Optional[Sled] = None, separate_commit_key: bool = False):
        # Chain::new, chain.rs:117-137.  `separate_commit_key` = JG_CFG_SEPARATE_COMMIT_KEY: the
        # engine option that keeps the "commit" key out of the block keyspace (no Q9)
        self.separate_commit_key = separate_commit_key
        self.db: Sled = Sled() if db is None else db           # sled::open(path)
        raw = self.db.get(COMMIT_KEY)                          # :119-123
        commit = int.from_bytes(raw, "big") if raw is not None else 0
        self.id_gen = commit                                   # :127
        self.commit = commit                                   # :128
        self.head = commit                                     # :129
        if commit == 0:                                        # :132-134
            self.init()

    def _next_id(self) -> int:                                 # IdGenerator::next, chain.rs:25-27
        v = self.id_gen
        self.id_gen += 1
        return v

    def init(self) -> None:                                    # chain.rs:139-153
        id_ = self._next_id()
        if id_ != 0:
            raise Panic("chain.rs:141 assert_eq!(id, 0)")
        self.db.insert(block_key(id_), Block(id_, id_))

    def has(self, block_id: int) -> bool:                      # chain.rs:155-157
        return self.db.contains_key(block_key(block_id))

    def append(self) -> int:                                   # chain.rs:160-175
        id_ = self._next_id()
        if not id_ > self.head:
            raise Panic("chain.rs:163 assert!(id > self.head)")
        block = Block(id_, self.head)
        self.db.insert(block_key(block.id), block)
        self.head = block.id
        return block.id

    def extend(self, block: Block) -> None:                    # chain.rs:178-192
        if not self.has(block.next):
            raise Panic("chain.rs:180-185 Err(block not found in chain)")
        self.db.insert(block_key(block.id), Block(block.id, block.next))
        self.head = block.id

    def commit_to(self, block_id: int) -> int:                 # Chain::commit, chain.rs:195-205
        if self.db.contains_key(block_key(block_id)):
            self.db.insert(COMMIT_KEY, block_key(block_id))
            self.commit = block_id
        else:
            raise Panic('chain.rs:201 panic!("")')
        return block_id

    def range(self, lo: Optional[int], hi: Optional[int], hi_inclusive: bool = False):
        """Chain::range (chain.rs:208-230): a LAZY iterator over the db in key order.  The map
        closures run per item pulled: an item that is not a bincode `Block` (the "commit" key's
        8-byte value) panics when it is reached (chain.rs:219-226), not before."""
        lo_k = block_key(lo) if lo is not None else None
        hi_k = block_key(hi) if hi is not None else None
        for _k, v in self.db.range(lo_k, hi_k, hi_inclusive):
            if not isinstance(v, Block):
                if self.separate_commit_key:
                    continue
                raise Panic("chain.rs:219-226 couldn't deserialize (range ran into the commit key)")
            yield v

    def compact(self) -> None:                                 # chain.rs:239-253
        next_id = None
        for b in reversed(list(self.range(0, self.commit))):   # range(0..commit).rev()
            if next_id is not None and b.id != next_id:
                self.db.remove(block_key(b.id))
            next_id = b.next


# ------------------------------------------------------------------------------- progress.rs
@dataclass
class Progress:  # progress.rs:119-125 (the type parameter is the `state` tag)
    node_id: int
    state: str   # "Probe" | "Replicate"
    active: bool = False
    head: int = 0

    def increment(self, block_id: int) -> bool:                # progress.rs:133-140
        if self.head < block_id:
            self.head = block_id
            return True
        return False

    def is_active(self) -> bool:
        # Probe: !paused, never paused (progress.rs:162-164); Replicate: capacity > len of an
        # inflight queue nothing ever pushes to (progress.rs:219-221)
        return True


def node_progress_advance(prog: Progress, block_id: int) -> Progress:  # NodeProgress::advance, progress.rs:76-94
    if prog.state == "Probe":
        if prog.increment(block_id):
            return Progress(prog.node_id, "Replicate", prog.active, prog.head)   # From<Probe>, :224-235
        return prog
    if prog.state == "Replicate":
        if prog.increment(block_id):
            return prog
        return Progress(prog.node_id, "Probe", prog.active, prog.head)           # From<Replicate>, :167-176
    raise Panic("progress.rs:92 panic!()")


class ReplicationProgress:  # progress.rs:9-60
    def __init__(self, nodes: List[int]):
        if not nodes:
            raise Panic("progress.rs:16 assert!(!nodes.is_empty())")
        self.progress: Dict[int, Progress] = {}
        for node_id in nodes:
            self.progress[node_id] = Progress(node_id, "Probe")                  # :19-21, Progress::new :152-160

    def get_mut(self, node_id: int) -> Optional[Progress]:
        return self.progress.get(node_id)

    def advance(self, node_id: int, block_id: int) -> None:                      # :42-46
        node = self.progress.pop(node_id, None)
        if node is None:
            raise Panic('progress.rs:43 expect("the node does not exist")')
        self.progress[node_id] = node_progress_advance(node, block_id)

    def committed_index(self) -> int:                                            # :48-60
        indices = [pr.head for pr in self.progress.values()]
        indices.sort(reverse=True)                                               # sort_by(|a, b| b.cmp(a))
        return indices[len(indices) // 2]


# ------------------------------------------------------------------------------- election.rs
class Election:  # election.rs:5-74
    def __init__(self, voter_ids: List[int]):
        self.voter_ids = list(voter_ids)
        self.votes: Dict[int, bool] = {}

    def reset(self) -> None:
        self.votes.clear()

    def vote(self, id_: int, vote: bool) -> None:                                # :33-35
        self.votes[id_] = vote

    def quorum_size(self) -> int:                                                # :66-73
        if len(self.voter_ids) == 1:
            return 0
        return len(self.voter_ids) // 2 + 1

    def election_status(self) -> str:                                            # :37-57
        votes = sum(1 for v in self.votes.values() if v)
        total = len(self.votes)
        if votes >= self.quorum_size():
            return "Elected"
        if total - votes == self.quorum_size():
            return "Defeated"
        return "Voting"


# ----------------------------------------------------------------------------- mod.rs, rpc.rs
@dataclass
class Command:  # mod.rs:160-227, flattened
    kind: str
    term: int = 0
    candidate_id: int = 0
    last_term: int = 0
    head: int = 0
    from_: int = 0
    granted: bool = False
    leader_id: int = 0
    blocks: List[Block] = field(default_factory=list)
    node_id: int = 0
    success: bool = False
    commit: int = 0
    has_committed: bool = False
    req_id: int = 0        # ClientRequest.id / ClientResponse.id (the payload stays with the host)
    range_start: int = 0   # harness annotation on AppendEntries: the `progress.head` the leader ranged from


Address = Tuple[str, int]  # ("Peers", 0) | ("Peer", id) | ("Local", 0) | ("Client", 0): rpc.rs:5-14


@dataclass
class Message:  # rpc.rs:17-27
    from_: Address
    to: Address
    command: Command


@dataclass
class State:  # mod.rs:271-322
    current_term: int = 0
    voted_for: Optional[int] = None
    election_time: Optional[int] = None
    election_timeout: Optional[int] = None
    min_election_timeout: int = 500
    max_election_timeout: int = 1000


@dataclass
class Follower:  # follower.rs:19-23
    leader_id: Optional[int] = None
    queued_reqs: List[int] = field(default_factory=list)
    name = "Follower"

    def term(self, _term: int) -> None:                                          # follower.rs:27-29
        self.leader_id = None


@dataclass
class Candidate:  # candidate.rs:17-21
    election: Election = None
    queued_reqs: List[int] = field(default_factory=list)
    name = "Candidate"

    def term(self, _term: int) -> None:                                          # candidate.rs:161-163
        self.election.reset()


@dataclass
class Leader:  # leader.rs:23-30
    progress: ReplicationProgress = None
    heartbeat_time: int = 0
    heartbeat_timeout: int = 100
    name = "Leader"

    def term(self, _term: int) -> None:                                          # leader.rs:33-35
        raise Panic("leader.rs:34 unimplemented!()")


class Raft:
    """Raft<T> (mod.rs:326-341) with the role object swapped in place where the Rust moves `self`
    through a `From` impl.  `rpc` / `fsm` are the two channels (mod.rs:337-340).  They also carry
    harness markers (queue bookkeeping, range bounds) for things the Rust does not put on a
    channel but the engine's row vocabulary spells out; the markers change nothing."""

    def __init__(self, id_: int, nodes: List[int], heartbeat_timeout: int, min_to: int, max_to: int,
                 rand_range: Callable[[int, int], int], db: Optional[Sled] = None, now: int = 0,
                 separate_commit_key: bool = False):
        # Raft::<Follower>::new, follower.rs:68-95
        self.id = id_
        self.nodes = list(nodes)          # config.nodes: the OTHER nodes, in configuration order
        self.heartbeat_timeout = heartbeat_timeout
        self.rand_range = rand_range
        self.now = now
        self.state = State(min_election_timeout=min_to, max_election_timeout=max_to)
        self.role = Follower()
        self.chain = Chain(db, separate_commit_key)
        self.rpc: list = []               # Message objects, in emission order (+ harness queue markers)
        self.fsm: List[tuple] = []        # ("Apply", block_id) / ("Notify", req_id, block_id) (+ "Range" markers)
        self.decisions = 0                # Leader::commit + election_status-on-vote evaluations (bench metric)
        self.set_election_timeout()       # init(), follower.rs:93-95

    # -- mod.rs ---------------------------------------------------------------------------
    def needs_election(self) -> bool:                                            # mod.rs:352-357
        st = self.state
        if st.election_time is not None and st.election_timeout is not None:
            return (self.now - st.election_time) > st.election_timeout
        return False

    def term(self, term: int) -> None:                                           # mod.rs:360-365
        self.state.voted_for = None
        self.state.current_term = term
        self.role.term(term)

    def send(self, to: Address, cmd: Command) -> None:                           # mod.rs:390-394
        self.rpc.append(Message(("Peer", self.id), to, cmd))

    def send_all(self, cmd: Command) -> None:                                    # mod.rs:396-400
        self.rpc.append(Message(("Peer", self.id), ("Peers", 0), cmd))

    def apply(self, cmd: Command) -> None:                                       # RaftHandle::apply, mod.rs:471-479
        r = self.role.name
        if r == "Follower":
            self.follower_apply(cmd)
        elif r == "Candidate":
            self.candidate_apply(cmd)
        else:
            self.leader_apply(cmd)

    # -- follower.rs ----------------------------------------------------------------------
    def follower_apply(self, cmd: Command) -> None:                              # follower.rs:36-64
        k = cmd.kind
        if k == "Tick":
            self.follower_apply_tick()
        elif k == "AppendEntries":
            self.follower_apply_append_entries(cmd.blocks, cmd.leader_id, cmd.term)
        elif k == "Heartbeat":
            self.follower_apply_heartbeat(cmd.leader_id, cmd.term, cmd.commit)
        elif k == "VoteRequest":
            self.follower_apply_vote_request(cmd.candidate_id, cmd.last_term, cmd.head)
        elif k == "Timeout":
            self.follower_apply_timeout()
        elif k == "ClientRequest":
            self.follower_apply_client_request(cmd.req_id)
        elif k == "ClientResponse":
            self.follower_apply_client_response(cmd.req_id)
        # _ => apply_self

    def can_vote(self, last_term: int, head: int) -> bool:                       # follower.rs:97-101
        return not (self.state.voted_for is not None
                    or self.state.current_term > last_term
                    or self.chain.commit > head)

    def set_election_timeout(self) -> None:                                      # follower.rs:103-113
        st = self.state
        if not st.min_election_timeout < st.max_election_timeout:
            raise Panic("follower.rs:105 gen_range on an empty range")
        st.election_timeout = self.rand_range(st.min_election_timeout, st.max_election_timeout)
        st.election_time = self.now

    def follower_apply_tick(self) -> None:                                       # follower.rs:121-128
        if self.needs_election():
            self.apply(Command("Timeout"))

    def follower_apply_append_entries(self, blocks: List[Block], leader_id: int, term: int) -> None:  # :130-176
        if self.state.voted_for is None and term >= self.state.current_term:    # :137
            self.term(term)                                                      # :138
            self.state.election_time = self.now                                  # :141
            self.role.leader_id = leader_id                                      # :142
            self.state.voted_for = leader_id                                     # :143
        if self.state.voted_for is not None:                                     # :147-154
            voted_for = self.state.voted_for
            if voted_for != leader_id and term < self.state.current_term:
                raise Panic("follower.rs:149 assert!(!(voted_for != leader_id && term < current_term))")
        if blocks:                                                               # :157
            for block in blocks:
                self.chain.extend(block)                                         # :159 (`?`)
            self.rpc.append(Message(("Peer", self.id), ("Peer", leader_id),     # :163-172
                                    Command("AppendResponse", node_id=self.id, term=self.state.current_term,
                                            head=self.chain.head, success=True)))

    def follower_apply_heartbeat(self, leader_id: int, term: int, commit: int) -> None:  # follower.rs:178-217
        self.set_election_timeout()                                              # :184
        self.term(term)                                                          # :185
        self.role.leader_id = leader_id                                          # :186
        self.state.voted_for = leader_id                                         # :187
        queued, self.role.queued_reqs = self.role.queued_reqs, []                # :190 mem::take
        for req in queued:                                                       # :190-197
            self.send(("Peer", leader_id), Command("ClientRequest", req_id=req))
        has_committed = self.chain.has(commit)                                   # :200
        if has_committed and commit > self.chain.commit:                         # :201
            prev = self.chain.commit
            self.chain.commit_to(commit)                                         # :203
            self.fsm.append(("Range", "follower", prev, commit))             # (harness marker: range(prev..commit))
            for block in self.chain.range(prev, commit):                         # :204 range(prev..commit)
                self.fsm.append(("Apply", block.id))
        self.send(("Peer", leader_id),                                           # :209-215
                  Command("HeartbeatResponse", commit=self.chain.commit, has_committed=has_committed))

    def follower_apply_vote_request(self, candidate_id: int, last_term: int, head: int) -> None:  # :219-246
        if self.can_vote(last_term, head):
            self.send(("Peer", candidate_id),
                      Command("VoteResponse", term=self.state.current_term, from_=self.id, granted=True))
            self.state.voted_for = candidate_id                                  # :234
        else:
            self.send(("Peer", candidate_id),
                      Command("VoteResponse", term=self.state.current_term, from_=self.id, granted=False))

    def follower_apply_timeout(self) -> None:                                    # follower.rs:248-256
        if self.state.voted_for is None:
            self.set_election_timeout()
            self.candidate_from_follower()
            self.seek_election()

    def follower_apply_client_request(self, req_id: int) -> None:                # follower.rs:258-270
        if self.role.leader_id is not None:
            self.send(("Peer", self.role.leader_id), Command("ClientRequest", req_id=req_id))
        else:
            self.role.queued_reqs.append(req_id)
            self.rpc.append(("queue_push", req_id))                               # (harness marker)

    def follower_apply_client_response(self, req_id: int) -> None:               # follower.rs:272-282
        self.send(("Client", 0), Command("ClientResponse", req_id=req_id))

    def candidate_from_follower(self) -> None:                                   # From, follower.rs:285-304
        node_ids = list(self.nodes)
        node_ids.append(self.id)
        if self.role.queued_reqs:
            self.rpc.append(("queue_drop", len(self.role.queued_reqs)))          # queued_reqs: Vec::new() (:295)
        self.role = Candidate(election=Election(node_ids), queued_reqs=[])

    # -- candidate.rs ---------------------------------------------------------------------
    def seek_election(self) -> None:                                             # candidate.rs:24-45
        self.state.voted_for = self.id
        self.state.current_term += 1
        from_ = self.id
        term = self.state.current_term
        for _node in self.nodes:                                                 # :30-37: one broadcast per configured node
            self.send_all(Command("VoteRequest", term=term, candidate_id=from_, last_term=term,
                                  head=self.chain.head))
        self.apply(Command("VoteResponse", from_=from_, term=term, granted=True))  # :40-44

    def candidate_apply(self, cmd: Command) -> None:                             # candidate.rs:170-196
        k = cmd.kind
        if k == "Tick":
            self.candidate_apply_tick()
        elif k == "VoteRequest":
            self.candidate_apply_vote_request(cmd.candidate_id, cmd.term)
        elif k == "VoteResponse":
            self.candidate_apply_vote_response(cmd.granted, cmd.from_)
        elif k == "AppendEntries":
            self.candidate_apply_append_entries(cmd.term)
        elif k == "Heartbeat":
            self.candidate_apply_heartbeat(cmd.term, cmd.leader_id, cmd.commit)
        elif k == "ClientRequest":
            self.role.queued_reqs.append(cmd.req_id)                             # :190-193
            self.rpc.append(("queue_push", cmd.req_id))                          # (harness marker)

    def candidate_apply_tick(self) -> None:                                      # candidate.rs:48-68
        if self.needs_election():
            status = self.role.election.election_status()
            if status in ("Voting", "Defeated"):
                self.state.voted_for = None                                      # :53 / :59
                self.follower_from_candidate()
                self.apply(Command("Timeout"))
                return
            raise Panic('candidate.rs:64 panic!("this should never happen")')

    def candidate_apply_vote_request(self, candidate_id: int, term: int) -> None:  # candidate.rs:71-88
        if term > self.state.current_term:
            self.term(term)
            self.follower_from_candidate()
            return
        self.send(("Peer", candidate_id),
                  Command("VoteResponse", from_=self.id, term=self.state.current_term, granted=False))

    def candidate_apply_vote_response(self, granted: bool, from_: int) -> None:  # candidate.rs:91-98
        self.role.election.vote(from_, granted)
        self.decisions += 1
        status = self.role.election.election_status()
        if status == "Elected":
            self.elect()
        elif status == "Defeated":
            self.defeat()

    def defeat(self) -> None:                                                    # candidate.rs:101-105
        self.state.voted_for = None
        self.follower_from_candidate()

    def elect(self) -> None:                                                     # candidate.rs:108-113
        self.leader_from_candidate()
        self.heartbeat()

    def candidate_apply_append_entries(self, term: int) -> None:                 # candidate.rs:116-134
        if term >= self.state.current_term:
            self.follower_from_candidate()

    def candidate_apply_heartbeat(self, term: int, leader_id: int, commit: int) -> None:  # candidate.rs:137-157
        has_committed = self.chain.has(commit)
        commit = self.chain.commit
        self.term(term)
        self.state.voted_for = leader_id
        self.follower_from_candidate()
        self.send(("Peer", leader_id), Command("HeartbeatResponse", commit=commit, has_committed=has_committed))

    def follower_from_candidate(self) -> None:                                   # From, candidate.rs:198-214
        self.role = Follower(leader_id=None, queued_reqs=self.role.queued_reqs)

    def leader_from_candidate(self) -> None:                                     # From, candidate.rs:216-238
        nodes = list(self.nodes)
        nodes.append(self.id)
        if self.role.queued_reqs:
            self.rpc.append(("queue_drop", len(self.role.queued_reqs)))          # Leader has no queue
        self.role = Leader(progress=ReplicationProgress(nodes), heartbeat_time=self.now,
                           heartbeat_timeout=self.heartbeat_timeout)

    # -- leader.rs ------------------------------------------------------------------------
    def heartbeat(self) -> None:                                                 # leader.rs:44-51
        self.send_all(Command("Heartbeat", term=self.state.current_term, commit=self.chain.commit,
                              leader_id=self.id))

    def needs_heartbeat(self) -> bool:                                           # leader.rs:78-80
        return (self.now - self.role.heartbeat_time) > self.role.heartbeat_timeout

    def leader_commit(self) -> int:                                              # leader.rs:87-99
        self.decisions += 1
        quorum_idx = self.role.progress.committed_index()
        if quorum_idx > self.chain.commit:
            prev = self.chain.commit
            new = self.chain.commit_to(quorum_idx)
            self.fsm.append(("Range", "leader", prev, new))                  # (harness marker: range(prev..=new).skip(1))
            first = True
            for block in self.chain.range(prev, new, hi_inclusive=True):         # range(prev..=new).skip(1)
                if first:
                    first = False
                    continue
                self.fsm.append(("Apply", block.id))
        return quorum_idx

    def replicate(self) -> None:                                                 # leader.rs:124-174
        for node in self.nodes:
            progress = self.role.progress.get_mut(node)
            if progress is None:
                continue
            if not progress.is_active():
                continue
            if progress.state == "Probe":                                        # :133-149
                it = self.chain.range(progress.head, None)                       # range(progress.head..)
                blocks = []
                nth1 = None
                for i, b in enumerate(it):                                       # .nth(1)
                    if i == 1:
                        nth1 = b
                        break
                if nth1 is not None:
                    blocks = [nth1]
            else:                                                                # Replicate, :150-168
                it = self.chain.range(progress.head, None)
                blocks = []
                skipped = False
                for b in it:                                                     # .skip(1).take(MAX_INFLIGHT)
                    if not skipped:
                        skipped = True
                        continue
                    blocks.append(b)
                    if len(blocks) == MAX_INFLIGHT:
                        break
            self.rpc.append(Message(("Peer", self.id), ("Peer", node),
                                    Command("AppendEntries", term=self.state.current_term, leader_id=self.id,
                                            blocks=blocks, range_start=progress.head)))

    def leader_apply_client_request(self, req_id: int) -> None:                  # leader.rs:177-197
        term = self.state.current_term
        block_id = self.chain.append()                                           # :180
        node_id = self.id
        self.fsm.append(("Notify", req_id, block_id))                            # :184-188
        head = self.chain.head
        self.apply(Command("AppendResponse", node_id=node_id, term=term, success=True, head=head))

    def leader_apply_append_entries(self, term: int) -> None:                    # leader.rs:200-208
        if term > self.state.current_term:
            self.term(term)
            self.follower_from_leader()

    def leader_apply_append_response(self, node_id: int, head: int) -> None:     # leader.rs:211-219
        self.role.progress.advance(node_id, head)
        self.leader_commit()

    def leader_apply_heartbeat_response(self, commit: int, has_committed: bool) -> None:  # leader.rs:222-231
        if not has_committed and commit > 0:
            self.replicate()

    def leader_apply_tick(self) -> None:                                         # leader.rs:234-245
        if self.needs_heartbeat():
            self.heartbeat()
            self.role.heartbeat_time = self.now                                  # reset_heartbeat_timer, :82-84
        self.replicate()

    def leader_apply(self, cmd: Command) -> None:                                # leader.rs:248-266
        k = cmd.kind
        if k == "Tick":
            self.leader_apply_tick()
        elif k == "HeartbeatResponse":
            self.leader_apply_heartbeat_response(cmd.commit, cmd.has_committed)
        elif k == "AppendResponse":
            self.leader_apply_append_response(cmd.node_id, cmd.head)
        elif k == "AppendEntries":
            self.leader_apply_append_entries(cmd.term)
        elif k == "ClientRequest":
            self.leader_apply_client_request(cmd.req_id)

    def follower_from_leader(self) -> None:                                      # From, leader.rs:268-284
        self.role = Follower(leader_id=None, queued_reqs=[])
