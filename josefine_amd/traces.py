"""Host-side builders of the synthetic command traces: pure row
generation with numpy — no Raft logic, no oracle.  Used by bench.py, the smoke test and
the parity tests so that all of them drive the engine with the same inputs.

`hash(seed, tick, g, r)` is the counter-based hash of DESIGN.md "Synthetic traces" (the
same function the device generator k_synth_acks evaluates), keyed by the *global* group id
so that any shard regenerates its own slice.
"""
from __future__ import annotations

import numpy as np

from . import _capi as capi


def mix64(z):
    """splitmix64 finaliser, vectorised (uint64 wrap-around arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_hash(seed, tick, gg, r):
    with np.errstate(over="ignore"):
        a = mix64(np.uint64(seed) + np.uint64(tick) * np.uint64(0x9E3779B97F4A7C15))
        return mix64(a ^ (np.asarray(gg, dtype=np.uint64) * np.uint64(8) + np.uint64(r)))


def synth_fill_acks_host(seed, mode, tick, group_base, slots, sim, n_replicas):
    """One dense [R][G] tick of the synthetic AppendEntries-ack stream (DESIGN.md "Synthetic
    traces") in numpy — the same stream k_synth_acks generates on the device, for any shard:
    `sim` ([R][G] uint64, zero-initialised by the caller) is the generator's follower model and is
    updated in place.  mode 0: one append, every follower acks the previous leader head; mode 1:
    hash % 3 appends, 5 % dropped, 5 % stale duplicates, else min(lead, prev + U{0..5})."""
    R, G = n_replicas, sim.shape[1]
    slots = np.asarray(slots, dtype=np.int64)
    gi = np.arange(G)
    gg = gi.astype(np.uint64) + np.uint64(group_base)
    acks = np.zeros((R, G), dtype=np.uint64)
    lead = sim[slots, gi].copy()
    with np.errstate(over="ignore"):
        # the own slot's hash: r differs per group
        n_app = np.ones(G, np.uint64) if mode == 0 else \
            mix64(mix64(np.uint64(seed) + np.uint64(tick) * np.uint64(0x9E3779B97F4A7C15)) ^ (gg * np.uint64(8) + slots.astype(np.uint64))) % np.uint64(3)
        for r in range(R):
            other = slots != r
            if mode == 0:
                a, new = lead, lead
            else:
                u = synth_hash(seed, tick, gg, r)
                p = u % np.uint64(100)
                adv = (u >> np.uint64(32)) % np.uint64(capi.MAX_INFLIGHT + 1)
                v = np.minimum(lead, sim[r] + adv)
                a = np.where(p < 5, np.uint64(capi.NO_ACK), np.where(p < 10, sim[r], v))
                new = np.where(p < 10, sim[r], v)
            acks[r] = np.where(other, a, acks[r])
            sim[r] = np.where(other, new, sim[r])
        acks[slots, gi] = n_app
        sim[slots, gi] = lead + n_app
    return acks


def elect_all(e, now_ms: int = 0) -> None:
    """Make the local instance of every group the leader at term 1 by the reference's own
    path: Timeout -> candidate + self-vote (follower.rs:248-256, candidate.rs:24-45), then
    granted VoteResponses from the next R/2 slots until quorum (candidate.rs:91-113)."""
    from .engine import Command

    e.apply_all(Command.Timeout(), now_ms)
    if e.R == 1:
        return
    slots = e.read("self_slot")
    ids = np.array(e.node_ids, dtype=np.uint32)
    n = e.G
    for k in range(1, e.R // 2 + 1):  # quorum = R/2 + 1 including the self-vote
        voter = ids[(slots.astype(np.int64) + k) % e.R]
        e.submit_columns(np.full(n, capi.CMD_VOTE_RESPONSE, np.uint8), np.arange(n, dtype=np.uint32),
                         from_=voter, term=np.ones(n, np.uint64), flag=np.ones(n, np.uint8))
        e.step(now_ms)


def elect_where(e, mask, now_ms: int = 0) -> None:
    """elect_all for the groups in `mask` only: Timeout + granted VoteResponses from the next R/2 slots."""
    g = np.nonzero(mask)[0].astype(np.uint32)
    n = len(g)
    if not n:
        return
    e.submit_columns(np.full(n, capi.CMD_TIMEOUT, np.uint8), g)
    e.step(now_ms)
    slots = e.read("self_slot")[g].astype(np.int64)
    ids = np.array(e.node_ids, dtype=np.uint32)
    for k in range(1, e.R // 2 + 1):
        e.submit_columns(np.full(n, capi.CMD_VOTE_RESPONSE, np.uint8), g, from_=ids[(slots + k) % e.R],
                         term=np.ones(n, np.uint64), flag=np.ones(n, np.uint8))
        e.step(now_ms)


def failure_rows(seed, tick, group_base, G, R, node_ids, self_slots, percent=1):
    """BASELINE.json configs[4]: the command rows of one tick's leader failures.

    Every group fails with probability percent/100 (hash salt 7): the local (leader) instance
    crashes and restarts (JG_CMD_RESTART = Raft::new + Chain::new on the persisted tree), times
    out with voted_for == None — the only way a node can campaign in the reference (SURVEY.md
    §7.3 Q4) — and receives granted VoteResponses from the next R/2 replicas.  Returns (columns
    for submit_columns / upload_rows, number of failing groups)."""
    gg = np.arange(G, dtype=np.uint64) + np.uint64(group_base)
    failing = np.nonzero(synth_hash(seed, tick, gg, 7) % np.uint64(100) < np.uint64(percent))[0].astype(np.uint32)
    n = len(failing)
    ids = np.array(node_ids, dtype=np.uint32)
    kind = [np.full(n, capi.CMD_RESTART, np.uint8), np.full(n, capi.CMD_TIMEOUT, np.uint8)]
    group = [failing, failing]
    frm = [np.zeros(n, np.uint32), np.zeros(n, np.uint32)]
    for k in range(1, R // 2 + 1):
        kind.append(np.full(n, capi.CMD_VOTE_RESPONSE, np.uint8))
        group.append(failing)
        frm.append(ids[(np.asarray(self_slots)[failing].astype(np.int64) + k) % R])
    kind, group, frm = np.concatenate(kind), np.concatenate(group), np.concatenate(frm)
    return dict(kind=kind, group=group, from_=frm, term=np.ones(len(kind), np.uint64),
                flag=np.ones(len(kind), np.uint8)), n


def cluster_failure_rows(seed, tick, G, R, percent=1, lead=0, candidate=1, also=(), group_base=0):
    """BASELINE.json configs[4] on a cluster of R nodes: every group fails with
    probability percent/100 per tick (the hash of failure_rows: same failing groups).  In a failing
    group the leader's replica crashes and restarts (State::default(), Chain::new on the persisted
    tree) and a designated follower — restarted as well, so that voted_for == None: nobody else may
    campaign — receives Timeout.  The other replicas answer its VoteRequests
    through can_vote when the transport delivers them, next round.
    `also`: more replicas that crash and restart with the leader (a rack going down).
    Returns one column dict (kind, group; group-sorted) or None per node, for
    jg_dense_cluster_round_routed's `inject`."""
    gg = np.arange(G, dtype=np.uint64) + np.uint64(group_base)
    failing = np.nonzero(synth_hash(seed, tick, gg, 7) % np.uint64(100) < np.uint64(percent))[0].astype(np.uint32)
    n = len(failing)
    out = [None] * R
    if n:
        restart = np.full(n, capi.CMD_RESTART, np.uint8)
        out[lead] = dict(kind=restart, group=failing)
        for r in also:
            out[r] = dict(kind=restart, group=failing)
        # group-sorted: Restart then Timeout of each failing group
        out[candidate] = dict(kind=np.stack([restart, np.full(n, capi.CMD_TIMEOUT, np.uint8)], axis=1).reshape(-1),
                              group=np.repeat(failing, 2))
    return out


def any_failure_rows(seed, tick, G, R, percent, leader_of, group_base=0, whole_group=True, skip=None, recreate=False):
    """configs[4] with PER-PARTITION LEADERSHIP (jg_dense_cluster_create, JG_CLUSTER_ANY_LEADER): every group fails with
    probability percent/100 per tick (the hash of failure_rows); in a failing group the leader's replica
    (leader_of[g]) crashes and restarts, the next replica - restarted too: voted_for == None, SURVEY.md 7.3 Q4 - receives
    Timeout and campaigns; whole_group: every other replica restarts as well (a rack going down: otherwise they
    remember their vote and refuse, and the group stays leaderless).  `skip`: groups left alone.  `recreate`: the replicas come
    back on EMPTY stores (JG_CMD_RECREATE instead of JG_CMD_RESTART: the rack's disks are gone) - the winner of the election is
    then a leader that can append (Q8 does not apply to a chain that starts over), so a group may fail again and again and the
    trace is stationary (bench.py --cluster --any-leader --failures p --recreate).
    Returns (one group-sorted column dict or None per node, the failing groups)."""
    gg = np.arange(G, dtype=np.uint64) + np.uint64(group_base)
    failing = synth_hash(seed, tick, gg, 7) % np.uint64(100) < np.uint64(percent)
    if skip is not None:
        failing &= ~np.asarray(skip, bool)
    failing = np.nonzero(failing)[0].astype(np.uint32)
    lead = np.asarray(leader_of)[failing].astype(np.int64)
    out = [None] * R
    for n in range(R):
        cand = (lead + 1) % R == n
        restart = (lead == n) | cand | bool(whole_group)
        # per failing group, in group order: Restart (if this node restarts), then Timeout (if it is the candidate)
        m = len(failing)
        kind = np.stack([np.full(m, capi.CMD_RECREATE if recreate else capi.CMD_RESTART, np.uint8), np.full(m, capi.CMD_TIMEOUT, np.uint8)], axis=1).reshape(-1)
        keep = np.stack([restart, cand], axis=1).reshape(-1)
        group = np.repeat(failing, 2)
        if keep.any():
            out[n] = dict(kind=kind[keep], group=group[keep])
    return out, failing


class FailureRepairTrace:
    """BASELINE.json configs[4] as a STATIONARY trace (cluster_failure_rows' failures + a repair schedule).

    Failure, every tick, every partition that is up with probability percent/100 (the hash of failure_rows): the leader's
    replica (`lead`) crashes and restarts, a designated follower (`candidate`) - restarted as well, so that voted_for ==
    None: nobody else may campaign - receives Timeout and campaigns; the other replicas remember their
    vote and refuse, so the partition stays leaderless and its candidate campaigns again at every election timeout -
    every VoteRequest / VoteResponse of that travels through the cluster's transport.  The client stops proposing to it.
    Repair, `repair_after` ticks later (an operator's runbook: SURVEY.md 7.3 Q4 allows any explicit trace).  What the
    reference leaves of a partition whose leader failed cannot be brought back to where it was: no replica that led or
    followed it can ever append again (Q8: chain.rs:163 - `id_gen` is re-seeded below the head), and a new leader re-sends
    the chain from block 1 (Q10), longer with every round the benchmark has run.  So the operator RE-CREATES the partition:
    every replica restarts on an empty data directory (JG_CMD_RECREATE: Raft::new + Chain::new's
    genesis, chain.rs:117-153) and replica `lead` receives Timeout: it campaigns, its VoteRequests are routed and answered
    through can_vote like any, the answers are routed back, and TWO ticks after the re-creation it counts them and is
    ELECTED by them (candidate.rs:91-113) - NO synthetic vote anywhere in the trace since round 6: the transport delivers
    every voter's first answer before anybody's second (jg_route.h), so the quorum of grants is seen before the refusals
    of the further copies overwrite them (election.rs:33-35), also with five nodes.  Its Heartbeat brings the others in,
    and the client proposes again.  The partition is then exactly what every partition was at tick 0: the trace is
    stationary in everything - leaderless fraction (about percent x (repair_after + 2) / 100), decisions per round, cost
    per round.
    rows(tick) must be called for tick = 0, 1, 2, ... in order; returns (per node one group-sorted column dict or None - for
    jg_dense_cluster_round_routed's `inject` -, the partitions failing this tick, the partitions whose new leader is seated
    in this tick's round);
    appends() = the ClientRequests per partition for the round of the last rows(): 0 where the partition is down (the
    cluster offers them where the lead node leads when the dense round begins: a partition repaired in this round has its
    leader by then)."""

    def __init__(self, seed, G, R, percent=1, repair_after=10, lead=0, candidate=1, group_base=0, node_ids=None):
        self.seed, self.G, self.R, self.percent, self.D = seed, G, R, percent, int(repair_after)
        self.lead, self.candidate, self.group_base = lead, candidate, group_base
        self.ids = np.arange(1, R + 1, dtype=np.uint32) if node_ids is None else np.asarray(node_ids, np.uint32)
        self.down_since = np.full(G, -1, np.int64)
        self.ever_failed = np.zeros(G, bool)
        self.campaigning = np.zeros(G, np.int8)  # 1: re-created in the last tick (its VoteRequests are being answered), 2: the tick before (elected in this one)
        self.next_tick = 0

    def leaderless(self):
        return self.down_since >= 0

    def appends(self):
        return np.where(self.down_since >= 0, 0, 1).astype(np.uint64)

    def rows(self, tick):
        assert tick == self.next_tick, "FailureRepairTrace.rows: ticks in order"
        self.next_tick += 1
        G, R = self.G, self.R
        gg = np.arange(G, dtype=np.uint64) + np.uint64(self.group_base)
        repaired = np.nonzero(self.campaigning == 2)[0].astype(np.uint32)  # the answers arrive in this round: elected, up again
        recreated = np.nonzero((self.down_since >= 0) & (self.down_since == tick - self.D))[0].astype(np.uint32)
        self.down_since[repaired] = -1
        self.campaigning[repaired] = 0
        self.campaigning[self.campaigning == 1] = 2
        self.campaigning[recreated] = 1
        hit = synth_hash(self.seed, tick, gg, 7) % np.uint64(100) < np.uint64(self.percent)
        if tick == 0:
            # (no failure before the cluster's first round: a follower that has not yet heard from its leader has not voted
            # - follower.rs:143,187 - and GRANTS; the designated candidate would simply be elected, at any R since the transport
            # interleaves the voters' answers, and the partition would not be the trace's "leaderless until repaired")
            hit[:] = False
        up = self.down_since < 0
        up[repaired] = False  # (not in the tick of its repair)
        failing = np.nonzero(hit & up)[0].astype(np.uint32)
        self.down_since[failing] = tick
        self.ever_failed[failing] = True
        out = [None] * R
        nf, nc = len(failing), len(recreated)
        if not nf and not nc:
            return out, failing, repaired
        for n in range(R):
            kinds, groups, froms, flags, within = [], [], [], [], []

            def add(g, *rows_):  # rows_: (kind, from, flag) per partition of g, in that order
                for k, (kind, frm, flag) in enumerate(rows_):
                    kinds.append(np.full(len(g), kind, np.uint8)), groups.append(g), within.append(np.full(len(g), k, np.int64))
                    froms.append(np.full(len(g), frm, np.uint32)), flags.append(np.full(len(g), flag, np.uint8))

            if nf and n == self.lead:
                add(failing, (capi.CMD_RESTART, 0, 0))
            elif nf and n == self.candidate:
                add(failing, (capi.CMD_RESTART, 0, 0), (capi.CMD_TIMEOUT, 0, 0))
            if nc and n == self.lead:
                add(recreated, (capi.CMD_RECREATE, 0, 0), (capi.CMD_TIMEOUT, 0, 0))
            elif nc:
                add(recreated, (capi.CMD_RECREATE, 0, 0))
            if not kinds:
                continue
            group = np.concatenate(groups)
            # group-sorted, a partition's rows in the order given above (the two sets are disjoint)
            order = np.lexsort((np.concatenate(within), group))
            out[n] = dict(kind=np.concatenate(kinds)[order], group=group[order], from_=np.concatenate(froms)[order],
                          term=np.ones(len(group), np.uint64), flag=np.concatenate(flags)[order])
        return out, failing, repaired
