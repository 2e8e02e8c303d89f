"""Random command traces for parity fuzzing.  Values are drawn near the current
state of the *oracle* (terms around current_term, ids around head, node ids from
the membership plus an occasional stranger) so that every branch of every role
is reached, including the reference's panic / Err paths."""
import weakref

import numpy as np

from josefine_amd import capi

KINDS = [capi.CMD_TICK, capi.CMD_PROPOSE, capi.CMD_VOTE_REQUEST, capi.CMD_VOTE_RESPONSE,
         capi.CMD_APPEND_ENTRIES, capi.CMD_APPEND_RESPONSE, capi.CMD_HEARTBEAT,
         capi.CMD_HEARTBEAT_RESPONSE, capi.CMD_TIMEOUT, capi.CMD_NOOP, capi.CMD_CLIENT_REQUEST,
         capi.CMD_CLIENT_RESPONSE, capi.CMD_RESTART]
# weights: keep RESTART / TIMEOUT rare enough that leaders live for a while
WEIGHTS = np.array([6, 1, 6, 10, 10, 16, 8, 4, 3, 1, 12, 2, 1], dtype=np.float64)
WEIGHTS /= WEIGHTS.sum()


_TRACK = {}  # id(budget array) -> what random_batch remembers about the chains it has built (see below)


def random_batch(rng: np.random.Generator, ora, n: int, foreign_voters: bool = False, budget=None):
    """n random command rows + block side arrays, as kwargs for submit_columns.

    `budget` ([G] int array, updated in place) bounds the gaps / forks generated per
    group so that chains stay within the engine's JG_CHAIN_WINDOW segments."""
    G, R = ora.G, ora.R
    track = None
    if budget is None:
        budget = np.full(G, 1 << 30)
    else:
        # per budget array: the blocks sent with a parent other than id-1, and the highest id the group
        # can hold (a Restart puts the head back at the commit index but keeps the stored blocks)
        track = _TRACK.get(id(budget))
        if track is None or track["of"]() is not budget:  # (ids are reused once an array is gone)
            track = _TRACK[id(budget)] = {"of": weakref.ref(budget), "forks": {}, "hi": np.zeros(G, np.int64)}
    ids = np.array(ora.node_ids, dtype=np.uint32)
    term_now = ora.read("term").astype(np.int64)
    head_now = ora.read("head").astype(np.int64)
    commit_now = ora.read("commit").astype(np.int64)
    group = rng.integers(0, G, n).astype(np.uint32)
    kind = rng.choice(KINDS, size=n, p=WEIGHTS).astype(np.uint8)
    from_ = ids[rng.integers(0, R, n)]
    stranger = rng.random(n) < 0.02
    if not foreign_voters:
        stranger &= kind != capi.CMD_VOTE_RESPONSE
    # strangers: one foreign id, or (foreign_voters) six of them - Election::vote counts whoever
    # answers (election.rs:33-35); the engine remembers up to JG_FOREIGN_VOTERS distinct ones
    sid = np.uint32(4242) if not foreign_voters else (4242 + rng.integers(0, 6, n)).astype(np.uint32)
    from_ = np.where(stranger, sid, from_).astype(np.uint32)
    term = np.maximum(term_now[group] + rng.integers(-1, 3, n), 0).astype(np.uint64)
    idv = np.maximum(head_now[group] + rng.integers(-2, 3, n), 0)
    use_commit = rng.random(n) < 0.3
    idv = np.where(use_commit, np.maximum(commit_now[group] + rng.integers(-1, 2, n), 0), idv).astype(np.uint64)
    aux = np.maximum(term_now[group] + rng.integers(-1, 2, n), 0).astype(np.uint64)  # last_term
    flag = (rng.random(n) < 0.7).astype(np.uint8)
    blk_id, blk_next = [], []
    id_col = idv.copy()
    for i in np.nonzero(kind == capi.CMD_APPEND_ENTRIES)[0]:
        nb = int(rng.integers(0, 4))
        id_col[i] = len(blk_id)
        aux[i] = nb
        h = int(head_now[group[i]])
        forks = track["forks"].setdefault(int(group[i]), {}) if track else {}
        hi = max(int(track["hi"][group[i]]), h) if track else h
        for _ in range(nb):
            r = rng.random()
            if r >= 0.75 and r < 0.9:  # a fork costs the engine at most two segments (it may land inside a run)
                if budget[group[i]] < 2:
                    r = 0.0
                else:
                    budget[group[i]] -= 2
            if r < 0.75:       # regular extension of the follower's chain
                nid, nxt = h + 1, h
                # ... which, after a Restart put the head back at the commit index, can re-send an id
                # that an earlier fork stored with another parent: an overwrite, up to two segments
                if forks.get(nid, nxt) != nxt:
                    if budget[group[i]] < 2:
                        blk_id.append(h + 2), blk_next.append(hi + 7)  # no budget: the Err case instead
                        h = h + 2
                        continue
                    budget[group[i]] -= 2
            elif r < 0.9:      # fork / gap with an existing parent
                nid, nxt = h + int(rng.integers(1, 4)), max(h - int(rng.integers(0, 2)), 0)
            else:              # missing parent -> Err (chain.rs:180-185)
                nid, nxt = h + 2, hi + 7
            hi = max(hi, nid)
            if r < 0.9:
                if nxt != nid - 1:
                    forks[nid] = nxt
                else:
                    forks.pop(nid, None)
            blk_id.append(nid)
            blk_next.append(nxt)
            h = nid
        head_now[group[i]] = h  # keep later rows of this batch plausible
        if track:
            track["hi"][group[i]] = hi
    return dict(kind=kind, group=group, from_=from_, term=term, id=id_col, aux=aux, flag=flag,
                blk_id=np.array(blk_id, dtype=np.uint64), blk_next=np.array(blk_next, dtype=np.uint64))


# ---- a state-aware stream ------------------------------------------------------------------------
# random_batch draws the command kind blind to the role, so most of its rows hit followers or groups
# that have already faulted (the reference's panics are sticky): it is the adversarial stream.  This
# generator asks the oracle what every group IS and sends it what such a node receives in a live
# cluster - leaders get AppendResponses / HeartbeatResponses / ClientRequests / Ticks, candidates get
# votes, followers get their leader's Heartbeats and AppendEntries (and now and then a Timeout, so
# that leaders keep appearing), faulted groups are restarted with p = 0.5 - with a few percent of
# everything else mixed in.  `stats` (dict, updated in place) counts what the tests assert liveness on.
def random_batch_aware(rng: np.random.Generator, ora, n: int, stats=None, p_wild=0.04):
    G, R = ora.G, ora.R
    ids = np.array(ora.node_ids, dtype=np.uint32)
    role = ora.read("role")
    fault = ora.read("fault")
    term_now = ora.read("term").astype(np.int64)
    head_now = ora.read("head").astype(np.int64)
    commit_now = ora.read("commit").astype(np.int64)
    id_gen = ora.read("id_gen").astype(np.int64)
    voted = ora.read("has_voted").astype(bool)
    voted_for = ora.read("voted_for")
    slot = ora.read("self_slot").astype(np.int64)
    group = rng.integers(0, G, n).astype(np.uint32)
    kind = np.zeros(n, np.uint8)
    from_ = np.zeros(n, np.uint32)
    term = np.zeros(n, np.uint64)
    idc = np.zeros(n, np.uint64)
    aux = np.zeros(n, np.uint64)
    flag = np.zeros(n, np.uint8)
    blk_id, blk_next = [], []
    heads = head_now.copy()  # keeps later AppendEntries rows of this batch plausible
    for i in range(n):
        g = int(group[i])
        t, h, c, s = int(term_now[g]), int(heads[g]), int(commit_now[g]), int(slot[g])
        other = int(ids[(s + 1 + int(rng.integers(0, max(R - 1, 1)))) % R])
        u = rng.random()
        if fault[g]:
            k = capi.CMD_RESTART if rng.random() < 0.5 else int(rng.choice(KINDS))
            kind[i], from_[i], term[i], idc[i] = k, other, t, h
            if k == capi.CMD_APPEND_ENTRIES:
                idc[i] = len(blk_id)  # (an empty block list)
            continue
        if u < p_wild:  # anything at all (the blind generator's distribution)
            kind[i], from_[i], term[i], idc[i], aux[i], flag[i] = int(rng.choice(KINDS, p=WEIGHTS)), other, max(t + int(rng.integers(-1, 3)), 0), \
                max(h + int(rng.integers(-2, 3)), 0), 0 if True else 0, int(rng.random() < 0.7)
            if kind[i] == capi.CMD_APPEND_ENTRIES:
                idc[i], aux[i] = len(blk_id), 0
            continue
        v = rng.random()
        if role[g] == capi.ROLE_LEADER:
            if v < 0.45:    # an acknowledgement at or below the head (progress.rs:42-46)
                kind[i], from_[i], term[i], idc[i], flag[i] = capi.CMD_APPEND_RESPONSE, other, t, max(h - int(rng.integers(0, 3)), 0), 1
            elif v < 0.57:
                kind[i], from_[i], idc[i], flag[i] = capi.CMD_HEARTBEAT_RESPONSE, other, max(c - int(rng.integers(0, 2)), 0), int(rng.random() < 0.8)
            elif v < 0.85 and id_gen[g] > h:  # (an append with id_gen <= head is the reference's Q8 panic)
                kind[i], idc[i] = capi.CMD_CLIENT_REQUEST, int(rng.integers(1, 1 << 40))
                heads[g] = max(int(id_gen[g]), h + 1)
                id_gen[g] = heads[g] + 1
            elif v < 0.95:
                kind[i] = capi.CMD_TICK
            else:           # a stale peer's traffic
                kind[i], from_[i], term[i], idc[i] = int(rng.choice([capi.CMD_APPEND_ENTRIES, capi.CMD_HEARTBEAT, capi.CMD_VOTE_REQUEST])), other, \
                    max(t - int(rng.integers(0, 2)), 0), h
                if kind[i] == capi.CMD_APPEND_ENTRIES:
                    idc[i], aux[i] = len(blk_id), 0
                else:
                    aux[i] = t
        elif role[g] == capi.ROLE_CANDIDATE:
            if v < 0.65:
                kind[i], from_[i], term[i], flag[i] = capi.CMD_VOTE_RESPONSE, other, t, int(rng.random() < 0.75)
            elif v < 0.8:
                kind[i] = capi.CMD_TICK
            elif v < 0.9:
                kind[i], from_[i], term[i], idc[i] = capi.CMD_HEARTBEAT, other, t + int(rng.integers(0, 2)), int(rng.integers(0, h + 1))
            else:
                kind[i], from_[i], term[i], idc[i], aux[i] = capi.CMD_VOTE_REQUEST, other, t + int(rng.integers(0, 2)), h, t
        else:
            lead = int(voted_for[g]) if voted[g] and int(voted_for[g]) in ids and int(voted_for[g]) != int(ids[s]) else other
            # a follower that has voted never campaigns again (follower.rs:249, Q4): leaders only come out of
            # followers that have not - their timer fires here before a Heartbeat can reach them - and out of
            # nodes that restart (voted_for is not persisted)
            if not voted[g] and v < 0.5:
                kind[i] = capi.CMD_TIMEOUT
                continue
            if voted[g] and v < 0.06:
                kind[i] = capi.CMD_RESTART
                continue
            if v < 0.25:
                kind[i], from_[i], term[i], idc[i] = capi.CMD_HEARTBEAT, lead, t + int(rng.random() < 0.1), int(rng.integers(0, h + 1))
            elif v < 0.55:  # the leader's window: a run that extends the chain (rarely a fork or a gap)
                nb = int(rng.integers(0, 4))
                kind[i], from_[i], term[i], idc[i], aux[i] = capi.CMD_APPEND_ENTRIES, lead, t + int(rng.random() < 0.1), len(blk_id), nb
                for _ in range(nb):
                    blk_id.append(h + 1), blk_next.append(h)
                    h += 1
                heads[g] = h
            elif v < 0.67:
                kind[i], from_[i], term[i], idc[i], aux[i] = capi.CMD_VOTE_REQUEST, other, t + int(rng.integers(0, 2)), max(h + int(rng.integers(-1, 2)), 0), t
            elif v < 0.80:  # the election timer (only a follower that has not voted campaigns: Q4)
                kind[i] = capi.CMD_TIMEOUT
            elif v < 0.92:
                kind[i] = capi.CMD_TICK
            else:
                kind[i], idc[i] = capi.CMD_CLIENT_REQUEST, int(rng.integers(1, 1 << 40))
    if stats is not None:
        live = fault[group] == 0
        stats["commands"] = stats.get("commands", 0) + n
        stats["to_live_groups"] = stats.get("to_live_groups", 0) + int(live.sum())
        stats.setdefault("led", []).append(float(((role == capi.ROLE_LEADER) & (fault == 0)).mean()))
    return dict(kind=kind, group=group, from_=from_, term=term, id=idc, aux=aux, flag=flag,
                blk_id=np.array(blk_id, dtype=np.uint64), blk_next=np.array(blk_next, dtype=np.uint64))


def assert_live(stats, decisions, R):
    """What makes a fuzz run worth its name: most commands reach groups that are still alive, a fair share
    of the groups is LED while it runs, and the commands turn into quorum decisions."""
    frac_live = stats["to_live_groups"] / stats["commands"]
    led = float(np.mean(stats["led"]))
    per_cmd = decisions / stats["commands"]
    print(f"fuzz liveness R={R}: {stats['commands']} commands, {100 * frac_live:.0f} % to un-faulted groups, "
          f"{100 * led:.0f} % of the groups led on average, {decisions} decisions = {per_cmd:.2f} per command")
    assert frac_live >= 0.30, frac_live
    if R >= 3:
        assert led >= 0.10, led
    assert per_cmd >= 0.10, per_cmd
