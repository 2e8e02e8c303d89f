"""Helpers for the jg_step_node tests (a node's whole tick from host rows, include/josefine_gpu.h):

* `node_traffic`: one tick's worth of inbound rows for a node that leads some partitions and follows
  others - the steady-state vocabulary (AppendResponse / HeartbeatResponse / ClientRequest for leaders,
  Heartbeat / AppendEntries for followers) plus everything that must push a partition onto the
  general path (duplicates, votes, Timeout, explicit Tick rows, Restart, forged senders, blocks that
  are not a run, ClientRequests at followers, heads a mailbox word cannot hold), shuffled;
* `plain_apply_equivalent`: the SPECIFICATION of the node step: every row through plain jg_submit +
  jg_step (Apply::apply, one command at a time) IN THE ORDER GIVEN (server.rs:120-161), then the
  Ticks.  What jo_step_node / jg_step_node must be indistinguishable from - state, faults, and per
  partition the order of every fsm_tx and rpc_tx row; `classify` restates which partitions the
  engine reports as rows_general (a matter of representation, not of results);
* `columns_as_rows`: the mailbox columns of a node outbox as the message rows they stand for.
"""
import numpy as np

from josefine_amd import capi

NO = capi.NO_ACK


def elect_some(e, mask, now_ms=0):
    """Timeout + granted votes for the partitions in `mask` only (traces.elect_all for a subset)."""
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


def node_traffic(rng, ora, token0=0, p_noise=0.03, quiet=0.15, p_reorder=0.08):
    """One tick's inbound rows for the node `ora` models, as kwargs for submit_columns."""
    G, R = ora.G, ora.R
    ids = np.array(ora.node_ids, dtype=np.uint32)
    role, head, commit, term = ora.read("role"), ora.read("head").astype(np.int64), ora.read("commit").astype(np.int64), \
        ora.read("term").astype(np.int64)
    slot = ora.read("self_slot").astype(np.int64)
    rows = []  # (kind, group, from, term, id, aux, flag, blocks)

    def add(kind, g, frm=0, t=0, i=0, aux=0, flag=0, blocks=None):
        rows.append((kind, g, frm, t, i, aux, flag, blocks))

    for g in range(G):
        if rng.random() < quiet:
            continue
        h, c, t, s = int(head[g]), int(commit[g]), int(term[g]), int(slot[g])
        if role[g] == capi.ROLE_LEADER:
            for r in range(R):
                if r == s:
                    continue
                if rng.random() < 0.7:  # AppendResponse: mostly at or just below the head, rarely forged above it
                    hh = max(h + int(rng.choice([-3, -2, -1, 0, 0, 0, 0, 1], p=[.05, .1, .2, .3, .15, .1, .07, .03])), 0)
                    add(capi.CMD_APPEND_RESPONSE, g, ids[r], t, hh, 0, int(rng.random() < 0.9))
                if rng.random() < 0.5:
                    has = rng.random() < 0.8
                    add(capi.CMD_HEARTBEAT_RESPONSE, g, ids[r], 0, max(c + int(rng.integers(-2, 2)), 0), 0, int(has))
            if rng.random() < 0.6:
                add(capi.CMD_CLIENT_REQUEST, g, 0, 0, token0 + g)
            if rng.random() < p_noise / 2:  # somebody else leads at a higher term (leader.rs:200-208: step down; :263: ignored)
                other = ids[(s + 1) % R]
                if rng.random() < 0.5:
                    add(capi.CMD_HEARTBEAT, g, other, t + 1, int(rng.integers(0, h + 2)))
                if rng.random() < 0.7:
                    add(capi.CMD_APPEND_ENTRIES, g, other, t + int(rng.integers(0, 3)), 0, 0, 0, [(h + 1, h)] if rng.random() < 0.5 else [])
        else:
            lead = ids[(s + 1) % R]
            lt = t + int(rng.choice([0, 0, 0, 1, 2]))
            both_same = rng.random() < 0.93
            if rng.random() < 0.7:
                add(capi.CMD_HEARTBEAT, g, lead, lt, int(rng.integers(0, h + 2)))
            if rng.random() < 0.7:
                n = int(rng.integers(0, 5))
                frm = h if rng.random() < 0.85 else max(h + int(rng.integers(-2, 3)), 0)
                blocks = [(frm + 1 + k, frm + k) for k in range(n)]
                if n and rng.random() < 0.04:  # not a run: a fork inside the window
                    blocks[-1] = (blocks[-1][0] + 1, max(blocks[-1][1] - 1, 0))
                add(capi.CMD_APPEND_ENTRIES, g, lead if both_same else ids[(s + 2) % R], lt if both_same or rng.random() < 0.5 else lt + 1,
                    0, 0, 0, blocks)
        if rng.random() < p_noise:  # what must take the partition off the column path
            k = int(rng.integers(0, 10))
            if k == 0:
                add(capi.CMD_TIMEOUT, g)
            elif k == 1:
                add(capi.CMD_TICK, g)
            elif k == 2:
                add(capi.CMD_VOTE_REQUEST, g, ids[(s + 1) % R], t + 1, h, t)
            elif k == 3:
                add(capi.CMD_VOTE_RESPONSE, g, ids[(s + 1) % R], t, 0, 0, int(rng.random() < 0.5))
            elif k == 4:
                add(capi.CMD_APPEND_RESPONSE, g, 4242, t, h)           # a stranger (progress.rs:43 panics)
            elif k == 5:
                add(capi.CMD_APPEND_RESPONSE, g, ids[(s + 1) % R], t, h)  # quite possibly a duplicate for that slot
            elif k == 6:
                add(capi.CMD_CLIENT_REQUEST, g, 0, 0, token0 + G + g)  # a second request / a request at a follower
            elif k == 7:
                add(capi.CMD_HEARTBEAT_RESPONSE, g, ids[s], 0, c, 0, 0)  # "from" the own id
            elif k == 8:
                add(capi.CMD_APPEND_RESPONSE, g, ids[(s + 1) % R], t, capi.MAILBOX_NONE + int(rng.integers(0, 2)))
            elif rng.random() < 0.3:
                add(capi.CMD_RESTART, g)
    order = rng.permutation(len(rows))
    rows = [rows[i] for i in order]
    # the interleaving of different peers' messages and of client requests is the network's; ONE peer's
    # messages mostly keep the order it sent them in (a follower answers the Heartbeat before the
    # AppendEntries of a Tick, leader.rs:234-245) - mostly: the other order happens too (a replicate()
    # triggered by a HeartbeatResponse, then the next Tick's Heartbeat) and must be served exactly as well
    first = {}
    for i, r in enumerate(rows):
        k, g, f = r[0], r[1], r[2]
        if k in (capi.CMD_HEARTBEAT, capi.CMD_HEARTBEAT_RESPONSE):
            first.setdefault((g, f), []).append(i)
    for i, r in enumerate(rows):
        k, g, f = r[0], r[1], r[2]
        if k in (capi.CMD_APPEND_ENTRIES, capi.CMD_APPEND_RESPONSE) and (g, f) in first:
            j = first[(g, f)][0]
            if (i < j) != (rng.random() < p_reorder):
                rows[i], rows[j] = rows[j], rows[i]
                first[(g, f)][0] = i
    return rows_to_columns(rows)


def rows_to_columns(rows):
    n = len(rows)
    kind = np.zeros(n, np.uint8)
    group = np.zeros(n, np.uint32)
    frm = np.zeros(n, np.uint32)
    term = np.zeros(n, np.uint64)
    idc = np.zeros(n, np.uint64)
    aux = np.zeros(n, np.uint64)
    flag = np.zeros(n, np.uint8)
    bi, bn = [], []
    for i, (k, g, f, t, d, a, fl, blocks) in enumerate(rows):
        kind[i], group[i], frm[i], term[i], flag[i] = k, g, f, t, fl
        if k == capi.CMD_APPEND_ENTRIES:
            idc[i], aux[i] = len(bi), len(blocks or [])
            for (b, nx) in blocks or []:
                bi.append(b), bn.append(nx)
        else:
            idc[i], aux[i] = d, a
    return dict(kind=kind, group=group, from_=frm, term=term, id=idc, aux=aux, flag=flag,
                blk_id=np.array(bi, np.uint64), blk_next=np.array(bn, np.uint64))


def classify(cols, role, self_slot, node_ids, leader=True, follower=True):
    """Per partition: do its rows take the general path?  (include/josefine_gpu.h, jg_step_node.)"""
    G = len(role)
    ids = list(node_ids)
    general = np.zeros(G, bool)
    seen = [set() for _ in range(G)]
    beat = [{} for _ in range(G)]
    at = [{} for _ in range(G)]
    for i in range(len(cols["kind"])):
        k, g, f = int(cols["kind"][i]), int(cols["group"][i]), int(cols["from_"][i])
        key = None
        if k in (capi.CMD_APPEND_RESPONSE, capi.CMD_HEARTBEAT_RESPONSE):
            s = ids.index(f) if f in ids else -1
            bad = not leader or s < 0 or s == self_slot[g] or (k == capi.CMD_APPEND_RESPONSE and int(cols["id"][i]) >= capi.MAILBOX_NONE)
            key = (k, s) if s >= 0 else None
        elif k == capi.CMD_CLIENT_REQUEST:
            bad = not leader or role[g] != capi.ROLE_LEADER
            key = (k,)
        elif k == capi.CMD_HEARTBEAT:
            # (a leader's answer to these is a role change or nothing, never an answer word: rows)
            bad = not follower or int(cols["id"][i]) == NO or f == 0 or role[g] == capi.ROLE_LEADER
            key = (k,)
            beat[g]["hb"] = (int(cols["term"][i]), f)
            at[g]["hb"] = i
        elif k == capi.CMD_APPEND_ENTRIES:
            first, n = int(cols["id"][i]), int(cols["aux"][i])
            b = cols["blk_id"][first:first + n].astype(object)
            nx = cols["blk_next"][first:first + n].astype(object)
            run = n <= 0xfe and (n == 0 or (int(b[0]) >= 1 and int(b[0]) - 1 + n < capi.MAILBOX_NONE and
                                          all(int(b[j]) == int(b[0]) + j and int(nx[j]) == int(b[0]) + j - 1 for j in range(n))))
            bad = not follower or f == 0 or not run or role[g] == capi.ROLE_LEADER
            key = (k,)
            beat[g]["ae"] = (int(cols["term"][i]), f)
            at[g]["ae"] = i
        else:
            bad = True
        if bad or (key is not None and key in seen[g]):
            general[g] = True
        if key is not None:
            seen[g].add(key)
    for g in range(G):
        # one answer word holds HeartbeatResponse then AppendResponse: an AppendEntries BEFORE the Heartbeat is rows
        if len(beat[g]) == 2 and (beat[g]["hb"] != beat[g]["ae"] or at[g]["ae"] < at[g]["hb"]):
            general[g] = True
    return general


def plain_apply_equivalent(e, cols, now_ms, leader=True, follower=True, tick=True):
    """Apply one node step to engine `e` through plain submit + step only: EVERY row in the order given
    (per partition: the stream order, mod.rs:471-479), then Command::Tick for the partitions whose half
    runs (the role after the rows decides which).  Returns which partitions the engine reports as general
    (`classify`); leaves the fsm rows of the step in `e.plain_fsm` (drained)."""
    general = classify(cols, e.read("role"), e.read("self_slot"), list(e.node_ids), leader, follower)
    if len(cols["kind"]):
        e.submit_columns(cols["kind"], cols["group"], cols["from_"], cols["term"], cols["id"], cols["aux"], cols["flag"],
                         cols["blk_id"], cols["blk_next"])
    e.step(now_ms)
    fsm = [e.drain_applies()]
    if tick:
        role = e.read("role")
        g = np.nonzero(((role == capi.ROLE_LEADER) & leader) | ((role != capi.ROLE_LEADER) & follower))[0].astype(np.uint32)
        if len(g):
            e.submit_columns(np.full(len(g), capi.CMD_TICK, np.uint8), g)
    e.step(now_ms)
    fsm.append(e.drain_applies())
    e.plain_fsm = np.concatenate(fsm)
    return general


def msgs_per_partition(rows):
    """message tuples -> {partition: [tuples in order]}"""
    d = {}
    for t in rows:
        d.setdefault(t[0], []).append(t)
    return d


def columns_as_rows(out, node_ids, self_slot, leader_of=None):
    """The mailbox columns of a node outbox as jg_msg_row-shaped tuples
    (group, kind, to_kind, to_id, flag, term, id, aux), without `from` (the caller's own id)."""
    rows = []
    ids = [int(x) for x in node_ids]
    if out.get("beat_term") is not None:
        bt, bc, ae = out["beat_term"], out["beat_commit"], out["ae"]
        R, G = ae.shape
        for g in range(G):
            if int(bc[g]) != NO:
                rows.append((g, capi.CMD_HEARTBEAT, capi.TO_PEERS, 0, 0, int(bt[g]), int(bc[g]), 0))
            for r in range(R):
                w = int(ae[r, g])
                if w != NO:
                    rows.append((g, capi.CMD_APPEND_ENTRIES, capi.TO_PEER, ids[r], 0, int(bt[g]), w >> 8, w & 0xff))
    if out.get("answer") is not None:
        an, hbc = out["answer"], out["hb_commit"]
        for g in range(len(an)):
            w = int(an[g])
            if w == NO:
                continue
            to = int(leader_of[g]) if leader_of is not None else 0
            if (w & 0xff) != capi.HB_NONE:
                rows.append((g, capi.CMD_HEARTBEAT_RESPONSE, capi.TO_PEER, to, w & 0xff, 0, int(hbc[g]), 0))
            if (w >> 8) != capi.MAILBOX_NONE:
                rows.append((g, capi.CMD_APPEND_RESPONSE, capi.TO_PEER, to, 1, None, w >> 8, 0))
    return rows


def compare_outboxes(a, b, what=""):
    """Two node outboxes (engine.step_node dicts): every column; hb_commit only where the answer
    carries a HeartbeatResponse."""
    for k in ("beat_term", "beat_commit", "ae", "answer"):
        x, y = a.get(k), b.get(k)
        assert (x is None) == (y is None), (what, k)
        if x is not None and not np.array_equal(x, y):
            bad = np.argwhere(x != y)[:6]
            raise AssertionError(f"{what}: outbox column {k} differs at {bad.tolist()}: {x[tuple(bad[0])]:#x} vs {y[tuple(bad[0])]:#x}")
    if a.get("answer") is not None:
        m = (a["answer"] != np.uint64(NO)) & ((a["answer"] & np.uint64(0xff)) != np.uint64(capi.HB_NONE))
        assert np.array_equal(a["hb_commit"][m], b["hb_commit"][m]), (what, "hb_commit")
    assert a["rows"] == b["rows"] and a["rows_general"] == b["rows_general"], (what, a["rows"], b["rows"], a["rows_general"], b["rows_general"])
