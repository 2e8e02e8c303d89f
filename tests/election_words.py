"""The ELECTION vocabulary as mailbox words (DESIGN.md "What comes next"): what a routed round's vote traffic looks
like as one word per (sender slot, partition) instead of rows - stated in numpy, as the mailbox columns of the
steady-state traffic were before their kernels existed (tests/dense_node.py).  Test infrastructure: nothing in the
product reads this yet; tests/test_election_words.py holds it to the rows the Python statement of the transport
(RoutedCluster) really delivers on the configs[4] traces, and counts what does not fit.

One round's inbound rows of one node, per partition in the transport's order (sender slot, emission order):

  request word   sender s campaigns (candidate.rs:24-44): `copies` identical VoteRequest{term, candidate_id, last_term = term,
                 head} - the reference sends one broadcast per configured peer (Q5: `for _node in &self.config.nodes`), so copies == R - 1
                   -> (term, head, copies)
  answer word    sender s answers this node's campaign (follower.rs:219-246, candidate.rs:66-84): its VoteResponses
                 {from = s's id, term, granted} to the copies, in order: the first through can_vote, every further one
                 `false` once the first was granted (voted_for is set by then) or the same refusal again
                   -> (term, first, rest, copies)   with rest == False whenever copies > 1 and first == True

What is neither inside a sender's run - the Heartbeat of a fresh leader, its answer - stays a row and keeps its place in
the run (the words carry where their stretch begins); a stretch that is not uniform (responses that change term halfway)
stays rows as a whole; so does a voter that answers two candidates of one partition in one round (two
addressees: the answer word has one)."""
import numpy as np

from josefine_amd import capi

REQ_DTYPE = np.dtype([("group", "<u4"), ("src", "<i8"), ("at", "<u4"), ("term", "<u8"), ("head", "<u8"), ("copies", "<u4")])
ANS_DTYPE = np.dtype([("group", "<u4"), ("src", "<i8"), ("at", "<u4"), ("term", "<u8"), ("first", "u1"), ("rest", "u1"), ("copies", "<u4")])


def encode(cols, src, member_ids):
    """cols: a node's inbound command columns of one round, sorted (group, sender, emission); src: the sender slot of
    every row (parallel array).  Returns (request words, answer words, mask of the rows that stay rows, every row's
    ordinal within its (sender, partition) run).  A word stands for a maximal stretch of one sender's VoteRequests /
    VoteResponses for one partition; `at` is where the stretch begins in the sender's run (a node that is elected and
    campaigns again within one round sends Heartbeat, then VoteRequests: the stretch begins at 1)."""
    n = len(cols["kind"])
    stay = np.ones(n, bool)
    ordinal = np.zeros(n, np.int64)
    reqs, anss = [], []
    if not n:
        return np.zeros(0, REQ_DTYPE), np.zeros(0, ANS_DTYPE), stay, ordinal
    key = cols["group"].astype(np.int64) * 16 + src
    starts = np.r_[0, np.nonzero(key[1:] != key[:-1])[0] + 1]
    ends = np.r_[starts[1:], n]
    for a, b in zip(starts, ends):
        ordinal[a:b] = np.arange(b - a)
        s = int(src[a])
        if s < 0 or s >= len(member_ids):
            continue  # injected rows: not mail
        sid = member_ids[s]
        k = cols["kind"][a:b]
        i = 0
        while i < b - a:  # maximal stretches of one vote kind
            j = i + 1
            while j < b - a and k[j] == k[i]:
                j += 1
            lo, hi = a + i, a + j
            if k[i] == capi.CMD_VOTE_REQUEST:
                same = (cols["term"][lo:hi] == cols["term"][lo]).all() and (cols["id"][lo:hi] == cols["id"][lo]).all() and \
                    (cols["aux"][lo:hi] == cols["term"][lo]).all() and (cols["from_"][lo:hi] == sid).all() and (cols["flag"][lo:hi] == 0).all()
                if same:
                    reqs.append((cols["group"][lo], s, i, cols["term"][lo], cols["id"][lo], hi - lo))
                    stay[lo:hi] = False
            elif k[i] == capi.CMD_VOTE_RESPONSE:
                f = cols["flag"][lo:hi]
                rest_ok = hi - lo == 1 or (f[1:] == f[1]).all()
                same = (cols["term"][lo:hi] == cols["term"][lo]).all() and (cols["from_"][lo:hi] == sid).all() and \
                    (cols["id"][lo:hi] == 0).all() and (cols["aux"][lo:hi] == 0).all()
                if same and rest_ok:
                    anss.append((cols["group"][lo], s, i, cols["term"][lo], f[0], f[1] if hi - lo > 1 else 0, hi - lo))
                    stay[lo:hi] = False
            i = j
    return np.array(reqs, REQ_DTYPE), np.array(anss, ANS_DTYPE), stay, ordinal


def decode(reqs, anss, rest, rest_src, rest_ord, member_ids):
    """the words back into rows, merged with the rows that stayed rows (`rest` columns, their sender slots and ordinals),
    in the transport's order - what a dense voter / candidate half has to apply"""
    parts, srcs, ords = [], [], []

    def rows(m, kind, group, frm, term, id_, aux, flag):
        return dict(kind=np.full(m, kind, np.uint8), group=np.full(m, group, np.uint32), from_=np.full(m, frm, np.uint32),
                    term=np.full(m, term, np.uint64), id=np.full(m, id_, np.uint64), aux=np.full(m, aux, np.uint64), flag=flag)
    for w in reqs:
        m = int(w["copies"])
        parts.append(rows(m, capi.CMD_VOTE_REQUEST, w["group"], member_ids[w["src"]], w["term"], w["head"], w["term"], np.zeros(m, np.uint8)))
        srcs.append(np.full(m, w["src"], np.int64)), ords.append(int(w["at"]) + np.arange(m))
    for w in anss:
        m = int(w["copies"])
        flag = np.full(m, w["rest"], np.uint8)
        flag[0] = w["first"]
        parts.append(rows(m, capi.CMD_VOTE_RESPONSE, w["group"], member_ids[w["src"]], w["term"], 0, 0, flag))
        srcs.append(np.full(m, w["src"], np.int64)), ords.append(int(w["at"]) + np.arange(m))
    if len(rest["kind"]):
        parts.append(rest)
        srcs.append(rest_src), ords.append(rest_ord)
    if not parts:
        return None, None
    cols = {k: np.concatenate([np.asarray(p[k]) for p in parts]) for k in parts[0]}
    src = np.concatenate(srcs)
    order = np.lexsort((np.concatenate(ords), src, cols["group"]))  # (group, sender, emission)
    return {k: v[order] for k, v in cols.items()}, src[order]
