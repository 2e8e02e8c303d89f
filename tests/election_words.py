"""The ELECTION vocabulary as mailbox words (DESIGN.md "What comes next"): what a routed round's vote traffic looks
like as one word per (sender slot, partition) instead of rows - stated in numpy, as the mailbox columns of the
steady-state traffic were before their kernels existed (tests/dense_node.py).  Test infrastructure: nothing in the
product reads this yet; tests/test_election_words.py holds it to the rows the Python statement of the transport
(RoutedCluster) really delivers on the configs[4] traces, and counts what does not fit.

One round's inbound rows of one node, per partition in the transport's order (phase, emission index, sender slot) - every
sender's stream in its own order, the senders interleaved (tests/dense_node.py::RoutedCluster); every row carries its
ord = phase << 8 | emission index within its sender's step:

  request word   sender s campaigns (candidate.rs:24-44): `copies` identical VoteRequest{term, candidate_id, last_term = term,
                 head} at consecutive ords - the reference sends one broadcast per configured peer (Q5: `for _node in
                 &self.config.nodes`), so copies == R - 1
                   -> (ord of the first, term, head, copies)
  answer word    sender s answers this node's campaign (follower.rs:219-246, candidate.rs:66-84): its VoteResponses
                 {from = s's id, term, granted} to the copies, at consecutive ords: the first through can_vote, every further
                 one `false` once the first was granted (voted_for is set by then) or the same refusal again
                   -> (ord of the first, term, first, rest, copies)   with rest == False whenever copies > 1 and first == True

What is neither - the Heartbeat of a fresh leader, its answer - stays a row and keeps its ord; a stretch that is not uniform
(responses that change term halfway) stays rows as a whole; so does a voter that answers two candidates of one partition in
one round (two addressees: the answer word has one; their copies arrive interleaved, so neither's answers are consecutive)."""
import numpy as np

from josefine_amd import capi

REQ_DTYPE = np.dtype([("group", "<u4"), ("src", "<i8"), ("at", "<u4"), ("term", "<u8"), ("head", "<u8"), ("copies", "<u4")])
ANS_DTYPE = np.dtype([("group", "<u4"), ("src", "<i8"), ("at", "<u4"), ("term", "<u8"), ("first", "u1"), ("rest", "u1"), ("copies", "<u4")])


def encode(cols, src, ord_, member_ids):
    """cols: a node's inbound command columns of one round (any order); src, ord_: the sender slot and the ord of every row
    (parallel arrays).  Returns (request words, answer words, mask of the rows that stay rows).  A word stands for a maximal
    stretch of one sender's VoteRequests / VoteResponses for one partition at CONSECUTIVE ords; `at` is the ord of its
    first copy."""
    n = len(cols["kind"])
    stay = np.ones(n, bool)
    reqs, anss = [], []
    if not n:
        return np.zeros(0, REQ_DTYPE), np.zeros(0, ANS_DTYPE), stay
    by = np.lexsort((ord_, src, cols["group"]))  # every (partition, sender) stream, in its own order
    c = {k: np.asarray(v)[by] for k, v in cols.items()}
    s_src, s_ord = np.asarray(src)[by], np.asarray(ord_)[by]
    key = c["group"].astype(np.int64) * 16 + s_src
    starts = np.r_[0, np.nonzero(key[1:] != key[:-1])[0] + 1]
    ends = np.r_[starts[1:], n]
    for a, b in zip(starts, ends):
        s = int(s_src[a])
        if s < 0 or s >= len(member_ids):
            continue  # (not a member's mail)
        sid = member_ids[s]
        k = c["kind"][a:b]
        i = 0
        while i < b - a:  # maximal stretches of one vote kind at consecutive ords
            j = i + 1
            while j < b - a and k[j] == k[i] and s_ord[a + j] == s_ord[a + j - 1] + 1:
                j += 1
            lo, hi = a + i, a + j
            if k[i] == capi.CMD_VOTE_REQUEST:
                same = (c["term"][lo:hi] == c["term"][lo]).all() and (c["id"][lo:hi] == c["id"][lo]).all() and \
                    (c["aux"][lo:hi] == c["term"][lo]).all() and (c["from_"][lo:hi] == sid).all() and (c["flag"][lo:hi] == 0).all()
                if same:
                    reqs.append((c["group"][lo], s, s_ord[lo], c["term"][lo], c["id"][lo], hi - lo))
                    stay[by[lo:hi]] = False
            elif k[i] == capi.CMD_VOTE_RESPONSE:
                f = c["flag"][lo:hi]
                rest_ok = hi - lo == 1 or (f[1:] == f[1]).all()
                same = (c["term"][lo:hi] == c["term"][lo]).all() and (c["from_"][lo:hi] == sid).all() and \
                    (c["id"][lo:hi] == 0).all() and (c["aux"][lo:hi] == 0).all()
                if same and rest_ok:
                    anss.append((c["group"][lo], s, s_ord[lo], c["term"][lo], f[0], f[1] if hi - lo > 1 else 0, hi - lo))
                    stay[by[lo:hi]] = False
            i = j
    return np.array(reqs, REQ_DTYPE), np.array(anss, ANS_DTYPE), stay


def decode(reqs, anss, rest, rest_src, rest_ord, member_ids):
    """the words back into rows, merged with the rows that stayed rows (`rest` columns, their sender slots and ords), in
    the transport's order - what a dense voter / candidate half has to apply.  Returns (columns, sender slots, ords)."""
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
        srcs.append(np.asarray(rest_src, np.int64)), ords.append(np.asarray(rest_ord, np.int64))
    if not parts:
        return None, None, None
    cols = {k: np.concatenate([np.asarray(p[k]) for p in parts]) for k in parts[0]}
    src, ord_ = np.concatenate(srcs), np.concatenate(ords)
    order = np.lexsort((src, ord_, cols["group"]))  # (group, ord, sender): a stable sort - a sender's request stretch before its answer stretch at equal ords
    return {k: v[order] for k, v in cols.items()}, src[order], ord_[order]
