"""The election vocabulary as mailbox words, with the TRANSPORT's side in the device's own functions (CPU): a routed cluster
of host-compiled nodes (tests/host_compiled.py) in which every emitted row goes through jg_votes.h's census
(jg_votes_census_row), the delivering pass asks jg_votes_row_travels per row and addressee, the answer words a partition
cannot take are expanded back to rows (jg_votes_expand_group), and the receiving half (jg_vote_half_group) reads the packed
mail of the round before - i.e. the routed round under JG_ROUTE_VOTE_WORDS=1 with the staging and its ordering played in
numpy - against the same cluster over oracle engines in which every message is a row: every column of every node after
every round, the rows delivered and the rows kept."""
import ctypes as C

import numpy as np
import pytest

from josefine_amd import capi
from dense_node import RoutedCluster, cluster_failure_rows, routable
from host_compiled import HostCompiled, VoteMail
from oracle_lib import oracle_engine
from parity import compare_snapshots

ROUTE_DTYPE = np.dtype([("src", "<i8"), ("step", "<i8"), ("k", "<i8")])


def ranks_within_groups(groups):
    """emission index of every row within its group (rows in emission order per group)"""
    k = np.zeros(len(groups), np.int64)
    seen = {}
    for i, g in enumerate(groups.tolist()):
        k[i] = seen.get(g, 0)
        seen[g] = k[i] + 1
    return k


class MailCluster(RoutedCluster):
    def __init__(self, G, R, need=None, **kw):
        super().__init__(HostCompiled, G, R, **kw)
        self.mail = [VoteMail(R, G), VoteMail(R, G)]
        self.t = 0
        self.need = R - 1 if need is None else need
        self.inrows = [[] for _ in range(R)]  # per addressee: (rows, route keys)
        self.in_words = self.in_rows = self.vote_rows_as_rows = self.expanded = 0

    def dests(self, rows, s):
        ok = routable(rows, self.member_ids)
        m = np.zeros(len(rows), np.uint32)
        for n in range(self.R):
            if n == s:
                continue
            to_n = ok & ((rows["to_kind"] == capi.TO_PEERS) | (rows["to_id"] == self.member_ids[n]))
            m |= np.where(to_n, np.uint32(1 << n), np.uint32(0))
        return m

    def round(self, appends, inject=None, dt_ms=100):
        G, R = self.G, self.R
        now = self.now + dt_ms
        prev, cur = self.mail[(self.t + 1) & 1], self.mail[self.t & 1]
        cur.clear()
        lib = self.nodes[0].lib
        emitted = [[] for _ in range(R)]  # per sender: (rows, step, k)
        for n in range(R):
            # step 1: what the transport delivered - the words (the receiving half) and, for the other partitions, the rows
            xrows, xk = self.nodes[n].vote_half_mail(n, now, prev, cur, step=1, need=self.need)
            if len(xrows):
                emitted[n].append((xrows, np.full(len(xrows), 1, np.uint32), xk.astype(np.uint32)))
            parts = list(self.inrows[n])
            self.inrows[n] = []
            rows = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, capi.MSG_DTYPE)
            keys = np.concatenate([p[1] for p in parts]) if parts else np.zeros(0, ROUTE_DTYPE)
            order = np.lexsort((keys["k"], keys["step"], keys["src"], rows["group"]))  # (the staging's ordering key)
            rows = rows[order]
            cols = dict(kind=rows["kind"], group=rows["group"], from_=rows["from"], term=rows["term"], id=rows["id"], aux=rows["aux"], flag=rows["flag"])
            self.in_rows += len(rows)
            self.vote_rows_as_rows += int(np.isin(rows["kind"], (capi.CMD_VOTE_REQUEST, capi.CMD_VOTE_RESPONSE)).sum())
            inj = inject[n] if inject else None
            if inj is not None and len(inj["kind"]):  # step 2 on the device: after everything delivered, per partition
                m = len(inj["kind"])
                z8, z4 = np.zeros(m, np.uint64), np.zeros(m, np.uint32)
                ic = dict(kind=inj["kind"], group=inj["group"], from_=inj.get("from_", z4), term=inj.get("term", z8), id=inj.get("id", z8),
                          aux=inj.get("aux", z8), flag=inj.get("flag", np.zeros(m, np.uint8)))
                cols = {k: np.concatenate([cols[k], np.asarray(ic[k])]) for k in cols}
                order = np.argsort(cols["group"], kind="stable")
                cols = {k: v[order] for k, v in cols.items()}
                self.delivered[n] += m
            if len(cols["kind"]):
                self.nodes[n].submit_columns(**cols)
                self.nodes[n].step(now)
                out = self.nodes[n].drain_messages()
                if len(out):
                    emitted[n].append((out, np.full(len(out), 2, np.uint32), ranks_within_groups(out["group"]).astype(np.uint32)))
        outs = self.dense_round(appends, dt_ms)
        drained = self.rows.pop()
        for s in range(R):
            if len(drained[s]):
                emitted[s].append((drained[s], np.full(len(drained[s]), 3, np.uint32), ranks_within_groups(drained[s]["group"]).astype(np.uint32)))
        # -- the transport: census, then the delivering pass and the expansion (jg_votes.h)
        flat = []
        for s in range(R):
            if not emitted[s]:
                flat.append(None)
                continue
            rows = np.ascontiguousarray(np.concatenate([e[0] for e in emitted[s]]))
            step = np.ascontiguousarray(np.concatenate([e[1] for e in emitted[s]]))
            k = np.ascontiguousarray(np.concatenate([e[2] for e in emitted[s]]))
            d = np.ascontiguousarray(self.dests(rows, s))
            self.kept[s] = np.concatenate([self.kept[s], rows[d == 0]])
            lib.hc_votes_census(C.addressof(cur.c), s, int(self.member_ids[s]), rows.ctypes.data, step.ctypes.data, k.ctypes.data, d.ctypes.data, len(rows))
            flat.append((rows, step, k, d))
        for s in range(R):
            if flat[s] is not None:
                rows, step, k, d = flat[s]
                travels = np.zeros(len(rows), np.uint32)
                lib.hc_votes_travels(C.addressof(cur.c), int(self.member_ids[s]), rows.ctypes.data, k.ctypes.data, d.ctypes.data, len(rows), self.need,
                                     travels.ctypes.data)
                for n in range(R):
                    to_n = (travels >> n) & 1 == 1
                    if to_n.any():
                        key = np.zeros(int(to_n.sum()), ROUTE_DTYPE)
                        key["src"], key["step"], key["k"] = s, step[to_n], k[to_n]
                        self.inrows[n].append((rows[to_n], key))
                    self.in_words += int((((d >> n) & 1 == 1) & ~to_n).sum())
                    self.delivered[n] += int(((d >> n) & 1 == 1).sum())  # (counted when sent; a word's copies count as the rows they stand for)
            cap = 8 * G
            xr, to, st, kk = np.zeros(cap, capi.MSG_DTYPE), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
            m = lib.hc_votes_expand(self.nodes[s]._h, C.addressof(cur.c), s, self.need, xr.ctypes.data, to.ctypes.data, st.ctypes.data, kk.ctypes.data, cap)
            assert m < cap
            self.expanded += m
            for n in range(R):
                to_n = to[:m] == n
                if to_n.any():
                    key = np.zeros(int(to_n.sum()), ROUTE_DTYPE)
                    key["src"], key["step"], key["k"] = s, st[:m][to_n], kk[:m][to_n]
                    self.inrows[n].append((xr[:m][to_n], key))
            ac = cur.a_ctl[s]  # the answer words' copies: delivered to their addressee, as words or expanded
            n_ans, to_ans = ac & 0xff, (ac >> 21) & 7
            for n in range(R):
                self.delivered[n] += int(n_ans[to_ans == n].sum())
            self.in_words += int(n_ans.sum()) - m
        self.t += 1
        return outs


@pytest.mark.parametrize("R,percent,also,T", [(5, 3, (), 40), (3, 4, (2,), 40), (5, 3, (2,), 40), (3, 4, (), 40), (3, 25, (1, 2), 60), (5, 25, (1, 2, 3), 60),
                                              (4, 30, (1, 2), 60)])
def test_routed_round_with_the_vote_mail_equals_the_row_transport(R, percent, also, T):
    G = 150
    ora = RoutedCluster(oracle_engine, G, R, seed=5)
    dev = MailCluster(G, R, seed=5)
    for t in range(T):
        inj = cluster_failure_rows(99, t, G, R, percent, also=also) if t >= 3 else [None] * R
        ora.round(np.ones(G, np.uint64), inject=inj)
        dev.round(np.ones(G, np.uint64), inject=[None if c is None else dict(c) for c in inj])
        for n in range(R):
            compare_snapshots(dev.nodes[n], ora.nodes[n], f"round {t} node {n}")
        # (what is delivered at the start of round t + 1 is counted there by the row transport and at the end of round t here)
        assert [k.tobytes() for k in ora.kept] == [k.tobytes() for k in dev.kept], t
    for n in range(R):
        assert dev.nodes[n].counters()["decisions"] == ora.nodes[n].counters()["decisions"], n
    pending = np.array([sum(len(rows) for _, rows in ora.inbound[n]) for n in range(R)])
    assert (ora.delivered + pending).tolist() == dev.delivered.tolist()
    if percent < 10:  # the configs[4] rates: nearly all of the vote traffic is words
        assert dev.in_words > 10 * dev.vote_rows_as_rows and dev.in_words > G, (dev.in_words, dev.vote_rows_as_rows)
    else:  # (chaos; the ways back to rows are taken in test_the_transports_functions_on_random_mail)
        assert dev.in_words > G
    print(f"R={R} also={also}: {dev.in_words} rows' worth in words, {dev.in_rows} rows delivered ({dev.vote_rows_as_rows} of them votes, {dev.expanded} expanded)")


def _row(g, kind, to_kind, to_id, frm, term, id_=0, aux=0, flag=0):
    r = np.zeros(1, capi.MSG_DTYPE)
    r["group"], r["kind"], r["to_kind"], r["to_id"], r["from"], r["term"], r["id"], r["aux"], r["flag"] = g, kind, to_kind, to_id, frm, term, id_, aux, flag
    return r


@pytest.mark.parametrize("R,seed", [(3, 1), (4, 2), (5, 3), (5, 4)])
def test_the_transports_functions_on_random_mail(R, seed):
    """census / travels / expand on emissions no cluster would produce together (two campaigns of one sender in a round,
    malformed requests, answers next to rows, rows to one addressee only): for every addressee and partition EITHER
    everything arrives as rows - then exactly the rows the plain transport delivers, answer words written out, in its
    order - OR nothing does, and the words read back (as the receiving half reads them) say exactly those rows"""
    G = 96
    rng = np.random.default_rng(seed)
    node = HostCompiled(G, R, seed=1, self_slots=np.zeros(G, np.uint8))
    ids = np.array(node.node_ids[:R], np.uint32)
    lib = node.lib
    need = R - 1
    n_words = n_rows = n_expanded = n_double = 0
    for it in range(6):
        mail = VoteMail(R, G)
        mail.q_term[:], mail.q_head[:], mail.a_term[:] = rng.integers(0, 1 << 60, (3, R, G), dtype=np.uint64)  # (garbage where no control word says otherwise)
        plain = {}  # (d, g) -> list of (s, step, k, row)
        emitted = [[] for _ in range(R)]
        for s in range(R):
            for g in range(G):
                if rng.random() < 0.55:
                    continue
                k = {1: 0, 2: 0, 3: 0}
                events = []
                if rng.random() < 0.3:  # (something said before the answers)
                    events.append(("row", 1, k[1], _row(g, capi.CMD_HEARTBEAT_RESPONSE, capi.TO_PEER, ids[(s + 1) % R], ids[s], 1, 5)))
                    k[1] += 1
                if rng.random() < 0.5:  # an answer word (the vote half's: step 1)
                    n, to = int(rng.integers(1, R + 1)), int((s + 1 + rng.integers(0, R - 1)) % R)
                    first, rest, term = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(1, 9))
                    events.append(("word", 1, k[1], n, to, first, rest, term))
                    k[1] += n
                for _ in range(int(rng.choice([0, 1, 1, 1, 2, 3]))):
                    step = int(rng.integers(1, 4))
                    kind = rng.choice(["campaign", "campaign", "campaign", "heartbeat", "response", "ae", "odd", "oddcampaign"])
                    term, head = int(rng.integers(1, 9)), int(rng.integers(0, 50))
                    if kind == "campaign":
                        for _c in range(need):
                            events.append(("row", step, k[step], _row(g, capi.CMD_VOTE_REQUEST, capi.TO_PEERS, 0, ids[s], term, head, term)))
                            k[step] += 1
                    elif kind == "oddcampaign":  # R - 1 broadcasts that are not a campaign's: last_term != term
                        for _c in range(need):
                            events.append(("row", step, k[step], _row(g, capi.CMD_VOTE_REQUEST, capi.TO_PEERS, 0, ids[s], term, head, term + 1)))
                            k[step] += 1
                    elif kind == "odd":  # a request that is not a campaign's copy: to one peer, or last_term != term
                        to_one = rng.random() < 0.5
                        events.append(("row", step, k[step], _row(g, capi.CMD_VOTE_REQUEST, capi.TO_PEER if to_one else capi.TO_PEERS,
                                                                   ids[(s + 1) % R] if to_one else 0, ids[s], term, head, term if to_one else term + 1)))
                        k[step] += 1
                    elif kind == "heartbeat":
                        events.append(("row", step, k[step], _row(g, capi.CMD_HEARTBEAT, capi.TO_PEERS, 0, ids[s], term, head)))
                        k[step] += 1
                    elif kind == "response":
                        events.append(("row", step, k[step], _row(g, capi.CMD_VOTE_RESPONSE, capi.TO_PEER, ids[(s + 1 + rng.integers(0, R - 1)) % R], ids[s], term, 0, 0,
                                                                   int(rng.integers(0, 2)))))
                        k[step] += 1
                    else:
                        events.append(("row", step, k[step], _row(g, capi.CMD_APPEND_ENTRIES, capi.TO_PEER, ids[(s + 1) % R], ids[s], term, head, 1)))
                        k[step] += 1
                campaigns = sum(1 for e in events if e[0] == "row" and e[3]["kind"][0] == capi.CMD_VOTE_REQUEST and e[3]["to_kind"][0] == capi.TO_PEERS
                                and e[3]["aux"][0] == e[3]["term"][0]) // need
                n_double += campaigns > 1
                for e in events:
                    if e[0] == "word":
                        _, step, k0, n, to, first, rest, term = e
                        mail.a_term[s, g] = term
                        mail.a_ctl[s, g] = n | (step << 8 | k0) << 8 | first << 19 | rest << 20 | to << 21
                        VoteMail.set_bits(mail.wordmail, to, [g])
                        for j in range(n):
                            plain.setdefault((to, g), []).append((s, step, k0 + j, _row(g, capi.CMD_VOTE_RESPONSE, capi.TO_PEER, ids[to], ids[s], term, 0, 0, rest if j else first)))
                    else:
                        _, step, kk, row = e
                        emitted[s].append((row, step, kk))
                        if row["kind"][0] in (capi.CMD_APPEND_ENTRIES, capi.CMD_CLIENT_REQUEST):
                            continue
                        for d in range(R):
                            if d != s and (row["to_kind"][0] == capi.TO_PEERS or row["to_id"][0] == ids[d]):
                                plain.setdefault((d, g), []).append((s, step, kk, row))
        got = {}
        flat = []
        for s in range(R):
            rows = np.ascontiguousarray(np.concatenate([e[0] for e in emitted[s]]))
            step = np.array([e[1] for e in emitted[s]], np.uint32)
            k = np.array([e[2] for e in emitted[s]], np.uint32)
            perm = rng.permutation(len(rows))  # (the census is a parallel pass: any order)
            rows, step, k = np.ascontiguousarray(rows[perm]), np.ascontiguousarray(step[perm]), np.ascontiguousarray(k[perm])
            ok = routable(rows, ids)
            d = np.zeros(len(rows), np.uint32)
            for n in range(R):
                if n != s:
                    d |= np.where(ok & ((rows["to_kind"] == capi.TO_PEERS) | (rows["to_id"] == ids[n])), np.uint32(1 << n), np.uint32(0))
            d = np.ascontiguousarray(d)
            lib.hc_votes_census(C.addressof(mail.c), s, int(ids[s]), rows.ctypes.data, step.ctypes.data, k.ctypes.data, d.ctypes.data, len(rows))
            flat.append((rows, step, k, d))
        for s in range(R):
            rows, step, k, d = flat[s]
            travels = np.zeros(len(rows), np.uint32)
            lib.hc_votes_travels(C.addressof(mail.c), int(ids[s]), rows.ctypes.data, k.ctypes.data, d.ctypes.data, len(rows), need, travels.ctypes.data)
            assert not (travels & ~d).any()
            for i in np.nonzero(travels)[0]:
                for n in range(R):
                    if (travels[i] >> n) & 1:
                        got.setdefault((n, int(rows["group"][i])), []).append((s, int(step[i]), int(k[i]), rows[i:i + 1]))
            cap = 8 * G
            xr, to, st, kk = np.zeros(cap, capi.MSG_DTYPE), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
            m = lib.hc_votes_expand(node._h, C.addressof(mail.c), s, need, xr.ctypes.data, to.ctypes.data, st.ctypes.data, kk.ctypes.data, cap)
            n_expanded += m
            for i in range(m):
                got.setdefault((int(to[i]), int(xr["group"][i])), []).append((s, int(st[i]), int(kk[i]), xr[i:i + 1]))
        for (d, g), want in plain.items():
            want = sorted(want, key=lambda e: e[:3])
            have = sorted(got.get((d, g), []), key=lambda e: e[:3])
            if have:  # as rows: all of them
                assert [e[:3] for e in have] == [e[:3] for e in want], (d, g)
                assert b"".join(e[3].tobytes() for e in have) == b"".join(e[3].tobytes() for e in want), (d, g)
                n_rows += len(have)
                continue
            assert VoteMail.bits(mail.wordmail, d, G)[g] and not VoteMail.bits(mail.rowmail, d, G)[g], (d, g)
            said = []  # the words, read as jg_vote_half_group reads them
            for s in range(R):
                if s == d:
                    continue
                qc, ac = int(mail.q_ctl[s, g]), int(mail.a_ctl[s, g])
                qn = qc & 0xff
                if qn:
                    assert qn == need
                    q_ord = ((qc >> 8) - qn * (qn - 1) // 2) // qn
                    for j in range(qn):
                        said.append((s, q_ord >> 8, (q_ord & 0xff) + j, _row(g, capi.CMD_VOTE_REQUEST, capi.TO_PEERS, 0, ids[s], int(mail.q_term[s, g]),
                                                                            int(mail.q_head[s, g]), int(mail.q_term[s, g]))))
                if ac & 0xff and (ac >> 21) & 7 == d:
                    a_ord = (ac >> 8) & 0x7ff
                    for j in range(ac & 0xff):
                        said.append((s, a_ord >> 8, (a_ord & 0xff) + j, _row(g, capi.CMD_VOTE_RESPONSE, capi.TO_PEER, ids[d], ids[s], int(mail.a_term[s, g]), 0, 0,
                                                                            (ac >> (20 if j else 19)) & 1)))
            said = sorted(said, key=lambda e: e[:3])
            assert [e[:3] for e in said] == [e[:3] for e in want], (d, g)
            assert b"".join(e[3].tobytes() for e in said) == b"".join(e[3].tobytes() for e in want), (d, g)
            n_words += len(said)
        assert not (set(got) - set(plain))
    assert n_words > 500 and n_rows > 500 and n_expanded > 50 and n_double > 10, (n_words, n_rows, n_expanded, n_double)
