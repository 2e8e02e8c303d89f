"""The election vocabulary as mailbox words, with the TRANSPORT's side in the device's own functions (CPU): a routed cluster
of host-compiled nodes (tests/host_compiled.py) in which every emitted row goes through jg_votes.h's census
(jg_votes_census_row), the delivering pass asks jg_votes_row_travels per row and addressee, the answer words a partition
cannot take are expanded back to rows (jg_votes_expand_count / _row), and the receiving half (jg_vote_half_group) reads the packed
mail of the round before - i.e. the routed round under JG_ROUTE_VOTE_WORDS=1 with the staging and its ordering played in
numpy - against the same cluster over oracle engines in which every message is a row: every column of every node after
every round, the rows delivered and the rows kept."""
import ctypes as C

import numpy as np
import pytest

from josefine_amd import capi
from dense_node import PHASE_DELIVERED, PHASE_INJECTED, RoutedCluster, cluster_failure_rows, emission_index, routable
from host_compiled import HostCompiled, VoteMail
from oracle_lib import oracle_engine
from parity import compare_snapshots

ROUTE_DTYPE = np.dtype([("src", "<i8"), ("step", "<i8"), ("k", "<i8")])


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
        emitted = [[] for _ in range(R)]  # per sender: (rows, phase of the round, emission index)
        for n in range(R):
            # phase 1: what the transport delivered - the words (the receiving half) and, for the other partitions, the rows
            xrows, xk = self.nodes[n].vote_half_mail(n, now, prev, cur, step=PHASE_DELIVERED, need=self.need)
            parts = list(self.inrows[n])
            self.inrows[n] = []
            rows = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, capi.MSG_DTYPE)
            keys = np.concatenate([p[1] for p in parts]) if parts else np.zeros(0, ROUTE_DTYPE)
            order = np.lexsort((keys["src"], keys["k"], keys["step"], rows["group"]))  # (the staging's ordering key: partition, phase, emission index, sender)
            rows = rows[order]
            self.in_rows += len(rows)
            self.vote_rows_as_rows += int(np.isin(rows["kind"], (capi.CMD_VOTE_REQUEST, capi.CMD_VOTE_RESPONSE)).sum())
            out = np.zeros(0, capi.MSG_DTYPE)
            if len(rows):
                self.nodes[n].submit_columns(**self.columns_of(rows))
                self.nodes[n].step(now)
                out = self.nodes[n].drain_messages()
            if len(out) + len(xrows):  # one step's emissions, partitions ascending (its two halves served different partitions)
                both, k = np.concatenate([xrows, out]), np.concatenate([xk.astype(np.int64), emission_index(out["group"])])
                order = np.argsort(both["group"], kind="stable")
                emitted[n].append((both[order], np.full(len(both), PHASE_DELIVERED, np.uint32), k[order].astype(np.uint32)))
            cols = self._inject_columns(inject[n] if inject else None)
            if cols is not None:  # phase 2: the injected rows
                self.delivered[n] += len(cols["kind"])
                self.nodes[n].submit_columns(**cols)
                self.nodes[n].step(now)
                out = self.nodes[n].drain_messages()
                if len(out):
                    emitted[n].append((out, np.full(len(out), PHASE_INJECTED, np.uint32), emission_index(out["group"]).astype(np.uint32)))
        outs = self.dense_round(appends, dt_ms)
        for s, parts in enumerate(self.dense_emitted()):
            emitted[s].extend((rows, phase.astype(np.uint32), k.astype(np.uint32)) for rows, phase, k in parts)
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
        lib.hc_votes_validate(C.addressof(cur.c), self.need)
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
    pending = np.array([ora.pending(n) for n in range(R)])
    assert (ora.delivered + pending).tolist() == dev.delivered.tolist()
    if percent < 10:  # the configs[4] rates: nearly all of the vote traffic is words
        assert dev.in_words > 10 * dev.vote_rows_as_rows and dev.in_words > G, (dev.in_words, dev.vote_rows_as_rows)
    else:  # (chaos; the ways back to rows are taken in test_the_transports_functions_on_random_mail)
        assert dev.in_words > G
    print(f"R={R} also={also}: {dev.in_words} rows' worth in words, {dev.in_rows} rows delivered ({dev.vote_rows_as_rows} of them votes, {dev.expanded} expanded)")


@pytest.mark.parametrize("R,seed", [(3, 1), (4, 2), (5, 3), (5, 4)])
def test_the_transports_functions_on_random_mail(R, seed):
    """census / travels / expand on emissions no cluster would produce together (tests/vote_mail_cases.py): for every
    addressee and partition EITHER everything arrives as rows - then exactly the rows the plain transport delivers, answer
    words written out, in its order - OR nothing does, and the words read back (as the receiving half reads them) say
    exactly those rows"""
    from vote_mail_cases import check_mail, random_emissions
    G = 96
    rng = np.random.default_rng(seed)
    node = HostCompiled(G, R, seed=1, self_slots=np.zeros(G, np.uint8))
    ids = np.array(node.node_ids[:R], np.uint32)
    lib = node.lib
    need = R - 1
    n_words = n_rows = n_expanded = n_double = 0
    for it in range(6):
        mail, emitted, plain, nd = random_emissions(R, G, ids, rng)
        n_double += nd
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
        lib.hc_votes_validate(C.addressof(mail.c), need)
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
        got_rows = {key: [e[3] for e in sorted(v, key=lambda e: (e[1], e[2], e[0]))] for key, v in got.items()}  # (phase, emission index, sender)
        nw, nr = check_mail(R, G, ids, mail, plain, got_rows, need)
        n_words, n_rows = n_words + nw, n_rows + nr
    assert n_words > 500 and n_rows > 500 and n_expanded > 50 and n_double > 10, (n_words, n_rows, n_expanded, n_double)
