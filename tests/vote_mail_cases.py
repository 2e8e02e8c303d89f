"""Emissions no cluster would produce together, for the election mail's transport (jg_votes.h): per sender and partition
a random mix of campaigns (sometimes two in a round), malformed requests, answers next to rows, rows to one addressee only,
rows the transport keeps - with the rows the PLAIN transport delivers per addressee and partition (answer words written out),
and the check that whatever arrived is EITHER exactly those rows, in its order, OR nothing, with words that read back (as
jg_vote_half_group reads them) to exactly those rows.  Shared by tests/test_vote_mail.py (the functions, lane by lane) and
tests/test_host_workgroups.py (the kernels)."""
import numpy as np

from josefine_amd import capi
from host_compiled import VoteMail


def row(g, kind, to_kind, to_id, frm, term, id_=0, aux=0, flag=0):
    r = np.zeros(1, capi.MSG_DTYPE)
    r["group"], r["kind"], r["to_kind"], r["to_id"], r["from"], r["term"], r["id"], r["aux"], r["flag"] = g, kind, to_kind, to_id, frm, term, id_, aux, flag
    return r


def random_emissions(R, G, ids, rng, words=True):
    """-> (mail with the answer words written, emitted[s] = list of (row, step, k), plain[(d, g)] = list of (s, step, k, row),
    number of (sender, partition) pairs with two campaigns in the round)"""
    need = R - 1
    mail = VoteMail(R, G)
    mail.q_term[:], mail.q_head[:], mail.a_term[:] = rng.integers(0, 1 << 60, (3, R, G), dtype=np.uint64)  # (garbage where no control word says otherwise)
    plain, emitted, n_double = {}, [[] for _ in range(R)], 0
    for s in range(R):
        for g in range(G):
            if rng.random() < 0.55:
                continue
            k = {1: 0, 2: 0, 3: 0}
            events = []
            if rng.random() < 0.3:  # (something said before the answers)
                events.append(("row", 1, k[1], row(g, capi.CMD_HEARTBEAT_RESPONSE, capi.TO_PEER, ids[(s + 1) % R], ids[s], 1, 5)))
                k[1] += 1
            if words and rng.random() < 0.5:  # an answer word (the vote half's: step 1)
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
                        events.append(("row", step, k[step], row(g, capi.CMD_VOTE_REQUEST, capi.TO_PEERS, 0, ids[s], term, head, term)))
                        k[step] += 1
                elif kind == "oddcampaign":  # R - 1 broadcasts that are not a campaign's: last_term != term
                    for _c in range(need):
                        events.append(("row", step, k[step], row(g, capi.CMD_VOTE_REQUEST, capi.TO_PEERS, 0, ids[s], term, head, term + 1)))
                        k[step] += 1
                elif kind == "odd":  # a request that is not a campaign's copy: to one peer, or last_term != term
                    to_one = rng.random() < 0.5
                    events.append(("row", step, k[step], row(g, capi.CMD_VOTE_REQUEST, capi.TO_PEER if to_one else capi.TO_PEERS,
                                                             ids[(s + 1) % R] if to_one else 0, ids[s], term, head, term if to_one else term + 1)))
                    k[step] += 1
                elif kind == "heartbeat":
                    events.append(("row", step, k[step], row(g, capi.CMD_HEARTBEAT, capi.TO_PEERS, 0, ids[s], term, head)))
                    k[step] += 1
                elif kind == "response":
                    events.append(("row", step, k[step], row(g, capi.CMD_VOTE_RESPONSE, capi.TO_PEER, ids[(s + 1 + rng.integers(0, R - 1)) % R], ids[s], term, 0, 0,
                                                             int(rng.integers(0, 2)))))
                    k[step] += 1
                else:
                    events.append(("row", step, k[step], row(g, capi.CMD_APPEND_ENTRIES, capi.TO_PEER, ids[(s + 1) % R], ids[s], term, head, 1)))
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
                        plain.setdefault((to, g), []).append((s, step, k0 + j, row(g, capi.CMD_VOTE_RESPONSE, capi.TO_PEER, ids[to], ids[s], term, 0, 0, rest if j else first)))
                else:
                    _, step, kk, r = e
                    emitted[s].append((r, step, kk))
                    if r["kind"][0] in (capi.CMD_APPEND_ENTRIES, capi.CMD_CLIENT_REQUEST):
                        continue
                    for d in range(R):
                        if d != s and (r["to_kind"][0] == capi.TO_PEERS or r["to_id"][0] == ids[d]):
                            plain.setdefault((d, g), []).append((s, step, kk, r))
    return mail, emitted, plain, n_double


COLS = ("kind", "flag", "group", "from", "term", "id", "aux")  # what an addressee's next step is given of a row


def command_bytes(rows):
    return b"".join(r[0][c].tobytes() for r in rows for c in COLS)


def check_mail(R, G, ids, mail, plain, got_rows, need):
    """got_rows[(d, g)]: the rows that arrived for addressee d and partition g, in the order they are applied.  Returns
    (rows' worth in words, rows as rows)"""
    n_words = n_rows = 0
    for (d, g), want in plain.items():
        want = [e[3] for e in sorted(want, key=lambda e: (e[1], e[2], e[0]))]  # the transport's order: (phase, emission index, sender)
        have = got_rows.get((d, g), [])
        if have:  # as rows: all of them, in the plain transport's order
            assert command_bytes(have) == command_bytes(want), (d, g, len(have), len(want))
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
                    said.append((s, q_ord >> 8, (q_ord & 0xff) + j, row(g, capi.CMD_VOTE_REQUEST, capi.TO_PEERS, 0, ids[s], int(mail.q_term[s, g]),
                                                                       int(mail.q_head[s, g]), int(mail.q_term[s, g]))))
            if ac & 0xff and (ac >> 21) & 7 == d:
                a_ord = (ac >> 8) & 0x7ff
                for j in range(ac & 0xff):
                    said.append((s, a_ord >> 8, (a_ord & 0xff) + j, row(g, capi.CMD_VOTE_RESPONSE, capi.TO_PEER, ids[d], ids[s], int(mail.a_term[s, g]), 0, 0,
                                                                       (ac >> (20 if j else 19)) & 1)))
        said = [e[3] for e in sorted(said, key=lambda e: (e[1], e[2], e[0]))]
        assert command_bytes(said) == command_bytes(want), (d, g)
        n_words += len(said)
    assert not (set(k for k, v in got_rows.items() if v) - set(plain))
    return n_words, n_rows
