"""The routed round's transport (jg_route.h) and the election mail's kernels (jg_votes.h) as KERNELS on the host (CPU):
tests/host_workgroups.py runs a workgroup as 256 cooperative fibers, so the delivering pass with its LDS staging and
tallies, the bucket pass, the in-LDS sort, the census, the word-aware delivering pass, the expansion and the receiving
half's launch execute as written - against the numpy statement of the same pass, and, inside a routed cluster of
host-compiled nodes, against the oracle cluster that moves every message as a row."""
import ctypes as C

import numpy as np
import pytest

from josefine_amd import capi
from dense_node import PHASE_DELIVERED, PHASE_FOLLOWER, PHASE_INJECTED, PHASE_LEADER, RoutedCluster, cluster_failure_rows, emission_index, routable
from host_compiled import HostCompiled, VoteMail
import host_workgroups as hw
from oracle_lib import oracle_engine
from parity import compare_snapshots
from vote_mail_cases import check_mail, random_emissions


def test_the_stand_in_itself():
    """a workgroup scan (wave shuffles, an LDS hop, barriers), ballots after some lanes have left, a wave reduction"""
    lib = hw.build()
    rng = np.random.default_rng(1)
    for n in (1, 63, 64, 65, 255, 256, 257, 1000):
        a = rng.integers(0, 9, n).astype(np.uint32)
        nb = (n + 255) // 256
        excl, total, ballots, odd = np.zeros(n, np.uint32), np.zeros(nb, np.uint32), np.zeros((n + 63) // 64, np.uint64), np.zeros(1, np.uint32)
        lib.hw_selftest(a.ctypes.data, n, excl.ctypes.data, total.ctypes.data, ballots.ctypes.data, odd.ctypes.data)
        for b in range(nb):
            seg = a[b * 256:(b + 1) * 256]
            assert np.array_equal(excl[b * 256:b * 256 + len(seg)], np.cumsum(seg) - seg) and total[b] == seg.sum(), (n, b)
        bits = (a & 1).astype(bool)
        for w in range((n + 63) // 64):
            assert int(ballots[w]) == sum(1 << i for i, x in enumerate(bits[w * 64:(w + 1) * 64]) if x), (n, w)
        assert int(odd[0]) == int(a[bits].sum()), n


def senders_of(emitted, R, rng, seq_base=40):
    """a round's emissions as the kernels find them: step 2 in a sparse step's output region (a slot per partition, its rows
    back to back), the other steps in the exceptional queue, in any order"""
    out = []
    for s in range(R):
        xq = [e for e in emitted[s] if e[1] != 2]
        rec = [e for e in emitted[s] if e[1] == 2]
        q = np.zeros(len(xq), hw.XQ_DTYPE)
        for i, (r, step, k) in enumerate(xq):
            q["row"][i], q["seq"][i], q["k"][i] = r[0], seq_base + step, k
        q = q[rng.permutation(len(q))]
        # (... and rows of an earlier round, left undrained: not this round's mail)
        old = np.zeros(2, hw.XQ_DTYPE)
        old["row"]["kind"], old["row"]["to_kind"], old["seq"] = capi.CMD_HEARTBEAT, capi.TO_PEERS, seq_base
        sd = dict(xq=np.concatenate([q, old]), seq_base=seq_base, rec=None)
        if rec:
            groups = sorted({int(e[0]["group"][0]) for e in rec})
            per_row = max(sum(1 for e in rec if int(e[0]["group"][0]) == g) for g in groups)
            cnt, msg = np.zeros(len(groups), np.uint32), np.zeros((len(groups), per_row), capi.MSG_DTYPE)
            for i, g in enumerate(groups):
                mine = sorted((e for e in rec if int(e[0]["group"][0]) == g), key=lambda e: e[2])
                assert [e[2] for e in mine] == list(range(len(mine)))  # (a slot's rows: emission index = position)
                cnt[i] = len(mine)
                for j, e in enumerate(mine):
                    msg[i, j] = e[0][0]
            sd["rec"] = (2, cnt, msg)
        out.append(sd)
    return out


def rows_by_partition(out, R):
    """the addressees' command columns -> {(d, g): rows in the order they will be applied}"""
    got = {}
    for d in range(R):
        c = out[d]
        arr = np.zeros(len(c["kind"]), capi.MSG_DTYPE)
        arr["kind"], arr["flag"], arr["group"], arr["from"], arr["term"], arr["id"], arr["aux"] = c["kind"], c["flag"], c["group"], c["from_"], c["term"], c["id"], c["aux"]
        assert (np.diff(arr["group"].astype(np.int64)) >= 0).all()  # (a node's batch is sorted by partition)
        for i in range(len(arr)):
            got.setdefault((d, int(arr["group"][i])), []).append(arr[i:i + 1])
    return got


@pytest.mark.parametrize("R,seed,words", [(3, 1, False), (5, 2, False), (3, 3, True), (4, 4, True), (5, 5, True), (5, 6, True)])
def test_the_transports_kernels_on_random_emissions(R, seed, words):
    """words = False: the product's default delivering pass + bucket pass + in-LDS sort deliver exactly the plain
    transport's rows, per addressee in (partition, phase, emission index, sender) order; words = True: the census, the word-aware
    delivering pass and the expansion - per addressee and partition EITHER all of the plain rows OR words that say them"""
    G = 96
    rng = np.random.default_rng(seed)
    node = HostCompiled(G, R, seed=1, self_slots=np.zeros(G, np.uint8))
    ids = np.array(node.node_ids[:R], np.uint32)
    tr = hw.Transport(R, G, ids, words)
    n_words = n_rows = 0
    for it in range(3):
        mail, emitted, plain, _ = random_emissions(R, G, ids, rng, words=words)
        out, per, kinds, stays = tr.route(senders_of(emitted, R, rng), mail)
        got = rows_by_partition(out, R)
        nw, nr = check_mail(R, G, ids, mail, plain, got, R - 1)
        n_words, n_rows = n_words + nw, n_rows + nr
        if not words:
            assert nw == 0 and nr == sum(len(v) for v in plain.values())
        # the tallies: rows per (sender, addressee), the kinds an addressee receives, the rows that stay with their sender
        for s in range(R):
            rows = np.concatenate([e[0] for e in emitted[s]])
            ok = routable(rows, ids)
            assert int(stays[s]) == int((~ok).sum()) + 2, (s, stays[s])  # (+ the two rows of an earlier round)
        for d in range(R):
            k = 0
            for key, v in got.items():
                if key[0] == d:
                    for r in v:
                        k |= 1 << int(r["kind"][0])
            assert int(kinds[d]) == k, (d, kinds[d], k)
            assert int(per[:, d].sum()) == sum(len(v) for key, v in got.items() if key[0] == d)
    assert n_rows > 300 and (not words or n_words > 200), (n_rows, n_words)
    assert tr.repeats >= 1  # (the staging started too small: a segment ran over, the pass was repeated with a larger one - the census was not)


class KernelMailCluster(RoutedCluster):
    """the routed round under JG_ROUTE_VOTE_WORDS as round_routed_impl launches it, kernels on the host: k_votes_clear,
    k_vote_half_multi (every node in one launch), the rows' steps, the dense round, then the transport (hw_route)"""

    def __init__(self, G, R, **kw):
        super().__init__(HostCompiled, G, R, **kw)
        self.lib = hw.build()
        self.mail = [VoteMail(R, G), VoteMail(R, G)]
        self.t = 0
        self.tr = hw.Transport(R, G, self.member_ids, True)
        self.next_rows = [None] * R
        self.handles = (C.c_void_p * R)(*[n._h for n in self.nodes])
        self.rows_moved = 0

    def round(self, appends, inject=None, dt_ms=100):
        G, R = self.G, self.R
        now = self.now + dt_ms
        prev, cur = self.mail[(self.t + 1) & 1], self.mail[self.t & 1]
        self.lib.hw_votes_clear(C.addressof(cur.c))  # (sparse: the control words of the partitions a wordmail bit names)
        assert not (cur.q_ctl.any() or cur.a_ctl.any() or cur.rowmail.any() or cur.wordmail.any()), "the mail of two rounds ago was not cleared"
        seq = np.zeros(R, np.uint32)
        rc = self.lib.hw_vote_half_multi(self.handles, R, seq.ctypes.data, PHASE_DELIVERED, now, C.addressof(prev.c), C.addressof(cur.c), 2)
        assert rc == 0, rc
        senders = []
        kept_now = [[] for _ in range(R)]
        for n in range(R):
            cap = (R + 3) * G + 64
            q = np.zeros(cap, hw.XQ_DTYPE)
            m = self.lib.hw_take_xq(self.nodes[n]._h, q.ctypes.data, cap)
            xq = [q[:m]]  # (the vote half's rows: the step numbered seq_base + 1, phase 1)
            seq_base = int(seq[n]) - 1
            # the delivered rows - phase 1 too, other partitions than the words' - as a sparse step's output region: a slot per
            # partition, its rows back to back
            cols = self.next_rows[n]
            rec = None
            if cols is not None:
                self.delivered[n] += len(cols["kind"])
                self.rows_moved += len(cols["kind"])
                self.nodes[n].submit_columns(**cols)
                self.nodes[n].step(now)
                out = self.nodes[n].drain_messages()
                kept_now[n].append(out)
                if len(out):
                    groups, first = np.unique(out["group"], return_index=True)
                    cnt = np.diff(np.r_[first, len(out)]).astype(np.uint32)
                    msg = np.zeros((len(groups), int(cnt.max())), capi.MSG_DTYPE)
                    for i, (a, c) in enumerate(zip(first, cnt)):
                        msg[i, :c] = out[a:a + c]
                    rec = (PHASE_DELIVERED, cnt, msg)
            # the injected rows: a step of their own (numbered seq_base + 2, phase 2); its rows reach the transport through the queue here
            ic = self._inject_columns(inject[n] if inject else None)
            if ic is not None:
                self.delivered[n] += len(ic["kind"])
                self.nodes[n].submit_columns(**ic)
                self.nodes[n].step(now)
                out = self.nodes[n].drain_messages()
                kept_now[n].append(out)
                q2 = np.zeros(len(out), hw.XQ_DTYPE)
                q2["row"], q2["seq"], q2["k"] = out, seq_base + 2, emission_index(out["group"])
                xq.append(q2)
            # (the node's steps of the round -> phases: 1, 2, then its dense half - numbered seq_base + 3 below)
            phases = PHASE_DELIVERED << 3 | PHASE_INJECTED << 6 | (PHASE_LEADER if n == self.lead else PHASE_FOLLOWER) << 9
            senders.append(dict(xq=xq, seq_base=seq_base, rec=rec, phases=phases))
        outs = self.dense_round(appends, dt_ms)
        drained = self.rows.pop()
        for s in range(R):
            d = drained[s]
            kept_now[s].append(d)
            q = np.zeros(len(d), hw.XQ_DTYPE)
            q["row"], q["seq"], q["k"] = d, senders[s]["seq_base"] + 3, emission_index(d["group"])
            senders[s]["xq"] = np.concatenate(senders[s]["xq"] + [q])
        for s in range(R):  # (the vote half's rows are all mail; what stays is what the rows' step and the dense round kept, in drain order)
            rows = np.concatenate(kept_now[s])
            self.kept[s] = np.concatenate([self.kept[s], rows[~routable(rows, self.member_ids)]])
        out, per, kinds, stays = self.tr.route(senders, cur)
        self.next_rows = [c if len(c["kind"]) else None for c in out]
        # the words' copies count as the rows they stand for
        need = R - 1
        for n in range(R):
            as_rows = VoteMail.bits(cur.rowmail, n, G).copy()
            for s in range(R):
                if s != n:
                    c = cur.q_ctl[s] & 0xff
                    as_rows |= (c != 0) & (c != need)
            for s in range(R):
                if s == n:
                    continue
                self.delivered[n] += int((cur.q_ctl[s] & 0xff)[~as_rows & ((cur.q_ctl[s] & 0xff) == need)].sum())
                ac = cur.a_ctl[s]
                self.delivered[n] += int((ac & 0xff)[~as_rows & (((ac >> 21) & 7) == n)].sum())
        self.t += 1
        return outs


@pytest.mark.parametrize("R,percent,also,T", [(5, 3, (), 40), (3, 4, (2,), 40), (5, 3, (2,), 30), (3, 25, (1, 2), 40), (5, 25, (1, 2, 3), 30), (4, 30, (1, 2), 30)])
def test_routed_round_with_the_vote_mail_in_the_devices_kernels(R, percent, also, T):
    G = 150
    ora = RoutedCluster(oracle_engine, G, R, seed=5)
    dev = KernelMailCluster(G, R, seed=5)
    for t in range(T):
        inj = cluster_failure_rows(99, t, G, R, percent, also=also) if t >= 3 else [None] * R
        ora.round(np.ones(G, np.uint64), inject=inj)
        dev.round(np.ones(G, np.uint64), inject=[None if c is None else dict(c) for c in inj])
        for n in range(R):
            compare_snapshots(dev.nodes[n], ora.nodes[n], f"round {t} node {n}")
        assert [k.tobytes() for k in ora.kept] == [k.tobytes() for k in dev.kept], t
    for n in range(R):
        assert dev.nodes[n].counters()["decisions"] == ora.nodes[n].counters()["decisions"], n
        assert dev.nodes[n].drain_faults().tobytes() == ora.nodes[n].drain_faults().tobytes()
    pending_dev = np.array([0 if c is None else len(c["kind"]) for c in dev.next_rows])
    pending_ora = np.array([ora.pending(n) for n in range(R)])
    # (rows are counted when applied, the words' copies when sent)
    assert (ora.delivered + pending_ora).tolist() == (dev.delivered + pending_dev).tolist()
    assert dev.rows_moved < int(ora.delivered.sum()) // 2 or percent >= 10
