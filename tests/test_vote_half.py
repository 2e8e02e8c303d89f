"""The election vocabulary as mailbox words, END TO END on the host (CPU): a routed cluster whose nodes are the device's state
machine compiled for the host (tests/host_compiled.py) and whose vote traffic does NOT travel as rows - every partition whose
inbound batch of a round is vote traffic only is encoded into request / answer words (tests/election_words.py), applied by the
receiving half (josefine_amd/csrc/jg_votes.h: the per-partition logic of the step that is to replace the row transport for
this traffic; not part of the engine yet), and answered in words - against the same cluster over oracle engines with every
message a row: every column of every node after every round, the rows delivered and the rows kept, on the configs[4]
traces (R = 5: campaigns refused; R = 3 and R = 5 with a second restarted replica: campaigns won, the winner's Heartbeat)."""
import numpy as np
import pytest

from josefine_amd import capi
from dense_node import PHASE_DELIVERED, PHASE_INJECTED, RoutedCluster, cluster_failure_rows, emission_index
from election_words import ANS_DTYPE, REQ_DTYPE, decode, encode
from host_compiled import HostCompiled
from oracle_lib import oracle_engine
from parity import compare_snapshots


def emitted_rows(out, xrows, xk, n, member_ids):
    """what node n emitted in its vote half: its answer words as the rows they stand for and its exceptional rows, per
    partition in emission order -> (rows, their emission indices)"""
    parts, keys = [], []
    for g in np.nonzero(out["n"])[0]:
        m = int(out["n"][g])
        r = np.zeros(m, dtype=capi.MSG_DTYPE)
        r["group"], r["kind"], r["to_kind"], r["to_id"] = g, capi.CMD_VOTE_RESPONSE, capi.TO_PEER, member_ids[int(out["to"][g])]
        r["from"], r["term"] = member_ids[n], out["term"][g]
        r["flag"] = (int(out["bits"][g]) >> 1) & 1
        r["flag"][0] = int(out["bits"][g]) & 1
        parts.append(r)
        keys.append(np.stack([np.full(m, g, np.int64), (int(out["at"][g]) & 0xff) + np.arange(m)], axis=1))
    if len(xrows):
        parts.append(xrows)
        keys.append(np.stack([xrows["group"].astype(np.int64), xk.astype(np.int64)], axis=1))
    if not parts:
        return np.zeros(0, dtype=capi.MSG_DTYPE), np.zeros(0, np.int64)
    rows, key = np.concatenate(parts), np.concatenate(keys)
    order = np.lexsort((key[:, 1], key[:, 0]))
    return rows[order], key[order, 1]


def empty_words(R, G):
    return dict(q_term=np.zeros((R, G), np.uint64), q_head=np.zeros((R, G), np.uint64), q_n=np.zeros((R, G), np.uint8), q_at=np.zeros((R, G), np.uint32),
                a_term=np.zeros((R, G), np.uint64), a_n=np.zeros((R, G), np.uint8), a_at=np.zeros((R, G), np.uint32), a_bits=np.zeros((R, G), np.uint8),
                a_to=np.zeros((R, G), np.uint8))


class WordCluster(RoutedCluster):
    """RoutedCluster over the host-compiled device source whose vote traffic goes through jg_votes.h"""

    def __init__(self, G, R, **kw):
        super().__init__(HostCompiled, G, R, **kw)
        self.word_rows = self.row_rows = self.vote_rows_as_rows = self.answered_in_words = self.exceptional = self.exceptional_answers = 0

    def round(self, appends, inject=None, dt_ms=100):
        G, R = self.G, self.R
        now = self.now + dt_ms
        emitted = [[] for _ in range(R)]
        for n in range(R):
            rows, src, ord_ = self.inbound_order(n)
            self.inbound[n] = []
            if len(rows):
                cols = self.columns_of(rows)
                self.delivered[n] += len(rows)
                reqs, anss, stay = encode(cols, src, ord_, self.member_ids)
                # a partition travels in words when everything it receives this round is words (and one answer word can hold
                # what it will say: one requester)
                per_group_rows = np.bincount(cols["group"], minlength=G)
                per_group_word_rows = np.bincount(cols["group"][~stay], minlength=G)
                requesters = np.bincount(reqs["group"], minlength=G) if len(reqs) else np.zeros(G, np.int64)
                in_words = (per_group_rows > 0) & (per_group_rows == per_group_word_rows) & (requesters <= 1)
                as_rows = ~in_words[cols["group"]]
                self.word_rows += int((~as_rows).sum())
                self.row_rows += int(as_rows.sum())
                self.vote_rows_as_rows += int((as_rows & ~stay).sum())
                # the delivered step: the rows through the state machine, the words through the receiving half (other partitions)
                self.nodes[n].submit_columns(**{k: v[as_rows] for k, v in cols.items()})
                self.nodes[n].step(now)
                out1 = self.nodes[n].drain_messages()
                w = empty_words(R, G)
                for q in reqs[in_words[reqs["group"]]] if len(reqs) else []:
                    s, g = int(q["src"]), int(q["group"])
                    w["q_term"][s, g], w["q_head"][s, g], w["q_n"][s, g], w["q_at"][s, g] = q["term"], q["head"], q["copies"], q["at"]
                for a in anss[in_words[anss["group"]]] if len(anss) else []:
                    s, g = int(a["src"]), int(a["group"])
                    w["a_term"][s, g], w["a_n"][s, g], w["a_at"][s, g] = a["term"], a["copies"], a["at"]
                    w["a_bits"][s, g], w["a_to"][s, g] = int(a["first"]) | int(a["rest"]) << 1, n  # (delivered to this node: addressed to it)
                out, xrows, xk = self.nodes[n].vote_half(n, now, w, step=PHASE_DELIVERED)
                out2, k2 = emitted_rows(out, xrows, xk, n, self.member_ids)
                self.answered_in_words += int(out["n"].sum())
                self.exceptional += len(xrows)
                self.exceptional_answers += int((xrows["kind"] == capi.CMD_VOTE_RESPONSE).sum())
                # one step's emissions, partitions ascending (the two halves of it served different partitions)
                both, k = np.concatenate([out1, out2]), np.concatenate([emission_index(out1["group"]), k2])
                order = np.argsort(both["group"], kind="stable")
                if len(both):
                    emitted[n].append((both[order], np.full(len(both), PHASE_DELIVERED, np.int64), k[order]))
            cols = self._inject_columns(inject[n] if inject else None)
            if cols is not None:
                self.delivered[n] += len(cols["kind"])
                self.nodes[n].submit_columns(**cols)
                self.nodes[n].step(now)
                out = self.nodes[n].drain_messages()
                if len(out):
                    emitted[n].append((out, np.full(len(out), PHASE_INJECTED, np.int64), emission_index(out["group"])))
        outs = self.dense_round(appends, dt_ms)
        for s, parts in enumerate(self.dense_emitted()):
            emitted[s].extend(parts)
        self.transport(emitted)
        return outs


@pytest.mark.parametrize("R,percent,also", [(5, 3, ()), (3, 4, (2,)), (5, 3, (2,)), (3, 4, ())])
def test_vote_traffic_in_words_end_to_end(R, percent, also):
    G, T = 150, 40
    ora = RoutedCluster(oracle_engine, G, R, seed=5)
    dev = WordCluster(G, R, seed=5)
    for t in range(T):
        inj = cluster_failure_rows(99, t, G, R, percent, also=also) if t >= 3 else [None] * R
        ora.round(np.ones(G, np.uint64), inject=inj)
        dev.round(np.ones(G, np.uint64), inject=[None if c is None else dict(c) for c in inj])
        for n in range(R):
            compare_snapshots(dev.nodes[n], ora.nodes[n], f"round {t} node {n}")
        assert ora.delivered.tolist() == dev.delivered.tolist(), t
        assert [k.tobytes() for k in ora.kept] == [k.tobytes() for k in dev.kept], t
    for n in range(R):
        assert dev.nodes[n].counters()["decisions"] == ora.nodes[n].counters()["decisions"], n
    # the vote traffic did travel in words: nearly all of it (the rest shared its partition's batch with a row of another kind)
    assert dev.word_rows > 10 * dev.vote_rows_as_rows and dev.word_rows > G
    # ... and was answered in words: what the half left as rows is what a winner says (Heartbeat) and little else
    assert dev.answered_in_words > 20 * dev.exceptional_answers and dev.answered_in_words > G, (dev.answered_in_words, dev.exceptional_answers)
    print(f"R={R} also={also}: {dev.answered_in_words} answers emitted in words, {dev.exceptional} rows beside them")
    print(f"R={R} also={also}: {dev.word_rows} vote rows delivered as words, {dev.vote_rows_as_rows} as rows; {dev.row_rows} rows in all")


def test_jg_votes_compiles_for_gfx950():
    """the header is not part of the engine's translation unit yet: the device compiler sees it here"""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tmp = tempfile.mkdtemp(prefix="jg_votes_")
    src = os.path.join(tmp, "votes.hip")
    open(src, "w").write('#define JG_BLOCK 256\n#include "jg_votes.h"\n')
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", src, f"-I{os.path.join(root, 'josefine_amd', 'csrc')}",
                        "-o", os.path.join(tmp, "votes.o")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("R,seed", [(3, 1), (5, 2), (5, 3)])
def test_random_words_on_every_role(R, seed):
    """words no cluster would send (any term around the receiver's, any head, 1..R+1 copies, either answer bit pattern, an
    answer before or after a request from the same sender, several requesters for one partition) on the states a failure
    run leaves behind (leaders, followers that voted or did not, candidates): the vote half against the oracle fed the
    rows the words stand for - every column, and everything emitted, row for row"""
    G, T = 96, 14
    rng = np.random.default_rng(seed)
    ora = RoutedCluster(oracle_engine, G, R, seed=7)
    dev = RoutedCluster(HostCompiled, G, R, seed=7)
    for t in range(T):
        inj = cluster_failure_rows(31, t, G, R, 12, also=(2,)) if t >= 2 else [None] * R
        ora.round(np.ones(G, np.uint64), inject=inj)
        dev.round(np.ones(G, np.uint64), inject=[None if c is None else dict(c) for c in inj])
    roles = np.concatenate([ora.nodes[n].read("role") for n in range(R)])
    assert len(np.unique(roles)) == 3, np.unique(roles)  # followers, candidates and leaders are all there
    ids = ora.member_ids
    now = ora.now
    emitted = 0
    for it in range(12):
        now += int(rng.integers(0, 300))
        for n in range(R):
            term = ora.nodes[n].read("term").astype(np.int64)
            head = ora.nodes[n].read("head").astype(np.int64)
            reqs, anss = [], []
            for g in np.nonzero(rng.random(G) < 0.5)[0]:
                for s in range(R):
                    if s == n or rng.random() < 0.4:
                        continue
                    has_q, has_a = rng.random() < 0.6, rng.random() < 0.6
                    nq, na = int(rng.integers(1, R + 2)), int(rng.integers(1, R + 2))
                    q_first = rng.random() < 0.5
                    q_at = 0 if q_first or not has_a else na
                    a_at = 0 if not q_first or not has_q else nq
                    if rng.random() < 0.2:  # (a Heartbeat went first: not mail for this half, the ordinal shifts)
                        q_at, a_at = q_at + 1, a_at + 1
                    if has_q:
                        reqs.append((g, s, q_at, max(0, term[g] + rng.integers(-1, 3)), max(0, head[g] + rng.integers(-2, 3)), nq))
                    if has_a:
                        anss.append((g, s, a_at, max(0, term[g] + rng.integers(-1, 2)), rng.integers(0, 2), rng.integers(0, 2), na))
            reqs, anss = np.array(reqs, REQ_DTYPE), np.array(anss, ANS_DTYPE)
            elsewhere = rng.random(len(anss)) < 0.15  # a sender's answer word for another node: this node must not read it
            anss_else, anss = anss[elsewhere], anss[~elsewhere]
            none = dict(kind=np.zeros(0, np.uint8), group=np.zeros(0, np.uint32), from_=np.zeros(0, np.uint32), term=np.zeros(0, np.uint64),
                        id=np.zeros(0, np.uint64), aux=np.zeros(0, np.uint64), flag=np.zeros(0, np.uint8))
            cols = decode(reqs, anss, none, np.zeros(0, np.int64), np.zeros(0, np.int64), ids)[0]
            if cols is None:
                continue
            ora.nodes[n].submit_columns(**cols)
            ora.nodes[n].step(now)
            w = empty_words(R, G)
            for q in reqs:
                s, g = int(q["src"]), int(q["group"])
                w["q_term"][s, g], w["q_head"][s, g], w["q_n"][s, g], w["q_at"][s, g] = q["term"], q["head"], q["copies"], q["at"]
            for a in anss:
                s, g = int(a["src"]), int(a["group"])
                w["a_term"][s, g], w["a_n"][s, g], w["a_at"][s, g] = a["term"], a["copies"], a["at"]
                w["a_bits"][s, g], w["a_to"][s, g] = int(a["first"]) | int(a["rest"]) << 1, n
            for a in anss_else:
                s, g = int(a["src"]), int(a["group"])
                w["a_term"][s, g], w["a_n"][s, g], w["a_at"][s, g] = a["term"], a["copies"], a["at"]
                w["a_bits"][s, g], w["a_to"][s, g] = int(a["first"]) | int(a["rest"]) << 1, (n + 1 + int(rng.integers(0, R - 1))) % R
            out, xrows, xk = dev.nodes[n].vote_half(n, now, w)
            compare_snapshots(dev.nodes[n], ora.nodes[n], f"iteration {it} node {n}")
            want, got = ora.nodes[n].drain_messages(), emitted_rows(out, xrows, xk, n, ids)[0]
            assert len(want) == len(got) and want.tobytes() == got.tobytes(), (it, n, len(want), len(got))
            assert ora.nodes[n].drain_faults().tobytes() == dev.nodes[n].drain_faults().tobytes()
            assert ora.nodes[n].drain_applies().tobytes() == dev.nodes[n].drain_applies().tobytes()
            emitted += len(got)
    assert emitted > 1000
    roles = np.concatenate([ora.nodes[n].read("role") for n in range(R)])
    assert len(np.unique(roles)) >= 2
