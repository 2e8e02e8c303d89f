"""tests/election_words.py held to the rows a routed cluster really exchanges (CPU, oracle engines): on the configs[4]
traces every node's inbound batch of every round is encoded into request / answer words and decoded again - the rows
must come back exactly, in the transport's order - and the share of the vote traffic that fits the words is counted."""
import numpy as np
import pytest

from josefine_amd import capi
from dense_node import RoutedCluster, cluster_failure_rows
from election_words import decode, encode
from oracle_lib import oracle_engine


class Recording(RoutedCluster):
    """RoutedCluster that keeps, per node and round, the delivered batch with the sender slot and the ord of every row"""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.batches = []

    def _inbound_columns(self, n):
        rows, src, ord_ = self.inbound_order(n)
        if len(rows):
            self.batches.append((n, self.columns_of(rows), src, ord_))
        return super()._inbound_columns(n)


@pytest.mark.parametrize("R,percent,also", [(5, 2, ()), (3, 3, (2,)), (5, 3, (2,))])
def test_vote_traffic_round_trips_through_the_words(R, percent, also):
    G, T = 400, 40
    cl = Recording(oracle_engine, G, R, seed=5)
    for t in range(T):
        inj = cluster_failure_rows(99, t, G, R, percent, also=also) if t >= 3 else None
        cl.round(np.ones(G, np.uint64), inject=inj)
    votes = fit = words = two_addressees = 0
    for n, cols, src, ord_ in cl.batches:
        is_vote = np.isin(cols["kind"], (capi.CMD_VOTE_REQUEST, capi.CMD_VOTE_RESPONSE))
        reqs, anss, stay = encode(cols, src, ord_, cl.member_ids)
        rest = {k: v[stay] for k, v in cols.items()}
        back, back_src, back_ord = decode(reqs, anss, rest, src[stay], ord_[stay], cl.member_ids)
        for k in cols:
            assert np.array_equal(back[k], cols[k]), (n, k)
        assert np.array_equal(back_src, src) and np.array_equal(back_ord, ord_)
        votes += int(is_vote.sum())
        fit += int((is_vote & ~stay).sum())
        words += len(reqs) + len(anss)
        assert (reqs["copies"] == R - 1).all()  # Q5: one broadcast per configured peer (config.nodes: the others)
        assert (anss["copies"] == R - 1).all()  # ... and one answer per copy
        assert (anss["rest"][anss["first"] == 1] == 0).all()  # after a granted vote every further copy is refused
        # a voter's answers go to ONE candidate per partition and round unless two campaign at once: count those
        if len(reqs):
            g, c = np.unique(reqs["group"], return_counts=True)
            two_addressees += int((c > 1).sum())
    assert votes > 20 * G // 10 and words * (R - 1) == fit
    assert fit == votes  # every VoteRequest / VoteResponse row of these traces is a whole run of R - 1 identical / first+rest copies
    # (two campaigns for one partition in one round: the voter half needs the row transport there - rare, but it happens)
    print(f"R={R}: {votes} vote rows = {words} words x {R - 1} copies; partitions with two campaigns in one round: {two_addressees}")
