"""Soak of the vote mail on the device (not collected by pytest: run it on a GPU box):
    python tests/soak_vote_words.py [first_seed] [n_seeds]
tests/test_gpu_vote_words.py's chaotic cases - 25-30 %/round of the partitions fail, up to three more replicas restart with
the leader, R = 3, 4, 5 - and the stationary trace, over OTHER seeds than the suite's: the routed round with
JG_CLUSTER_OPT_VOTE_WORDS against the oracle cluster that moves every message as a row, every column of every node after
every round, the rows kept, faults, applies and decisions at the end."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from josefine_amd import BatchedRaft, DenseCluster as LibCluster, capi  # noqa: E402
from josefine_amd.traces import FailureRepairTrace  # noqa: E402
from dense_node import RoutedCluster, cluster_failure_rows  # noqa: E402
from oracle_lib import oracle_engine  # noqa: E402
from parity import compare_snapshots, elect_all  # noqa: E402


def one(seed, R, percent, also, G, T, stationary=False):
    ora = RoutedCluster(oracle_engine, G, R, seed=seed)
    nodes = [BatchedRaft(G, R, seed=seed + r, self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY) for r in range(R)]
    elect_all(nodes[0])
    nodes[0].drain_messages(), nodes[0].drain_applies()
    lib = LibCluster(nodes, vote_words=True)
    lib.set_appends(1)
    tr = FailureRepairTrace(seed * 7 + 1, G, R, percent, 5, node_ids=ora.member_ids) if stationary else None
    moved = as_rows = 0
    for t in range(T):
        lists = []
        if stationary:
            inj, failing, repaired = tr.rows(t)
            appends = tr.appends()
            lists = [nodes[0].upload_u32(x) if len(x) else None for x in (failing, repaired)]
            if lists[0] is not None:
                lib.withdraw_appends(lists[0].ptr, len(failing))
            if lists[1] is not None:
                lib.offer_appends(lists[1].ptr, len(repaired), 1)
        else:
            inj = cluster_failure_rows(seed * 13 + 5, t, G, R, percent, also=also) if t >= 3 else [None] * R
            appends = np.ones(G, np.uint64)
        up = [None if c is None else nodes[n].upload_rows(**c) for n, c in enumerate(inj)]
        st = lib.round_routed((t + 1) * 100, up)
        ora.round(appends, inject=inj)
        for n in range(R):
            compare_snapshots(nodes[n], ora.nodes[n], f"seed {seed} round {t} node {n}")
        want = [ora.pending(n) for n in range(R)]
        assert all(a <= b for a, b in zip(st["delivered"], want)), (seed, t, st["delivered"], want)
        as_rows += sum(st["delivered"])
        moved += sum(want)
        for rows in up + lists:
            if rows is not None:
                rows.free()
    for n in range(R):
        assert nodes[n].drain_messages().tobytes() == ora.kept[n].tobytes(), (seed, n)
        assert nodes[n].drain_faults().tobytes() == ora.nodes[n].drain_faults().tobytes(), (seed, n)
        assert nodes[n].drain_applies().tobytes() == ora.nodes[n].drain_applies().tobytes(), (seed, n)
        assert nodes[n].counters()["decisions"] == ora.nodes[n].counters()["decisions"], (seed, n)
    lib.close()
    return moved, as_rows


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cases = [(3, 25, (1, 2), 1500, 50, False), (5, 25, (1, 2, 3), 1500, 50, False), (4, 30, (1, 2), 1500, 50, False), (5, 2, (), 2500, 50, False),
             (5, 3, (), 2500, 60, True), (3, 4, (), 2500, 60, True)]
    total = 0
    for seed in range(first, first + count):
        for R, percent, also, G, T, stationary in cases:
            moved, as_rows = one(seed, R, percent, also, G, T, stationary)
            total += 1
            print(f"seed {seed} R {R} {percent} % {also} {'stationary' if stationary else ''}: ok, {moved} messages, {as_rows} as rows", flush=True)
    print(f"soak ok: {total} clusters")
