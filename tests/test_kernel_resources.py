"""The dense kernels' register budget, from the compiler's own resource report for the gfx950 code object (no GPU: hipcc
cross-compiles).  The bandwidth-bound halves live on their occupancy - capping the closed loop's halves at 3 / 4 waves per
SIMD costs 10 us per round (profiles/r04/ab_dense_occupancy.txt) - and a register that creeps in shows up nowhere but
here and in the next profile."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel (as c++filt prints it, cut at the argument list) -> (max VGPRs, min waves/SIMD); no scratch in any of them
BUDGET = {
    "void k_leader_tick_dense<5, false>": (64, 8),   # the headline
    "void k_leader_tick_dense<5, true>": (64, 8),
    "void k_leader_tick_dense<3, false>": (64, 8),
    "void k_leader_node_tick<5, false>": (72, 7),    # the closed loop's leader half
    "void k_leader_node_tick<3, false>": (64, 7),
    "k_follower_tick_dense_multi": (72, 7),          # ... and its follower halves
    "k_follower_tick_dense": (72, 7),
    "k_follower_tick_dense_any": (72, 7),            # per-partition leadership
    "void k_leader_node_tick_any<3>": (64, 7),
    "void k_leader_node_tick_any<5>": (80, 6),
    "k_cluster_claim": (32, 8),
    "k_vote_half_multi": (128, 4),                   # the vote mail's receiving half: no scratch (its jobs are kernel arguments)
    "k_round_head_multi": (168, 3),                  # ... beside the delivered rows' step, one launch: the general state machine's registers, no scratch
    "k_votes_census_multi": (40, 8),                 # the census of a round's emissions, one launch
    "k_route_deliver_multi_words": (80, 3),          # the delivering pass, one launch (its LDS staging bounds the waves: 49 KB)
    "k_node_classify": (40, 7),                      # jg_step_node's row passes
    "k_node_route": (40, 7),
}


@pytest.fixture(scope="module")
def report():
    r = subprocess.run(["bash", os.path.join(ROOT, "profiles", "kernel_resources.sh")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("#") or ln.startswith("kernel,"):
            continue
        f = [x.strip() for x in ln.rsplit(",", 6)]
        if len(f) == 7 and f[1].isdigit():
            rows[f[0]] = dict(vgprs=int(f[1]), scratch=int(f[4]), waves=int(f[6]))
    assert len(rows) > 60 and not any("rocprim" in k for k in rows), r.stdout[:2000]
    return rows


def test_dense_kernels_keep_their_register_budget(report):
    for k, (max_vgprs, min_waves) in BUDGET.items():
        assert k in report, f"{k} is not in the code object any more: {sorted(report)[:5]} ..."
        got = report[k]
        assert got["scratch"] == 0, (k, got)
        assert got["vgprs"] <= max_vgprs and got["waves"] >= min_waves, (k, got, (max_vgprs, min_waves))


def test_the_committed_report_is_the_code_objects(report):
    """profiles/r06/kernel_resources.txt is what DESIGN.md quotes: it must be this source's"""
    path = os.path.join(ROOT, "profiles", "r06", "kernel_resources.txt")
    want = {}
    for ln in open(path):
        f = [x.strip() for x in ln.rsplit(",", 6)]
        if len(f) == 7 and f[1].isdigit():
            want[f[0]] = (int(f[1]), int(f[4]), int(f[6]))
    got = {k: (v["vgprs"], v["scratch"], v["waves"]) for k, v in report.items()}
    diff = {k: (want.get(k), got.get(k)) for k in set(want) | set(got) if want.get(k) != got.get(k)}
    assert not diff, f"re-run `bash profiles/kernel_resources.sh > profiles/r06/kernel_resources.txt`: {dict(list(diff.items())[:6])}"
