"""bench.py's pure parts, without a GPU: the bytes the roofline is priced with, and the rule that a PMC traffic figure
read from profiles/traffic.json is REFUSED once the kernel source it was collected on has changed."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench():
    """bench.py as a module - imported HERE, not when the file is collected: importing it decides how the runtime waits
    (HSA_ENABLE_INTERRUPT), which is nobody's business in a test process that only collects this file"""
    keep = {k: os.environ.get(k) for k in ("HSA_ENABLE_INTERRUPT", "JG_BENCH_POLLING_DEFAULTED")}
    sys.path.insert(0, ROOT)
    import bench as b
    yield b
    for k, v in keep.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_priced_bytes(bench):
    assert bench.alg_bytes_per_group_step(5) == 68 and bench.alg_bytes_per_group_step(3) == 52
    assert bench.alg_bytes_per_group_step(5, mode=1) == 80
    assert bench.survey_bytes_per_group_step(5) == 156  # SURVEY.md 8(d)'s B(R), reported beside the roofline


def test_traffic_is_refused_when_the_kernel_source_changed(bench, monkeypatch):
    key = "G1000000_R5_mode0"
    data = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    t = bench.pmc_traffic(key)
    assert t["source"] == "profiles/traffic.json" and t["collected_from"].startswith("profiles/")
    # as committed: either collected on this very source (bytes on the line) or refused - never a stale figure
    if t["kernel_unchanged_since_collection"]:
        assert t["bytes"] == data[key] and "stale_bytes" not in t
        assert 0.98 * 68e6 < t["bytes"] < 1.05 * 68e6  # (PMC: 1.009 x the bytes priced)
    else:
        assert t["bytes"] is None and t["stale_bytes"] == data[key]
    monkeypatch.setattr(bench, "kernel_source_sha", lambda name="jg_dense.h": "0" * 16)
    t = bench.pmc_traffic(key)
    assert t["bytes"] is None and t["stale_bytes"] == data[key] and t["kernel_unchanged_since_collection"] is False
    assert "re-run" in t["note"]
    assert bench.pmc_traffic("no such workload") is None


def test_the_host_wait_is_on_every_line(bench):
    class A:
        devices_bound = [0]
    c = bench.devices_config(A(), 1)
    assert c["devices"] == [0] and c["devices_aliased"] is False
    assert ("polled" in c["host_wait"]) == (os.environ.get("HSA_ENABLE_INTERRUPT") == "0")
