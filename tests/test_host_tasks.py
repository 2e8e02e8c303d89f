"""The fork-join pool of josefine_amd/host/bench_event_loop.cpp (the decoder / consumer tasks beside a pipelined event
loop): every job of every run exactly once - runs without a GPU."""
import json
import subprocess

from josefine_amd.build import build_event_loop_bench


def test_task_pool_runs_every_job_once():
    exe = build_event_loop_bench()
    r = subprocess.run([exe, "tasks-selftest"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert json.loads(r.stdout.strip())["ok"] is True
