"""bench.py's contract with the driver: one JSON line on rank 0 with the agreed keys, the
`roofline` and `cpu_baseline` objects, whole-job aggregation for N > 1 (two ranks on the one
GPU of the box, gloo for the barrier / reductions), and the secondary modes."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config"}


def run(args, env=None, launcher=None):
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env={**os.environ, **(env or {})})
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_default_shape_small():
    d = run(["--groups", "50000", "--steps", "40", "--warmup", "5", "--cpu-budget", "1"])
    assert KEYS <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["unit"] == "decisions/s" and d["dtype"] == "u64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "polled" in d["config"]["host_wait"] or "interrupt" in d["config"]["host_wait"]  # (how the host waits is on the line)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0
    assert r["alg_bytes_per_group_step"] == 8 * 5 + 28 and r["survey_priced"]["bytes_per_group_step"] == 156
    sc = r["stream_ceiling"]  # the same bytes through a kernel with no logic, measured in the same run
    assert sc["avg_launch_us"] > 0 and abs(sc["kernel_vs_ceiling"] - sc["avg_launch_us"] / r["avg_launch_us"]) < 1e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    # value = decisions of all ranks / wall time: 5 decisions per group-step in the steady-state stream
    assert abs(d["value"] / d["group_steps_per_s"] - 5.0) < 1e-6
    assert d["batched_ticks"]["ticks_per_launch"] == 16


def test_two_ranks_aggregate():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(port)]
    d = run(["--gpus", "2", "--groups", "50000", "--steps", "30", "--warmup", "5", "--cpu-budget", "1"], env={"JG_BENCH_BACKEND": "gloo"},
            launcher=launcher)
    assert d["n_gpus"] == 2 and d["config"]["partitions_total"] == 100000
    assert abs(d["value"] / d["group_steps_per_s"] - 5.0) < 1e-6
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0  # rank 0 measures it for every N
    pr = d["per_rank"]
    assert len(pr["decisions_per_s"]) == 2 and len(pr["avg_launch_us"]) == 2 and min(pr["decisions_per_s"]) > 0
    # a short run: the timed region is repeated and the median reported (steps keeps its meaning: one region)
    tr = d["timed_regions"]
    assert tr["n"] == 9 and tr["reported"] == "median" and len(tr["ms_per_step_each"]) == 9
    assert sorted(tr["ms_per_step_each"])[4] == pytest.approx(d["ms_per_step"])


def launch_two(args, timeout=900, n=2):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_scaling_presets_two_ranks():
    """--config 3 / 4: BASELINE.json configs[3] and configs[4] as the scaling run sees them (two ranks, aliased on a
    one-GPU box): one GPU's share each, per-rank rates on the line."""
    d = launch_two(["--config", "3", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["config"]["partitions_per_gpu"] == 1_250_000 and d["config"]["replicas"] == 3
    assert d["config"]["partitions_total"] == 2_500_000 and len(d["per_rank"]["decisions_per_s"]) == 2
    assert abs(d["value"] / d["group_steps_per_s"] - 3.0) < 1e-6
    d = launch_two(["--config", "4", "--steps", "12", "--warmup", "4", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["config"]["partitions_per_gpu"] == 125_000 and d["config"]["replicas"] == 5
    assert "configs[4]" in d["config"]["workload"] and len(d["per_rank"]["decisions_per_s"]) == 2
    d = launch_two(["--config", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-secondary"])
    assert d["config"]["partitions_per_gpu"] == 1_000_000 and d["config"]["replicas"] == 5


def test_scale_run_rehearsal_eight_ranks():
    """The driver's scaling command at N = 8 - `python bench.py --gpus 8 --config {2, 3, 4}` - on whatever box this is (eight
    ranks ALIASED onto its devices, barrier and reductions over gloo): the plumbing of the first real 8-GPU lease. Eight
    per-rank entries, the aggregate is their sum over the slowest rank's clock, rank 0 brings the CPU baseline, the line says
    which devices the ranks bound and that they were shared."""
    import torch
    aliased = torch.cuda.device_count() < 8
    for cfg, groups, R, extra in ((2, 1_000_000, 5, ["--cpu-budget", "1"]), (3, 1_250_000, 3, ["--no-cpu-baseline"]), (4, 125_000, 5, ["--no-cpu-baseline"])):
        d = launch_two(["--config", str(cfg), "--steps", "12", "--warmup", "4", "--no-secondary"] + extra, n=8, timeout=1200)
        assert KEYS <= set(d) and d["n_gpus"] == 8 and d["steps"] == 12 and d["scaling"] == "weak", cfg
        c = d["config"]
        assert c["partitions_per_gpu"] == groups and c["replicas"] == R and c["partitions_total"] == 8 * groups, cfg
        assert len(c["devices"]) == 8 and c["devices_aliased"] == aliased, c
        pr = d["per_rank"]
        assert len(pr["decisions_per_s"]) == 8 and len(pr["avg_launch_us"]) == 8 and min(pr["decisions_per_s"]) > 0, cfg
        # the aggregate: all ranks' decisions over the slowest rank's time - between 8 x the slowest and the sum of the per-rank rates
        assert 8 * min(pr["decisions_per_s"]) * 0.98 <= d["value"] <= sum(pr["decisions_per_s"]) * 1.02, (cfg, d["value"], pr)
        if cfg == 2:
            assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and "roofline" in d
        if cfg == 4:
            assert "configs[4]" in c["workload"] and c["stationary"].startswith("yes")


def test_default_line_carries_the_secondaries():
    """The driver's N = 1 line also carries the other modes (short sub-runs of bench.py itself), each with its own roofline
    fraction - so that BENCH_rNN.json holds more than the headline."""
    d = run(["--groups", "100000", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"], env={"JG_BENCH_SECONDARY": "1"})
    sec = d["secondary"]
    assert set(sec) == {"closed_loop", "routed_round", "routed_round_rows_only", "routed_round_no_repairs", "per_partition_leadership",
                        "per_partition_leadership_failures", "per_partition_leadership_five_nodes", "failures_tick", "event_loop"}
    # round 6: no synthetic vote anywhere - configs[4]'s re-created partitions (R = 5) elect their leaders through the transport,
    # and so do the five nodes of the per-partition-leadership cluster
    assert sec["routed_round"]["elections_won_through_the_transport"] > 0 and sec["routed_round"]["synthetic_votes"] == 0
    assert sec["per_partition_leadership_five_nodes"]["elections"]["won_through_the_transport"] is True
    ppf = sec["per_partition_leadership_failures"]  # re-created groups: every election won through the transport, the winners append
    assert ppf["rows_left_for_the_host"] == 0 and ppf["stationary"] == "yes" and ppf["winners_appending_again_fraction_of_failed_groups"] > 0.8
    assert ppf["elections_won_after_failures"] > 0
    for k, v in sec.items():
        assert "error" not in v, (k, v)
    assert sec["closed_loop"]["round_us"] > 0 and 0 < sec["closed_loop"]["frac"] < 1
    assert sec["routed_round"]["round_ms"] > 0 and sec["per_partition_leadership"]["elections"]["won_through_the_transport"] is True
    assert sec["routed_round"]["vote_words"] is True and sec["routed_round"]["stationary"] == "yes" and sec["routed_round_rows_only"]["vote_words"] is False
    lf = sec["routed_round"]["leaderless_fraction"]
    assert abs(lf["at_start_of_timed_region"] - lf["at_end"]) < 0.02 and sec["routed_round_no_repairs"]["stationary"] == "no"
    assert sec["event_loop"]["decisions_per_s"] > 0 and sec["event_loop"]["rows_on_the_general_path"] == 0
    assert sec["failures_tick"]["tick_ms"] > 0


def test_plain_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's scaling command
    when it does not wrap it): bench.py re-executes itself under torch.distributed.run; the line says
    n_gpus 2, twice the partitions, and which devices the ranks bound (aliased on this 1-GPU box)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--groups", "50000", "--steps", "30", "--warmup", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["partitions_total"] == 100000 and d["config"]["partitions_per_gpu"] == 50000
    assert len(d["config"]["devices"]) == 2
    import torch
    assert d["config"]["devices_aliased"] == (torch.cuda.device_count() < 2)
    assert abs(d["value"] / d["group_steps_per_s"] - 5.0) < 1e-6
    one = run(["--groups", "50000", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"])
    assert one["n_gpus"] == 1 and one["config"]["devices"] == [0] and one["config"]["devices_aliased"] is False
    if not d["config"]["devices_aliased"]:  # distinct devices: the job is two GPUs' worth of decisions per second
        assert 1.5 < d["value"] / one["value"] < 2.5


def test_world_size_must_equal_gpus():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--groups", "1000", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env={**os.environ, "JG_BENCH_BACKEND": "gloo"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stdout + r.stderr


def test_cluster_and_failure_modes():
    d = run(["--cluster", "--groups", "30000", "--steps", "20", "--warmup", "5"])
    assert KEYS <= set(d) and "closed loop" in d["config"]["workload"] and d["value"] > 0
    d = run(["--failures", "1", "--groups", "100000", "--steps", "24", "--warmup", "8", "--no-cpu-baseline"])
    assert KEYS <= set(d) and d["value"] > 0
    r = d["roofline"]  # the dominant kernel under failures, priced with its own HIP event pairs
    assert r["bound"] == "hbm" and r["avg_launch_us"] > 0 and r["launches_timed"] == 24 // 4  # (every 4th dense launch is timed)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["rows_delivered_to_host"]["messages"] > 0 and d["rows_delivered_to_host"]["faults"] > 0
    # configs[4] as specified: real votes, routed between the nodes on the device
    d = run(["--cluster", "--failures", "2", "--groups", "40000", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    assert KEYS <= set(d) and "configs[4] as specified" in d["config"]["workload"] and d["value"] > 0
    lf = d["leaderless_fraction"]  # the stationary trace (failures + re-creation after --repair-after rounds): flat at about p x (D + 1)
    assert d["rows_routed_per_round"] > 0 and 0.1 < lf["at_start_of_timed_region"] < 0.3 and abs(lf["at_start_of_timed_region"] - lf["at_end"]) < 0.02
    assert d["config"]["stationary"].startswith("yes") and d["config"]["repair_after_rounds"] == 10 and d["steps"] == 20 and d["warmup"] == 5
    ms = [w["ms_per_round"] for w in d["ms_per_round_by_leaderless_fraction"]]
    assert len(ms) == 4 and min(ms) > 0
    # ... and without repairs (rounds 2-4's trace), with the election vocabulary as mailbox words: the leaderless fraction grows
    d = run(["--cluster", "--failures", "2", "--groups", "40000", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--repair-after", "0", "--vote-words", "1"])
    lf = d["leaderless_fraction"]
    assert d["vote_words"] is True and d["config"]["stationary"].startswith("no") and 0 < lf["at_start_of_timed_region"] < lf["at_end"] < 1
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["leader_kernel"]["avg_launch_us"] > 0


def test_single_process_multi_device_agrees_with_process_per_gpu():
    """bench.py --single-process: one engine handle over N shards (aliased onto the one GPU here).
    Same JSON contract; at N = 1 it is the classic path's workload and its value must be in the same
    range; at N = 3 (aliased) the job is three times the groups."""
    a = run(["--groups", "200000", "--steps", "60", "--warmup", "10", "--no-cpu-baseline"])
    b = run(["--single-process", "--gpus", "1", "--groups", "200000", "--steps", "60", "--warmup", "10", "--no-cpu-baseline"])
    assert KEYS <= set(b) and b["n_gpus"] == 1
    assert abs(b["value"] / b["group_steps_per_s"] - 5.0) < 1e-6
    assert 0.6 < b["roofline"]["avg_launch_us"] / a["roofline"]["avg_launch_us"] < 1.6, (a["roofline"], b["roofline"])
    c = run(["--single-process", "--alias-devices", "--gpus", "3", "--groups", "100000", "--steps", "40", "--warmup", "10",
             "--no-cpu-baseline"])
    assert c["n_gpus"] == 3 and c["config"]["partitions_total"] == 300000 and "3 shard(s)" in c["config"]["parallelism"]
    assert abs(c["value"] / c["group_steps_per_s"] - 5.0) < 1e-6


def test_event_loop_line():
    """bench.py --event-loop: decisions/s through josefine::BatchedEventLoop over jg_step_node, with the PCIe
    bytes per tick, the A/B against round 2's loop (one Tick row per partition through the general state
    machine) and the roofline of the dense leader half as timed inside the loop."""
    d = run(["--event-loop", "--groups", "20000", "--loops", "4", "--steps", "20", "--warmup", "5", "--cpu-budget", "1"])
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["value"] > 0 and "Apply surface" in d["config"]["workload"]
    ev = d["event_loop"]
    assert ev["rows_on_the_general_path"] == 0 and ev["rows_in_per_tick"] > 20000 * 5 and ev["fsm_rows_per_tick"] == 2 * 20000
    assert ev["pcie_bytes_per_tick"]["h2d"] > 0 and ev["pcie_bytes_per_tick"]["d2h"] > 0
    assert ev["speedup_over_round2_loop"] > 2, ev
    assert ev["loops"] == 4 and d["config"]["loops"] == 4 and "4 event loop(s) of 5000 partitions" in d["config"]["parallelism"]
    assert ev["loop_only_decisions_per_s"] >= ev["one_loop"]["decisions_per_s"] > 0
    assert ev["column_inbound"]["loops"] == 4 and ev["column_inbound"]["one_loop_decisions_per_s"] > 0
    tk = ev["one_loop_with_transport_and_consumer_tasks"]  # (the binary checks the closed form of the stream in every mode)
    assert tk["task_threads_beside_the_loop"] == 4 and tk["decisions_per_s"] > 0 and tk["column_inbound_decisions_per_s"] > 0
    assert tk["rows_on_the_general_path"] == 0
    wd = tk["wire_decode"]  # the peers' traffic as length-delimited serde_json frames through host/formats.hpp's decoder
    assert wd["decisions_per_s"] > 0 and wd["rows_on_the_general_path"] == 0 and wd["frames_per_tick"] >= 20000 * 4 and wd["wire_bytes_decoded_per_tick"] > 100 * 20000 * 4
    assert tk["host_wait"]["interrupt_decisions_per_s"] > 0 and tk["host_wait"]["polled_decisions_per_s"] > 0
    cb = tk["compact_bus"]  # ABI v7's formats: packed kind byte, common AppendEntries word, fused fsm row - the same stream, fewer bytes
    assert cb["decisions_per_s"] > 0 and cb["rows_on_the_general_path"] == 0 and cb["fsm_rows_per_tick"] == 20000 == cb["fsm_rows_per_tick_plain"] / 2
    assert cb["pcie_bytes_per_decision"] <= 32 < cb["pcie_bytes_per_decision_plain"], cb
    assert cb["column_inbound_pcie_bytes_per_decision"] < cb["pcie_bytes_per_decision"]
    two = cb["two_ticks_in_flight"]  # JG_NODE_KEEP (ABI v9): the same stream (the binary checks its closed form), the same bytes, two ticks in flight
    assert two["decisions_per_s"] > 0 and two["polled_decisions_per_s"] > 0 and two["column_inbound_decisions_per_s"] > 0 and two["plain_bus_decisions_per_s"] > 0
    assert two["rows_on_the_general_path"] == 0 and abs(two["pcie_bytes_per_decision"] - cb["pcie_bytes_per_decision"]) < 1.0, two
    assert two["with_eight_task_threads"]["decisions_per_s"] > 0 and two["with_eight_task_threads"]["one_in_flight_decisions_per_s"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["avg_launch_us"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["cpu_baseline"]["kind"] == "port"
