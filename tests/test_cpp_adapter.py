"""Build and run the C++ host-adapter tests (tests/cpp/test_raft_handle.cpp) against
the HIP engine: the reference's L1 tests through josefine_amd/host/raft_handle.hpp and
BASELINE config #1 (3-node plumbing)."""
import os
import subprocess

import pytest

from josefine_amd.build import CSRC, build_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_raft_handle.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_raft_handle")


def compile_adapter_test():
    build_hip()
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-o", EXE, SRC, f"-L{CSRC}", "-ljosefine_gpu",
                    f"-Wl,-rpath,{CSRC}", "-Wl,-rpath,/opt/rocm/lib"], check=True)


def test_cpp_adapter_compiles():
    """CPU: the header-only adapter and its tests compile and link against the C ABI."""
    compile_adapter_test()
    assert os.path.exists(EXE)


def test_cpp_host_logic_on_oracle():
    """CPU: the host logic above the ABI (BatchedRaft pump, BlockStore, fsm::Driver and
    server::event_loop for many partitions) with the oracle library standing in for the engine:
    the reference's L1 tests, its event_loop test and the 3-node plumbing of BASELINE config #1."""
    odir = os.path.join(ROOT, "oracle")
    subprocess.run(["make", "-C", odir], check=True, capture_output=True)
    exe = EXE + "_oracle"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-DJG_TEST_AGAINST_ORACLE", "-o", exe, SRC, f"-L{odir}",
                    "-ljosefine_oracle", f"-Wl,-rpath,{odir}"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpp adapter ok" in r.stdout


def test_host_formats_kats():
    """CPU: josefine_amd/host/formats.hpp — the sled key / bincode value encoding of the chain store
    and the length-delimited serde_json peer protocol, against the
    reference's own vectors (chain.rs:345-350, tcp.rs:172-232) and one vector per Command."""
    src = os.path.join(ROOT, "tests", "cpp", "test_formats.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "test_formats")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-o", exe, src], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "formats ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_adapter_runs_reference_tests():
    compile_adapter_test()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpp adapter ok" in r.stdout


# ---- a whole cluster through BatchedEventLoop (the dense kernels behind the Apply surface) -----------
CL_SRC = os.path.join(ROOT, "tests", "cpp", "test_event_loop_cluster.cpp")
CL_EXE = os.path.join(ROOT, "tests", "cpp", "test_event_loop_cluster")


def build_cluster_test(oracle: bool) -> str:
    if oracle:
        odir = os.path.join(ROOT, "oracle")
        subprocess.run(["make", "-C", odir], check=True, capture_output=True)
        exe = CL_EXE + "_oracle"
        subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-DJG_TEST_AGAINST_ORACLE", "-o", exe, CL_SRC, f"-L{odir}",
                        "-ljosefine_oracle", f"-Wl,-rpath,{odir}"], check=True)
        return exe
    build_hip()
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-o", CL_EXE, CL_SRC, f"-L{CSRC}", "-ljosefine_gpu",
                    f"-Wl,-rpath,{CSRC}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return CL_EXE


def run_cluster(exe, *args, timeout=900, env=None):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, env={**os.environ, **(env or {})})
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("cluster ")][0]
    assert line.startswith("cluster ok"), line
    return line


def test_event_loop_cluster_host_logic_on_oracle():
    """CPU: R BatchedEventLoops over R oracle engines, wired through the host transport: the steady
    state after a scripted election (every row in column form) and an election by the timers alone
    (votes travel as rows through the general path)."""
    exe = build_cluster_test(oracle=True)
    line = run_cluster(exe, 2000, 5, 50, "scripted")
    assert "leaders=2000" in line and " rows_general=0 " in line and "max_head=50" in line
    line = run_cluster(exe, 500, 3, 80, "elect")
    assert "leaders=500" in line and "faults=0" in line and " rows_general=0 " not in line
    leadership_moved_and_stays_dense(line, 500)
    # FIVE nodes elect by their timers too: the host transport of this harness delivers a partition's mail with the senders
    # interleaved (emission index, sender) - the device transport's order since round 6; sender after sender no candidate of
    # five ever holds a quorum of grants (candidate.rs:30-37, election.rs:33-35)
    line = run_cluster(exe, 500, 5, 120, "elect")
    assert "leaders=500" in line and "faults=0" in line
    leadership_moved_and_stays_dense(line, 500)
    # ... and what the reference does when settled leaders crash and restart (Q4, follower.rs:249): their followers have
    # voted in the current term and never campaign, the restarted replica campaigns at term 1 and is refused - the
    # partitions stay leaderless (and fault-free) for good.  Restated here so that nobody "fixes" it on the device.
    line = run_cluster(exe, 500, 3, 150, "failover")
    assert "restarted_leaders=500" in line and "leaders=0 " in line and "faults=0" in line and "led_by_another_node_now=0" in line, line
    # ABI v7's bus formats (JG_NODE_COMMON_AE + JG_NODE_FSM_FUSED): the same protocol - every message, decision, leader -
    # with fewer fsm rows on the channel (a leader's Apply + Notify of a tick is one row)
    def fields(line):
        return dict(kv.split("=") for kv in line.split() if "=" in kv)
    for args in ((2000, 5, 50, "scripted"), (500, 3, 80, "elect")):
        plain, compact = fields(run_cluster(exe, *args)), fields(run_cluster(exe, *args, env={"JG_CLUSTER_COMPACT": "1"}))
        for k in ("leaders", "faults", "proposals", "msg_rows", "column_messages", "rows_in", "rows_general", "decisions", "max_head", "leaders_by_node"):
            assert plain[k] == compact[k], (args, k, plain[k], compact[k])
        assert int(compact["fsm_rows"]) < int(plain["fsm_rows"])
    # two ticks in flight (BatchedEventLoop::in_flight = 2 over the oracle's JG_NODE_KEEP): a step's outputs reach the
    # channels a step later - the same protocol, the same ends
    line = run_cluster(exe, 2000, 5, 50, "scripted", env={"JG_CLUSTER_PIPELINED": "1", "JG_CLUSTER_IN_FLIGHT": "2"})
    assert "leaders=2000" in line and " rows_general=0 " in line and "max_head=50" in line
    line = run_cluster(exe, 500, 3, 120, "elect", env={"JG_CLUSTER_PIPELINED": "1", "JG_CLUSTER_IN_FLIGHT": "2"})
    assert "leaders=500" in line and "faults=0" in line
    leadership_moved_and_stays_dense(line, 500)
    build_cluster_test(oracle=False)  # (links against the C ABI: compile check without a GPU)


def leadership_moved_and_stays_dense(line, G):
    """Elect mode (R = 3, configs[3]'s replica count): the timers put the leaders on EVERY node (per-partition
    leadership, not one lead node), and in the last quarter of the run every partition keeps committing under
    the leader it elected with no row on the general path: AppendEntries / Heartbeat / the answers all travel in
    column form through the dense halves of jg_step_node (candidate.rs:108-113 -> leader.rs:124-174)."""
    f = dict(kv.split("=") for kv in line.split() if "=" in kv)
    by_node = [int(x) for x in f["leaders_by_node"].split("/")]
    assert sum(by_node) == G and all(n > 0 for n in by_node), line
    assert int(f["tail_ticks"]) >= 10 and int(f["tail_rows_general"]) == 0 and int(f["tail_partitions_committing"]) == G, line


@pytest.mark.gpu
@pytest.mark.parametrize("args", [(100_000, 5, 50, "scripted"), (3000, 3, 80, "elect"), (20_000, 3, 40, "scripted"),
                                  (1000, 3, 90, "failover"), (1000, 5, 160, "elect")])
def test_event_loop_cluster_equals_the_oracle_backed_loops(args):
    """VERDICT r2 #2 'Done': 100 k x 5 partitions for 50 ticks through BatchedEventLoop - five loops, five
    engines, every message between them through the host - with EVERY rpc_tx / fsm_tx row and outbox word
    equal to the same loops over the oracle library (one hash over all of them, tick by tick, plus the final
    state columns), and an election by timers at R = 3 the same way."""
    dev = run_cluster(build_cluster_test(oracle=False), *args)
    ora = run_cluster(build_cluster_test(oracle=True), *args)
    assert dev == ora, (dev, ora)
    if args[3] == "scripted":
        assert f"leaders={args[0]}" in dev and " rows_general=0 " in dev and f"max_head={args[2]}" in dev
    elif args[3] == "elect":
        leadership_moved_and_stays_dense(dev, args[0])
    else:  # failover: the reference's dead end (Q4), identically on both
        assert f"restarted_leaders={args[0]}" in dev and "leaders=0 " in dev and "faults=0" in dev


@pytest.mark.gpu
@pytest.mark.parametrize("in_flight", [1, 2])
@pytest.mark.parametrize("args", [(50_000, 5, 40, "scripted"), (3000, 3, 80, "elect")])
def test_pipelined_event_loops_equal_the_oracle_backed_ones(args, in_flight):
    """BatchedEventLoop::pipelined (ONE loop that overlaps with itself: a step returns once its rows are on the device and
    classified, its outputs are delivered at the start of the next step) - the same cluster of loops, every rpc_tx / fsm_tx
    row and outbox word equal to the pipelined loops over the oracle library; and the run still does what the synchronous
    one does (leaders elected, every partition committing).  in_flight = 2 (BatchedEventLoop::in_flight, JG_NODE_KEEP): TWO
    ticks in flight - a step's outputs are delivered at the start of the step after the next; the elections of the `elect`
    run put rows on the general path, exceptional rows and the late fetch of the AppendEntries rows through the kept steps."""
    env = {"JG_CLUSTER_PIPELINED": "1", "JG_CLUSTER_IN_FLIGHT": str(in_flight)}
    if in_flight == 2 and args[3] == "elect":
        args = (args[0], args[1], 120, args[3])  # (every hop takes a tick longer: so do the elections)
    dev = run_cluster(build_cluster_test(oracle=False), *args, env=env)
    ora = run_cluster(build_cluster_test(oracle=True), *args, env=env)
    assert dev == ora, (dev, ora)
    assert f"leaders={args[0]}" in dev and "faults=0" in dev


@pytest.mark.gpu
@pytest.mark.parametrize("args,pipelined", [((50_000, 5, 40, "scripted"), False), ((3000, 3, 80, "elect"), False), ((3000, 3, 80, "elect"), True),
                                            ((3000, 3, 80, "elect"), 2), ((20_000, 5, 40, "scripted"), 2)])
def test_compact_bus_event_loops_equal_the_oracle_backed_ones(args, pipelined):
    """ABI v7's bus formats through the loops (BatchedEventLoop::bus = JG_NODE_COMMON_AE | JG_NODE_FSM_FUSED): the Tick's
    AppendEntries words as one word per partition where the followers' agree, a leader's fsm_tx rows of a step as one row -
    every fsm row as it is, every outbox word it stands for and every rpc_tx row equal to the same loops over the oracle
    library, which restates the formats itself (oracle/oracle_engine.cpp)."""
    env = {"JG_CLUSTER_COMPACT": "1", **({"JG_CLUSTER_PIPELINED": "1"} if pipelined else {}), **({"JG_CLUSTER_IN_FLIGHT": "2"} if pipelined == 2 else {})}
    if pipelined == 2 and args[3] == "elect":
        args = (args[0], args[1], 120, args[3])  # (two ticks in flight: every hop takes a tick longer, so do the elections)
    dev = run_cluster(build_cluster_test(oracle=False), *args, env=env)
    ora = run_cluster(build_cluster_test(oracle=True), *args, env=env)
    assert dev == ora, (dev, ora)
    assert f"leaders={args[0]}" in dev and "faults=0" in dev
