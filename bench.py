#!/usr/bin/env python3
"""bench.py — Raft quorum decisions/sec on the dense leader-tick hot path.

Workload (N=1): BASELINE.json configs[2] — 1M partitions x 5 replicas,
steady-state append + commit: every tick each leader appends one block and
receives the 4 follower acks of the previous tick.  The
synthetic AppendEntries-ack stream for all warm-up + timed ticks is generated
on the device BEFORE the timed region, so inputs are resident in HBM.  For
N>1 (one process per GPU: launched by torch.distributed.run, or - when plain
`python bench.py --gpus N` is run without a launcher - by bench.py re-executing
itself under torch.distributed.run; WORLD_SIZE must equal --gpus either way)
every rank owns its own 1M x 5 shard (weak scaling, contiguous global group ids, no data-path
collective: Raft groups are independent — SURVEY.md §8(e)).

A "step" = one tick = one launch of k_leader_tick_dense<5> over the rank's
groups.  Timing: barrier + synchronize, then exactly K steps; a rank's clock stops
when its own K steps are complete (HIP event + synchronize), the closing barrier
follows, and the MAX over ranks is what is reported (the latency of the barrier
collective itself is not part of anybody's K steps).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# How the host learns that the device is done: the ROCm runtime can take an interrupt per completion signal (its
# default) or poll the signal.  A timed region of K steps ends with one such wait, and at K = 20 the interrupt's
# wake-up is 3 % of the region (profiles/r04/ab_hsa_interrupt.txt: 3.90 -> 4.01 x 10^11 decisions/s on the driver's
# command; nothing changes for the long regions).  The engine's thread polls unless the caller says otherwise; the
# line says which (config.host_wait).  Set before anything loads the runtime.
if "HSA_ENABLE_INTERRUPT" not in os.environ:
    os.environ["HSA_ENABLE_INTERRUPT"] = "0"
    os.environ["JG_BENCH_POLLING_DEFAULTED"] = "1"

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def alg_bytes_per_group_step(R: int, mode: int = 0) -> int:
    """Bytes the dense leader tick has to move per group-step with the engine's state layout
    (DESIGN.md "k_leader_tick_dense"): read R ack heads (8 B each) + the packed progress / commit
    word 8 + head 8 + flags 4; write head 8  =>  B*(R) = 8R + 28 (68 B at R = 5).
    The ragged stream (mode 1: drops, duplicates, 0-2 appends) changes the lags and the
    Probe / Replicate bits of nearly every group every tick, so the packed word (8 B) and the flag
    word (4 B) have to be written back as well: B*_ragged(R) = 8R + 40 (80 B at R = 5; PMC: 80.5 MB
    per 1 M x 5 launch, profiles/traffic.json)."""
    return 8 * R + 28 + (12 if mode == 1 else 0)


def survey_bytes_per_group_step(R: int) -> int:
    """SURVEY.md §8(d)'s B(R): every progress head an 8-byte absolute value, read and written each
    tick: read R ack heads + R match heads + commit + head + term (8 B each) + 4 B flags; write R
    match heads + commit = 24R + 36 (156 B at R = 5).  Reported beside the roofline for continuity."""
    return 24 * R + 36


def kernel_source_sha(name: str = "jg_dense.h") -> str:
    """sha256 (first 16 hex digits) of the header that holds the dense kernels: what a PMC traffic figure in
    profiles/traffic.json is valid for."""
    import hashlib
    with open(os.path.join(ROOT, "josefine_amd", "csrc", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def pmc_traffic(key: str):
    """roofline.traffic: the HBM bytes per launch that rocprofv3's PMC passes measured for this workload
    (profiles/traffic.json, written by profiles/update_traffic.py from the committed counter files), together
    with where it comes from - and REFUSED (bytes = null) when the kernel source has changed since the
    counters were collected: a constant read from a file is not a measurement of this run."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None
    with open(tpath) as f:
        data = json.load(f)
    v = data.get(key)
    if v is None:
        return None
    meta = data.get("_collected", {}).get(key, {})
    now = kernel_source_sha()
    same = meta.get("kernel_sha") == now
    out = {"bytes": v if same else None, "source": "profiles/traffic.json", "collected_from": meta.get("from"),
           "kernel_sha_at_collection": meta.get("kernel_sha"), "kernel_sha_now": now, "kernel_unchanged_since_collection": same}
    if not same:
        out["stale_bytes"] = v
        out["note"] = "jg_dense.h changed since these counters were collected: re-run profiles/collect_round.sh + update_traffic.py"
    return out


def effective_cores() -> int:
    """Host cores this process may actually use: the scheduler affinity mask and the cgroup CPU
    quota (the GPU box advertises 256 hardware threads and grants 16 CPUs' worth of time)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(R: int, seed: int, budget_s: float):
    """Time the CPU oracle (the reference-shaped C++ port: per ack HashMap remove/insert +
    Vec sort, progress.rs:42-60) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from oracle_lib import oracle_engine
    from josefine_amd.traces import elect_all
    from parity import synth_tick_host

    # ~10-15 s of single-core work: 100k groups x R replicas, as many ticks of the same
    # steady-state stream as fit the budget (ack blocks are generated outside the timed region)
    Gs, max_ticks = 100_000, 400
    ora = oracle_engine(Gs, R, seed=seed)
    elect_all(ora)
    sim = np.zeros((R, Gs), dtype=np.uint64)
    d0 = ora.counters()["decisions"]
    dt, done = 0.0, 0
    while done < max_ticks and dt < budget_s:
        acks = synth_tick_host(ora, 0, done, sim)
        t0 = time.perf_counter()
        ora.step_dense_acks(acks)
        dt += time.perf_counter() - t0
        done += 1
    dec = ora.counters()["decisions"] - d0
    # The same stream on all host cores: one persistent Python thread per core (ctypes releases the
    # GIL for the whole tick), each with a PRIVATE engine over its own 10 k groups - Raft groups are
    # independent, so the threads never meet: no per-tick barrier, no thread creation in the timed
    # region.  Every thread applies the same number of ticks; the clock is the slowest thread's.
    cores = effective_cores()
    mt = None
    if cores > 1:
        import threading
        Gt = 10_000
        Tt = max(4, min(40, int(4.0 * (dec / dt) / (Gt * R))))  # ~4 s per thread at single-core speed
        start = threading.Barrier(cores + 1)
        res = [None] * cores

        def worker(i):
            e = oracle_engine(Gt, R, seed=seed, group_base=i * Gt)
            elect_all(e)
            acks = np.zeros((R, Gt), dtype=np.uint64)
            acks[0] = 1
            d_0 = e.counters()["decisions"]
            start.wait()
            t_0 = time.perf_counter()
            for t in range(Tt):  # mode 0 closed form: one append, every follower acks the previous head
                acks[1:] = t
                e.step_dense_acks(acks)
            res[i] = (time.perf_counter() - t_0, e.counters()["decisions"] - d_0)

        th = [threading.Thread(target=worker, args=(i,)) for i in range(cores)]
        for t in th:
            t.start()
        start.wait()
        for t in th:
            t.join()
        slowest = max(r[0] for r in res)
        mt = sum(r[1] for r in res) / slowest
    out = {
        "value": dec / dt, "unit": "decisions/s", "cores": 1, "kind": "port",
        "sample": f"{Gs} groups x {R} replicas x {done} ticks of the same steady-state stream, "
                  f"C++ oracle (Rust reference not buildable here: no cargo)",
        "all_cores_value": mt, "all_cores": cores, "hardware_threads": os.cpu_count(),
    }
    if mt:
        out["all_cores_sample"] = f"{cores} threads x private engine of {Gt} groups x {R} replicas x {Tt} ticks"
        out["all_cores_scaling_efficiency"] = mt / (cores * dec / dt)
    return out


def workload_name(G: int, R: int, mode: int, failures: int) -> str:
    """Which BASELINE.json config this run is (or which one it is a GPU's share of)."""
    if failures:
        tag = f"BASELINE.json configs[4]: {failures} %/tick leader failures + re-elections" + \
              ("" if (G, R) == (1_000_000, 5) else " at another size")
    elif mode == 1:
        tag = "BASELINE.json configs[1]: ragged AppendEntries-ack stream (drops, duplicates, 0-2 appends)" + \
              ("" if (G, R) == (10_000, 3) else " at another size")
    elif (G, R) == (1_000_000, 5):
        tag = "BASELINE.json configs[2]: steady-state append+commit"
    elif (G, R) == (1_250_000, 3):
        tag = "one GPU's share of BASELINE.json configs[3] (10 M x 3 over 8 GPUs): steady-state append+commit"
    else:
        tag = "steady-state append+commit at a non-BASELINE size"
    return f"{G} partitions x {R} replicas per GPU, {tag}; device-resident synthetic ack stream, mode {mode}"


def node_alg_bytes(R: int, common_ae: bool = True):
    """Algorithmic bytes per group of one closed-loop protocol round (DESIGN.md "Dense node tick").
    Leader half: the 8R + 28 of the ack tick (the inbox's answer words are its ack block: the
    HeartbeatResponse code rides in their low byte) + term 8 + heartbeat_time 8, outbox beat 16 + the
    AppendEntries words: ONE 8-byte word per group where every follower's is the same (JgLeaderNode::o_aec: what a
    jg_dense_cluster and jg_step_node with JG_NODE_COMMON_AE write in the steady state; round 5 still priced the R - 1
    words the kernel no longer moves - 132 B at R = 5 where PMC counts 112.7 MB per 1 M groups) = 108 B at R = 5,
    or (common_ae = False: the plain jg_step_node outbox) R - 1 words.  Follower half, per follower: state read 56 +
    inbox 24 (beat 16 + ae 8), written head 8 + answer 8, and every other tick (heartbeat)
    HeartbeatResponse.commit 8 + commit 8 + election timer 16."""
    leader = (8 * R + 28) + 16 + 16 + (8 if common_ae else 8 * (R - 1))
    follower = 56 + 24 + 8 + 8 + 16
    return leader, follower


def cluster_main(args, torch, dist, rank, world, dev_index, red_dev):
    """--cluster: all R replicas of every partition on this GPU as R engines (one per node) that
    exchange nothing but dense mailbox columns, chained with jg_stream_wait.  A step = one full
    protocol round: leader half (acks + appends + Tick) on the leader node, follower half
    (Heartbeat + AppendEntries + Tick) on the R-1 others; no synthetic acks anywhere."""
    import numpy as np
    from josefine_amd import BatchedRaft, capi
    from josefine_amd.traces import elect_all

    if args.any_leader:
        return cluster_any_main(args, torch, dist, rank, world, dev_index, red_dev)
    if args.failures:
        return cluster_failures_main(args, torch, dist, rank, world, dev_index, red_dev)
    G, R, K, W = args.groups, args.replicas, args.steps, args.warmup
    nodes = [BatchedRaft(G, R, seed=args.seed + r, device_id=dev_index, group_base=rank * G,
                         self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY) for r in range(R)]
    L = nodes[0]
    elect_all(L)
    L.drain_messages(), L.drain_applies()
    api = L.api

    # the round is driven from inside the library (jg_dense_cluster_*: leader half, then the follower
    # halves, streams chained with events): a round costs its launches, not a dozen ctypes calls
    arr = (C.c_void_p * R)(*[n._h for n in nodes])
    cl = C.c_void_p()
    L._check(api.dense_cluster_create(arr, R, 0, C.byref(cl)))
    L._check(api.dense_cluster_set_appends(cl, 1, None))  # one ClientRequest per group per round
    for e in nodes:
        e._check(api.sync(e._h))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        for e in nodes:
            e._check(api.sync(e._h))

    now = [0]

    def rounds(n):
        L._check(api.dense_cluster_rounds(cl, now[0] + 100, 100, n))
        now[0] += 100 * n

    rounds(W)
    barrier()
    c0 = L.counters()
    barrier()
    t0 = time.perf_counter()
    L._check(api.timer_start(L._h))
    rounds(K)
    ev_ms = C.c_float(0)
    L._check(api.timer_stop(L._h, C.byref(ev_ms)))
    for e in nodes:
        e._check(api.sync(e._h))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0  # this rank's K rounds are done; MAX over ranks below
    barrier()
    decisions = float(L.counters()["decisions"] - c0["decisions"])
    T_timed = W + K
    # the leader half alone: a few more rounds, one call each (eager launches), with event pairs
    # around k_leader_node_tick
    L._check(api.kernel_timing(L._h, 1))
    for _ in range(20):
        rounds(1)
    k_us, k_n = C.c_float(0), C.c_uint32(0)
    L._check(api.kernel_timing_read(L._h, C.byref(k_us), C.byref(k_n)))
    L._check(api.kernel_timing(L._h, 0))

    # full-size property check: real protocol rounds, so the commit index trails the head by the
    # round trip (append -> replicate -> ack -> majority) and every follower tracks the leader
    T = T_timed + 20
    head, commit = L.read("head"), L.read("commit")
    assert (head == T).all() and (commit >= T - 3).all() and not L.read("fault").any(), "closed loop: leader state"
    for r in range(1, R):
        e = nodes[r]
        assert (e.read("head") >= T - 1).all() and (e.read("commit") >= T - 5).all() and not e.read("fault").any(), \
            f"closed loop: follower {r} state"
        assert (e.read("voted_for") == L.node_ids[0]).all()
    assert sum(len(e.drain_messages()) for e in nodes) == 0, "rows left the mailbox vocabulary"

    if world > 1:
        tw = torch.tensor([wall, ev_ms.value], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        td = torch.tensor([decisions], dtype=torch.float64, device=red_dev)
        dist.all_reduce(td, op=dist.ReduceOp.SUM)
        wall, decisions = tw[0].item(), td[0].item()
    if rank == 0:
        lb, fb = node_alg_bytes(R)
        alg = (lb + (R - 1) * fb) * G
        round_s = ev_ms.value / 1e3 / K
        out = {
            "metric": "Raft quorum decisions/sec over N partitions; achieved HBM GB/s vs roofline",
            "value": decisions / wall, "unit": "decisions/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"closed loop: {R} nodes x {G} partitions on one GPU, 1 append per partition per round, "
                                   "leader half + follower halves over dense mailboxes (no synthetic acks)",
                       "partitions_per_gpu": G, "replicas": R, "partitions_total": G * world,
                       "q9": "off (JG_CFG_SEPARATE_COMMIT_KEY: bit-exact vs the oracle with the same switch; the reference would "
                             "panic at the first replicate() to a caught-up follower, leader.rs:152-157)",
                       "parallelism": f"{world} independent shard(s), no collective", **devices_config(args, world)},
            "group_rounds_per_s": G * world * K / wall,
            "roofline": {"bound": "hbm", "achieved": alg / round_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / round_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": f"one round: k_leader_node_tick<{R}> + {R - 1} x k_follower_tick_dense (+ empty slow kernels)",
                         "alg_bytes_per_launch": alg, "avg_launch_us": round_s * 1e6,
                         "alg_bytes_per_group": {"leader_half": lb, "follower_half": fb},
                         "frac_of_measured_copy": alg / round_s / 1e9 / 6290.0,
                         # the leader half alone, by its own HIP event pairs (jg_kernel_timing)
                         "leader_kernel": {"kernel": f"k_leader_node_tick<{R}>", "avg_launch_us": k_us.value,
                                           "launches_timed": k_n.value, "alg_bytes_per_launch": lb * G,
                                           "achieved": lb * G / (k_us.value * 1e-6) / 1e9 if k_us.value else None,
                                           "frac": lb * G / (k_us.value * 1e-6) / 1e9 / HBM_PEAK_GBS if k_us.value else None}},
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cluster_any_main(args, torch, dist, rank, world, dev_index, red_dev):
    """--cluster --any-leader: PER-PARTITION LEADERSHIP inside the device-resident cluster (jg_dense_cluster_create with
    JG_CLUSTER_ANY_LEADER).  Leaders are ELECTED, through the library's transport: every partition's designated
    candidate (--leadership blocked: node g*R/G, interleaved: node g % R) receives Timeout, campaigns, its VoteRequests
    reach the peers as routed rows, they answer through can_vote, the first majority makes it leader
    (candidate.rs:101-113) - then every node leads G/R partitions and follows the rest, and the timed region is the closed
    loop over the cluster's mailbox columns with every winner replicating in column form (leader.rs:124-174).
    With --failures p: per round p % of the partitions lose their leader - at R = 3 the whole group restarts (a rack),
    the next replica campaigns and WINS through the transport, leadership moves and stays in the columns; the client
    withdraws its proposals from a partition whose leader it lost (what the reference makes of a re-elected leader is
    Q8: its first append panics, chain.rs:163)."""
    import numpy as np
    from josefine_amd import BatchedRaft, DenseCluster, capi
    from josefine_amd.traces import any_failure_rows

    G, R, K, W = args.groups, args.replicas, args.steps, args.warmup
    nodes = [BatchedRaft(G, R, seed=args.seed + r, device_id=dev_index, group_base=rank * G,
                         self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY) for r in range(R)]
    L = nodes[0]
    api = L.api
    lib = DenseCluster(nodes, lead=None, vote_words=bool(args.vote_words))
    lib.set_appends(0)
    g = np.arange(G, dtype=np.int64)
    leader_of = (g * R // G) if args.leadership == "blocked" else g % R

    def sync_all():
        for e in nodes:
            e._check(api.sync(e._h))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        sync_all()

    now = [0]

    def routed(inject=None):
        now[0] += 100
        return lib.round_routed(now[0], inject)

    # -- the elections, through the transport - at any R since round 6: it delivers every voter's first answer before
    # anybody's second (jg_route.h: phase, emission index, sender), so the candidate sees its quorum of grants before the
    # refusals of its further copies overwrite them (candidate.rs:30-37, election.rs:33-35; until round 5 each sender's
    # answers came back to back and R > 3 needed synthetic votes).  --synthetic-votes 1 brings those back (an A/B).
    through_transport = not args.synthetic_votes
    votes_routed = 0
    sync_all()
    t0 = time.perf_counter()
    if through_transport:
        inj = []
        for n in range(R):
            mine = np.nonzero(leader_of == n)[0].astype(np.uint32)
            inj.append(nodes[n].upload_rows(kind=np.full(len(mine), capi.CMD_TIMEOUT, np.uint8), group=mine) if len(mine) else None)
        sync_all()
        t0 = time.perf_counter()
        votes_routed = sum(routed(inj)["delivered"])
        for _ in range(3):  # VoteRequest -> VoteResponse -> elect()
            votes_routed += sum(routed()["delivered"])
        sync_all()
        for rows in inj:
            if rows is not None:
                rows.free()
    else:
        from josefine_amd.traces import elect_where
        for n, e in enumerate(nodes):
            elect_where(e, leader_of == n)
            e.drain_messages(), e.drain_applies()
    election_ms = (time.perf_counter() - t0) * 1e3
    for n, e in enumerate(nodes):
        role = e.read("role")
        assert (role[leader_of == n] == capi.ROLE_LEADER).all() and (role[leader_of != n] != capi.ROLE_LEADER).all(), \
            f"node {n}: the elections did not make it leader of its partitions"
    lib.set_appends(1)
    appended_from = now[0]

    failed = np.zeros(G, bool)
    trace, withdraw = [], []
    recreate = bool(args.recreate)
    whole = R == 3 or recreate
    if args.failures:
        for t in range(W + K):
            # --recreate: the failing group comes back on EMPTY stores (JG_CMD_RECREATE): the election's winner can append
            # (no Q8), the client keeps proposing, a group may fail any number of times - a stationary trace whose every vote
            # is real (R = 3: the first grant is the quorum, the campaign is won through the transport)
            cols, failing = any_failure_rows(args.seed, t, G, R, args.failures, leader_of, group_base=rank * G, whole_group=whole,
                                             skip=None if recreate else failed, recreate=recreate)
            failed[failing] = True
            trace.append([None if c is None else nodes[n].upload_rows(**c) for n, c in enumerate(cols)])
            wl = None
            if len(failing) and not recreate:
                p = C.c_void_p()
                L._check(api.device_alloc(L._h, max(failing.nbytes, 16), C.byref(p)))
                L._check(api.device_upload(L._h, p, failing.ctypes.data, failing.nbytes))
                wl = (p, len(failing))
            withdraw.append(wl)
    sync_all()

    delivered = [0, 0]
    kept = [0]

    def rounds(t0_, t1_, which):
        if not args.failures:
            lib.rounds(now[0] + 100, 100, t1_ - t0_)
            now[0] += 100 * (t1_ - t0_)
            return
        for t in range(t0_, t1_):
            if withdraw[t] is not None:
                lib.withdraw_appends(withdraw[t][0].value, withdraw[t][1])
            st = routed(trace[t])
            delivered[which] += sum(st["delivered"])
            kept[0] += st["kept"]

    rounds(0, W, 0)
    barrier()
    c0 = sum(e.counters()["decisions"] for e in nodes)
    led_at_start = None
    if args.failures:
        led0 = np.zeros(G, bool)
        for e in nodes:
            led0 |= (e.read("role") == capi.ROLE_LEADER) & (e.read("fault") == 0)
        led_at_start = float((~led0).mean())
    failed_at_start = float(failed_before(args, np, G, R, leader_of, rank, W).mean()) if args.failures and not recreate else 0.0
    barrier()
    t0 = time.perf_counter()
    L._check(api.timer_start(L._h))
    rounds(W, W + K, 1)
    ev_ms = C.c_float(0)
    L._check(api.timer_stop(L._h, C.byref(ev_ms)))
    sync_all()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    barrier()
    decisions = float(sum(e.counters()["decisions"] for e in nodes) - c0)

    # full-size properties (window parity against the numpy statement over oracle engines: tests/test_gpu_fullsize.py)
    T = (now[0] - appended_from) // 100
    healthy = ~failed
    appending_again = 0.0
    led = np.zeros(G, bool)
    won = 0
    for n, e in enumerate(nodes):
        role, head, commit, fault = e.read("role"), e.read("head"), e.read("commit"), e.read("fault")
        mine = healthy & (leader_of == n)
        assert (role[mine] == capi.ROLE_LEADER).all() and (head[mine] == T).all() and (commit[mine] >= T - 3).all(), f"node {n}: its partitions"
        other = healthy & (leader_of != n)
        assert (role[other] == capi.ROLE_FOLLOWER).all() and (head[other] >= T - 1).all() and (commit[other] >= T - 5).all(), f"node {n}: as a follower"
        assert not fault[healthy].any()
        lead_now = (role == capi.ROLE_LEADER) & (fault == 0)
        led |= lead_now
        won += int((lead_now & failed & ((leader_of + 1) % R == n)).sum())
        assert not recreate or not fault.any(), "a re-created group must not fault"
        if recreate:  # the winners of re-created groups append again: a block per round since they were elected
            mine_again = lead_now & failed & ((leader_of + 1) % R == n)
            appending_again += float((mine_again & (head > 0)).sum()) / max(int(failed.sum()), 1)
    rows_left = sum(len(e.drain_messages()) for e in nodes)
    if not args.failures:
        assert rows_left == 0, "rows left the mailbox vocabulary"

    if world > 1:
        tw = torch.tensor([wall, ev_ms.value], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        td = torch.tensor([decisions, float(delivered[1])], dtype=torch.float64, device=red_dev)
        dist.all_reduce(td, op=dist.ReduceOp.SUM)
        wall, decisions, delivered[1] = tw[0].item(), td[0].item(), td[1].item()
    if rank == 0:
        lb, fb = node_alg_bytes(R)
        alg = (lb + (R - 1) * fb) * G
        round_s = (ev_ms.value / 1e3 if not args.failures else wall) / K
        out = {
            "metric": "Raft quorum decisions/sec over N partitions; achieved HBM GB/s vs roofline",
            "value": decisions / wall, "unit": "decisions/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall * 1e3 / K, "ms_per_step_events": ev_ms.value / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": (f"per-partition leadership: {R} nodes x {G} partitions on one GPU, leaders elected " + ("THROUGH the device transport" if through_transport else "(synthetic votes)") + " "
                                    f"({args.leadership}: every node leads G/R partitions and follows the rest), closed loop over the "
                                    "cluster's mailbox columns, 1 client request per led partition per round"
                                    + (f"; {args.failures} %/round of the partitions lose their leader"
                                       + (" (the whole group comes back on EMPTY stores, JG_CMD_RECREATE - again and again), the next replica campaigns and wins "
                                          "through the transport - every vote real -, leads and APPENDS (a chain that starts over: no Q8); the client "
                                          "keeps proposing: a stationary trace" if recreate else
                                          " (the whole group restarts), the next replica campaigns and wins through the transport, leadership "
                                          "moves and stays in column form; the client withdraws its proposals from such a partition (Q8)"
                                          if whole else " (crash + restart; the other replicas remember their vote: Q4, the partition stays leaderless)")
                                       if args.failures else "")),
                       "partitions_per_gpu": G, "replicas": R, "partitions_total": G * world, "leadership": args.leadership,
                       "q9": "off (JG_CFG_SEPARATE_COMMIT_KEY: bit-exact vs the oracle with the same switch; the reference would "
                             "panic at the first replicate() to a caught-up follower, leader.rs:152-157)",
                       "parallelism": f"{world} independent shard(s), no collective", **devices_config(args, world)},
            "group_rounds_per_s": G * world * K / wall,
            "elections": {"partitions": G, "won_through_the_transport": through_transport, "rounds": 4 if through_transport else None, "wall_ms": election_ms,
                          "rows_routed": votes_routed, "leaders_per_node": [int((leader_of == n).sum()) for n in range(R)]},
            "roofline": {"bound": "hbm", "achieved": alg / round_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / round_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": f"one round: k_cluster_claim + k_leader_node_tick_any<{R}> (all nodes) + k_follower_tick_dense_any (all nodes) "
                                   "+ the two slow kernels" + (" + the sparse steps over the routed rows + the transport" if args.failures else ""),
                         "alg_bytes_per_launch": alg, "avg_launch_us": round_s * 1e6,
                         "alg_bytes_per_group": {"leader_half": lb, "follower_half": fb},
                         "frac_of_measured_copy": alg / round_s / 1e9 / 6290.0,
                         "note": "priced with what ONE leader step and R - 1 follower steps per partition must move (the single-lead "
                                 "closed loop's bytes): every node runs both halves here, each over the partitions its role selects"},
        }
        if args.failures:
            out["leaderless_fraction"] = {"at_start_of_timed_region": led_at_start, "at_end": float((~led).mean())}
            if recreate:
                out["config"]["stationary"] = "yes: groups fail at any time, any number of times; every election is won through the transport and its winner appends"
                out["winners_appending_again_fraction_of_failed_groups"] = appending_again
            out["failed_fraction"] = {"at_start_of_timed_region": failed_at_start, "at_end": float(failed.mean())}
            out["elections_won_after_failures"] = won
            out["rows_routed_per_round"] = delivered[1] / K / world
            out["rows_left_for_the_host"] = rows_left
            out["decisions_in_timed_region"] = decisions
            out["vote_words"] = bool(args.vote_words)  # (JG_CLUSTER_OPT_VOTE_WORDS: jg_votes.h)
        print(json.dumps(out), flush=True)
    lib.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def failed_before(args, np, G, R, leader_of, rank, upto):
    from josefine_amd.traces import any_failure_rows
    failed = np.zeros(G, bool)
    for t in range(upto):
        _, failing = any_failure_rows(args.seed, t, G, R, args.failures, leader_of, group_base=rank * G, skip=failed)
        failed[failing] = True
    return failed


def cluster_failures_main(args, torch, dist, rank, world, dev_index, red_dev):
    """--cluster --failures p: BASELINE.json configs[4] as SURVEY.md §8(d) #5 specifies it, as a STATIONARY trace
    (josefine_amd.traces.FailureRepairTrace).  The closed loop of --cluster, and per round p % of the partitions that are
    up lose their leader: its replica crashes and restarts, a restarted follower (voted_for == None, §7.3 Q4) times out
    and campaigns, the other replicas answer its VoteRequests through can_vote on the device - they remember their
    vote and refuse, the partition is leaderless and its candidate campaigns again at every election timeout; every vote
    travels between the engines through the library's device-side transport (jg_dense_cluster_round_routed) and is
    applied the round after.  The client stops proposing to it.  --repair-after D rounds later the partition is RE-CREATED
    (what the reference leaves of it can never append again, Q8, and would be re-fed from block 1, Q10: every replica
    restarts on an empty store - JG_CMD_RECREATE -, replica 0 receives Timeout, campaigns, and is ELECTED two rounds later by
    the answers the transport brings back - no synthetic vote anywhere since round 6: every voter's first answer is
    delivered before anybody's second, so five nodes elect as three do) and is what every partition was at round 0, the
    client proposing again: leaderless fraction (p x (D + 2)), decisions per round and cost per round are flat.
    --repair-after 0: no repairs (rounds 2-4's trace: the leaderless fraction grows by p per round)."""
    import numpy as np
    from josefine_amd import BatchedRaft, DenseCluster, capi
    from josefine_amd.traces import FailureRepairTrace, elect_all

    G, R, K, W = args.groups, args.replicas, args.steps, args.warmup
    D = args.repair_after
    nodes = [BatchedRaft(G, R, seed=args.seed + r, device_id=dev_index, group_base=rank * G,
                         self_slots=np.full(G, r, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY) for r in range(R)]
    L = nodes[0]
    elect_all(L)
    L.drain_messages(), L.drain_applies()
    api = L.api
    lib = DenseCluster(nodes, vote_words=bool(args.vote_words))
    lib.set_appends(1)
    # The trace is resident in HBM before the timed region: per round and node one group-sorted batch, and the list of the
    # partitions the client stops proposing to.  The trace has run SETTLE rounds before the first timed one - W of them
    # as the warm-up, the rest before that - so that the timed region starts in the trace's steady state (a repair
    # schedule of D rounds needs D rounds to fill; election timeouts are 5-10 rounds)
    settle = max(0, (2 * D + 10 if D else 0) - W)
    tr = FailureRepairTrace(args.seed, G, R, args.failures, D if D else 1 << 40, group_base=rank * G, node_ids=[nodes[r].node_ids[r] for r in range(R)])
    trace, withdraw, offer, frac, seated = [], [], [], [], []
    for t in range(settle + W + K):
        cols, failing, repaired = tr.rows(t)
        seated.append(len(repaired))
        trace.append([None if c is None else nodes[n].upload_rows(**c) for n, c in enumerate(cols)])
        withdraw.append(L.upload_u32(failing) if len(failing) else None)   # the client stops proposing to a partition that lost its leader ...
        offer.append(L.upload_u32(repaired) if len(repaired) else None)    # ... and proposes again to one that was re-created
        frac.append(float(tr.leaderless().mean()))
    W += settle  # (from here on the settle rounds are warm-up rounds)
    for e in nodes:
        e._check(api.sync(e._h))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        for e in nodes:
            e._check(api.sync(e._h))

    delivered = [0, 0]  # [warm-up, timed] rows the transport moved
    kept = [0]
    fsm = [0, 0]  # [queued by the rounds, handed to the host]: a repaired follower's first Heartbeat advances its commit index -
    #               the Apply range (follower.rs:204-206) is the partition's state machine's, drained like any (jg_drain_applies_view)

    def rounds(t0_, t1_, which):
        for t in range(t0_, t1_):
            # (the client's proposals: one per round for every partition, offered where replica 0 LEADS when the dense round begins -
            # JgLeaderNode::mask_offers.  A failed partition has no leader there until it is re-created and its election is won, so
            # nothing has to be withdrawn or offered again per round: rounds 5's two launches per round are gone.  --offer-lists 1
            # issues them as before - jg_dense_cluster_withdraw_appends / _offer_appends, what tests/test_dense_node.py drives)
            if args.offer_lists:
                if withdraw[t] is not None:
                    lib.withdraw_appends(withdraw[t].ptr, withdraw[t].n)
                if offer[t] is not None:
                    lib.offer_appends(offer[t].ptr, offer[t].n, 1)
            tr0 = time.perf_counter()
            st = lib.round_routed((t + 1) * 100, trace[t])
            if os.environ.get("JG_BENCH_TRACE_ROUNDS"):
                print(f"[bench] round {t}: {(time.perf_counter() - tr0) * 1e3:.3f} ms, delivered {sum(st['delivered'])}, kept {st['kept']}, fsm {st['fsm_rows']}", file=sys.stderr)
            delivered[which] += sum(st["delivered"])
            kept[0] += st["kept"]
            fsm[0] += st["fsm_rows"]
            if st["fsm_rows"] and args.drain_applies:
                for e in nodes:
                    fsm[1] += len(e.drain_applies(copy=False))

    rounds(0, W, 0)
    barrier()
    c0 = sum(e.counters()["decisions"] for e in nodes)
    L._check(api.kernel_timing(L._h, 8))  # (every 8th leader half: an event pair around a launch costs the stream ~10 us - rocprofv3's trace shows the two gaps)
    barrier()
    # The K rounds are also timed in four consecutive windows (a routed round synchronises with the host anyway: the
    # window marks add nothing), each reported against the leaderless fraction it ran at: flat with the repair schedule,
    # growing without it.
    marks = [W + (K * q) // 4 for q in range(5)]
    window_s = []
    t0 = time.perf_counter()
    L._check(api.timer_start(L._h))
    for q in range(4):
        tq = time.perf_counter()
        rounds(marks[q], marks[q + 1], 1)
        window_s.append(time.perf_counter() - tq)
    ev_ms = C.c_float(0)
    L._check(api.timer_stop(L._h, C.byref(ev_ms)))
    for e in nodes:
        e._check(api.sync(e._h))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    barrier()
    decisions = float(sum(e.counters()["decisions"] for e in nodes) - c0)
    k_us, k_n = C.c_float(0), C.c_uint32(0)
    L._check(api.kernel_timing_read(L._h, C.byref(k_us), C.byref(k_n)))
    L._check(api.kernel_timing(L._h, 0))

    # full-size properties (parity against oracle clusters: tests/test_dense_node.py::test_stationary_failure_repair_trace_*,
    # tests/test_gpu_fullsize.py): what the reference's rules make of this trace - a failing partition
    # is leaderless (every replica that was not restarted remembers its vote and refuses) until its repair, the partitions
    # that never failed keep committing, nothing faults, no row leaves the transport's vocabulary
    T = W + K
    down, never = tr.leaderless(), ~tr.ever_failed
    role = L.read("role")
    assert (role[down] != capi.ROLE_LEADER).all() and (role[~down] == capi.ROLE_LEADER).all(), "leadership"
    assert (L.read("head")[never] == T).all() and (L.read("commit")[never] >= T - 3).all(), "healthy partitions commit"
    recreated = tr.ever_failed & ~down
    assert not D or (recreated.any() and (L.read("head")[recreated] > 0).all() and (L.read("head")[recreated] < T).all()), "re-created partitions append again"
    for e in nodes[1:]:
        assert (e.read("role") != capi.ROLE_LEADER).all()
    for e in nodes:
        assert not e.read("fault").any()
    assert kept[0] == 0 and sum(len(e.drain_messages()) for e in nodes) == 0, "rows left the transport's vocabulary"
    fsm[1] += sum(len(e.drain_applies(copy=False)) for e in nodes)
    assert fsm[0] == fsm[1], "FSM rows"

    per_rank = None
    if world > 1:
        mine = [None] * world
        dist.all_gather_object(mine, (decisions / wall, wall * 1e6 / K))
        per_rank = {"decisions_per_s": [m[0] for m in mine], "avg_launch_us": [m[1] for m in mine]}
        tw = torch.tensor([wall, ev_ms.value], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        td = torch.tensor([decisions, float(delivered[1])], dtype=torch.float64, device=red_dev)
        dist.all_reduce(td, op=dist.ReduceOp.SUM)
        wall, decisions, delivered[1] = tw[0].item(), td[0].item(), td[1].item()
    if rank == 0:
        lb, fb = node_alg_bytes(R)
        alg = (lb + (R - 1) * fb) * G
        round_s = wall / K
        out = {
            "metric": "Raft quorum decisions/sec over N partitions; achieved HBM GB/s vs roofline",
            "value": decisions / wall, "unit": "decisions/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / K, "ms_per_step_events": ev_ms.value / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[4] as specified: closed loop of {R} nodes x {G} partitions on one GPU, "
                                   f"{args.failures} %/round of the partitions that are up lose their leader (crash + restart), a restarted "
                                   "follower times out and campaigns (again at every election timeout), votes answered through can_vote "
                                   "and routed between the nodes on the device, applied the round after"
                                   + (f"; {D} rounds after its failure a partition is RE-CREATED: every replica restarts on an empty store, replica 0 "
                                      "receives Timeout, campaigns and is ELECTED through the transport two rounds later (no synthetic vote: its "
                                      "VoteRequests, the voters' answers and its Heartbeat are all routed)"
                                      if D else "; no repairs: a failed partition stays leaderless")
                                   + "; 1 client request per round for every partition that has a leader",
                       "partitions_per_gpu": G, "replicas": R, "partitions_total": G * world,
                       "q9": "off (JG_CFG_SEPARATE_COMMIT_KEY: bit-exact vs the oracle with the same switch; the reference would "
                             "panic at the first replicate() to a caught-up follower, leader.rs:152-157)",
                       "parallelism": f"{world} independent shard(s), no collective", **devices_config(args, world)},
            "group_rounds_per_s": G * world * K / wall,
            "per_rank": per_rank,
            "rows_routed_per_round": delivered[1] / K / world,
            "decisions_in_timed_region": decisions,
            "vote_words": bool(args.vote_words),  # (JG_CLUSTER_OPT_VOTE_WORDS: jg_votes.h; rows_routed_per_round counts rows only)
            "roofline": {"bound": "hbm", "achieved": alg / round_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / round_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": f"one round: k_leader_node_tick<{R}> + {R - 1} x k_follower_tick_dense + slow kernels "
                                   f"+ the sparse steps over the routed rows (k_apply_vote_runs_multi / k_apply_rows_multi) + the transport "
                                   "(k_route_rec_multi / _xq_multi, bucket pass + k_route_sort_build: no library sort)",
                         "alg_bytes_per_launch": alg, "avg_launch_us": round_s * 1e6,
                         "alg_bytes_per_group": {"leader_half": lb, "follower_half": fb},
                         "frac_of_measured_copy": alg / round_s / 1e9 / 6290.0,
                         "note": "priced with the dense halves only (every partition's tick); the routed rows and the "
                                 "general state machine they run through are the overhead this line shows; one host "
                                 "synchronisation per round (the transport's row counts)",
                         "leader_kernel": {"kernel": f"k_leader_node_tick<{R}> (leaderless partitions deferred to k_dense_slow)",
                                           "avg_launch_us": k_us.value, "launches_timed": k_n.value,
                                           "alg_bytes_per_launch": lb * G,
                                           "achieved": lb * G / (k_us.value * 1e-6) / 1e9 if k_us.value else None,
                                           "frac": lb * G / (k_us.value * 1e-6) / 1e9 / HBM_PEAK_GBS if k_us.value else None}},
        }
        lf = lambda t: frac[t - 1] if t else 0.0  # the leaderless fraction BEFORE round t
        out["leaderless_fraction"] = {"at_start_of_timed_region": lf(W), "at_end": frac[W + K - 1]}
        out["ms_per_round_by_leaderless_fraction"] = [
            {"rounds": [marks[q] - W, marks[q + 1] - W], "leaderless_fraction": [lf(marks[q]), frac[marks[q + 1] - 1]],
             "ms_per_round": window_s[q] * 1e3 / max(marks[q + 1] - marks[q], 1)} for q in range(4)]
        out["partitions_that_failed_at_least_once"] = float(tr.ever_failed.mean())
        # every one of them checked above: a partition the trace counts as up is LED by replica 0 (`leadership`), and nothing
        # but the transport's mail can have elected it - the trace injects Restart / Recreate / Timeout rows only
        out["elections_won_through_the_transport"] = int(sum(seated[W:W + K])) if D else 0
        out["synthetic_votes"] = 0
        out["fsm_rows_per_round"] = fsm[0] / (W + K)
        out["config"]["fsm_rows"] = ("drained by the host every round (jg_drain_applies_view: the pinned queue, no copy)" if args.drain_applies
                                     else "left queued until the end of the run (--drain-applies 0)")
        out["config"]["repair_after_rounds"] = D
        out["config"]["rounds_before_the_timed_region"] = W
        out["config"]["stationary"] = (
            f"yes: failures at {args.failures} %/round of the partitions that are up, each re-created {D} rounds later and then what every "
            "partition was at round 0 (Q8 / Q10: what the reference leaves of a failed partition can neither append nor be "
            "caught up at a cost that does not grow with the run) - leaderless fraction, decisions per round and cost per "
            "round are flat (leaderless_fraction, ms_per_round_by_leaderless_fraction: the same run in four windows)"
            if D else
            "no (--repair-after 0): a partition that lost its leader stays leaderless (SURVEY.md 7.3 Q4) and campaigns at every "
            "election timeout; ms_per_round_by_leaderless_fraction times the same run in four windows")
        print(json.dumps(out), flush=True)
    lib.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def event_loop_main(args):
    """--event-loop: decisions/s THROUGH the reference's driver surface.  One node that leads G partitions,
    driven by josefine::BatchedEventLoop (the C++ mirror of server::event_loop, src/raft/server.rs:103-165) over
    jg_step_node: inbound rows (one ClientRequest, R-1 AppendResponses and, every other tick, R-1
    HeartbeatResponses per partition and tick, shuffled) decoded straight into the engine's pinned columns,
    classified and scattered into the dense mailboxes ON THE DEVICE, the Tick as a flag; fsm_tx rows and the
    Tick's outbox columns come back over PCIe and are read by batch sinks.  The followers are synthetic, as
    the ack stream of the headline line is.  The binary is josefine_amd/host/bench_event_loop.cpp.
    `value` is the process hosting the G partitions on several such loops (--loops: one host thread, one engine,
    two HIP streams each - the reference runs a task per partition; a batched loop is synchronous per tick, so
    loops side by side are what overlaps decode, both directions of the bus and kernels); the one-loop figures,
    the column-inbound figures (batched peers) and round 2's loop are in the `event_loop` object."""
    import subprocess
    from josefine_amd.build import build_event_loop_bench
    exe = build_event_loop_bench()
    G, R, K, W = args.groups, args.replicas, args.steps, args.warmup
    cands = sorted({max(1, int(x)) for x in str(args.loops).split(",") if x.strip()})
    cands = [c for c in cands if G % c == 0] or [1]  # (the partitions divide evenly over the loops)

    def hw_queues(loops):
        """Every loop has two streams (steps, drains); the HIP runtime multiplexes a process's streams onto 4 hardware
        queues unless told otherwise (GPU_MAX_HW_QUEUES), and streams that share one run one behind the other."""
        return os.environ.get("GPU_MAX_HW_QUEUES") or (str(2 * loops) if loops > 2 else None)

    def run(mode, k, w, loops=1, polled=False, compact=False, in_flight=1, helpers=None):
        env = dict(os.environ)
        env["JG_BENCH_IN_FLIGHT"] = str(in_flight)
        if env.get("JG_BENCH_POLLING_DEFAULTED"):  # (several loop threads waiting side by side: the runtime's default, interrupts)
            env.pop("HSA_ENABLE_INTERRUPT", None)
        if polled:  # (the A/B: the loop's thread spins on the completion signal instead of sleeping on an interrupt)
            env["HSA_ENABLE_INTERRUPT"] = "0"
        if hw_queues(loops):
            env["GPU_MAX_HW_QUEUES"] = hw_queues(loops)
        r = subprocess.run([exe, str(G), str(R), str(k), str(w), mode, "0", str(loops)] + ([str(helpers if helpers is not None else R - 1), "compact"] if compact else []),
                           capture_output=True, text=True, timeout=1200, env=env)
        if r.returncode != 0:
            raise SystemExit(f"bench_event_loop {mode} x {loops} failed: {r.stdout} {r.stderr}")
        return json.loads(r.stdout.strip().splitlines()[-1])

    def best(mode, one):  # the process hosts the partitions on L loops (one thread + one engine each): the best of the candidates
        runs = [one if c == 1 else run(mode, K, W, c) for c in cands]
        return max(runs, key=lambda r: r["decisions_per_s"]), {str(r["loops"]): r["decisions_per_s"] for r in runs}

    p1 = run("pipe", K, W)         # ONE loop that overlaps with itself: tick t + 1 is decoded while tick t runs and lands
    pc1 = run("pipecolumns", K, W)
    pt1 = run("pipetasks", K, W)   # ... with the reference's other tasks (connection readers, channel consumers) on threads of their own
    ptc1 = run("pipetaskscolumns", K, W)
    pt1_polled = run("pipetasks", K, W, polled=True)
    # ... and with ABI v7's bus formats: sender slot and flag in the kind byte (13 B per inbound row, not 18), the Tick's
    # AppendEntries words as one word per partition (8 B, not 8 (R - 1)), a leader's Apply + Notify of a tick as one fsm row
    ptk = run("pipetasks", K, W, compact=True)
    ptck = run("pipetaskscolumns", K, W, compact=True)
    pk1 = run("pipe", K, W, compact=True)
    # ... and with TWO ticks in flight (JG_NODE_KEEP): tick t + 1 begun before tick t's outputs are read
    ptk2 = run("pipetasks", K, W, compact=True, in_flight=2)
    ptck2 = run("pipetaskscolumns", K, W, compact=True, in_flight=2)
    pt2 = run("pipetasks", K, W, in_flight=2)
    ptk2_polled = run("pipetasks", K, W, compact=True, in_flight=2, polled=True)
    ptk2_8 = run("pipetasks", K, W, compact=True, in_flight=2, helpers=8)  # (the reference's runtime has a worker per core: eight task threads beside the loop's instead of R - 1)
    ptk1_8 = run("pipetasks", K, W, compact=True, in_flight=1, helpers=8)
    # ... and with the peers' traffic as the reference's BYTES (length-delimited serde_json frames, tcp.rs:139-170) through
    # host/formats.hpp's decoder in the connection tasks: a few ticks (the senders' encoding of every tick comes first)
    ptw = run("pipetaskswire", max(2, min(K, 4)), 2)
    d1 = run("inplace", K, W)   # ONE loop owns every partition
    d, d_by_loops = best("inplace", d1)
    colm1 = run("columns", K, W)  # the followers' answers as columns (batched peers), only the client requests as rows
    colm, colm_by_loops = best("columns", colm1)
    L, Lc = d["loops"], colm["loops"]
    copy = run("copy", K, W)
    old = run("general", max(3, min(K, 10)), 2)  # round 2's loop: one Tick ROW per partition, the general state machine only
    lb = node_alg_bytes(R, common_ae=False)[0] + 4  # the leader half of the node tick (d1: the plain outbox, R - 1 AppendEntries words) + the fsm delta word it leaves behind
    k_us = d1["leader_kernel_us"]  # (from the one-loop run: the kernel over all G partitions, nothing beside it)
    ach = lb * G / (k_us * 1e-6) / 1e9 if k_us else 0.0
    loop_ms = d1["ms_submit"] + d1["ms_step_and_drain"]
    out = {
        "metric": "Raft quorum decisions/sec over N partitions; achieved HBM GB/s vs roofline",
        "value": d["decisions_per_s"], "unit": "decisions/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": d["ms_per_tick"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": f"through the Apply surface: josefine::BatchedEventLoop over jg_step_node, one node leading {G} partitions x "
                               f"{R} replicas; per tick and partition 1 ClientRequest + {R - 1} AppendResponses (+ {R - 1} HeartbeatResponses "
                               "every other tick) as shuffled host rows, decoded in place into the engine's pinned columns; Tick = one flag; "
                               "fsm_tx rows + outbox columns back over PCIe into batch sinks that read every byte; synthetic followers",
                   "partitions_per_gpu": G, "replicas": R, "partitions_total": G,
                   "parallelism": f"{L} event loop(s) of {G // L} partitions each (one host thread + one engine + one HIP stream per loop), 1 GPU",
                   "loops": L,
                   "q9": "off (JG_CFG_SEPARATE_COMMIT_KEY)", "devices": [0], "devices_aliased": False,
                   "host_wait": "completion signals by interrupt (the runtime's default)" if os.environ.get("JG_BENCH_POLLING_DEFAULTED")
                   or os.environ.get("HSA_ENABLE_INTERRUPT") != "0" else "completion signals polled (HSA_ENABLE_INTERRUPT=0)"},
        "event_loop": {
            "loops": L, "decisions_per_s_by_loops": d_by_loops,
            "hip_hardware_queues": {"GPU_MAX_HW_QUEUES": hw_queues(L), "note": "2 per loop (its step stream and its drain stream); the runtime's default of 4 "
                                    "makes the loops' streams share queues: 8.5-8.9e8/s instead of 1.0e9 with row inbound on 8 loops"},
            "ms_per_tick_per_loop": {"transport_decode_into_pinned_columns": d["ms_fill"], "submit_commit_validation": d["ms_submit"],
                                     "step_node_and_drains": d["ms_step_and_drain"], "all_loops_side_by_side": d["ms_per_tick"]},
            "one_loop_pipelined": {
                "what": "ONE loop (one host thread, one engine) that overlaps with itself: a step returns once its rows are on the device "
                        "and classified, the transport decodes the next tick into the freed pinned columns while the dense halves run and "
                        "the outputs travel home, fsm_tx / rpc_tx are fed at the start of the next step; rows validated on the device "
                        "(JG_COL_UNCHECKED)",
                "decisions_per_s": p1["decisions_per_s"], "ms_per_tick": p1["ms_per_tick"],
                "ms_per_tick_parts": {"transport_decode_into_pinned_columns": p1["ms_fill"], "submit_commit": p1["ms_submit"],
                                      "step_begin_and_previous_outputs": p1["ms_step_and_drain"]},
                "column_inbound_decisions_per_s": pc1["decisions_per_s"], "column_inbound_ms_per_tick": pc1["ms_per_tick"],
                "speedup_over_the_synchronous_loop": p1["decisions_per_s"] / d1["decisions_per_s"],
                "rows_on_the_general_path": p1["rows_general"]},
            "one_loop_with_transport_and_consumer_tasks": {
                "what": "the same ONE pipelined event loop (one engine, every engine call on the loop's thread) with the work the reference "
                        "does not do on the loop's task taken off its thread: frames are decoded by the per-connection read tasks "
                        "(src/raft/tcp.rs:139-170; the loop receives Commands from a channel, server.rs:120-137), fsm_tx is consumed by "
                        "the driver task (src/raft/fsm.rs), rpc_tx by the per-peer senders (tcp.rs:87-137) - here "
                        f"{pt1['task_threads_beside_each_loop']} helper threads beside the loop's, fork-join: the decoders fill their "
                        "slices of the pinned columns, the consumers read their slices of the output batches; the committed batch leaves "
                        "for the device at once (JG_COL_UPLOAD_NOW: a copy stream of its own, while the previous step's outputs travel the other way)",
                "task_threads_beside_the_loop": pt1["task_threads_beside_each_loop"],
                "decisions_per_s": pt1["decisions_per_s"], "ms_per_tick": pt1["ms_per_tick"],
                "ms_per_tick_parts": {"transport_decode_into_pinned_columns": pt1["ms_fill"], "submit_commit": pt1["ms_submit"],
                                      "step_begin_and_previous_outputs": pt1["ms_step_and_drain"]},
                "column_inbound_decisions_per_s": ptc1["decisions_per_s"], "column_inbound_ms_per_tick": ptc1["ms_per_tick"],
                "column_inbound_ms_per_tick_parts": {"transport_decode_into_pinned_columns": ptc1["ms_fill"], "submit_commit": ptc1["ms_submit"],
                                                     "step_begin_and_previous_outputs": ptc1["ms_step_and_drain"]},
                "rows_on_the_general_path": pt1["rows_general"] + ptc1["rows_general"],
                "host_wait": {"what": "these figures wait for completion signals by interrupt (the runtime's default: the loop's thread sleeps while the "
                                      "device works, as a host that also runs the broker needs it); polled (HSA_ENABLE_INTERRUPT=0) a core spins per waiting thread",
                              "interrupt_decisions_per_s": pt1["decisions_per_s"], "polled_decisions_per_s": pt1_polled["decisions_per_s"]},
                "compact_bus": {
                    "what": "the same loop and tasks with the node step's compact formats (ABI v7): JG_COL_PACKED_KIND - the transport's decoder writes "
                            "kind | sender slot << 4 | flag << 7 into one byte, no from / flag columns (13 B per inbound row instead of 18) -, "
                            "JG_NODE_COMMON_AE - the Tick's AppendEntries words come home as ONE word per partition where every follower's is the "
                            "same (8 B instead of 8 (R - 1); the rows are fetched only in a tick where some partition's words differ) - and "
                            "JG_NODE_FSM_FUSED - a leader's Apply + Notify (+ Apply) of a tick as one 24-byte fsm row",
                    "decisions_per_s": ptk["decisions_per_s"], "ms_per_tick": ptk["ms_per_tick"],
                    "ms_per_tick_parts": {"transport_decode_into_pinned_columns": ptk["ms_fill"], "submit_commit": ptk["ms_submit"],
                                          "step_begin_and_previous_outputs": ptk["ms_step_and_drain"]},
                    "pcie_bytes_per_tick": {"h2d": ptk["pcie_h2d_bytes_per_tick"], "d2h": ptk["pcie_d2h_bytes_per_tick"]},
                    "pcie_bytes_per_decision": (ptk["pcie_h2d_bytes_per_tick"] + ptk["pcie_d2h_bytes_per_tick"]) * ptk["ticks"] / ptk["decisions"],
                    "pcie_bytes_per_decision_plain": (pt1["pcie_h2d_bytes_per_tick"] + pt1["pcie_d2h_bytes_per_tick"]) * pt1["ticks"] / pt1["decisions"],
                    "fsm_rows_per_tick": ptk["fsm_rows_per_tick"], "fsm_rows_per_tick_plain": pt1["fsm_rows_per_tick"],
                    "column_inbound_decisions_per_s": ptck["decisions_per_s"], "column_inbound_ms_per_tick": ptck["ms_per_tick"],
                    "column_inbound_pcie_bytes_per_decision": (ptck["pcie_h2d_bytes_per_tick"] + ptck["pcie_d2h_bytes_per_tick"]) * ptck["ticks"] / ptck["decisions"],
                    "one_thread_decisions_per_s": pk1["decisions_per_s"], "one_thread_ms_per_tick": pk1["ms_per_tick"],
                    "rows_on_the_general_path": ptk["rows_general"] + ptck["rows_general"] + pk1["rows_general"],
                    "two_ticks_in_flight": {
                        "what": "BatchedEventLoop::in_flight = 2 (JG_NODE_KEEP, ABI v9): tick t + 1 is begun - its rows on their way up, its kernels "
                                "enqueued - BEFORE tick t's outbox is viewed; the view waits for tick t's outputs only, its fsm rows came home behind "
                                "its own kernels (the copy sized from the tick before: no drain issued by the host, no count waited for).  Same rows, "
                                "same sinks, same order (tests/test_node_step.py::test_node_step_two_in_flight_parity, the cluster of loops)",
                        "decisions_per_s": ptk2["decisions_per_s"], "ms_per_tick": ptk2["ms_per_tick"],
                        "ms_per_tick_parts": {"transport_decode_into_pinned_columns": ptk2["ms_fill"], "submit_commit": ptk2["ms_submit"],
                                              "step_begin_and_previous_outputs": ptk2["ms_step_and_drain"]},
                        "pcie_bytes_per_decision": (ptk2["pcie_h2d_bytes_per_tick"] + ptk2["pcie_d2h_bytes_per_tick"]) * ptk2["ticks"] / ptk2["decisions"],
                        "pcie_GB_per_s_both_ways": (ptk2["pcie_h2d_bytes_per_tick"] + ptk2["pcie_d2h_bytes_per_tick"]) / (ptk2["ms_per_tick"] * 1e-3) / 1e9,
                        "polled_decisions_per_s": ptk2_polled["decisions_per_s"],
                        "with_eight_task_threads": {"what": "the same loop with 8 task threads beside it instead of R - 1 = 4 (the transport's decoding and the sinks are "
                                                            "what bounds it): two ticks in flight / one",
                                                    "decisions_per_s": ptk2_8["decisions_per_s"], "ms_per_tick": ptk2_8["ms_per_tick"],
                                                    "one_in_flight_decisions_per_s": ptk1_8["decisions_per_s"],
                                                    "pcie_GB_per_s_both_ways": (ptk2_8["pcie_h2d_bytes_per_tick"] + ptk2_8["pcie_d2h_bytes_per_tick"]) / (ptk2_8["ms_per_tick"] * 1e-3) / 1e9},
                        "column_inbound_decisions_per_s": ptck2["decisions_per_s"], "column_inbound_ms_per_tick": ptck2["ms_per_tick"],
                        "plain_bus_decisions_per_s": pt2["decisions_per_s"], "plain_bus_ms_per_tick": pt2["ms_per_tick"],
                        "rows_on_the_general_path": ptk2["rows_general"] + ptck2["rows_general"] + pt2["rows_general"]}},
                "wire_decode": {
                    "what": "the same loop and tasks, but the peers' AppendResponses / HeartbeatResponses arrive as what a stock josefine peer sends - "
                            "LengthDelimitedCodec frames around serde_json(Message), one byte stream per connection (src/raft/tcp.rs:40-51,139-170) - "
                            "and the connection tasks run host/formats.hpp's decode_message on every frame before they write the row (the other "
                            "figures' decoder writes rows it computes: what a columnar transport would hand over)",
                    "decisions_per_s": ptw["decisions_per_s"], "ms_per_tick": ptw["ms_per_tick"], "ticks": ptw["ticks"],
                    "ms_per_tick_parts": {"transport_decode_into_pinned_columns": ptw["ms_fill"], "submit_commit": ptw["ms_submit"],
                                          "step_begin_and_previous_outputs": ptw["ms_step_and_drain"]},
                    "wire_bytes_decoded_per_tick": ptw["wire_bytes_decoded_per_tick"], "frames_per_tick": ptw["rows_in_per_tick"] - G,
                    "decode_ns_per_frame_per_thread": ptw["ms_fill"] * 1e6 * (ptw["task_threads_beside_each_loop"] + 1) / max(ptw["rows_in_per_tick"] - G, 1),
                    "rows_on_the_general_path": ptw["rows_general"]}},
            "one_loop": {"what": "ONE loop (one host thread, one engine) owns every partition: nothing overlaps",
                         "decisions_per_s": d1["decisions_per_s"],
                         "ms_per_tick": {"transport_decode_into_pinned_columns": d1["ms_fill"], "submit_commit_validation": d1["ms_submit"],
                                         "step_node_and_drains": d1["ms_step_and_drain"], "total": d1["ms_per_tick"]}},
            "loop_only_decisions_per_s": d1["decisions"] / (loop_ms * d1["ticks"] / 1e3),
            "rows_in_per_tick": d["rows_in_per_tick"], "rows_on_the_general_path": d["rows_general"],
            "fsm_rows_per_tick": d["fsm_rows_per_tick"],
            "pcie_bytes_per_tick": {"h2d": d["pcie_h2d_bytes_per_tick"], "d2h": d["pcie_d2h_bytes_per_tick"]},
            "pcie_bytes_per_decision": (d["pcie_h2d_bytes_per_tick"] + d["pcie_d2h_bytes_per_tick"]) * d["ticks"] / d["decisions"],
            "with_two_host_copies_decisions_per_s": copy["decisions_per_s"],
            "column_inbound": {
                "what": "the R - 1 followers are batched peers: each ships its answers as ONE column of JG_ANSWER words "
                        "(jg_node_inbox_columns, written in place), only the ClientRequests are rows",
                "decisions_per_s": colm["decisions_per_s"], "ms_per_tick": colm["ms_per_tick"], "loops": Lc, "decisions_per_s_by_loops": colm_by_loops,
                "one_loop_decisions_per_s": colm1["decisions_per_s"], "one_loop_ms_per_tick": colm1["ms_per_tick"],
                "loop_only_decisions_per_s": colm1["decisions"] / ((colm1["ms_submit"] + colm1["ms_step_and_drain"]) * colm1["ticks"] / 1e3),
                "pcie_bytes_per_tick": {"h2d": colm["pcie_h2d_bytes_per_tick"], "d2h": colm["pcie_d2h_bytes_per_tick"]},
                "pcie_bytes_per_decision": (colm["pcie_h2d_bytes_per_tick"] + colm["pcie_d2h_bytes_per_tick"]) * colm["ticks"] / colm["decisions"]},
            "round2_loop_decisions_per_s": old["decisions_per_s"],
            "round2_loop": "BatchedEventLoop.dense = false: every row and one Tick ROW per partition through jg_submit + jg_step",
            "speedup_over_round2_loop": d1["decisions_per_s"] / old["decisions_per_s"],
            "target": "VERDICT r2 asked for >= 1e9/s through the loop: " +
                      ("reached" if max(d["decisions_per_s"], colm["decisions_per_s"]) >= 1e9 else "not reached") +
                      f" ({max(d['decisions_per_s'], colm['decisions_per_s']):.3g}/s best of row inbound on {L} loop(s) / column inbound on {Lc}; "
                      "row inbound moves 46 B over PCIe per decision, column inbound 30 B: DESIGN.md 'Through the Apply surface')",
        },
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                     "kernel": f"k_leader_node_tick<{R}, FSM> (the dense leader half under jg_step_node; HIP event pairs, jg_kernel_timing)",
                     "alg_bytes_per_launch": lb * G, "avg_launch_us": k_us, "launches_timed": d1["leader_kernel_launches"],
                     "frac_of_measured_copy": ach / 6290.0,
                     "note": "the loop as a whole is PCIe- and host-bound, not HBM-bound: the kernel is a few percent of a tick"},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(R, args.seed, args.cpu_budget)
    print(json.dumps(out), flush=True)


def single_process_main(args):
    """--single-process: ONE process, ONE engine handle, N shards (jg_config.n_devices /
    device_ids) - the shape a josefine process driving N GPUs has (one event loop owns the handle,
    src/raft/server.rs:103-165).  Same workload per GPU as the process-per-GPU mode, same JSON line;
    every shard's ack stream lives in its own device's memory and a step is one
    jg_step_dense_acks_shards call (one launch per shard, each from the shard's host thread)."""
    import numpy as np
    import torch  # noqa: F401  (first: the engine library then resolves against the same HIP runtime)
    from josefine_amd import BatchedRaft
    from josefine_amd.traces import elect_all

    N, G, R, K, W = args.gpus, args.groups, args.replicas, args.steps, args.warmup
    devs = [0] * N if args.alias_devices else list(range(N))
    eng = BatchedRaft(G * N, R, seed=args.seed, device_ids=devs)
    assert eng.n_shards == N
    elect_all(eng)
    eng.drain_messages(), eng.drain_applies()
    api = eng.api
    shards = [eng.shard(d) for d in range(N)]
    tick_bytes = R * G * 8
    bufs = []
    for sh in shards:  # the whole stream resident in each device's HBM before the timed region
        sim, buf = sh.alloc(tick_bytes), sh.alloc(tick_bytes * (W + K))
        for t in range(W + K):
            sh._check(api.synth_fill_acks_device(sh._h, args.mode, t, sim, C.c_void_p(buf.value + t * tick_bytes)))
        bufs.append(buf)
    eng._check(api.sync(eng._h))
    ptrs = [(C.c_void_p * N)(*[C.c_void_p(b.value + t * tick_bytes) for b in bufs]) for t in range(W + K)]
    for t in range(W):
        eng._check(api.step_dense_acks_shards(eng._h, ptrs[t], 1))
    eng._check(api.sync(eng._h))
    c0 = eng.counters()
    t0 = time.perf_counter()
    eng._check(api.timer_start(eng._h))
    for t in range(W, W + K):
        eng._check(api.step_dense_acks_shards(eng._h, ptrs[t], 1))
    ev_ms = C.c_float(0)
    eng._check(api.timer_stop(eng._h, C.byref(ev_ms)))  # the slowest shard's stream
    wall = time.perf_counter() - t0
    decisions = eng.counters()["decisions"] - c0["decisions"]
    if args.mode == 0:
        head, commit, fault = eng.read("head"), eng.read("commit"), eng.read("fault")
        assert (head == W + K).all() and (commit == W + K - 1).all() and not fault.any(), "steady-state closed form violated"
    launch_s = ev_ms.value / 1e3 / K
    alg = alg_bytes_per_group_step(R, args.mode) * G
    achieved = alg / launch_s / 1e9
    out = {
        "metric": "Raft quorum decisions/sec over N partitions; achieved HBM GB/s vs roofline",
        "value": decisions / wall, "unit": "decisions/s", "n_gpus": N, "steps": K, "warmup": W,
        "ms_per_step": wall * 1e3 / K, "ms_per_step_events": ev_ms.value / K, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload_name(G, R, args.mode, 0), "partitions_per_gpu": G, "replicas": R,
                   "partitions_total": G * N,
                   "parallelism": f"1 process, 1 engine handle, {N} shard(s) on devices {devs} (jg_config.n_devices), no collective"},
        "group_steps_per_s": G * N * K / wall,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": f"k_leader_tick_dense<{R}> (per shard; the slowest shard's stream)",
                     "alg_bytes_per_launch": alg, "avg_launch_us": launch_s * 1e6, "frac_of_measured_copy": achieved / 6290.0},
    }
    if not args.no_cpu_baseline and N == 1:
        out["cpu_baseline"] = cpu_baseline(R, args.seed, args.cpu_budget)
    print(json.dumps(out), flush=True)


def secondary_lines(args):
    """The other modes, measured in the same invocation so that the driver's BENCH file carries them: short runs of
    bench.py itself (a sub-process each, a time cap each), reduced to the figures the review tracks.  Every entry says
    which command it is; a mode that fails or runs out of time is reported as such, never silently dropped."""
    me = os.path.abspath(__file__)
    R = args.replicas

    def run(name, extra, cap=150):
        cmd = [sys.executable, me, "--no-cpu-baseline", "--no-secondary", "--groups", str(args.groups), "--replicas", str(R),
               "--seed", hex(args.seed)] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=cap, env={**os.environ, "JG_SELF_LAUNCHED": "1"})
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                return {"command": " ".join(cmd[1:]), "error": (r.stderr or r.stdout)[-400:]}
            return {"command": " ".join(["bench.py"] + cmd[2:]), "seconds": time.perf_counter() - t0, "line": json.loads(lines[-1])}
        except subprocess.TimeoutExpired:
            return {"command": " ".join(cmd[1:]), "error": f"no result within {cap} s"}

    out = {}
    x = run("closed_loop", ["--cluster", "--steps", "100", "--warmup", "10"])
    out["closed_loop"] = x if "error" in x else {
        "command": x["command"], "round_us": x["line"]["ms_per_step_events"] * 1e3 if "ms_per_step_events" in x["line"] else x["line"]["ms_per_step"] * 1e3,
        "frac": x["line"]["roofline"]["frac"], "leader_kernel_us": x["line"]["roofline"]["leader_kernel"]["avg_launch_us"],
        "leader_kernel_frac": x["line"]["roofline"]["leader_kernel"]["frac"], "decisions_per_s": x["line"]["value"]}
    # configs[4] as specified, the STATIONARY trace (failures + re-creation after 10 rounds): the election vocabulary as
    # mailbox words (JG_CLUSTER_OPT_VOTE_WORDS) and, beside it, everything as rows; then rounds 2-4's trace (no repairs:
    # the leaderless fraction grows through the region) for continuity with their lines
    def routed(x):
        return x if "error" in x else {
            "command": x["command"], "round_ms": x["line"]["ms_per_step"], "frac": x["line"]["roofline"]["frac"], "vote_words": x["line"]["vote_words"],
            "rows_routed_per_round": x["line"]["rows_routed_per_round"], "leaderless_fraction": x["line"]["leaderless_fraction"],
            "stationary": x["line"]["config"]["stationary"].split(":")[0].split(" ")[0], "round_ms_by_window": [w["ms_per_round"] for w in x["line"]["ms_per_round_by_leaderless_fraction"]],
            "elections_won_through_the_transport": x["line"].get("elections_won_through_the_transport"), "synthetic_votes": x["line"].get("synthetic_votes"),
            "decisions_per_s": x["line"]["value"]}
    out["routed_round"] = routed(run("routed_round", ["--cluster", "--failures", "1", "--steps", "40", "--warmup", "10", "--vote-words", "1"]))
    out["routed_round_rows_only"] = routed(run("routed_round_rows_only", ["--cluster", "--failures", "1", "--steps", "40", "--warmup", "10", "--vote-words", "0"]))
    out["routed_round_no_repairs"] = routed(run("routed_round_no_repairs", ["--cluster", "--failures", "1", "--steps", "40", "--warmup", "10", "--vote-words", "1", "--repair-after", "0"]))
    x = run("any_leader", ["--cluster", "--any-leader", "--replicas", "3", "--steps", "100", "--warmup", "10"])
    out["per_partition_leadership"] = x if "error" in x else {
        "command": x["command"], "round_us": x["line"]["ms_per_step_events"] * 1e3, "frac": x["line"]["roofline"]["frac"],
        "elections": x["line"]["elections"], "decisions_per_s": x["line"]["value"]}
    # (R = 3, the failing groups re-created - JG_CMD_RECREATE -, any number of times: every election WON through the transport,
    # every vote real, the winners append: a stationary trace; without --recreate it is round 4's, Q8 and a withdrawing client)
    x = run("any_leader_failures", ["--cluster", "--any-leader", "--replicas", "3", "--failures", "1", "--recreate", "--steps", "60", "--warmup", "30"])
    out["per_partition_leadership_failures"] = x if "error" in x else {
        "command": x["command"], "round_ms": x["line"]["ms_per_step"], "frac": x["line"]["roofline"]["frac"],
        "stationary": x["line"]["config"].get("stationary", "no").split(":")[0],
        "winners_appending_again_fraction_of_failed_groups": x["line"].get("winners_appending_again_fraction_of_failed_groups"),
        "elections_won_after_failures": x["line"]["elections_won_after_failures"], "leaderless_fraction": x["line"]["leaderless_fraction"],
        "rows_left_for_the_host": x["line"]["rows_left_for_the_host"], "decisions_per_s": x["line"]["value"]}
    # (round 6: FIVE nodes elect their leaders through the device transport like three do - 1 M elections, 40 M routed rows in four rounds)
    x = run("any_leader_x5", ["--cluster", "--any-leader", "--replicas", "5", "--steps", "60", "--warmup", "10"])
    out["per_partition_leadership_five_nodes"] = x if "error" in x else {
        "command": x["command"], "round_us": x["line"]["ms_per_step_events"] * 1e3, "frac": x["line"]["roofline"]["frac"],
        "elections": x["line"]["elections"], "decisions_per_s": x["line"]["value"]}
    x = run("failures_tick", ["--failures", "1", "--steps", "160", "--warmup", "64"])  # (profiles/*/bench_failures_1pct.json's command)
    out["failures_tick"] = x if "error" in x else {
        "command": x["command"], "tick_ms": x["line"]["ms_per_step"], "dense_kernel_us": x["line"]["roofline"]["avg_launch_us"],
        "tick_ms_without_final_flush": x["line"]["drain_pipeline"]["ms_per_step_without_final_flush"], "final_flush_ms": x["line"]["drain_pipeline"]["final_flush_ms"],
        "frac": x["line"]["roofline"]["frac"], "decisions_per_s": x["line"]["value"]}
    x = run("event_loop", ["--event-loop", "--steps", "12", "--warmup", "3", "--loops", "4"], cap=240)
    out["event_loop"] = x if "error" in x else {
        "command": x["command"], "decisions_per_s": x["line"]["value"], "loops": x["line"]["event_loop"]["loops"],
        "one_loop_decisions_per_s": x["line"]["event_loop"]["one_loop"]["decisions_per_s"],
        "one_loop_pipelined_decisions_per_s": x["line"]["event_loop"]["one_loop_pipelined"]["decisions_per_s"],
        "one_loop_pipelined_column_inbound_decisions_per_s": x["line"]["event_loop"]["one_loop_pipelined"]["column_inbound_decisions_per_s"],
        "one_loop_with_tasks_decisions_per_s": x["line"]["event_loop"]["one_loop_with_transport_and_consumer_tasks"]["decisions_per_s"],
        "one_loop_with_tasks_column_inbound_decisions_per_s": x["line"]["event_loop"]["one_loop_with_transport_and_consumer_tasks"]["column_inbound_decisions_per_s"],
        "column_inbound_decisions_per_s": x["line"]["event_loop"]["column_inbound"]["decisions_per_s"],
        "rows_on_the_general_path": x["line"]["event_loop"]["rows_on_the_general_path"],
        "pcie_bytes_per_decision": x["line"]["event_loop"]["pcie_bytes_per_decision"],
        "compact_bus_decisions_per_s": x["line"]["event_loop"]["one_loop_with_transport_and_consumer_tasks"]["compact_bus"]["decisions_per_s"],
        "compact_bus_pcie_bytes_per_decision": x["line"]["event_loop"]["one_loop_with_transport_and_consumer_tasks"]["compact_bus"]["pcie_bytes_per_decision"],
        # (round 6: two ticks in flight - JG_NODE_KEEP - on the same ONE loop)
        "two_ticks_in_flight_decisions_per_s": x["line"]["event_loop"]["one_loop_with_transport_and_consumer_tasks"]["compact_bus"]["two_ticks_in_flight"]["decisions_per_s"],
        "two_ticks_in_flight_polled_decisions_per_s": x["line"]["event_loop"]["one_loop_with_transport_and_consumer_tasks"]["compact_bus"]["two_ticks_in_flight"]["polled_decisions_per_s"],
        "two_ticks_in_flight_column_inbound_decisions_per_s": x["line"]["event_loop"]["one_loop_with_transport_and_consumer_tasks"]["compact_bus"]["two_ticks_in_flight"]["column_inbound_decisions_per_s"],
        "two_ticks_in_flight_eight_task_threads_decisions_per_s": x["line"]["event_loop"]["one_loop_with_transport_and_consumer_tasks"]["compact_bus"]["two_ticks_in_flight"]["with_eight_task_threads"]["decisions_per_s"]}
    return out


def self_launch(args):
    """`python bench.py --gpus N` (N > 1, no launcher around it): re-exec this script under
    torch.distributed.run with N ranks on this node, one per GPU - the command the driver uses.  On a
    box with fewer than N devices the ranks share devices (rank r -> device r mod #devices) and the
    barrier / reductions go over gloo (RCCL refuses two ranks on one device); the JSON line then says
    `"devices_aliased": true`: an exercise of the N-rank path, not a scaling number."""
    import socket
    import subprocess

    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    env = dict(os.environ)
    if ndev < args.gpus:
        env.setdefault("JG_BENCH_BACKEND", "gloo")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without a launcher: {' '.join(cmd)} ({ndev} device(s) visible)", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def devices_config(args, world):
    """config entries that say where the ranks ran"""
    devs = getattr(args, "devices_bound", [0])
    polled = os.environ.get("HSA_ENABLE_INTERRUPT") == "0"
    return {"devices": devs, "devices_aliased": len(set(devs)) < world,
            "host_wait": ("completion signals polled (HSA_ENABLE_INTERRUPT=0" + (", set by bench.py" if os.environ.get("JG_BENCH_POLLING_DEFAULTED") else "") + ")")
            if polled else "completion signals by interrupt (the runtime's default)"}


def main():
    # A short-lived measuring process: the cyclic collector stays off.  A generation-2 pass over the few thousand objects a
    # pre-built trace holds took 36 ms in the middle of a timed region (one routed round of 0.4 ms measured as 36.4 ms,
    # always the same round: the allocation count that triggers it is deterministic) - reference counting frees what matters.
    import gc
    gc.disable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--groups", type=int, default=1_000_000, help="partitions per GPU")
    ap.add_argument("--replicas", type=int, default=5)
    ap.add_argument("--mode", type=int, default=0, help="0 steady-state (#3/#4), 1 ragged (#2)")
    ap.add_argument("--ticks-per-launch", type=int, default=1,
                    help="T>1: temporal fusion (jg_step_dense_acks_device_n), state read/written once per T ticks")
    ap.add_argument("--failures", type=int, default=0,
                    help="percent of groups per tick whose leader crashes and is re-elected (BASELINE configs[4]; "
                         "RESTART/Timeout/VoteResponse rows through jg_submit + jg_step)")
    ap.add_argument("--cluster", action="store_true",
                    help="closed loop: all R replicas of every partition on this GPU as R engines exchanging dense "
                         "mailbox columns (jg_step_dense_leader / jg_step_dense_follower); secondary measurement")
    ap.add_argument("--any-leader", action="store_true",
                    help="with --cluster: per-partition leadership (JG_CLUSTER_ANY_LEADER) - leaders elected through the device transport on "
                         "every node, every node runs both halves over the cluster's mailbox columns")
    ap.add_argument("--vote-words", type=int, choices=[0, 1], default=None,
                    help="--cluster with --failures: JG_CLUSTER_OPT_VOTE_WORDS - an election's traffic as mailbox words (csrc/jg_votes.h) instead of "
                         "rows.  Default: 1 for configs[4] as specified (single lead: campaigns that are refused recur at every election timeout - "
                         "0.39 against 0.47 ms per round), 0 with --any-leader at R = 3 (whole groups restart: every campaign is WON at once, the "
                         "traffic is the winners' Heartbeats - rows - and the word passes buy nothing: 0.25 against 0.24 ms), 1 with --any-leader "
                         "at R >= 5 (a campaign's 4 x 4 copies and their answers are most of a round's rows there: 0.378 against 0.415 ms)")
    ap.add_argument("--drain-applies", type=int, choices=[0, 1], default=1,
                    help="--cluster --failures: hand the rounds' FSM rows (the Apply ranges of repaired followers) to the host every round")
    ap.add_argument("--synthetic-votes", type=int, choices=[0, 1], default=0,
                    help="--cluster --any-leader: 1 = the initial leaders are seated with injected VoteResponses instead of being elected "
                         "through the device transport (what R > 3 needed until round 6; kept as an A/B)")
    ap.add_argument("--recreate", action="store_true",
                    help="--cluster --any-leader --failures p: the failing group comes back on EMPTY stores (JG_CMD_RECREATE) - the campaign is won "
                         "through the transport, the winner appends, groups may fail again: a stationary trace with no synthetic vote")
    ap.add_argument("--offer-lists", type=int, choices=[0, 1], default=0,
                    help="--cluster --failures (single lead): 1 = the client's proposals withdrawn from a failed partition and offered again to a "
                         "re-created one by two launches per round (jg_dense_cluster_withdraw_appends / _offer_appends: round 5); 0 = one proposal per "
                         "partition and round throughout, taken only where replica 0 leads (the same appends: a failed partition has no leader there)")
    ap.add_argument("--repair-after", type=int, default=10,
                    help="--cluster --failures (single lead): rounds after which a failed partition is repaired (every replica restarts, "
                         "replica 0 is re-seated) - the stationary configs[4] trace; 0: never (the leaderless fraction grows)")
    ap.add_argument("--leadership", choices=["blocked", "interleaved"], default="blocked",
                    help="with --any-leader: node g*R/G (contiguous blocks: what an adapter that numbers its partitions by preferred "
                         "leader gets) or node g %% R leads partition g")
    ap.add_argument("--single-process", action="store_true",
                    help="one process, one engine handle over --gpus shards (jg_config.n_devices) instead of one "
                         "process per GPU; run it directly, not under torch.distributed.run")
    ap.add_argument("--event-loop", action="store_true",
                    help="decisions/s through the reference's driver surface: josefine::BatchedEventLoop (C++) over jg_step_node, "
                         "host rows in, fsm_tx rows + outbox columns out")
    ap.add_argument("--loops", default="4,8",
                    help="with --event-loop: the process hosts the partitions on this many event loops (one host thread, one engine, "
                         "one HIP stream each); a comma-separated list = measure each, report the best; the one-loop figures are "
                         "reported beside them")
    ap.add_argument("--alias-devices", action="store_true",
                    help="with --single-process: put every shard on device 0 (exercises the multi-device path on a 1-GPU box)")
    ap.add_argument("--config", type=int, choices=[2, 3, 4], default=None,
                    help="BASELINE.json configs[N] presets for the scaling run: 2 = 1 M x 5 steady state per GPU (the default); "
                         "3 = 1.25 M x 3 per GPU (10 M x 3 over 8); 4 = 125 k x 5 per GPU (1 M x 5 over 8) as the routed cluster with "
                         "1 %%/round leader failures (--cluster --failures 1)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements the default N = 1 line carries (closed loop, routed round, event loop, "
                         "failure tick, per-partition leadership: a few seconds each, short runs of the modes themselves)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x6A6F736566696E65)
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.vote_words is None:
        args.vote_words = (1 if args.replicas >= 5 else 0) if args.any_leader else 1
    if args.config == 3:
        args.groups, args.replicas = 1_250_000, 3
    elif args.config == 4:
        args.groups, args.replicas, args.cluster, args.failures = 125_000, 5, True, max(args.failures, 1)
    if args.single_process:
        return single_process_main(args)
    if args.event_loop:
        if args.gpus != 1:
            raise SystemExit("--event-loop is a single-GPU measurement")
        return event_loop_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU)
        return self_launch(args)

    import torch  # first: the engine library then resolves against the same HIP runtime
    import torch.distributed as dist
    import numpy as np

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: every rank is one GPU's share, "
                         "launch exactly --gpus ranks (or run `python bench.py --gpus N` and let it launch them)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # RCCL ("nccl") is the backend of record; JG_BENCH_BACKEND=gloo exists only so that the
    # N>1 code path can be exercised with several ranks on a single-GPU box.
    backend = os.environ.get("JG_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    red_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    # which device ordinal every rank bound (in rank order): part of the JSON line, so that an N-GPU
    # number that was really measured on fewer devices says so
    args.devices_bound = [dev_index]
    if world > 1:
        got = [None] * world
        dist.all_gather_object(got, dev_index)
        args.devices_bound = [int(x) for x in got]
    print(f"[bench rank {rank}/{world}] bound to HIP device {dev_index} of {torch.cuda.device_count()}", file=sys.stderr, flush=True)
    if backend == "nccl" and len(set(args.devices_bound)) != world:  # RCCL = one rank per GPU: an aliased job is not an N-GPU number
        raise SystemExit(f"bench.py: {world} ranks bound devices {args.devices_bound}: every rank needs a GPU of its own "
                         "(JG_BENCH_BACKEND=gloo exists to exercise the N > 1 path on fewer devices)")

    if args.cluster:
        return cluster_main(args, torch, dist, rank, world, dev_index, red_dev)

    from josefine_amd import BatchedRaft
    from josefine_amd.traces import elect_all

    G, R, K, W = args.groups, args.replicas, args.steps, args.warmup
    eng = BatchedRaft(G, R, seed=args.seed, device_id=dev_index, group_base=rank * G)
    elect_all(eng)  # Timeout -> votes -> leader through the general kernel
    eng.drain_messages(), eng.drain_applies()
    api, h = eng.api, eng._h

    # pre-generate the whole ack stream in HBM: (W+K) ticks x [R][G] u64
    tick_bytes = R * G * 8
    sim = C.c_void_p()
    stream_buf = C.c_void_p()
    eng._check(api.device_alloc(h, tick_bytes, C.byref(sim)))
    eng._check(api.device_alloc(h, tick_bytes * (W + K), C.byref(stream_buf)))
    for t in range(W + K):
        eng._check(api.synth_fill_acks_device(h, args.mode, t, sim, C.c_void_p(stream_buf.value + t * tick_bytes)))
    eng._check(api.sync(h))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng._check(api.sync(h))

    T = max(1, args.ticks_per_launch) if not args.failures else 1
    fail_rows = None
    if args.failures:
        from josefine_amd.traces import failure_rows
        slots = eng.read("self_slot")
        # the failure trace, like the ack stream, is resident in HBM before the timed region:
        # one group-sorted device batch per tick (jg_step_device_rows, no host pass per tick)
        fail_rows = [eng.upload_rows(**failure_rows(args.seed, t, rank * G, G, R, eng.node_ids, slots, args.failures)[0])
                     for t in range(W + K)]

    DRAIN_EVERY = int(os.environ.get("JG_BENCH_DRAIN_EVERY", "16"))  # ticks per drain batch of the failure trace

    def run_ticks(t0_, t1_):
        """Apply ticks [t0_, t1_) — exactly t1_-t0_ steps — in launches of up to T ticks."""
        n_launch = 0
        t = t0_
        while t < t1_:
            n = min(T, t1_ - t)
            ptr = C.c_void_p(stream_buf.value + t * tick_bytes)
            if n == 1:
                eng._check(api.step_dense_acks_device(h, ptr))
            else:
                eng._check(api.step_dense_acks_device_n(h, ptr, n))
            if fail_rows is not None and fail_rows[t].n:
                eng.step_device_rows(fail_rows[t], now_ms=100 * (t + 1))
                if t % DRAIN_EVERY == DRAIN_EVERY - 1:
                    # the host consumes the outbound messages as it goes (pinned views): the batch
                    # whose transfer was started 16 ticks ago, then the next transfer is started
                    # (jg_drain_prefetch) - PCIe and host time overlap the device time of the next ticks
                    eng.drain_wait()   # back-pressure: never more than one batch ahead of our own output
                    consume()
                    eng.drain_prefetch()
            t += n
            n_launch += 1
        if fail_rows is not None:  # everything stepped is delivered before the clock stops
            tf = time.perf_counter()
            eng.drain_flush()
            consume()
            tail_s[0] = time.perf_counter() - tf  # (the region's last batch: its transfer overlaps nothing)
        return n_launch

    tail_s = [0.0]

    drained = {"messages": 0, "applies": 0, "faults": 0}

    def consume():
        drained["messages"] += len(eng.drain_messages(copy=False))
        drained["applies"] += len(eng.drain_applies(copy=False))
        drained["faults"] += len(eng.drain_faults())

    run_ticks(0, W)
    if fail_rows is not None:  # one more full drain cycle outside the timed region (pinned queues at their working size)
        eng.drain_flush()
        consume()
        eng._check(api.kernel_timing(h, 4))  # (every 4th dense launch: the event pair is not free in a tick of three small kernels)
    barrier()
    c0 = eng.counters()
    barrier()
    t0 = time.perf_counter()
    eng._check(api.timer_start(h))
    n_launches = run_ticks(W, W + K)
    ev_ms = C.c_float(0)
    eng._check(api.timer_stop(h, C.byref(ev_ms)))  # HIP events on the engine's stream (synchronises it)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0  # this rank's K steps are done; MAX over ranks below
    barrier()
    c1 = eng.counters()

    decisions = c1["decisions"] - c0["decisions"]
    # A timed region of a few hundred microseconds is mostly the launch latency of its first kernel and the completion
    # wake-up of its last (~45 us whatever K is): when K steps take less than 50 ms the region is REPEATED - the same K
    # steps over the next K ticks of the stream, barrier and synchronisation on both sides each time - and the MEDIAN
    # region is reported (`steps` stays K: one region; every region is listed in `timed_regions`).
    ticks_done = W + K
    regions = [(wall, ev_ms.value)]
    if T == 1 and not args.failures and args.mode == 0:
        all_short = torch.tensor([1.0 if wall < 0.05 else 0.0], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(all_short, op=dist.ReduceOp.MIN)  # (every rank takes the same number of regions)
        n_more = 8 if all_short.item() > 0 else 0
        for _ in range(n_more):
            for t in range(K):  # the next K ticks of the stream into the slots of W .. W+K-1
                eng._check(api.synth_fill_acks_device(h, 0, ticks_done + t, sim, C.c_void_p(stream_buf.value + (W + t) * tick_bytes)))
            barrier()
            tr0 = time.perf_counter()
            eng._check(api.timer_start(h))
            run_ticks(W, W + K)
            ev_r = C.c_float(0)
            eng._check(api.timer_stop(h, C.byref(ev_r)))
            torch.cuda.synchronize()
            regions.append((time.perf_counter() - tr0, ev_r.value))
            barrier()
            ticks_done += K
        if n_more:
            c1 = eng.counters()
            decisions = c1["decisions"] - c0["decisions"]
    if world > 1:  # MAX over ranks, region by region
        tr = torch.tensor(regions, dtype=torch.float64, device=red_dev)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        regions_all = [tuple(x) for x in tr.tolist()]
    else:
        regions_all = regions
    order = sorted(range(len(regions_all)), key=lambda i: regions_all[i][0])
    mid = order[(len(order) - 1) // 2]
    wall_own, ev_own = regions[mid]  # this rank's figures of the region that is the job's median
    wall, ev_ms = regions_all[mid][0], C.c_float(regions_all[mid][1])
    # parity property at full size (closed form of the steady-state stream, mode 0):
    # after T ticks every leader has head == T and commit == T-1, no faults.
    if args.mode == 0 and not args.failures:
        total = ticks_done
        head, commit, fault = eng.read("head"), eng.read("commit"), eng.read("fault")
        assert (head == total).all() and (commit == total - 1).all() and not fault.any(), \
            "steady-state closed form violated"

    # Secondary, clearly separated measurement: the same K ticks' worth of stream applied 16
    # ticks per launch (temporal fusion, jg_step_dense_acks_device_n).  Not the headline: a tick
    # is a unit of latency, and the headline kernel is the one whose bytes match B(R).
    batched = None
    if T == 1 and not args.failures and args.mode == 0 and K >= 32:
        TB = 16
        for t in range(K):  # the next K ticks into the buffer slots of W .. W+K-1
            eng._check(api.synth_fill_acks_device(h, 0, ticks_done + t, sim,
                                                  C.c_void_p(stream_buf.value + (W + t) * tick_bytes)))
        barrier()
        tb0 = time.perf_counter()
        t = 0
        while t < K:
            n = min(TB, K - t)
            eng._check(api.step_dense_acks_device_n(h, C.c_void_p(stream_buf.value + (W + t) * tick_bytes), n))
            t += n
        barrier()
        wall_b = time.perf_counter() - tb0
        head, commit = eng.read("head"), eng.read("commit")
        assert (head == ticks_done + K).all() and (commit == ticks_done + K - 1).all(), "closed form violated (batched ticks)"
        dec_b = float(eng.counters()["decisions"] - c1["decisions"])
        if world > 1:
            tb = torch.tensor([wall_b], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            tdb = torch.tensor([dec_b], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tdb, op=dist.ReduceOp.SUM)
            wall_b, dec_b = tb[0].item(), tdb[0].item()
        batched = {"ticks_per_launch": TB, "steps": K, "ms_per_step": wall_b * 1e3 / K,
                   "decisions_per_s": dec_b / wall_b,
                   "bytes_moved_per_group_step": 8 * R + 28 / TB,
                   "note": "same results bit for bit; state read/written once per launch"}

    decisions = decisions / len(regions)  # (every region takes the same decisions: K steps of the same stream)
    per_rank = None
    if world > 1:
        td = torch.tensor([float(decisions)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(td, op=dist.ReduceOp.SUM)
        ev_max_ms, decisions_all = ev_ms.value, td[0].item()
        mine = [None] * world
        dist.all_gather_object(mine, (float(decisions) / wall_own, ev_own * 1e3 / n_launches))
        per_rank = {"decisions_per_s": [m[0] for m in mine], "avg_launch_us": [m[1] for m in mine]}
    else:
        ev_max_ms, decisions_all = ev_ms.value, float(decisions)

    if rank == 0:
        launch_s = (ev_own / 1e3) / n_launches  # average dense-kernel launch on this rank's stream
        ticks_per_launch = K / n_launches
        alg = alg_bytes_per_group_step(R, args.mode) * G * ticks_per_launch
        achieved = alg / launch_s / 1e9
        traffic = pmc_traffic(f"G{G}_R{R}_mode{args.mode}" + ("" if T == 1 else f"_T{T}")
                              + (f"_failures{args.failures}" if args.failures else ""))
        out = {
            "metric": "Raft quorum decisions/sec over N partitions; achieved HBM GB/s vs roofline",
            "value": decisions_all / wall,
            "unit": "decisions/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": wall * 1e3 / K,
            "ms_per_step_events": ev_max_ms / K,  # the same K steps by HIP events on the engine's stream (max over ranks)
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": workload_name(G, R, args.mode, args.failures),
                "partitions_per_gpu": G, "replicas": R, "partitions_total": G * world,
                "parallelism": f"{world} independent shard(s), no collective", **devices_config(args, world),
            },
            "group_steps_per_s": G * world * K / wall,
            "timed_regions": {"n": len(regions_all), "reported": "median" if len(regions_all) > 1 else "the one region",
                              "ms_per_step_each": [r[0] * 1e3 / K for r in regions_all],
                              "ms_per_step_events_each": [r[1] / K for r in regions_all],
                              "why": "K steps take less than 50 ms: the region is repeated over the following ticks of the stream and "
                                     "the median is reported (a region's first launch and last wake-up cost ~45 us whatever K is)"
                                     if len(regions_all) > 1 else None},
            "per_rank": per_rank,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": f"k_leader_tick_dense<{R}>" if T == 1 else f"k_leader_tick_dense_n<{R}> (T={T} ticks/launch)",
                "alg_bytes_per_launch": alg, "ticks_per_launch": ticks_per_launch,
                "avg_launch_us": launch_s * 1e6, "peak_basis": "8.0 TB/s spec (6.29 TB/s measured copy)",
                "frac_of_measured_copy": achieved / 6290.0,
                "alg_bytes_per_group_step": alg_bytes_per_group_step(R, args.mode),
                # the same launch priced with SURVEY.md's B(R) = 24R + 36 (8-byte absolute progress heads
                # read and written every tick); > 1 possible: the engine stores them delta-packed
                "survey_priced": {"bytes_per_group_step": survey_bytes_per_group_step(R),
                                  "achieved": survey_bytes_per_group_step(R) * G * ticks_per_launch / launch_s / 1e9,
                                  "frac": survey_bytes_per_group_step(R) * G * ticks_per_launch / launch_s / 1e9 / HBM_PEAK_GBS},
            },
        }
        if hasattr(api, "calibrate_stream") and T == 1 and not args.failures:
            # what a launch of this shape costs on this machine with no Raft logic in it: a plain
            # streaming kernel with the same bytes per group, grid, workgroup size and stream
            cal_us = C.c_float(0)
            eng._check(api.calibrate_stream(h, max(K, 50), C.byref(cal_us)))
            out["roofline"]["stream_ceiling"] = {
                "kernel": f"k_stream_calib<{R}> (same bytes per group, no logic)",
                "avg_launch_us": cal_us.value, "achieved": alg / (cal_us.value * 1e-6) / 1e9,
                "unit": "GB/s", "kernel_vs_ceiling": cal_us.value / (launch_s * 1e6)}
        if args.failures:
            # several kernels per tick (dense + deferred-group replay + k_apply_rows) and host
            # drains inside the timed region: the stream's event time is the whole tick; the
            # dominant kernel is priced with its own event pairs (jg_kernel_timing: the last 256
            # launches of k_leader_tick_dense, ~half of the groups dead by then)
            k_us, k_n = C.c_float(0), C.c_uint32(0)
            eng._check(api.kernel_timing_read(h, C.byref(k_us), C.byref(k_n)))
            alg1 = alg_bytes_per_group_step(R) * G
            ach = alg1 / (k_us.value * 1e-6) / 1e9 if k_us.value else 0.0
            out["roofline"] = {
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic, "kernel": f"k_leader_tick_dense<{R}> under failures (dead / deferred groups in every wave)",
                "alg_bytes_per_launch": alg1, "avg_launch_us": k_us.value, "launches_timed": k_n.value,
                "frac_of_measured_copy": ach / 6290.0,
                "note": "every group's bytes are priced, dead or not: the kernel loads them before it can know; launches "
                        "that overlap the pipelined drain's PCIe gathers run ~4x slower than the 13.8 us the same kernel "
                        "takes alone (profiles/r02/kernel_stats_failures_1pct.csv)"}
            out["tick_us"] = launch_s * 1e6
            out["rows_delivered_to_host"] = drained
            # the region ends with a full drain - everything stepped is in host memory before the clock stops - whose transfer
            # overlaps no tick: a fixed cost (3-4 ms), so ms_per_step depends on --steps (0.081 at 160, 0.118 at 96).  Beside it: the
            # tick with that tail taken out (what a longer run converges to) and the dense kernel alone.
            out["drain_pipeline"] = {"ticks_per_batch": DRAIN_EVERY, "final_flush_ms": tail_s[0] * 1e3,
                                     "ms_per_step_without_final_flush": (wall - tail_s[0]) * 1e3 / K, "dense_kernel_us": k_us.value}
        if batched is not None:
            out["batched_ticks"] = batched
        if not args.no_cpu_baseline:  # (rank 0, after the timed regions and their barriers; the other ranks wait below)
            out["cpu_baseline"] = cpu_baseline(R, args.seed, args.cpu_budget)
        if world == 1 and not args.no_secondary and not args.failures and T == 1 and args.mode == 0 and \
                (G >= 500_000 or os.environ.get("JG_BENCH_SECONDARY")):  # (small test runs: only on request)
            del eng  # (the engine's memory goes back before the other modes build theirs)
            out["secondary"] = secondary_lines(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
