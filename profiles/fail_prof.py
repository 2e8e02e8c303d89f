"""Where a tick of BASELINE configs[4] (1 %/tick leader failures) spends its time on the host:
enqueue of the two steps, device work (sync), and the three drains, as bench.py --failures runs
them (zero-copy views, drain every 16 ticks)."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from josefine_amd import BatchedRaft  # noqa: E402
from parity import elect_all, DeviceSynth  # noqa: E402
from failures import failure_rows  # noqa: E402

G, R, SEED = 1_000_000, 5, 0x6A6F736566696E65
DRAIN = int(os.environ.get("DRAIN_EVERY", "16"))
PIPE = os.environ.get("PIPE", "1") == "1"  # jg_drain_prefetch pattern (bench.py --failures) vs synchronous drains
e = BatchedRaft(G, R, seed=SEED)
elect_all(e)
e.drain_messages(), e.drain_applies()
synth = DeviceSynth(e)
slots = e.read("self_slot")
N = int(os.environ.get("TICKS", "80"))
rows = [e.upload_rows(**failure_rows(SEED, t, 0, G, R, e.node_ids, slots, 1)[0]) for t in range(N)]
tt = {"dense": 0.0, "rows": 0.0, "sync": 0.0, "drain_msg": 0.0, "drain_fsm": 0.0, "drain_fault": 0.0, "prefetch": 0.0}
t_start = None
nmsg = 0
for t in range(N):
    synth.fill(0, t)
    e._check(e.api.sync(e._h))
    a = time.perf_counter()
    e._check(e.api.step_dense_acks_device(e._h, synth.acks))
    b = time.perf_counter()
    e.step_device_rows(rows[t], 100 * (t + 1))
    c = time.perf_counter()
    if t >= 16:
        tt["dense"] += b - a
        tt["rows"] += c - b
    if t == 16:
        t_start = time.perf_counter()
    if t % DRAIN == DRAIN - 1:
        if not PIPE:
            e._check(e.api.sync(e._h))
        d = time.perf_counter()
        m = e.drain_messages(copy=False)
        f = time.perf_counter()
        e.drain_applies(copy=False)
        g = time.perf_counter()
        e.drain_faults()
        h = time.perf_counter()
        if PIPE:
            e.drain_prefetch()
            tt["prefetch"] += time.perf_counter() - h
        if t >= 16:
            nmsg += len(m)
            tt["sync"] += d - c
            tt["drain_msg"] += f - d
            tt["drain_fsm"] += g - f
            tt["drain_fault"] += h - g
e._check(e.api.sync(e._h))
print("pipelined" if PIPE else "synchronous", "wall per tick (ms):", round((time.perf_counter() - t_start) / (N - 16) * 1e3, 4))
print({k: round(v / (N - 16) * 1e3, 4) for k, v in tt.items()}, "ms per tick; rows/tick", rows[20].n,
      "msg rows/tick", nmsg // (N - 16))
print(np.bincount(e.read("fault"))[:6], e.counters())
