import sys, time, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from josefine_amd import BatchedRaft
from parity import elect_all, DeviceSynth
from failures import failure_rows
G, R, SEED = 1_000_000, 5, 0x6A6F736566696E65
e = BatchedRaft(G, R, seed=SEED); elect_all(e); e.drain_messages(); e.drain_applies()
synth = DeviceSynth(e); slots = e.read("self_slot")
rows = [e.upload_rows(**failure_rows(SEED, t, 0, G, R, e.node_ids, slots, 1)[0]) for t in range(48)]
tt = {"dense": 0.0, "rows": 0.0, "drain": 0.0, "sync": 0.0}
for t in range(48):
    synth.fill(0, t)
    e._check(e.api.sync(e._h))
    a = time.perf_counter()
    e._check(e.api.step_dense_acks_device(e._h, synth.acks)); b = time.perf_counter()
    e.step_device_rows(rows[t], 100 * (t + 1)); c = time.perf_counter()
    e._check(e.api.sync(e._h)); d = time.perf_counter()
    if t % 16 == 15:
        e.drain_messages(); e.drain_applies(); e.drain_faults()
    f = time.perf_counter()
    if t >= 16:
        tt["dense"] += b - a; tt["rows"] += c - b; tt["sync"] += d - c; tt["drain"] += f - d
print({k: round(v / 32 * 1e3, 4) for k, v in tt.items()}, "ms per tick; rows/tick", rows[20].n)
print(np.bincount(e.read("fault"))[:6], e.counters())
