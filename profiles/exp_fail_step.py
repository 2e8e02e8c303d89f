"""Per-tick duration of the dense leader tick around one batch of configs[4] failure rows
(HIP events around each launch), to see what a sparse step does to the ticks after it."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from josefine_amd import BatchedRaft, capi  # noqa: E402
from josefine_amd.traces import elect_all, failure_rows  # noqa: E402

G, R, SEED = 1_000_000, 5, 0x6A6F736566696E65
MODE = os.environ.get("MODE", "fail")  # fail | restart | none
e = BatchedRaft(G, R, seed=SEED)
elect_all(e)
e.drain_messages(), e.drain_applies()
api, h = e.api, e._h
tb = R * G * 8
N = 14
sim, buf = C.c_void_p(), C.c_void_p()
e._check(api.device_alloc(h, tb, C.byref(sim)))
e._check(api.device_alloc(h, tb * N, C.byref(buf)))
for t in range(N):
    e._check(api.synth_fill_acks_device(h, 0, t, sim, C.c_void_p(buf.value + t * tb)))
slots = e.read("self_slot")
cols, n = failure_rows(SEED, 0, 0, G, R, e.node_ids, slots, 1)
if MODE == "restart":
    k = cols["kind"] == capi.CMD_RESTART
    cols = {kk: v[k] for kk, v in cols.items()}
rows = e.upload_rows(**cols)
every = [e.upload_rows(**failure_rows(SEED, t, 0, G, R, e.node_ids, slots, 1)[0]) for t in range(N)] if MODE == "every" else None
e._check(api.sync(h))
out = []
for t in range(N):
    ms = C.c_float(0)
    e._check(api.timer_start(h))
    e._check(api.step_dense_acks_device(h, C.c_void_p(buf.value + t * tb)))
    e._check(api.timer_stop(h, C.byref(ms)))
    out.append(round(ms.value * 1e3, 1))
    if MODE == "every":
        e.step_device_rows(every[t], now_ms=100 * (t + 1))
        e._check(api.sync(h))
    elif t == 3 and MODE != "none":
        e.step_device_rows(rows, now_ms=100)
        e._check(api.sync(h))
head, idg, fault, role = e.read("head"), e.read("id_gen"), e.read("fault"), e.read("role")
live = (fault == 0) & (role == capi.ROLE_LEADER)
print(MODE, "us per tick (dense + slow kernel behind it):", out)
print("live leaders", int(live.sum()), "of them FAST-shaped (id_gen == head+1):", int((idg[live] == head[live] + 1).sum()),
      "faults", np.bincount(fault)[:4], "failing groups", n)
