"""Where a failure tick's wall time goes: the host's issue time per tick (jg_step_dense_acks_device +
jg_step_device_rows, no drains) against the device time of the same ticks (events) - is the tick host-bound?"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: F401
from josefine_amd import BatchedRaft
from josefine_amd.traces import elect_all, failure_rows

G, R, K = 1_000_000, 5, 64
eng = BatchedRaft(G, R, seed=1)
elect_all(eng)
eng.drain_messages(), eng.drain_applies()
api, h = eng.api, eng._h
tick_bytes = R * G * 8
sim, buf = C.c_void_p(), C.c_void_p()
eng._check(api.device_alloc(h, tick_bytes, C.byref(sim)))
eng._check(api.device_alloc(h, tick_bytes * K, C.byref(buf)))
for t in range(K):
    eng._check(api.synth_fill_acks_device(h, 0, t, sim, C.c_void_p(buf.value + t * tick_bytes)))
slots = eng.read("self_slot")
rows = [eng.upload_rows(**failure_rows(1, t, 0, G, R, eng.node_ids, slots, 1)[0]) for t in range(K)]
eng._check(api.sync(h))
for what in ("dense only", "rows only", "both", "both"):
    eng._check(api.sync(h))
    eng._check(api.timer_start(h))
    t0 = time.perf_counter()
    for t in range(K):
        if what != "rows only":
            eng._check(api.step_dense_acks_device(h, C.c_void_p(buf.value + t * tick_bytes)))
        if what != "dense only":
            eng.step_device_rows(rows[t], now_ms=100 * (t + 1))
    t_issue = time.perf_counter() - t0
    ms = C.c_float(0)
    eng._check(api.timer_stop(h, C.byref(ms)))
    t_all = time.perf_counter() - t0
    print(f"{what:10s}: host issue {t_issue / K * 1e6:6.1f} us per tick, device (events) {ms.value / K * 1e3:6.1f} us per tick, wall {t_all / K * 1e6:6.1f} us")
    eng.drain_flush(), eng.drain_messages(copy=False), eng.drain_applies(copy=False), eng.drain_faults()
