"""configs[4] diagnostic: run the failure trace as bench.py --failures does (dense tick + device-resident
failure rows per tick) and, every few ticks, time the dense launch alone (HIP events) and take a census
of the groups (role / fault / chain form / escaped lag fields) — what makes the dense kernel slow late
in the trace?"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from josefine_amd import BatchedRaft, capi  # noqa: E402
from josefine_amd.traces import elect_all, failure_rows  # noqa: E402

G, R, SEED = int(os.environ.get("G", 1_000_000)), 5, 0x6A6F736566696E65
N = int(os.environ.get("TICKS", 110))
PCT = int(os.environ.get("PCT", 1))
e = BatchedRaft(G, R, seed=SEED)
elect_all(e)
e.drain_messages(), e.drain_applies()
api, h = e.api, e._h
tb = R * G * 8
sim, buf = C.c_void_p(), C.c_void_p()
e._check(api.device_alloc(h, tb, C.byref(sim)))
e._check(api.device_alloc(h, tb * N, C.byref(buf)))
for t in range(N):
    e._check(api.synth_fill_acks_device(h, 0, t, sim, C.c_void_p(buf.value + t * tb)))
slots = e.read("self_slot")
rows = [e.upload_rows(**failure_rows(SEED, t, 0, G, R, e.node_ids, slots, PCT)[0]) for t in range(N)]
e._check(api.sync(h))
for t in range(N):
    ms = C.c_float(0)
    e._check(api.timer_start(h))
    e._check(api.step_dense_acks_device(h, C.c_void_p(buf.value + t * tb)))
    e._check(api.timer_stop(h, C.byref(ms)))
    e.step_device_rows(rows[t], now_ms=100 * (t + 1))
    if t % 16 == 15:
        e.drain_messages(copy=False), e.drain_applies(copy=False), e.drain_faults()
    if t % 10 == 9 or t < 4:
        role, fault, head, idg = e.read("role"), e.read("fault"), e.read("head"), e.read("id_gen")
        live = (fault == 0)
        lead = live & (role == capi.ROLE_LEADER)
        fastish = lead & (idg == head + 1)
        print(f"tick {t}: dense+slow {ms.value * 1e3:.1f} us; dead {int((~live).sum())} live leaders {int(lead.sum())} "
              f"(id_gen==head+1: {int(fastish.sum())}) followers {int((live & (role == 0)).sum())} candidates {int((live & (role == 1)).sum())} "
              f"faults {np.bincount(fault)[:9].tolist()} head==t+1: {int((head == t + 1).sum())}", flush=True)
