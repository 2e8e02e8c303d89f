"""Like exp_dense.py for the T-tick kernel (jg_step_dense_acks_device_n): us per tick."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from josefine_amd import BatchedRaft  # noqa: E402
from josefine_amd.traces import elect_all  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
K, W = 160, 32
e = BatchedRaft(G, R, seed=1)
elect_all(e)
e.drain_messages(), e.drain_applies()
api, h = e.api, e._h
tb = R * G * 8
sim, buf = C.c_void_p(), C.c_void_p()
e._check(api.device_alloc(h, tb, C.byref(sim)))
e._check(api.device_alloc(h, tb * (W + K), C.byref(buf)))
for t in range(W + K):
    e._check(api.synth_fill_acks_device(h, 0, t, sim, C.c_void_p(buf.value + t * tb)))
for t in range(0, W, T):
    e._check(api.step_dense_acks_device_n(h, C.c_void_p(buf.value + t * tb), min(T, W - t)))
e._check(api.sync(h))
e._check(api.timer_start(h))
for t in range(W, W + K, T):
    e._check(api.step_dense_acks_device_n(h, C.c_void_p(buf.value + t * tb), min(T, W + K - t)))
ms = C.c_float(0)
e._check(api.timer_stop(h, C.byref(ms)))
head, commit = e.read("head"), e.read("commit")
ok = bool((head == W + K).all() and (commit == W + K - 1).all() and not e.read("fault").any())
print(f"T={T} G={G} R={R} {ms.value * 1e3 / K:.2f} us/tick closed_form={'ok' if ok else 'VIOLATED'}", flush=True)
