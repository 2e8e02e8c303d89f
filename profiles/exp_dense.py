"""Micro-experiment driver: average launch time of the dense leader tick for the library in
JOSEFINE_GPU_LIB (variant builds under build/) and the grid cap in JG_DENSE_GRID.
    python profiles/exp_dense.py [G] [R] [ticks]
Prints one line: lib, grid, G, R, us per launch (HIP events on the engine's stream)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from josefine_amd import BatchedRaft  # noqa: E402
from josefine_amd.traces import elect_all  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
K = int(sys.argv[3]) if len(sys.argv) > 3 else 120
W = 20
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
for t in range(W):
    e._check(api.step_dense_acks_device(h, C.c_void_p(buf.value + t * tb)))
e._check(api.sync(h))
e._check(api.timer_start(h))
for t in range(W, W + K):
    e._check(api.step_dense_acks_device(h, C.c_void_p(buf.value + t * tb)))
ms = C.c_float(0)
e._check(api.timer_stop(h, C.byref(ms)))
head, commit = e.read("head"), e.read("commit")
ok = bool((head == W + K).all() and (commit == W + K - 1).all() and not e.read("fault").any())
print(f"{os.path.basename(os.environ.get('JOSEFINE_GPU_LIB', 'default'))} grid={os.environ.get('JG_DENSE_GRID', '-')} "
      f"G={G} R={R} {ms.value * 1e3 / K:.2f} us/launch closed_form={'ok' if ok else 'VIOLATED'}", flush=True)
