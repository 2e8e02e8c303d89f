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
DOWN = int(os.environ.get("DOWN_TICKS", "0"))  # slot 1 silent: DOWN_TICKS warm-up ticks, then the timed ones
if DOWN:
    import numpy as np
    from josefine_amd import capi
    none = np.full(G, capi.NO_ACK, dtype=np.uint64)
    tmp = C.c_void_p()
    e._check(api.device_alloc(h, tb, C.byref(tmp)))
    sim2 = C.c_void_p()
    e._check(api.device_alloc(h, tb, C.byref(sim2)))
    for t in range(DOWN):
        e._check(api.synth_fill_acks_device(h, 0, t, sim2, tmp))
        e._check(api.device_upload(h, C.c_void_p(tmp.value + 8 * G), none.ctypes.data, 8 * G))
        e._check(api.step_dense_acks_device(h, tmp))
    e._check(api.sync(h))
    for t in range(W + K):  # the pre-generated stream continues from there, slot 1 still silent
        e._check(api.synth_fill_acks_device(h, 0, DOWN + t, sim2, C.c_void_p(buf.value + t * tb)))
        e._check(api.device_upload(h, C.c_void_p(buf.value + t * tb + 8 * G), none.ctypes.data, 8 * G))
    e._check(api.sync(h))
    print("slot 1 down for", DOWN, "ticks; match[1] max", int(e.read("match", 1).max()), "head", int(e.read("head").max()), flush=True)
DEAD = float(os.environ.get("DEAD_FRAC", "0"))
if DEAD > 0:  # kill a fraction of the groups first (two forged acks above the head: chain.commit panics)
    import numpy as np
    from josefine_amd import capi
    kill = np.full((R, G), capi.NO_ACK, dtype=np.uint64)
    kill[0] = 0
    dead = np.random.default_rng(5).random(G) < DEAD
    kill[1][dead] = 10**9
    kill[2][dead] = 10**9
    kill[3 % R][dead] = 10**9
    e.step_dense_acks(kill)
    e.drain_faults()
    print("dead groups:", int((e.read("fault") != 0).sum()), flush=True)
for t in range(W):
    e._check(api.step_dense_acks_device(h, C.c_void_p(buf.value + t * tb)))
e._check(api.sync(h))
e._check(api.timer_start(h))
for t in range(W, W + K):
    e._check(api.step_dense_acks_device(h, C.c_void_p(buf.value + t * tb)))
ms = C.c_float(0)
e._check(api.timer_stop(h, C.byref(ms)))
head, commit = e.read("head"), e.read("commit")
ok = bool((head == W + K).all() and (commit == W + K - 1).all() and not e.read("fault").any()) or DEAD > 0 or DOWN > 0
print(f"{os.path.basename(os.environ.get('JOSEFINE_GPU_LIB', 'default'))} grid={os.environ.get('JG_DENSE_GRID', '-')} "
      f"G={G} R={R} {ms.value * 1e3 / K:.2f} us/launch closed_form={'ok' if ok else 'VIOLATED'}", flush=True)
