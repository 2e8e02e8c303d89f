"""Micro-experiment: average launch time of the leader half of the dense node tick
(k_leader_node_tick, HIP event pairs via jg_kernel_timing) for the library in JOSEFINE_GPU_LIB.
    python profiles/exp_node.py [G] [R] [rounds]
Steady state: one append per round, every follower acks the previous head, HeartbeatResponses with
has_committed = 1 from everybody (the closed loop's traffic without running the followers)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from josefine_amd import BatchedRaft, capi  # noqa: E402
from josefine_amd.traces import elect_all  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
e = BatchedRaft(G, R, seed=1, flags=capi.CFG_SEPARATE_COMMIT_KEY)
elect_all(e)
e.drain_messages(), e.drain_applies()
api, h = e.api, e._h


def dalloc(n):
    p = C.c_void_p()
    e._check(api.device_alloc(h, n, C.byref(p)))
    return p


NT = 8  # rotating inbox blocks (tick t acks head t-1)
acks = [dalloc(8 * R * G) for _ in range(NT)]
hbr_commit = dalloc(8 * R * G)
o_beat, o_ae = dalloc(16 * G), dalloc(8 * R * G)
outbox = capi.LeaderOutbox(o_beat.value, o_ae.value)
e._check(api.kernel_timing(h, 1))
now = 0
blk = np.zeros((R, G), np.uint64)
has = np.ones((R, G), np.uint8)  # HeartbeatResponse{has_committed: true} from everybody
for t in range(K + 10):
    blk[:] = max(t - 1, 0) if t else capi.NO_ACK
    blk[0] = 1
    if t == 0:
        blk[1:] = capi.NO_ACK
    words = np.ascontiguousarray(capi.pack_answers(blk, has))
    e._check(api.device_upload(h, acks[t % NT], words.ctypes.data, words.nbytes))
    now += 100
    inbox = capi.LeaderInbox(acks[t % NT].value, hbr_commit.value)
    e._check(api.step_dense_leader(h, now, C.byref(inbox), C.byref(outbox)))
us, n = C.c_float(0), C.c_uint32(0)
e._check(api.kernel_timing_read(h, C.byref(us), C.byref(n)))
ok = bool((e.read("head") == K + 10).all() and not e.read("fault").any() and len(e.drain_messages()) == 0)
print(f"{os.path.basename(os.environ.get('JOSEFINE_GPU_LIB', 'default'))} G={G} R={R} k_leader_node_tick {us.value:.2f} us/launch "
      f"({n.value} launches) steady={'ok' if ok else 'NO'} commit_min={int(e.read('commit').min())}", flush=True)
