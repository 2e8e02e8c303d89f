"""Timing of jg_chain_compact_resident (Chain::compact on the engine's own chains, chain.rs:239-253):
G follower groups, each with a main chain and five dead branches (six segments), commit at the head.
    python profiles/exp_compact.py [G]
Prints the kernel time (HIP events around the launch + the count readback) and the whole call
(including the copy of the removed-block rows to the host and their ordering)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from josefine_amd import BatchedRaft, capi  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
e = BatchedRaft(G, 3, seed=1)
e.chain_compact_resident()  # (nothing to remove yet: allocates the row list outside the timed calls)
# main chain 0<-1<-3<-5<-7<-9<-11<-12, dead one-block branches 2, 4, 6, 8, 10 hanging off it
blocks = [(1, 0), (2, 1), (3, 1), (4, 3), (5, 3), (6, 5), (7, 5), (8, 7), (9, 7), (10, 9), (11, 9), (12, 11)]
kind = np.r_[np.full(G, capi.CMD_APPEND_ENTRIES, np.uint8), np.full(G, capi.CMD_HEARTBEAT, np.uint8)]
group = np.r_[np.arange(G, dtype=np.uint32), np.arange(G, dtype=np.uint32)]
rows = e.upload_rows(kind, group, from_=np.full(2 * G, 2, np.uint32), term=np.ones(2 * G, np.uint64),
                     id=np.r_[np.zeros(G, np.uint64), np.full(G, 12, np.uint64)],
                     aux=np.r_[np.full(G, len(blocks), np.uint64), np.zeros(G, np.uint64)],
                     blk_id=np.array([b[0] for b in blocks], np.uint64), blk_next=np.array([b[1] for b in blocks], np.uint64))
e.step_device_rows(rows, now_ms=10)
e.drain_messages(), e.drain_applies()
assert (e.read("commit") == 12).all() and not e.read("fault").any()
api, h = e.api, e._h
for it in range(3):
    n = C.c_size_t(0)
    ms = C.c_float(0)
    t0 = time.perf_counter()
    e._check(api.timer_start(h))
    e._check(api.chain_compact_resident(h, C.byref(n)))
    e._check(api.timer_stop(h, C.byref(ms)))
    wall = time.perf_counter() - t0
    got = C.c_size_t(0)
    e._check(api.drain_compacted(h, None, 0, C.byref(got)))
    out = np.zeros(got.value, dtype=capi.COMPACT_DTYPE)
    if got.value:
        e._check(api.drain_compacted(h, out.ctypes.data, got.value, C.byref(got)))
    print(f"pass {it}: G={G} removed {n.value} blocks; stream (kernel + count readback) {ms.value * 1e3:.1f} us, whole call {wall * 1e3:.1f} ms",
          "first rows:", [(int(r['group']), int(r['id'])) for r in out[:5]], flush=True)
assert not e.read("fault").any()
