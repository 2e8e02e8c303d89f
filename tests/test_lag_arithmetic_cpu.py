"""The arithmetic of the headline kernel WITHOUT a GPU: jg_lag_tick (jg_dense.h) - the whole leader tick on packed lags
below the chain head - is cut out of the header as it stands, compiled for the host with a ten-line shim, and held to the
oracle's Leader::commit / ReplicationProgress::advance over an exhaustive small domain (R = 3: every reachable progress
state with head <= 3 x every tick of 0-2 appends and acknowledgements up to one above the head), a sampled one (R = 5)
and the escape codes (a follower too far BEHIND for its field).  Where the function refuses a tick (returns false: the
general path takes it), the reason must be the documented one."""
import ctypes as C
import itertools
import os
import subprocess
import tempfile

import numpy as np
import pytest

from josefine_amd import capi
from oracle_lib import oracle_engine
from parity import elect_all

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO = np.uint64(capi.NO_ACK)

SHIM = r'''
#include <cstdint>
#include <cstring>
#include "%(root)s/include/josefine_gpu.h"
#define __device__
#define __host__
#define __forceinline__ inline
#define JGF_COMMIT_KEY (1u << 5)
#define JGF_REPL_SHIFT 8
static inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
'''
WRAP = r'''
template <int R>
static int run(uint32_t s, uint32_t f, uint64_t w0, uint64_t head0, uint64_t n_app, const uint64_t* a, uint64_t* out) {
  uint64_t aa[R];
  for (int r = 0; r < R; r++) aa[r] = a[r];
  JgLagTick<R> o;
  std::memset(&o, 0, sizeof(o));
  uint32_t dec = 0;
  const bool ok = jg_lag_tick<R>(s, f, w0, head0, n_app, aa, o, dec);
  out[0] = o.w1, out[1] = o.head1, out[2] = o.nf, out[3] = dec, out[4] = o.adv, out[5] = o.cwide;
  return ok;
}
extern "C" int lag_tick(int R, uint32_t s, uint32_t f, uint64_t w0, uint64_t head0, uint64_t n_app, const uint64_t* a, uint64_t* out) {
  switch (R) {
    case 2: return run<2>(s, f, w0, head0, n_app, a, out);
    case 3: return run<3>(s, f, w0, head0, n_app, a, out);
    case 4: return run<4>(s, f, w0, head0, n_app, a, out);
    case 5: return run<5>(s, f, w0, head0, n_app, a, out);
    case 7: return run<7>(s, f, w0, head0, n_app, a, out);
    default: return -1;
  }
}
template <int R>
static int run_pre(uint32_t s, uint64_t w0, uint64_t head0, const uint64_t* a, uint32_t pre, uint32_t* adv) {
  uint64_t aa[R];
  for (int r = 0; r < R; r++) aa[r] = a[r];
  return jg_lag_pre<R>(s, w0, head0, aa, pre, *adv);
}
extern "C" int lag_pre(int R, uint32_t s, uint64_t w0, uint64_t head0, const uint64_t* a, uint32_t pre, uint32_t* adv) {
  switch (R) {
    case 3: return run_pre<3>(s, w0, head0, a, pre, adv);
    case 5: return run_pre<5>(s, w0, head0, a, pre, adv);
    default: return -1;
  }
}
extern "C" uint64_t lag_encode(uint64_t v, uint64_t base, uint32_t R) { return jg_lag_encode(v, base, R); }
extern "C" uint64_t lag_field(uint64_t w, uint32_t r, uint32_t R) { return jg_lag_field(w, r, R); }
extern "C" uint64_t lag_with(uint64_t w, uint32_t r, uint32_t R, uint64_t field) { return jg_lag_with(w, r, R, field); }
extern "C" int lag_wide(uint64_t field, uint32_t R) { return jg_lag_wide(field, R); }
extern "C" uint64_t lag_behind(uint32_t R) { return jg_lag_behind(R); }
'''


@pytest.fixture(scope="module")
def lag():
    dense = open(os.path.join(ROOT, "josefine_amd", "csrc", "jg_dense.h")).read()
    dev = open(os.path.join(ROOT, "josefine_amd", "csrc", "jg_device.h")).read()
    a = dense.index("template <int R>\nstruct JgLagTick {")
    b = dense.index("// The same tick on lags that are already unpacked")  # (jg_lag_tick and jg_lag_pre)
    c = dev.index("__host__ __device__ __forceinline__ uint32_t jg_lag_bits")
    d = dev.index("// Registers of one group while a lane walks its command segment.")
    src = SHIM % {"root": ROOT} + dev[c:d] + dense[a:b] + WRAP
    tmp = tempfile.mkdtemp(prefix="jg_lag_")
    cpp, so = os.path.join(tmp, "lag.cpp"), os.path.join(tmp, "liblag.so")
    open(cpp, "w").write(src)
    subprocess.run(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-o", so, cpp], check=True)
    lib = C.CDLL(so)
    lib.lag_tick.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    for f in (lib.lag_encode, lib.lag_field, lib.lag_with, lib.lag_behind):
        f.restype = C.c_uint64
    lib.lag_encode.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
    lib.lag_field.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    lib.lag_with.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64]
    lib.lag_wide.argtypes = [C.c_uint64, C.c_uint32]
    lib.lag_behind.argtypes = [C.c_uint32]
    lib.lag_pre.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    return lib


def _leaders(G, R, H, v, own):
    """an oracle engine whose G leaders (own slot `own`) have head H[g] and progress heads v[k][g] for the other slots, ascending"""
    e = oracle_engine(G, R, seed=1, self_slots=np.full(G, own, np.uint8), flags=capi.CFG_SEPARATE_COMMIT_KEY)
    elect_all(e)
    others = [r for r in range(R) if r != own]
    acks = np.full((R, G), NO, np.uint64)
    acks[own] = H
    e.step_dense_acks(acks)  # H appends, self-acked (leader.rs:177-197)
    acks = np.full((R, G), NO, np.uint64)
    acks[own] = 0
    acks[others] = v
    e.step_dense_acks(acks)  # the followers' acknowledgements (progress.rs:133-140), one Leader::commit each
    e.drain_messages(), e.drain_applies()
    assert not e.read("fault").any()
    return e


def _check(lag, R, H, v, n_app, a, own=0):
    """states (H, v) x inputs (n_app, a): the kernel's arithmetic against the oracle, group by group"""
    G = len(H)
    others = [r for r in range(R) if r != own]
    e = _leaders(G, R, H, v, own)
    head0, commit0 = e.read("head"), e.read("commit")
    match0 = np.stack([e.read("match", replica=r) for r in range(R)])
    repl0 = e.read("repl_state")
    assert (head0 == H).all() and (match0[own] == H).all()
    acks = np.full((R, G), NO, np.uint64)
    acks[own] = n_app
    acks[others] = a
    # the kernel's side first: packed word from the oracle's state, the tick, the word unpacked again
    ok = np.zeros(G, bool)
    got = np.zeros((G, R + 3), np.uint64)  # match[0..R), commit, head, repl bits
    dec = np.zeros(G, np.int64)
    out = (C.c_uint64 * 6)()
    for g in range(G):
        w0 = 0
        for r in range(R):
            w0 = lag.lag_with(w0, r, R, lag.lag_encode(int(match0[r][g]), int(head0[g]), R))
        w0 = lag.lag_with(w0, R, R, lag.lag_encode(int(commit0[g]), int(head0[g]), R))
        col = np.ascontiguousarray(acks[:, g])
        ok[g] = lag.lag_tick(R, own, int(repl0[g]) << 8, w0, int(head0[g]), int(n_app[g]), col.ctypes.data, out) == 1
        if ok[g]:
            w1, head1 = out[0], out[1]
            for r in range(R + 1):
                fld = lag.lag_field(w1, r, R)
                got[g][r] = np.uint64(0xFFFFFFFFFFFFFFFF) if lag.lag_wide(fld, R) else np.uint64(head1 - fld)
            got[g][R + 1] = head1
            got[g][R + 2] = (out[2] >> 8) & 0xFF
            dec[g] = out[3]
    # refused exactly where documented: an acknowledgement above the head it meets (the appends come first), or one for a
    # slot whose lag is an escape code (the caller's domain: none in the small states, the BEHIND slot in the large one)
    above = ((a != NO) & (a > (H + n_app)[None, :])).any(axis=0)
    wide0 = np.array([[lag.lag_wide(lag.lag_encode(int(match0[r][g]), int(head0[g]), R), R) for g in range(G)] for r in range(R)], bool)
    acked_wide = (wide0[others] & (a != NO)).any(axis=0)
    assert np.array_equal(~ok, above | acked_wide), np.nonzero(~ok != (above | acked_wide))[0][:8]
    # the oracle's side: the same tick through Leader::commit; groups the kernel refused get an empty tick (no decisions)
    acks[:, ~ok] = NO
    acks[own, ~ok] = 0
    d0 = e.counters()["decisions"]
    e.step_dense_acks(acks)
    assert not e.read("fault").any()
    want_match = np.stack([e.read("match", replica=r) for r in range(R)])
    sel = ok
    for r in range(R):
        m = sel & (got[:, r] != np.uint64(0xFFFFFFFFFFFFFFFF))  # (a BEHIND slot stays in its escape code: nothing to compare)
        assert np.array_equal(got[m, r], want_match[r][m]), (r, np.nonzero(got[:, r] != want_match[r])[0][:5])
    assert np.array_equal(got[sel, R], e.read("commit")[sel]) and np.array_equal(got[sel, R + 1], e.read("head")[sel])
    assert np.array_equal(got[sel, R + 2].astype(np.uint8), e.read("repl_state")[sel])
    assert int(dec[sel].sum()) == e.counters()["decisions"] - d0  # one decision per append and per acknowledgement
    return int(ok.sum())


def test_lag_tick_exhaustive_small_domain_r3(lag):
    R = 3
    states = [(H, v1, v2) for H in range(4) for v1 in range(H + 1) for v2 in range(H + 1)]
    rows = []
    for (H, v1, v2) in states:
        for n in (0, 1, 2):
            opts = [int(NO)] + list(range(H + n + 2))  # none, every head up to one ABOVE the head the ack meets
            for a1, a2 in itertools.product(opts, opts):
                rows.append((H, v1, v2, n, a1, a2))
    t = np.array(rows, dtype=np.uint64).T
    for own in range(R):  # (the majority network treats the slots differently: every place of the own slot)
        served = _check(lag, R, t[0], t[1:3], t[3], t[4:6], own=own)
        assert len(rows) > 3000 and served > 0.6 * len(rows)


def test_lag_tick_sampled_r5_and_r7(lag):
    rng = np.random.default_rng(5)
    for R, hmax in ((5, 6), (5, 40), (7, 40), (2, 40), (4, 40)):
        for own in range(R):  # (the majority networks treat the slots differently: every place of the own slot)
            G = 3000
            H = rng.integers(0, hmax, G).astype(np.uint64)
            v = np.stack([rng.integers(0, H + 1) for _ in range(R - 1)]).astype(np.uint64)
            n = rng.integers(0, 3, G).astype(np.uint64)
            a = np.stack([np.where(rng.random(G) < 0.25, NO, rng.integers(0, H + n + 2).astype(np.uint64)) for _ in range(R - 1)]).astype(np.uint64)
            served = _check(lag, R, H, v, n, a, own=own)
            assert served > G // 2


def test_lag_tick_a_follower_too_far_behind_stays_in_lag_space_r3(lag):
    """lag >= 65 534 at R = 3 (16-bit fields): the BEHIND escape - served while no acknowledgement arrives for that slot"""
    R = 3
    big = int(lag.lag_behind(R)) + 100
    rows = []
    for n in (0, 1):
        for a1 in (int(NO), big - 1, big + n):
            for a2 in (int(NO), 5):
                rows.append((big, big - 1, 0, n, a1, a2))
    t = np.array(rows, dtype=np.uint64).T
    served = _check(lag, R, t[0], t[1:3], t[3], t[4:6])
    assert served == len(rows) // 2  # exactly the ticks without an acknowledgement for the BEHIND slot


def test_lag_tick_a_reelected_leader_keeps_its_commit_index_r3(lag):
    """leader.rs:89-92: the commit index only grows.  A restarted, re-elected leader starts with every progress head at 0
    (candidate.rs:218-220, Q10) - too far BEHIND for its field here - and its commit index where the store had it: the
    majority is "unknown, below the commit index", the commit index stays.  Its lags are below the TOP of the run its store
    kept (jg_lane_base), not below its head."""
    R, G = 3, 4
    top = int(lag.lag_behind(R)) + 50
    e = oracle_engine(G, R, seed=2, flags=capi.CFG_SEPARATE_COMMIT_KEY)
    elect_all(e)
    acks = np.full((R, G), NO, np.uint64)
    acks[0] = top
    e.step_dense_acks(acks)
    acks[0], acks[1] = 0, top - 3
    e.step_dense_acks(acks)  # commit index top - 3
    g = np.arange(G, dtype=np.uint32)
    e.submit_columns(np.full(G, capi.CMD_RESTART, np.uint8), g)
    e.step(100)
    e.submit_columns(np.full(G, capi.CMD_TIMEOUT, np.uint8), g)
    e.submit_columns(np.full(G, capi.CMD_VOTE_RESPONSE, np.uint8), g, from_=np.full(G, 2, np.uint32), term=np.ones(G, np.uint64), flag=np.ones(G, np.uint8))
    e.step(100)
    e.drain_messages(), e.drain_applies()
    assert (e.read("role") == capi.ROLE_LEADER).all() and (e.read("head") == top - 3).all() and (e.read("commit") == top - 3).all()
    assert all((e.read("match", replica=r) == 0).all() for r in range(R))
    w0 = 0
    for r in range(R):
        w0 = lag.lag_with(w0, r, R, lag.lag_encode(0, top, R))
    w0 = lag.lag_with(w0, R, R, lag.lag_encode(top - 3, top, R))
    assert lag.lag_field(w0, R, R) == 3 and all(lag.lag_field(w0, r, R) == lag.lag_behind(R) for r in range(R))
    out = (C.c_uint64 * 6)()
    quiet = np.full(R, NO, np.uint64)
    quiet[0] = 0
    assert lag.lag_tick(R, 0, int(e.read("repl_state")[0]) << 8, w0, top, 0, quiet.ctypes.data, out) == 1
    assert out[0] == w0 and out[1] == top and out[3] == 0 and out[4] == 0  # nothing moves: the word, the base, no decision
    acks = np.full((R, G), NO, np.uint64)
    acks[0] = 0
    d0 = e.counters()["decisions"]
    e.step_dense_acks(acks)
    assert (e.read("commit") == top - 3).all() and e.counters()["decisions"] == d0 and not e.read("fault").any()
    # an acknowledgement for a slot in its escape code is the general path's (exact compare)
    acked = quiet.copy()
    acked[1] = top - 1
    assert lag.lag_tick(R, 0, 0, w0, top, 0, acked.ctypes.data, out) == 0


@pytest.mark.parametrize("R,own", [(3, 0), (3, 2), (5, 1)])
def test_lag_pre_is_what_fsm_tx_saw_before_the_notify(lag, R, own):
    """jg_lag_pre: the acknowledgements that ARRIVED BEFORE the tick's ClientRequest met the old head; the commit index
    they reached is the Apply range fsm_tx carries in front of the Notify (leader.rs:93,184-188).  Against the oracle's
    jg_step_node = plain arrival-order Apply: rows [AppendResponse of the early slots, ClientRequest, the others]."""
    rng = np.random.default_rng(R * 10 + own)
    others = [r for r in range(R) if r != own]
    G = 2500
    H = rng.integers(0, 6, G).astype(np.uint64)
    v = np.stack([rng.integers(0, H + 1) for _ in others]).astype(np.uint64)
    e = _leaders(G, R, H, v, own)
    head0, commit0 = e.read("head"), e.read("commit")
    match0 = np.stack([e.read("match", replica=r) for r in range(R)])
    a = np.stack([np.where(rng.random(G) < 0.3, NO, rng.integers(0, H + 1).astype(np.uint64)) for _ in others]).astype(np.uint64)  # (at or below the OLD head)
    early = rng.random((len(others), G)) < 0.5
    ids = np.array(e.node_ids, np.uint32)
    kind, group, frm, idc, flag = [], [], [], [], []
    for phase in (True, None, False):  # early acknowledgements, the ClientRequest, the late ones
        if phase is None:
            kind.append(np.full(G, capi.CMD_CLIENT_REQUEST, np.uint8)), group.append(np.arange(G, dtype=np.uint32))
            frm.append(np.zeros(G, np.uint32)), idc.append(np.arange(G, dtype=np.uint64) + 7), flag.append(np.zeros(G, np.uint8))
            continue
        for k, r in enumerate(others):
            m = (a[k] != NO) & (early[k] == phase)
            g = np.nonzero(m)[0].astype(np.uint32)
            kind.append(np.full(len(g), capi.CMD_APPEND_RESPONSE, np.uint8)), group.append(g)
            frm.append(np.full(len(g), ids[r], np.uint32)), idc.append(a[k][m]), flag.append(np.ones(len(g), np.uint8))
    n = sum(len(x) for x in kind)
    e.submit_columns(np.concatenate(kind), np.concatenate(group), from_=np.concatenate(frm), term=np.ones(n, np.uint64),
                     id=np.concatenate(idc), flag=np.concatenate(flag))
    e.step_node(1000, leader=True, follower=False, tick=False)
    assert not e.read("fault").any() and (e.read("head") == head0 + 1).all()
    rows = e.drain_applies()
    mid = commit0.copy()  # the commit index at the moment of the Notify: the end of the Apply range in front of it
    seen_notify = np.zeros(G, bool)
    for r in rows:
        g = int(r["group"])
        if r["kind"] == capi.FSM_NOTIFY:
            seen_notify[g] = True
        elif r["kind"] == capi.FSM_APPLY_LEADER and not seen_notify[g]:
            assert int(r["a"]) == int(commit0[g])
            mid[g] = r["b"]
    assert seen_notify.all()
    adv = C.c_uint32()
    checked = 0
    for g in range(G):
        w0 = 0
        for r in range(R):
            w0 = lag.lag_with(w0, r, R, lag.lag_encode(int(match0[r][g]), int(head0[g]), R))
        w0 = lag.lag_with(w0, R, R, lag.lag_encode(int(commit0[g]), int(head0[g]), R))
        col = np.full(R, NO, np.uint64)
        pre = 0
        for k, r in enumerate(others):
            col[r] = a[k][g]
            if a[k][g] != NO and early[k][g]:
                pre |= 1 << r
        assert lag.lag_pre(R, own, w0, int(head0[g]), col.ctypes.data, pre, C.byref(adv)) == 1
        assert adv.value == int(mid[g]) - int(commit0[g]), (g, adv.value, int(mid[g]), int(commit0[g]))
        checked += adv.value > 0
    assert checked > G // 20  # (early acknowledgements did move commit indices)
