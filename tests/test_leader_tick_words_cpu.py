"""The other half of the leader node tick's hot path WITHOUT a GPU: jg_dense_leader_tick (jg_dense.h) - Command::Tick of a
healthy leader as mailbox words: heartbeat() if due, then replicate() per other slot (the range start key and the number of
blocks after it: Probe nth(1) -> 1, Replicate skip(1).take(5), leader.rs:124-174,234-245), and the Q9 panic where the range
runs into the "commit" key - cut out of the header as it stands, compiled for the host, and held to the oracle's
jg_step_dense_leader outbox word for word."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from josefine_amd import capi
from host_compiled import CSRC, HIP_SHIM
from oracle_lib import oracle_engine
from parity import elect_all

NO = np.uint64(capi.NO_ACK)

WRAP = r'''
#include "jg_device.h"
struct JgDenseHot {
  uint32_t* flags;
  uint64_t* mlag;
  uint64_t* head;
  uint64_t* blk_decisions;
  uint32_t G;
  uint32_t hb_timeout, cfg_flags;
  uint64_t* term;
  uint64_t* heartbeat_time;
};
@STRUCTS@
@TICK@
template <int R>
static uint32_t emit(uint32_t s, uint64_t term, uint64_t hbt, uint64_t head, uint64_t commit, uint32_t nf, const uint64_t* match, uint64_t now,
                     uint32_t hb_timeout, uint32_t cfg_flags, uint64_t* out) {
  uint64_t hbt_col[1] = {hbt};
  JgDenseHot h{};
  h.G = 1, h.hb_timeout = hb_timeout, h.cfg_flags = cfg_flags, h.heartbeat_time = hbt_col;
  JgFaultRec fq[4];
  uint32_t fqn = 0;
  JgDev d{};
  d.G = 1, d.R = R, d.fault_q = fq, d.fault_q_n = &fqn, d.fault_q_cap = 4;
  jg_leader_beat beat{};
  uint64_t ae[R];
  for (int r = 0; r < R; r++) ae[r] = JG_NO_ACK;
  JgLeaderNode nd{};
  nd.now = now, nd.o_beat = &beat, nd.o_ae = ae;
  const uint32_t nf1 = jg_dense_leader_tick<R, true>(h, &d, nd, 0, 7, s, term, hbt, head, commit, nf, [&](int r) { return match[r]; });
  out[0] = beat.term, out[1] = beat.hb_commit, out[2] = nf1, out[3] = hbt_col[0], out[4] = fqn ? fq[0].code : 0;
  for (int r = 0; r < R; r++) out[5 + r] = ae[r];
  // ... and as a jg_dense_cluster's mailboxes get it (JgLeaderNode::o_aec): ONE word where the followers' words agree, the rows
  // only behind JG_AEC_INDIVIDUAL - read back the way the follower half reads it, it must be the same words
  uint64_t hbt2[1] = {hbt}, aec = 0x1234, ae2[R];
  uint32_t fqn2 = 0;
  for (int r = 0; r < R; r++) ae2[r] = 0x5678;  // (a row that is not written must not be read)
  h.heartbeat_time = hbt2, d.fault_q_n = &fqn2;
  jg_leader_beat beat2{};
  nd.o_beat = &beat2, nd.o_ae = ae2, nd.o_aec = &aec;
  const uint32_t nf2 = jg_dense_leader_tick<R, true>(h, &d, nd, 0, 7, s, term, hbt, head, commit, nf, [&](int r) { return match[r]; });
  uint64_t bad = nf2 != nf1 || beat2.term != beat.term || beat2.hb_commit != beat.hb_commit || hbt2[0] != hbt_col[0];
  for (int r = 0; r < R; r++)
    if ((uint32_t)r != s) bad |= (aec == JG_AEC_INDIVIDUAL ? ae2[r] : aec) != ae[r];
  out[5 + R] = bad, out[6 + R] = aec == JG_AEC_INDIVIDUAL;
  return nf1;
}
extern "C" uint32_t tick_words(int R, uint32_t s, uint64_t term, uint64_t hbt, uint64_t head, uint64_t commit, uint32_t nf, const uint64_t* match,
                               uint64_t now, uint32_t hb_timeout, uint32_t cfg_flags, uint64_t* out) {
  switch (R) {
    case 3: return emit<3>(s, term, hbt, head, commit, nf, match, now, hb_timeout, cfg_flags, out);
    case 5: return emit<5>(s, term, hbt, head, commit, nf, match, now, hb_timeout, cfg_flags, out);
    default: return ~0u;
  }
}
'''


@pytest.fixture(scope="module")
def lib():
    dense = open(os.path.join(CSRC, "jg_dense.h")).read()
    a = dense.index("struct JgClockVal {")
    b = dense.index("#define JG_OWNER_NONE")
    structs = dense[a:b]  # JgClockVal, JgClock, jg_clock_read, JgLeaderNode
    c = dense.index("template <int R, bool SKIP_OWN>\n__device__ __forceinline__ void jg_dense_outbox_none")
    d = dense.index("// ---- one tick per launch", c)
    tmp = tempfile.mkdtemp(prefix="jg_tick_words_")
    os.makedirs(os.path.join(tmp, "shim", "hip"))
    open(os.path.join(tmp, "shim", "hip", "hip_runtime.h"), "w").write(HIP_SHIM)
    cpp, so = os.path.join(tmp, "tick.cpp"), os.path.join(tmp, "libtick.so")
    open(cpp, "w").write(WRAP.replace("@STRUCTS@", structs).replace("@TICK@", dense[c:d]))
    subprocess.run(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-Wno-unused-function", f"-I{os.path.join(tmp, 'shim')}", f"-I{CSRC}", "-o", so, cpp], check=True)
    lb = C.CDLL(so)
    lb.tick_words.restype = C.c_uint32
    lb.tick_words.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32,
                              C.c_uint32, C.c_void_p]
    return lb


@pytest.mark.parametrize("R,cfg,own", [(3, capi.CFG_SEPARATE_COMMIT_KEY, 0), (5, capi.CFG_SEPARATE_COMMIT_KEY, 2), (3, 0, 1), (5, 0, 0)])
def test_tick_words_equal_the_oracles_outbox(lib, R, cfg, own):
    rng = np.random.default_rng(100 + R + cfg + own)
    G = 3000
    others = [r for r in range(R) if r != own]
    H = rng.integers(0, 12, G).astype(np.uint64)
    v = np.stack([rng.integers(0, H + 1) for _ in others]).astype(np.uint64)
    e = oracle_engine(G, R, seed=3, self_slots=np.full(G, own, np.uint8), flags=cfg)
    elect_all(e)
    acks = np.full((R, G), NO, np.uint64)
    acks[own] = H
    e.step_dense_acks(acks)
    acks[own] = 0
    acks[others] = v
    # (acknowledgements at different times: the Probe / Replicate states differ from slot to slot)
    late = rng.random((len(others), G)) < 0.5
    a1 = acks.copy()
    a1[others] = np.where(late, NO, v)
    e.step_dense_acks(a1)
    a2 = np.full((R, G), NO, np.uint64)
    a2[own] = 0
    a2[others] = np.where(late, v, NO)
    e.step_dense_acks(a2)
    e.drain_messages(), e.drain_applies()
    assert not e.read("fault").any()
    st = {k: e.read(k) for k in ("term", "head", "commit", "repl_state", "heartbeat_time")}
    match = np.stack([e.read("match", replica=r) for r in range(R)])
    now = rng.choice(np.array([0, 60, 99, 100, 101, 102, 180, 400], np.uint64), G)  # around the heartbeat period (100 ms): not due, the boundary, due
    # one Tick per `now` value on the oracle (the call takes one time): group by time
    want = {k: np.zeros(G, np.uint64) for k in ("term", "hb_commit")}
    want_ae = np.full((R, G), NO, np.uint64)
    out = (C.c_uint64 * (7 + R))()
    common = individual = 0
    got_fault = np.zeros(G, np.uint32)
    got_hbt = np.zeros(G, np.uint64)
    for g in range(G):
        nf = (int(st["repl_state"][g]) << 8) | (capi_commit_key() if st["commit"][g] > 0 else 0)
        col = np.ascontiguousarray(match[:, g])
        lib.tick_words(R, own, int(st["term"][g]), int(st["heartbeat_time"][g]), int(st["head"][g]), int(st["commit"][g]), nf, col.ctypes.data,
                       int(now[g]), 100, cfg, out)
        want["term"][g], want["hb_commit"][g] = out[0], out[1]
        assert out[5 + R] == 0, (g, list(out))  # the cluster's common-word form reads back to the same words
        individual += int(out[6 + R])
        common += 1 - int(out[6 + R])
        got_fault[g] = out[4]
        got_hbt[g] = out[3]
        for r in range(R):
            want_ae[r][g] = out[5 + r]
    # the oracle ticks every group at ONE time per call: walk the distinct times, comparing the groups ticked at each
    # (a Tick changes a leader's state only through heartbeat_time and a Q9 fault: the others are re-created per time)
    checked = faults = 0
    for t in np.unique(now):
        o = oracle_engine(G, R, seed=3, self_slots=np.full(G, own, np.uint8), flags=cfg)
        elect_all(o)
        acks = np.full((R, G), NO, np.uint64)
        acks[own] = H
        o.step_dense_acks(acks)
        o.step_dense_acks(a1)
        o.step_dense_acks(a2)
        o.drain_messages(), o.drain_applies()
        quiet = np.full((R, G), NO, np.uint64)
        quiet[own] = 0
        ob = o.step_dense_leader(int(t), quiet, tick=True)
        m = now == t
        f = o.read("fault")
        assert np.array_equal(got_fault[m], f[m].astype(np.uint32)), t
        ok = m & (f == 0)
        assert np.array_equal(want["term"][ok], ob["term"][ok]) and np.array_equal(want["hb_commit"][ok], ob["hb_commit"][ok]), t
        assert np.array_equal(got_hbt[ok], o.read("heartbeat_time")[ok]), t  # (leader.rs:78-84: reset when the heartbeat went out)
        words = (ob["ae_from"].astype(np.uint64) << np.uint64(8)) | ob["ae_n"].astype(np.uint64)
        words = np.where(ob["ae_n"] == capi.AE_NONE, NO, words)
        for r in others:
            assert np.array_equal(want_ae[r][ok], words[r][ok]), (t, r)
        checked += int(ok.sum())
        faults += int((m & (f != 0)).sum())
    assert checked + faults == G and checked > 50
    assert common > 0 and individual > 0, (common, individual)  # both forms of the cluster's AppendEntries column occurred
    if cfg == 0:
        assert faults > 0  # Q9: a caught-up follower's range runs into the "commit" key (chain.rs:219-226)


def capi_commit_key():
    return 1 << 5  # JGF_COMMIT_KEY (jg_device.h)
