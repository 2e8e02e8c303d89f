"""The device's general state machine - josefine_amd/csrc/jg_device.h as it stands: jg_load / jg_apply / jg_store, every
role, the chain, the elections, the lag-packed progress heads - compiled for the HOST by g++ behind a twenty-line stand-in
for <hip/hip_runtime.h>, and driven row by row the way k_apply_rows's owner lane drives it.

TEST INFRASTRUCTURE, and nothing else: the library is built at test time into a temporary directory, nothing under
josefine_amd/ can reach it, and it is no engine - no dense kernels, no transport, no drains of the product.  What it is
for: the CPU suite (which runs where there is no GPU) holds the device SOURCE of the state machine to the oracle with
the same fuzz the GPU suite runs through the real kernels (tests/test_host_compiled_state_machine.py): a change to
jg_device.h is checked before a GPU-minute is spent on it.

Sanitizers: JG_HOST_SANITIZE=address,undefined ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so)
python -m pytest tests/test_host_compiled_state_machine.py builds the harness instrumented (every column, list and queue
is a std::vector here: an access outside one is reported)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

from josefine_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "josefine_amd", "csrc")

HIP_SHIM = r'''
#pragma once
// stand-in for <hip/hip_runtime.h> when the device headers are compiled for the host (tests/host_compiled.py).  ONE lane at
// a time: a ballot is the lane's own bit at its own position in the wave, a shuffle from another lane finds nothing - good
// for code whose lanes work on their own (the general state machine, the slow kernels' list walks with JG_BLOCK = 1, the
// dense kernels' per-group logic with the lane where the group puts it), NOT for anything lanes do together (the
// transport's LDS staging and sorts, the wave reductions of the counters - those are added up by the harness).
#include <cstdint>
#include <cstddef>
#include <algorithm>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
#define __shared__ static
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 { uint32_t x = 1, y = 1, z = 1; };
static dim3 threadIdx{0, 0, 0}, blockIdx{0, 0, 0}, blockDim{1, 1, 1}, gridDim{1, 1, 1};
static inline void __syncthreads() {}
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T, class U> static inline T atomicOr(T* p, U v) { T o = *p; *p = o | (T)v; return o; }
template <class T, class U> static inline T atomicAnd(T* p, U v) { T o = *p; *p = o & (T)v; return o; }
template <class T, class U> static inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T> static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
template <class T, class U> static inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
static inline int __ffs(uint32_t v) { return __builtin_ffs((int)v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline uint64_t __ballot(bool x) { return (x ? 1ull : 0ull) << (threadIdx.x & 63u); }  // (the only active lane of its wave)
template <class T> static inline T __shfl(T v, int, int = 64) { return v; }
template <class T> static inline T __shfl_down(T, int, int = 64) { return T(0); }
template <class T> static inline T __shfl_xor(T, int, int = 64) { return T(0); }
template <class T> static inline T __shfl_up(T, int, int = 64) { return T(0); }
static inline uint32_t __builtin_amdgcn_readlane(uint32_t v, int) { return v; }
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_fetch_or(p, v, order, scope) atomicOr((p), (v))
#define __hip_atomic_fetch_and(p, v, order, scope) atomicAnd((p), (v))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (decltype(*(p) + 0))(v))
using std::max;
using std::min;
'''

HOST_H = r'''
#pragma once
#include <vector>
#include "jg_device.h"
struct Host {
  JgDev d{};
  std::vector<uint64_t> term, commit, head, id_gen, run_hi, mlag, match_wide, hbt, win_lo, win_hi, win_next, blk_dec;
  std::vector<uint32_t> flags, fvote;
  std::vector<uint4> cold_t, cold_v;
  std::vector<JgFaultRec> fq;
  std::vector<uint64_t> defer_bits, fdefer_bits;
  std::vector<uint32_t> slow_list, slow_cnt;
  std::vector<JgXqRec> xq;
  uint32_t status[8] = {};
  uint32_t seq = 0;
  uint64_t n_cmds = 0, decisions = 0;
  std::vector<jg_msg_row> msgs;
  size_t msgs_mark = 0;  // (hc_cluster_any_round: the rows queued when the node's leader half was done - the follower half's come behind)
  std::vector<jg_fsm_row> fsm;
  std::vector<JgFaultRec> faults;
  int err = 0;
};
#define VIS extern "C" __attribute__((visibility("default")))
'''

HARNESS = r'''
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>
#define JG_BLOCK 1
#include "jg_kernels.h"   // (jg_device.h, jg_dense.h, jg_sparse.h: the slow leader kernel's body)
#include "jg_follower.h"  // (... and the follower's)
#include "jg_node.h"      // (jg_step_node's row passes: prefill, classify, route, fsm build)
#include "jg_votes.h"     // (the election vocabulary as words: the receiving half - not part of the engine yet)

#include "host.h"

extern "C" Host* hc_create(uint32_t G, uint32_t R, const uint32_t* node_ids, const uint8_t* self_slots, uint64_t seed, uint64_t group_base,
                           uint32_t flags, uint32_t hb, uint32_t el_min, uint32_t el_max) {
  Host* h = new Host;
  JgDev& d = h->d;
  d.G = G, d.R = R;
  for (uint32_t r = 0; r < R; r++) d.node_ids[r] = node_ids[r];
  d.hb_timeout = hb, d.el_min = el_min, d.el_max = el_max, d.cfg_flags = flags, d.seed = seed, d.group_base = group_base;
  auto a64 = [&](std::vector<uint64_t>& v, size_t n) { v.assign(n, 0); return v.data(); };
  d.term = a64(h->term, G), d.commit = a64(h->commit, G), d.head = a64(h->head, G), d.id_gen = a64(h->id_gen, G);
  d.run_hi = a64(h->run_hi, G), d.mlag = a64(h->mlag, G), d.match_wide = a64(h->match_wide, (size_t)R * G);
  d.heartbeat_time = a64(h->hbt, G);
  d.win_lo = a64(h->win_lo, (size_t)JG_CHAIN_WINDOW * G), d.win_hi = a64(h->win_hi, (size_t)JG_CHAIN_WINDOW * G);
  d.win_next = a64(h->win_next, (size_t)JG_CHAIN_WINDOW * G);
  h->flags.assign(G, 0), d.flags = h->flags.data();
  h->cold_t.assign(G, uint4{}), h->cold_v.assign(G, uint4{}), d.cold.t = h->cold_t.data(), d.cold.v = h->cold_v.data();
  h->fvote.assign((size_t)JG_FOREIGN_VOTERS * G, 0), d.fvote_id = h->fvote.data();
  h->fq.assign((size_t)8 * G + 4096, JgFaultRec{}), d.fault_q = h->fq.data(), d.fault_q_cap = (uint32_t)h->fq.size();
  d.err = &h->status[0], d.irregular_seen = &h->status[1], d.deferred_seen = &h->status[2], d.fault_q_n = &h->status[3];
  d.xq_n = &h->status[4], d.cold_seen = &h->status[5];
  d.xq = nullptr, d.xq_cap = 0;
  h->blk_dec.assign(std::max<size_t>(4096, G / 64 + 2), 0), d.blk_decisions = h->blk_dec.data();
  h->defer_bits.assign((G + 63) / 64, 0), d.defer_bits = h->defer_bits.data();
  h->fdefer_bits.assign(2 * ((G + 63) / 64), 0), d.fdefer_bits = h->fdefer_bits.data();
  d.slow_cap = G + 64;
  h->slow_list.assign((size_t)JG_SHARDS * d.slow_cap, 0), d.slow_list = h->slow_list.data();
  h->slow_cnt.assign(JG_SHARDS, 0), d.slow_cnt = h->slow_cnt.data();
  h->xq.assign((size_t)(R + 3) * G + 64, JgXqRec{});
  for (uint32_t g = 0; g < G; g++) {
@INIT_BODY@
  }
  return h;
}
extern "C" void hc_destroy(Host* h) { delete h; }

// jg_submit + jg_step of one batch: rows in any order, applied per group in the order given (stable by group), as the
// owner lane of a run does in k_apply_rows (jg_sparse.h)
extern "C" int hc_step(Host* h, uint32_t n, const uint8_t* kind, const uint32_t* group, const uint32_t* from, const uint64_t* term,
                       const uint64_t* id, const uint64_t* aux, const uint8_t* flag, uint64_t nb, const uint64_t* blk_id,
                       const uint64_t* blk_next, uint64_t now) {
  const JgDev& d = h->d;
  if (!n) return 0;
  h->seq++;
  std::vector<uint32_t> order(n);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return group[a] < group[b]; });
  const uint32_t msg_per_row = d.R + 1 < 2 ? 2 : d.R + 1, fsm_per_row = 2;
  std::vector<jg_msg_row> mbuf;
  std::vector<jg_fsm_row> fbuf;
  for (uint32_t i = 0; i < n;) {
    const uint32_t g = group[order[i]];
    uint32_t run = 1;
    while (i + run < n && group[order[i + run]] == g) run++;
    if (g >= d.G) return 3;
    mbuf.assign((size_t)run * msg_per_row, jg_msg_row{});
    fbuf.assign((size_t)run * fsm_per_row, jg_fsm_row{});
    JgLane L;
    jg_load(d, L, g);
    L.now = now;
    L.seq = h->seq;
    L.mp = mbuf.data(), L.mend = mbuf.data() + mbuf.size();
    L.fp = fbuf.data(), L.fend = fbuf.data() + fbuf.size();
    for (uint32_t t = 0; t < run; t++) {
      const uint32_t k = order[i + t];
      JgCmd c;
      c.kind = kind[k], c.from = from[k], c.flag = flag[k], c.term = term[k], c.id = id[k], c.aux = aux[k];
      if (c.kind == JG_CMD_APPEND_ENTRIES && (c.aux > nb || c.id > nb - c.aux)) return 5;
      jg_apply<JG_KINDS_ALL>(d, L, c, blk_id, blk_next);
    }
    h->msgs.insert(h->msgs.end(), mbuf.data(), L.mp);
    h->fsm.insert(h->fsm.end(), fbuf.data(), L.fp);
    if (L.overflow) return 1;
    h->decisions += L.decisions;
    jg_store<true>(d, L);
    i += run;
  }
  h->n_cmds += n;
  // the step's fault records, in (step, group) order behind the earlier ones
  const uint32_t nf = *d.fault_q_n;
  std::vector<JgFaultRec> f(d.fault_q, d.fault_q + nf);
  std::stable_sort(f.begin(), f.end(), [](const JgFaultRec& a, const JgFaultRec& b) { return a.seq != b.seq ? a.seq < b.seq : a.group < b.group; });
  h->faults.insert(h->faults.end(), f.begin(), f.end());
  *d.fault_q_n = 0;
  return (int)h->status[0];
}
extern "C" size_t hc_drain_messages(Host* h, jg_msg_row* out, size_t cap) {
  const size_t n = h->msgs.size();
  if (out && cap >= n) {
    if (n) std::memcpy(out, h->msgs.data(), n * sizeof(jg_msg_row));
    h->msgs.clear();
  }
  return n;
}
extern "C" size_t hc_drain_applies(Host* h, jg_fsm_row* out, size_t cap) {
  const size_t n = h->fsm.size();
  if (out && cap >= n) {
    if (n) std::memcpy(out, h->fsm.data(), n * sizeof(jg_fsm_row));
    h->fsm.clear();
  }
  return n;
}
extern "C" size_t hc_drain_faults(Host* h, jg_fault_row* out, size_t cap) {
  const size_t n = h->faults.size();
  if (out && cap >= n) {
    for (size_t i = 0; i < n; i++) out[i] = jg_fault_row{h->faults[i].group, h->faults[i].code};
    h->faults.clear();
  }
  return n;
}
extern "C" size_t hc_msgs_mark(Host* h) { return h->msgs_mark; }
extern "C" void hc_counters(Host* h, uint64_t* out) { out[0] = h->n_cmds, out[1] = h->decisions, out[2] = 0, out[3] = 0; }
// jg_read_state's decoding, through the state machine's own jg_load
extern "C" int hc_read(Host* h, int field, uint32_t replica, void* out) {
  const JgDev& d = h->d;
  uint64_t* o64 = (uint64_t*)out;
  uint32_t* o32 = (uint32_t*)out;
  uint8_t* o8 = (uint8_t*)out;
  for (uint32_t g = 0; g < d.G; g++) {
    JgLane L;
    jg_load(d, L, g);
    const uint32_t role = jg_role(L);
    switch (field) {
      case JG_FIELD_TERM: o64[g] = L.term; break;
      case JG_FIELD_COMMIT: o64[g] = L.commit; break;
      case JG_FIELD_HEAD: o64[g] = L.head; break;
      case JG_FIELD_ID_GEN: o64[g] = L.id_gen; break;
      case JG_FIELD_MATCH: o64[g] = role == JG_ROLE_LEADER ? jg_match_get(d, L, replica) : 0; break;
      case JG_FIELD_ELECTION_TIME: o64[g] = L.election_time; break;
      case JG_FIELD_HEARTBEAT_TIME: o64[g] = role == JG_ROLE_LEADER ? L.heartbeat_time : 0; break;
      case JG_FIELD_VOTED_FOR: o32[g] = (L.flags & JGF_VOTED) ? L.voted_for : 0; break;
      case JG_FIELD_LEADER_ID: o32[g] = (role == JG_ROLE_FOLLOWER && (L.flags & JGF_HAS_LEADER)) ? L.leader_id : 0; break;
      case JG_FIELD_ELECTION_TIMEOUT: o32[g] = L.election_timeout; break;
      case JG_FIELD_QUEUED_REQS: o32[g] = L.queued; break;
      case JG_FIELD_VOTE_SEEN: o8[g] = role == JG_ROLE_CANDIDATE ? (uint8_t)(L.votes & 0xff) : 0; break;
      case JG_FIELD_VOTE_GRANTED: o8[g] = role == JG_ROLE_CANDIDATE ? (uint8_t)((L.votes >> 8) & 0xff) : 0; break;
      case JG_FIELD_HAS_VOTED: o8[g] = (L.flags & JGF_VOTED) ? 1 : 0; break;
      case JG_FIELD_ROLE: o8[g] = (uint8_t)role; break;
      case JG_FIELD_REPL_STATE: o8[g] = role == JG_ROLE_LEADER ? (uint8_t)((L.flags & JGF_REPL_MASK) >> JGF_REPL_SHIFT) : 0; break;
      case JG_FIELD_FAULT: o8[g] = (uint8_t)jg_fault(L); break;
      case JG_FIELD_HAS_LEADER: o8[g] = (role == JG_ROLE_FOLLOWER && (L.flags & JGF_HAS_LEADER)) ? 1 : 0; break;
      case JG_FIELD_SELF_SLOT: o8[g] = (uint8_t)jg_self(L); break;
      default: return -1;
    }
  }
  return 0;
}

// ---- the slow kernels' bodies, one lane, every shard in turn ----------------------------------------------------
static void collect_after_dense(Host* h) {
  JgDev& d = h->d;
  const uint32_t nx = *d.xq_n;
  std::vector<JgXqRec> x(h->xq.data(), h->xq.data() + nx);
  std::sort(x.begin(), x.end(), [](const JgXqRec& a, const JgXqRec& b) {  // the drain's merge order (josefine_gpu.hip::drain_finish)
    if (a.seq != b.seq) return a.seq < b.seq;
    if (a.row.group != b.row.group) return a.row.group < b.row.group;
    return a.k < b.k;
  });
  for (const JgXqRec& r : x) h->msgs.push_back(r.row);
  *d.xq_n = 0;
  const uint32_t nf = *d.fault_q_n;
  std::vector<JgFaultRec> f(d.fault_q, d.fault_q + nf);
  std::stable_sort(f.begin(), f.end(), [](const JgFaultRec& a, const JgFaultRec& b) { return a.seq != b.seq ? a.seq < b.seq : a.group < b.group; });
  h->faults.insert(h->faults.end(), f.begin(), f.end());
  *d.fault_q_n = 0;
  for (uint64_t& v : h->blk_dec) h->decisions += v, v = 0;
}
// jg_step_dense_leader with EVERY healthy leader handed to k_dense_slow<true> (the dense kernel's part for the others:
// an empty outbox row); answers: [R][G] JG_ANSWER words or null, o_beat / o_ae: null = no Tick
typedef int (*FastLeader)(Host*, const uint64_t* acks, uint32_t seq, int us, const JgLeaderNode* nd, int any);
typedef int (*FastFollower)(Host*, const JgFollowerArgs* a, int any);
static const uint64_t g_ones[2] = {~0ull, ~0ull};  // (an absent input column is a stride-0 view of one all-ones word)
extern "C" int hc_leader_half(Host* h, uint64_t now, const uint64_t* answers, const uint64_t* hbr_commit, jg_leader_beat* o_beat, uint64_t* o_ae,
                              FastLeader fast, int us) {
  JgDev& d = h->d;
  h->seq++;
  d.xq = h->xq.data(), d.xq_cap = (uint32_t)h->xq.size();
  JgLeaderNode nd{};
  nd.hbr_commit = hbr_commit, nd.packed = 1, nd.now = now, nd.ack_stride = answers ? 1 : 0;
  nd.o_beat = o_beat, nd.o_ae = o_ae;
  if (fast) {  // the dense kernel itself, a group at a time; what it cannot serve it marks for the slow body below
    const int rc = fast(h, answers ? answers : g_ones, h->seq, us, &nd, 0);
    if (rc) return rc;
  } else {
    for (uint32_t g = 0; g < d.G; g++) {
      const uint32_t f = d.flags[g];
      if (o_beat) {
        o_beat[g] = jg_leader_beat{0, JG_NO_ACK};
        for (uint32_t r = 0; r < d.R; r++) o_ae[(size_t)r * d.G + g] = JG_NO_ACK;
      }
      if ((f & (JGF_ROLE_MASK | JGF_FAULT_MASK)) == JG_ROLE_LEADER) d.defer_bits[g >> 6] |= 1ull << (g & 63u);
    }
  }
  gridDim.x = JG_SHARDS;
  for (uint32_t b = 0; b < JG_SHARDS; b++) {
    blockIdx.x = b;
    jg_dense_slow_body<true>(d, answers, 1, (size_t)d.R * d.G, h->seq, nd, false);
  }
  blockIdx.x = 0, gridDim.x = 1;
  collect_after_dense(h);
  d.xq = nullptr, d.xq_cap = 0;
  return (int)h->status[0];
}
// jg_step_dense_follower with EVERY live group handed to k_follower_slow
extern "C" int hc_follower_half(Host* h, uint64_t now, const jg_leader_beat* beat, const uint64_t* ae, const uint32_t* leader, uint32_t leader_id,
                                int tick, uint64_t* o_answer, uint64_t* o_hbc, FastFollower fast) {
  JgDev& d = h->d;
  h->seq++;
  d.xq = h->xq.data(), d.xq_cap = (uint32_t)h->xq.size();
  JgFollowerArgs a{};
  a.leader = leader, a.leader_id = leader_id, a.beat = beat, a.ae = ae, a.o_answer = o_answer, a.o_hbc = o_hbc;
  a.now = now, a.seq = h->seq, a.tick = tick ? 1 : 0;
  if (fast) {
    const int rc = fast(h, &a, 0);
    if (rc) return rc;
  } else {
    for (uint32_t g = 0; g < d.G; g++) {
      o_answer[g] = JG_NO_ACK;
      if (!(d.flags[g] & JGF_FAULT_MASK)) d.fdefer_bits[g >> 6] |= 1ull << (g & 63u);
    }
  }
  gridDim.x = JG_SHARDS;
  for (uint32_t b = 0; b < JG_SHARDS; b++) {
    blockIdx.x = b;
    jg_follower_slow_body(d, a);
  }
  blockIdx.x = 0, gridDim.x = 1;
  collect_after_dense(h);
  d.xq = nullptr, d.xq_cap = 0;
  return (int)h->status[0];
}

// ---- jg_step_node, synchronous: the device's row passes as they are, the two halves through the slow kernels' bodies ----
// josefine_gpu.hip::node_step in a few lines: k_node_prefill, k_node_classify, k_node_route over the unsorted rows; the
// general-path rows in (group, arrival) order through the state machine (the step's first sequence number); the leader half
// (arrival replay by nd.arr, fsm words) and the follower half over the mailbox columns the route pass filled; k_node_fsm_build.
struct NodeScratch {
  std::vector<uint64_t> answers, hbr_commit, token, f_ae, lt_max, lt_min, fsm_prev, fsm_mid, sparse_bits, sp_key;
  std::vector<jg_leader_beat> f_beat;
  std::vector<uint32_t> f_leader, cls, lf_max, lf_min, fsm_delta, arr, fo, sp_idx;
};
extern "C" int hc_step_node(Host* h, uint32_t n, const uint8_t* kind, const uint32_t* group, const uint32_t* from, const uint64_t* term,
                            const uint64_t* id, const uint64_t* aux, const uint8_t* flag, uint64_t nb, const uint64_t* blk_id,
                            const uint64_t* blk_next, uint64_t now, uint32_t flags, int uniform_self, uint32_t kinds_seen,
                            jg_leader_beat* o_beat, uint64_t* o_ae, uint64_t* o_answer, uint64_t* o_hbc, uint64_t* n_general,
                            FastLeader fast_l, FastFollower fast_f) {
  JgDev& d = h->d;
  const uint32_t G = d.G, R = d.R;
  const uint32_t halves = flags & (JG_NODE_LEADER_HALF | JG_NODE_FOLLOWER_HALF);
  const bool tick = (flags & JG_NODE_TICK) != 0;
  NodeScratch s;
  s.answers.assign((size_t)R * G, 0), s.hbr_commit.assign((size_t)R * G, 0), s.token.assign(G, 0), s.f_ae.assign(G, 0);
  s.lt_max.assign(G, 0), s.lt_min.assign(G, 0), s.fsm_prev.assign(G, 0), s.fsm_mid.assign(G, 0), s.sparse_bits.assign((G + 63) / 64, 0);
  s.f_beat.assign(G, jg_leader_beat{}), s.f_leader.assign(G, 0), s.cls.assign(G, 0), s.lf_max.assign(G, 0), s.lf_min.assign(G, 0);
  s.fsm_delta.assign(G, 0), s.arr.assign((size_t)2 * R * G, 0), s.fo.assign((size_t)2 * G, 0);
  s.sp_key.assign(n + 1, 0), s.sp_idx.assign(n + 1, 0);
  JgNodeCols c{};
  c.answers = s.answers.data(), c.hbr_commit = s.hbr_commit.data(), c.token = s.token.data(), c.f_beat = s.f_beat.data(), c.f_ae = s.f_ae.data();
  c.f_leader = s.f_leader.data(), c.cls = s.cls.data(), c.lt_max = s.lt_max.data(), c.lt_min = s.lt_min.data(), c.lf_max = s.lf_max.data();
  c.lf_min = s.lf_min.data(), c.arr = s.arr.data(), c.fo = s.fo.data(), c.sparse_bits = s.sparse_bits.data(), c.fsm_delta = s.fsm_delta.data();
  c.fsm_prev = s.fsm_prev.data(), c.fsm_mid = s.fsm_mid.data();
  const uint32_t seq0 = h->seq;
  const uint32_t both_beats = (kinds_seen & 3u) == 3u;
  blockIdx.x = 0, gridDim.x = 1;
  k_node_prefill(d, c, uniform_self, halves & JG_NODE_LEADER_HALF, halves & JG_NODE_FOLLOWER_HALF, both_beats, 0);
  h->seq = seq0 + 1;  // the general path's number: taken whether or not it runs
  *n_general = 0;
  if (n) {
    JgNodeRows rows{};
    rows.n = n, rows.group = group, rows.kind = kind, rows.from = from, rows.term = term, rows.id = id, rows.aux = aux, rows.flag = flag;
    rows.blk_id = blk_id, rows.blk_next = blk_next, rows.n_blocks = nb;
    uint32_t nsp[2] = {0, 0};
    k_node_classify(d, c, rows, uniform_self, halves, both_beats, 0);
    k_node_route(d, c, rows, uniform_self, both_beats, s.sp_key.data(), s.sp_idx.data(), nsp);
    if (h->status[0]) return (int)h->status[0];
    const uint32_t ns = nsp[0];
    *n_general = ns;
    if (ns) {  // (group, arrival index): what the bucket pass + k_bucket_order produce on the device
      std::vector<uint32_t> order(ns);
      std::iota(order.begin(), order.end(), 0u);
      std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return s.sp_key[a] < s.sp_key[b]; });
      std::vector<uint8_t> k2(ns), f2(ns);
      std::vector<uint32_t> g2(ns), fr2(ns);
      std::vector<uint64_t> t2(ns), i2(ns), a2(ns);
      for (uint32_t p = 0; p < ns; p++) {
        const uint32_t i = s.sp_idx[order[p]];
        k2[p] = kind[i], g2[p] = group[i], fr2[p] = from[i], t2[p] = term[i], i2[p] = id[i], a2[p] = aux[i], f2[p] = flag[i];
      }
      h->seq = seq0;  // (hc_step takes the next number itself)
      const int rc = hc_step(h, ns, k2.data(), g2.data(), fr2.data(), t2.data(), i2.data(), a2.data(), f2.data(), nb, blk_id, blk_next, now);
      if (rc) return rc;
    }
  }
  h->seq = seq0 + 1;
  d.xq = h->xq.data(), d.xq_cap = (uint32_t)h->xq.size();
  gridDim.x = JG_SHARDS;
  if (halves & JG_NODE_LEADER_HALF) {
    h->seq++;
    JgLeaderNode ln{};
    ln.hbr_commit = c.hbr_commit, ln.packed = 1, ln.ack_stride = 1, ln.now = now;
    if (tick) ln.o_beat = o_beat, ln.o_ae = o_ae;
    ln.fsm_delta = c.fsm_delta, ln.fsm_prev = c.fsm_prev, ln.fsm_mid = c.fsm_mid, ln.arr = c.arr, ln.col_mask = 0;
    if (fast_l) {
      gridDim.x = 1;
      const int rc = fast_l(h, c.answers, h->seq, uniform_self, &ln, 0);
      if (rc) return rc;
      gridDim.x = JG_SHARDS;
    } else {
      for (uint32_t g = 0; g < G; g++) {
        if (tick) {
          o_beat[g] = jg_leader_beat{0, JG_NO_ACK};
          for (uint32_t r = 0; r < R; r++) o_ae[(size_t)r * G + g] = JG_NO_ACK;
        }
        if ((d.flags[g] & (JGF_ROLE_MASK | JGF_FAULT_MASK)) == JG_ROLE_LEADER) d.defer_bits[g >> 6] |= 1ull << (g & 63u);
      }
    }
    for (uint32_t b = 0; b < JG_SHARDS; b++) {
      blockIdx.x = b;
      jg_dense_slow_body<true>(d, c.answers, 1, (size_t)R * G, h->seq, ln, false);
    }
  }
  if (halves & JG_NODE_FOLLOWER_HALF) {
    h->seq++;
    JgFollowerArgs a{};
    a.leader = c.f_leader, a.beat = c.f_beat, a.ae = c.f_ae, a.o_answer = o_answer, a.o_hbc = o_hbc;
    a.now = now, a.seq = h->seq, a.tick = tick ? 1 : 0, a.fsm_delta = c.fsm_delta, a.fsm_prev = c.fsm_prev;
    if (fast_f) {
      gridDim.x = 1;
      const int rc = fast_f(h, &a, 0);
      if (rc) return rc;
      gridDim.x = JG_SHARDS;
    } else {
      for (uint32_t g = 0; g < G; g++) {
        o_answer[g] = JG_NO_ACK;
        if (!(d.flags[g] & JGF_FAULT_MASK)) d.fdefer_bits[g >> 6] |= 1ull << (g & 63u);
      }
    }
    for (uint32_t b = 0; b < JG_SHARDS; b++) {
      blockIdx.x = b;
      jg_follower_slow_body(d, a);
    }
  }
  blockIdx.x = 0, gridDim.x = 1;
  collect_after_dense(h);
  d.xq = nullptr, d.xq_cap = 0;
  // the halves' fsm words -> rows, partitions ascending (k_node_fsm_build: one group per lane)
  std::vector<jg_fsm_row> fr((size_t)G * JGN_FSM_ROWS);
  std::vector<uint32_t> cnt(G, 0);
  std::vector<uint64_t> bsum(G + 1, 0);
  for (uint32_t g = 0; g < G; g++) {
    blockIdx.x = g;
    k_node_fsm_build(d, c, fr.data(), cnt.data(), bsum.data(), 0u);
    for (uint32_t k = 0; k < cnt[g]; k++) h->fsm.push_back(fr[(size_t)g * JGN_FSM_ROWS + k]);
  }
  blockIdx.x = 0;
  return (int)h->status[0];
}

// ---- a round of a cluster with per-partition leadership (josefine_gpu.hip::cluster_tables_any / cluster_launch_any): the
// claim kernel as it is, every node's leader half and follower half through the slow kernels' bodies with the cluster's
// mailboxes (owner[g] / offered[g]: JgLeaderNode, JgFollowerArgs) - the any-leader branches of those bodies on the host
extern "C" int hc_cluster_any_round(Host** hs, uint32_t R, uint64_t now, uint64_t* acks, uint64_t* hbr_commit, jg_leader_beat* o_beat,
                                    uint64_t* o_ae, uint8_t* owner, const uint64_t* offered, FastLeader fast_l, FastFollower fast_f) {
  const uint32_t G = hs[0]->d.G;
  JgClaimArgs ca{};
  ca.R = R, ca.G = G, ca.owner = owner;
  for (uint32_t r = 0; r < R; r++) ca.flags[r] = hs[r]->d.flags;
  blockIdx.x = 0, gridDim.x = 1;
  k_cluster_claim(ca);
  for (uint32_t r = 0; r < R; r++) hs[r]->seq += 2;  // leader half: seq - 1, follower half: seq
  for (uint32_t r = 0; r < R; r++) {
    Host* h = hs[r];
    JgDev& d = h->d;
    d.xq = h->xq.data(), d.xq_cap = (uint32_t)h->xq.size();
    JgLeaderNode nd{};
    nd.ack_stride = 1, nd.packed = 1, nd.hbr_commit = hbr_commit, nd.o_beat = o_beat, nd.o_ae = o_ae, nd.now = now;
    nd.owner = owner, nd.offered = offered;
    if (fast_l) {  // k_leader_node_tick_any's own per-group logic first
      const int rc = fast_l(h, acks, h->seq - 1, (int)r, &nd, 1);
      if (rc) return rc;
    } else {
      for (uint32_t g = 0; g < G; g++)
        if ((d.flags[g] & (JGF_ROLE_MASK | JGF_FAULT_MASK)) == JG_ROLE_LEADER) {
          d.defer_bits[g >> 6] |= 1ull << (g & 63u);
          if (owner[g] == r) {  // (what the dense half does before it hands an owned group over: jg_dense_outbox_none)
            o_beat[g] = jg_leader_beat{0, JG_NO_ACK};
            for (uint32_t q = 0; q < R; q++)
              if (q != r) o_ae[(size_t)q * G + g] = JG_NO_ACK;
          }
        }
    }
    gridDim.x = JG_SHARDS;
    for (uint32_t b = 0; b < JG_SHARDS; b++) {
      blockIdx.x = b;
      jg_dense_slow_body<true>(d, acks, 1, 0, h->seq - 1, nd, false);
    }
    blockIdx.x = 0, gridDim.x = 1;
    collect_after_dense(h);
    h->msgs_mark = h->msgs.size();
    d.xq = nullptr, d.xq_cap = 0;
    if (h->status[0]) return (int)h->status[0];
  }
  for (uint32_t r = 0; r < R; r++) {
    Host* h = hs[r];
    JgDev& d = h->d;
    d.xq = h->xq.data(), d.xq_cap = (uint32_t)h->xq.size();
    JgFollowerArgs a{};
    a.beat = o_beat, a.ae = o_ae + (size_t)r * G, a.o_answer = acks + (size_t)r * G, a.o_hbc = hbr_commit + (size_t)r * G;
    a.now = now, a.seq = h->seq, a.tick = 1, a.owner = owner, a.self_slot = r;
    if (fast_f) {  // k_follower_tick_dense_any's
      const int rc = fast_f(h, &a, 1);
      if (rc) return rc;
    } else {
      for (uint32_t g = 0; g < G; g++) {
        const uint32_t f = d.flags[g];
        const bool own_led = (f & (JGF_ROLE_MASK | JGF_FAULT_MASK)) == JG_ROLE_LEADER && owner[g] == r;
        if (own_led) continue;  // (the own slot's word of a group this node owns is nobody's: the dense half leaves it alone)
        if (f & JGF_FAULT_MASK) a.o_answer[g] = JG_NO_ACK;
        else d.fdefer_bits[g >> 6] |= 1ull << (g & 63u);
      }
    }
    gridDim.x = JG_SHARDS;
    for (uint32_t b = 0; b < JG_SHARDS; b++) {
      blockIdx.x = b;
      jg_follower_slow_body(d, a);
    }
    blockIdx.x = 0, gridDim.x = 1;
    collect_after_dense(h);
    d.xq = nullptr, d.xq_cap = 0;
    if (h->status[0]) return (int)h->status[0];
  }
  return 0;
}

// jg_step_dense_acks: the ack-only tick - the headline kernel's per-group body (its in-kernel general path over LDS included),
// k_dense_slow<false> for the leaders it hands over (chains that are not in FAST form)
extern "C" int hc_dense_acks(Host* h, const uint64_t* acks, FastLeader fast, int us) {
  JgDev& d = h->d;
  h->seq++;
  if (fast) {
    const int rc = fast(h, acks, h->seq, us, nullptr, 0);
    if (rc) return rc;
  } else {
    for (uint32_t g = 0; g < d.G; g++)
      if ((d.flags[g] & (JGF_ROLE_MASK | JGF_FAULT_MASK)) == JG_ROLE_LEADER) d.defer_bits[g >> 6] |= 1ull << (g & 63u);
  }
  JgLeaderNode none{};
  gridDim.x = JG_SHARDS;
  for (uint32_t b = 0; b < JG_SHARDS; b++) {
    blockIdx.x = b;
    jg_dense_slow_body<false>(d, acks, 1, (size_t)d.R * d.G, h->seq, none, false);
  }
  blockIdx.x = 0, gridDim.x = 1;
  collect_after_dense(h);
  return (int)h->status[0];
}

// jg_chain_compact: one lane per tree (k_chain_compact as it is)
extern "C" void hc_chain_compact(size_t n_trees, const uint64_t* off, const uint64_t* ids, const uint64_t* nexts, const uint64_t* commits, uint8_t* removed) {
  blockIdx.x = 0, threadIdx.x = 0, blockDim.x = 1, gridDim.x = 1;
  k_chain_compact(n_trees, off, ids, nexts, commits, removed);
}
// jg_step_dense_acks_device_n: T ticks per launch - the T-tick kernel's per-group body (state in registers across the ticks),
// k_dense_slow<false> replaying all T ticks for what it hands over
typedef int (*FastTicks)(Host*, const uint64_t* acks, uint32_t n_ticks, uint32_t seq0, int us);
extern "C" int hc_dense_acks_n(Host* h, const uint64_t* acks, uint32_t n_ticks, FastTicks fast, int us) {
  JgDev& d = h->d;
  h->seq++;
  const int rc = fast(h, acks, n_ticks, h->seq, us);
  if (rc) return rc;
  JgLeaderNode none{};
  gridDim.x = JG_SHARDS;
  for (uint32_t b = 0; b < JG_SHARDS; b++) {
    blockIdx.x = b;
    jg_dense_slow_body<false>(d, acks, n_ticks, (size_t)d.R * d.G, h->seq, none, false);
  }
  blockIdx.x = 0, gridDim.x = 1;
  collect_after_dense(h);
  h->seq += n_ticks - 1;
  return (int)h->status[0];
}

// the vote half (jg_votes.h) over one node: the last round's mail in, this round's mail (its answer words) and its
// exceptional rows (with their emission index) out - the rows are handed to the caller, who plays the transport
extern "C" int hc_vote_half(Host* h, uint32_t self, uint64_t now, uint32_t step, uint32_t need, const JgVoteMail* in, const JgVoteMail* out,
                            jg_msg_row* x_rows, uint32_t* x_k, size_t x_cap, size_t* x_n) {
  JgDev& d = h->d;
  h->seq++;
  d.xq = h->xq.data(), d.xq_cap = (uint32_t)h->xq.size();
  uint32_t st[JG_VOTE_ST_WORDS];  // (a lane's stretch cursors: LDS on the device)
  for (uint32_t g = 0; g < d.G; g++) h->decisions += jg_vote_half_group(d, g, self, *in, *out, need, now, h->seq, step, st, 1);
  const uint32_t nx = *d.xq_n;
  std::vector<JgXqRec> x(h->xq.data(), h->xq.data() + nx);
  std::sort(x.begin(), x.end(), [](const JgXqRec& a, const JgXqRec& b) { return a.row.group != b.row.group ? a.row.group < b.row.group : a.k < b.k; });
  *x_n = nx;
  for (size_t i = 0; i < nx && i < x_cap; i++) x_rows[i] = x[i].row, x_k[i] = x[i].k;
  *d.xq_n = 0;
  d.xq = nullptr, d.xq_cap = 0;
  const uint32_t nf = *d.fault_q_n;
  std::vector<JgFaultRec> f(d.fault_q, d.fault_q + nf);
  std::stable_sort(f.begin(), f.end(), [](const JgFaultRec& a, const JgFaultRec& b) { return a.seq != b.seq ? a.seq < b.seq : a.group < b.group; });
  h->faults.insert(h->faults.end(), f.begin(), f.end());
  *d.fault_q_n = 0;
  return (int)h->status[0];
}
// the transport's side of the mail, row by row as the kernels' lanes call it: the census of one sender's emitted rows ...
extern "C" void hc_votes_census(const JgVoteMail* m, uint32_t src, uint32_t sender_id, const jg_msg_row* rows, const uint32_t* step, const uint32_t* k,
                                const uint32_t* dests, size_t n) {
  for (size_t i = 0; i < n; i++) jg_votes_census_row(*m, src, sender_id, rows[i], step[i], k[i], dests[i]);
}
// ... the validation of the counts, once the census is complete ...
extern "C" void hc_votes_validate(const JgVoteMail* m, uint32_t need) {
  for (uint32_t g = 0; g < m->G; g++) jg_votes_validate_group(*m, g, need);
}
// ... which of them (per addressee: a bit mask within `dests`) still travel as rows after that ...
extern "C" void hc_votes_travels(const JgVoteMail* m, uint32_t sender_id, const jg_msg_row* rows, const uint32_t* k, const uint32_t* dests, size_t n,
                                 uint32_t need, uint32_t* out) {
  for (size_t i = 0; i < n; i++) {
    uint32_t t = 0;
    for (uint32_t b = dests[i]; b; b &= b - 1)
      if (jg_votes_row_travels(*m, sender_id, rows[i], k[i], (uint32_t)__builtin_ctz(b), need)) t |= b & (~b + 1u);
    out[i] = t;
  }
}
// ... and the answer words of sender s that have to go as rows after all (out: up to 255 rows per partition)
extern "C" size_t hc_votes_expand(Host* h, const JgVoteMail* m, uint32_t s, uint32_t need, jg_msg_row* rows, uint32_t* to, uint32_t* step, uint32_t* k,
                                  size_t cap) {
  size_t n = 0;
  for (uint32_t g = 0; g < m->G; g++) {
    uint32_t t = 0, st = 0, k0 = 0;
    const uint32_t c = jg_votes_expand_count(*m, s, g, need, &t, &st, &k0);
    for (uint32_t j = 0; j < c && n < cap; j++, n++) rows[n] = jg_votes_expand_row(*m, h->d.node_ids, s, g, j), to[n] = t, step[n] = st, k[n] = k0 + j;
  }
  return n;
}
'''

FAST = r'''
// the dense kernels' per-group logic on the host: JG_BLOCK = 64, one group per call with the lane where the group puts it
// (blockIdx.x = g / 64, threadIdx.x = g % 64), so that the deferral bitmaps and the counters' ballots come out as on the
// device; the wave reductions at the kernels' ends (jg_wave_count) are replaced by adding up what the bodies return
#define JG_BLOCK 64
#include "jg_dense.h"
#include "jg_follower.h"
#include "host.h"

template <int R>
static void leader_group(Host* h, const uint64_t* acks, uint32_t seq, int us, const JgLeaderNode* nd, int any) {
  const JgDev& d = h->d;
  const JgDenseHot hot = jg_dense_hot_of(d);
  static uint64_t sm[2 * R][JG_BLOCK];
  JgDecCount dec;
  if (any) dec = jg_dense_tick_body<R, true, true, true, false, true>(hot, &d, acks, seq, (uint32_t)us, *nd, nullptr);
  else if (nd && nd->fsm_delta) dec = us >= 0 ? jg_dense_tick_body<R, true, true, true, true>(hot, &d, acks, seq, (uint32_t)us, *nd, nullptr)
                                              : jg_dense_tick_body<R, false, true, true, true>(hot, &d, acks, seq, 0, *nd, nullptr);
  else if (nd) dec = us >= 0 ? jg_dense_tick_body<R, true, true, true, false>(hot, &d, acks, seq, (uint32_t)us, *nd, nullptr)
                             : jg_dense_tick_body<R, false, true, true, false>(hot, &d, acks, seq, 0, *nd, nullptr);
  else {
    JgLeaderNode none{};
    dec = us >= 0 ? jg_dense_tick_body<R, true, false, false>(hot, &d, acks, seq, (uint32_t)us, none, sm)
                  : jg_dense_tick_body<R, false, false, false>(hot, &d, acks, seq, 0, none, sm);
  }
  h->decisions += dec.lane;
}
VIS int hf_leader_tick(Host* h, const uint64_t* acks, uint32_t seq, int us, const JgLeaderNode* nd, int any) {
  const uint32_t G = h->d.G;
  gridDim.x = (G + 63) / 64;
  for (uint32_t g = 0; g < G; g++) {
    blockIdx.x = g >> 6, threadIdx.x = g & 63u;
    switch (h->d.R) {
      case 1: leader_group<1>(h, acks, seq, us, nd, any); break;
      case 2: leader_group<2>(h, acks, seq, us, nd, any); break;
      case 3: leader_group<3>(h, acks, seq, us, nd, any); break;
      case 4: leader_group<4>(h, acks, seq, us, nd, any); break;
      case 5: leader_group<5>(h, acks, seq, us, nd, any); break;
      case 6: leader_group<6>(h, acks, seq, us, nd, any); break;
      case 7: leader_group<7>(h, acks, seq, us, nd, any); break;
      default: leader_group<8>(h, acks, seq, us, nd, any); break;
    }
  }
  blockIdx.x = 0, threadIdx.x = 0, gridDim.x = 1;
  return (int)h->status[0];
}

template <int R>
static void ticks_group(Host* h, const uint64_t* acks, uint32_t n_ticks, uint32_t seq0, int us) {
  const size_t stride = (size_t)h->d.R * h->d.G;
  h->decisions += us >= 0 ? jg_dense_ticks_body<R, true>(h->d, acks, n_ticks, stride, seq0, (uint32_t)us)
                          : jg_dense_ticks_body<R, false>(h->d, acks, n_ticks, stride, seq0, 0);
}
VIS int hf_leader_ticks(Host* h, const uint64_t* acks, uint32_t n_ticks, uint32_t seq0, int us) {
  const uint32_t G = h->d.G;
  gridDim.x = (G + 63) / 64;
  for (uint32_t g = 0; g < G; g++) {
    blockIdx.x = g >> 6, threadIdx.x = g & 63u;
    switch (h->d.R) {
      case 1: ticks_group<1>(h, acks, n_ticks, seq0, us); break;
      case 2: ticks_group<2>(h, acks, n_ticks, seq0, us); break;
      case 3: ticks_group<3>(h, acks, n_ticks, seq0, us); break;
      case 4: ticks_group<4>(h, acks, n_ticks, seq0, us); break;
      case 5: ticks_group<5>(h, acks, n_ticks, seq0, us); break;
      case 6: ticks_group<6>(h, acks, n_ticks, seq0, us); break;
      case 7: ticks_group<7>(h, acks, n_ticks, seq0, us); break;
      default: ticks_group<8>(h, acks, n_ticks, seq0, us); break;
    }
  }
  blockIdx.x = 0, threadIdx.x = 0, gridDim.x = 1;
  return (int)h->status[0];
}
VIS int hf_follower_tick(Host* h, const JgFollowerArgs* a, int any) {
  const uint32_t G = h->d.G;
  gridDim.x = (G + 63) / 64;
  for (uint32_t g = 0; g < G; g++) {
    blockIdx.x = g >> 6, threadIdx.x = g & 63u;
    if (any) jg_follower_fast_body<true>(h->d, *a);
    else jg_follower_fast_body<false>(h->d, *a);
  }
  blockIdx.x = 0, threadIdx.x = 0, gridDim.x = 1;
  return (int)h->status[0];
}
'''

_lib = None
_fast = None


def _patched_sources():
    """a copy of the device headers with ONE textual substitution: jg_block_count sizes its per-wave scratch as
    JG_BLOCK / 64 words - zero with the one-lane JG_BLOCK this build uses"""
    import shutil
    tmp = tempfile.mkdtemp(prefix="jg_host_src_")
    dst = os.path.join(tmp, "josefine_amd", "csrc")
    shutil.copytree(CSRC, dst, ignore=shutil.ignore_patterns("*.so", "*.o"))
    os.makedirs(os.path.join(tmp, "include"))
    shutil.copy(os.path.join(ROOT, "include", "josefine_gpu.h"), os.path.join(tmp, "include"))
    p = os.path.join(dst, "jg_dense.h")
    text = open(p).read()
    a = text.index("__device__ __forceinline__ void jg_block_count(")
    b = text.index("}\n", text.index("__hip_atomic_fetch_add(&slots[blockIdx.x]", a))
    body = text[a:b]
    assert body.count("JG_BLOCK / 64") == 2
    open(p, "w").write(text[:a] + body.replace("JG_BLOCK / 64", "((JG_BLOCK + 63) / 64)") + text[b:])
    # ... and the transport's scan kernel asserts its tile of 4 buckets per thread of a 256-thread workgroup: not called here
    p = os.path.join(dst, "jg_route.h")
    text = open(p).read()
    line = '  static_assert(JG_ROUTE_SCAN_TILE == 4 * JG_BLOCK, "4 buckets per thread");'
    assert text.count(line) == 2  # (k_route_scan_all and the two-launch form's k_route_scan)
    open(p, "w").write(text.replace(line, "  // (one-lane host build: " + line.strip() + ")"))
    return dst


def build():
    """g++ -> a shared library in a temporary directory (once per process)"""
    global _lib
    if _lib is not None:
        return _lib
    src = _patched_sources()
    kernels = open(os.path.join(src, "jg_kernels.h")).read()
    a = kernels.index("__global__ void k_init_groups")
    a = kernels.index("JgLane L;", a)
    b = kernels.index("jg_store(d, L);", a) + len("jg_store(d, L);")
    init_body = kernels[a:b]  # RaftHandle::new for one group, as the init kernel does it
    tmp = tempfile.mkdtemp(prefix="jg_host_compiled_")
    os.makedirs(os.path.join(tmp, "shim", "hip"))
    open(os.path.join(tmp, "shim", "hip", "hip_runtime.h"), "w").write(HIP_SHIM)
    cpp, so = os.path.join(tmp, "host_compiled.cpp"), os.path.join(tmp, "libhost_compiled.so")
    open(os.path.join(tmp, "host.h"), "w").write(HOST_H)
    open(cpp, "w").write(HARNESS.replace("@INIT_BODY@", init_body))
    cc = ["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-Wno-unused-function", "-Wl,-Bsymbolic", f"-I{os.path.join(tmp, 'shim')}", f"-I{src}", f"-I{tmp}"]
    if os.environ.get("JG_HOST_SANITIZE"):  # e.g. "undefined" or "address,undefined" (the latter wants LD_PRELOAD=libasan.so for python)
        cc += ["-g", "-fno-omit-frame-pointer", f"-fsanitize={os.environ['JG_HOST_SANITIZE']}", "-fno-sanitize-recover=all"]
    subprocess.run(cc + ["-o", so, cpp], check=True)
    # the dense kernels' per-group logic: a library of its own (JG_BLOCK = 64 there, 1 here: nothing of the two may be merged)
    global _fast
    cpp2, so2 = os.path.join(tmp, "fast.cpp"), os.path.join(tmp, "libhost_fast.so")
    open(cpp2, "w").write(FAST)
    subprocess.run(cc + ["-fvisibility=hidden", "-o", so2, cpp2], check=True)
    _fast = C.CDLL(so2)
    lib = C.CDLL(so)
    lib.hc_create.restype = C.c_void_p
    lib.hc_create.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.hc_destroy.argtypes = [C.c_void_p]
    lib.hc_step.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 7 + [C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    for f in (lib.hc_drain_messages, lib.hc_drain_applies, lib.hc_drain_faults):
        f.restype = C.c_size_t
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.hc_counters.argtypes = [C.c_void_p, C.c_void_p]
    lib.hc_msgs_mark.restype = C.c_size_t
    lib.hc_msgs_mark.argtypes = [C.c_void_p]
    lib.hc_read.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    lib.hc_leader_half.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.hc_follower_half.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hc_vote_half.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32] + [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]
    lib.hc_votes_census.restype = None
    lib.hc_votes_census.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 4 + [C.c_size_t]
    lib.hc_votes_validate.restype = None
    lib.hc_votes_validate.argtypes = [C.c_void_p, C.c_uint32]
    lib.hc_votes_travels.restype = None
    lib.hc_votes_travels.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 3 + [C.c_size_t, C.c_uint32, C.c_void_p]
    lib.hc_votes_expand.restype = C.c_size_t
    lib.hc_votes_expand.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 4 + [C.c_size_t]
    lib.hc_dense_acks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.hc_dense_acks_n.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
    lib.hc_chain_compact.argtypes = [C.c_size_t] + [C.c_void_p] * 5
    lib.hc_cluster_any_round.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64] + [C.c_void_p] * 8
    lib.hc_step_node.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 7 + [C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32] + \
        [C.c_void_p] * 7
    _lib = lib
    return lib


class _JgVoteMail(C.Structure):
    _fields_ = [("R", C.c_uint32), ("G", C.c_uint32), ("words", C.c_uint32), ("pad", C.c_uint32), ("rec", C.c_void_p), ("rowmail", C.c_void_p),
                ("wordmail", C.c_void_p)]


VOTE_REC = np.dtype([("q_term", "<u8"), ("q_head", "<u8"), ("a_term", "<u8"), ("q_ctl", "<u4"), ("a_ctl", "<u4")])  # jg_votes.h: JgVoteRec
assert VOTE_REC.itemsize == 32


class VoteMail:
    """one round's election mail (jg_votes.h: JgVoteMail) in host arrays"""

    def __init__(self, R, G):
        self.R, self.G, self.words = R, G, (G + 63) // 64
        # the device's layout is one 32-byte record per (partition, sender), partition-major ([G][R], jg_vote_at); the tests
        # index [sender, partition] field by field: transposed views of the record array's fields
        self._rec = np.zeros((G, R), VOTE_REC)
        self.q_term, self.q_head, self.q_ctl, self.a_term, self.a_ctl = (self._rec[f].T for f in ("q_term", "q_head", "q_ctl", "a_term", "a_ctl"))
        self.rowmail, self.wordmail = np.zeros((R, self.words), np.uint64), np.zeros((R, self.words), np.uint64)
        self.c = _JgVoteMail(R, G, self.words, 0, self._rec.ctypes.data, self.rowmail.ctypes.data, self.wordmail.ctypes.data)

    def clear(self):  # (k_votes_clear: the control words and the bitmaps; the term / head columns keep their garbage)
        self.q_ctl[:], self.a_ctl[:], self.rowmail[:], self.wordmail[:] = 0, 0, 0, 0

    @staticmethod
    def set_bits(bitmap, d, groups):
        for g in groups:
            bitmap[d, g >> 6] |= np.uint64(1) << np.uint64(g & 63)

    @staticmethod
    def bits(bitmap, d, G):
        return ((bitmap[d, np.arange(G) >> 6] >> (np.arange(G) & 63).astype(np.uint64)) & np.uint64(1)).astype(bool)


class HostCompiled:
    """the subset of BatchedRaft's interface the sparse parity suites use, over the host-compiled state machine"""

    fast = True  # the dense halves: the dense kernels' own per-group logic first, the slow bodies for what it defers (False: all slow)

    def __init__(self, n_groups, n_replicas=1, node_ids=None, self_slots=None, seed=0, device_id=0, group_base=0, flags=0,
                 heartbeat_timeout_ms=100, election_timeout_ms=(500, 1000)):
        self.lib = build()
        self.G, self.R = int(n_groups), int(n_replicas)
        self.node_ids = list(node_ids) if node_ids is not None else list(range(1, self.R + 1))
        ids = np.array(self.node_ids, np.uint32)
        ss = None if self_slots is None else np.ascontiguousarray(self_slots, np.uint8)
        self._h = self.lib.hc_create(self.G, self.R, ids.ctypes.data, None if ss is None else ss.ctypes.data, int(seed), int(group_base),
                                     int(flags), int(heartbeat_timeout_ms), int(election_timeout_ms[0]), int(election_timeout_ms[1]))
        self._pending = []

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.hc_destroy(self._h)
            self._h = None

    def submit_columns(self, kind, group, from_=None, term=None, id=None, aux=None, flag=None, blk_id=None, blk_next=None):
        n = len(kind)
        z = lambda v, dt: np.zeros(n, dt) if v is None else np.ascontiguousarray(v, dt)  # noqa: E731
        self._pending.append(dict(kind=np.ascontiguousarray(kind, np.uint8), group=np.ascontiguousarray(group, np.uint32), from_=z(from_, np.uint32),
                                  term=z(term, np.uint64), id=z(id, np.uint64), aux=z(aux, np.uint64), flag=z(flag, np.uint8),
                                  blk_id=np.zeros(0, np.uint64) if blk_id is None else np.ascontiguousarray(blk_id, np.uint64),
                                  blk_next=np.zeros(0, np.uint64) if blk_next is None else np.ascontiguousarray(blk_next, np.uint64)))

    def step(self, now_ms=0):
        if not self._pending:
            return
        parts, self._pending = self._pending, []
        shift = 0
        for p in parts:  # (AppendEntries rows index the batch's block side arrays: jg_submit shifts them as batches are appended)
            ae = p["kind"] == capi.CMD_APPEND_ENTRIES
            p["id"] = np.where(ae, p["id"] + np.uint64(shift), p["id"])
            shift += len(p["blk_id"])
        cols = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        n, nb = len(cols["kind"]), len(cols["blk_id"])
        bid = cols["blk_id"] if nb else np.zeros(1, np.uint64)
        bnx = cols["blk_next"] if nb else np.zeros(1, np.uint64)
        rc = self.lib.hc_step(self._h, n, cols["kind"].ctypes.data, cols["group"].ctypes.data, cols["from_"].ctypes.data, cols["term"].ctypes.data,
                              cols["id"].ctypes.data, cols["aux"].ctypes.data, cols["flag"].ctypes.data, nb, bid.ctypes.data, bnx.ctypes.data, int(now_ms))
        assert rc == 0, f"host-compiled state machine: error {rc}"

    def submit(self, group, cmd):
        """one Command for one group (josefine_amd.engine.Command), applied at the next step in submit order"""
        c = cmd
        if c.kind == capi.CMD_APPEND_ENTRIES:
            bi, bn = [b[0] for b in c.blocks], [b[1] for b in c.blocks]
            self.submit_columns([c.kind], [group], [c.from_], [c.term], [0], [len(bi)], [c.flag], bi, bn)
        else:
            self.submit_columns([c.kind], [group], [c.from_], [c.term], [c.id], [c.aux], [c.flag])

    def apply(self, group, cmd, now_ms=0):
        self.submit(group, cmd)
        self.step(now_ms)
        return self.handle(group)

    def handle(self, group):
        from josefine_amd.engine import RaftHandle
        return RaftHandle(self, group)

    def apply_all(self, cmd, now_ms=0):
        n = self.G
        self.submit_columns(np.full(n, cmd.kind, np.uint8), np.arange(n, dtype=np.uint32), np.full(n, cmd.from_, np.uint32),
                            np.full(n, cmd.term, np.uint64), np.full(n, cmd.id, np.uint64), np.full(n, cmd.aux, np.uint64), np.full(n, cmd.flag, np.uint8))
        self.step(now_ms)

    def _drain(self, fn, dtype):
        n = fn(self._h, None, 0)
        out = np.zeros(n, dtype=dtype)
        if n:
            fn(self._h, out.ctypes.data, n)
        return out

    def drain_messages(self, copy=True):
        return self._drain(self.lib.hc_drain_messages, capi.MSG_DTYPE)

    def drain_applies(self, copy=True):
        return self._drain(self.lib.hc_drain_applies, capi.FSM_DTYPE)

    def drain_faults(self):
        return self._drain(self.lib.hc_drain_faults, capi.FAULT_DTYPE)

    def counters(self):
        arr = (C.c_uint64 * 4)()
        self.lib.hc_counters(self._h, arr)
        return {"commands": arr[0], "decisions": arr[1], "dense_group_steps": 0, "launches": 0}

    def read(self, field_name, replica=0, g0=0, n=None):
        fld = capi.FIELD_NAMES[field_name]
        out = np.zeros(self.G, dtype=capi.FIELD_DTYPES[fld])
        assert self.lib.hc_read(self._h, fld, int(replica), out.ctypes.data) == 0
        n = self.G - g0 if n is None else n
        return out[g0:g0 + n]

    def _us(self):
        slots = self.read("self_slot")
        return int(slots[0]) if (slots == slots[0]).all() else -1

    def _fast_leader(self):
        return C.cast(_fast.hf_leader_tick, C.c_void_p) if self.fast else None

    def _fast_follower(self):
        return C.cast(_fast.hf_follower_tick, C.c_void_p) if self.fast else None

    # -- the dense halves: the dense kernels' per-group logic, then the slow kernels' bodies (the interface of BatchedRaft's column forms) --
    def vote_half(self, self_slot, now_ms, words, R=None, step=0):
        """jg_votes.h's receiving half on words given as plain columns (dict of [R, G] arrays: q_term, q_head, q_n, q_at, a_term,
        a_n, a_at, a_bits, a_to; q_at / a_at: the ord - phase << 8 | emission index - of a stretch's first copy; any number of
        copies) -> (this node's answer word columns - `at`: its ord, `step` << 8 | emission index -, its exceptional rows, their
        emission indices)"""
        G, R = self.G, self.R
        inn, out = VoteMail(R, G), VoteMail(R, G)
        n = words["q_n"].astype(np.uint32)
        at = words["q_at"].astype(np.uint32)
        inn.q_term[:], inn.q_head[:] = words["q_term"], words["q_head"]
        inn.q_ctl[:] = n | (n * at + n * (n - np.minimum(n, 1)) // 2) << 8
        inn.a_term[:] = words["a_term"]
        a_n, bits = words["a_n"].astype(np.uint32), words["a_bits"].astype(np.uint32)
        inn.a_ctl[:] = np.where(a_n != 0, a_n | words["a_at"].astype(np.uint32) << 8 | (bits & 1) << 19 | (bits >> 1 & 1) << 20 | words["a_to"].astype(np.uint32) << 21, 0)
        mine = (a_n != 0) & (words["a_to"] == self_slot)
        mine[self_slot] = False
        has = ((n != 0) | mine)
        has[self_slot] = False
        inn.set_bits(inn.wordmail, self_slot, np.nonzero(has.any(axis=0))[0])
        xrows, xk = self.vote_half_mail(self_slot, now_ms, inn, out, step=step, need=0)
        c = out.a_ctl[self_slot]
        res = dict(term=out.a_term[self_slot].copy(), n=(c & 0xff).astype(np.uint8), at=(c >> 8 & 0x7ff).astype(np.uint32),
                   bits=((c >> 19) & 3).astype(np.uint8), to=((c >> 21) & 7).astype(np.uint8))
        return res, xrows, xk

    def vote_half_mail(self, self_slot, now_ms, mail_in, mail_out, step, need):
        """the receiving half on the packed mail (VoteMail): last round's in, this round's out -> (exceptional rows, their emission indices)"""
        assert not self._pending
        cap = (self.R + 3) * self.G + 64
        xr = np.zeros(cap, dtype=capi.MSG_DTYPE)
        xk = np.zeros(cap, np.uint32)
        xn = C.c_size_t(0)
        rc = self.lib.hc_vote_half(self._h, int(self_slot), int(now_ms), int(step), int(need), C.addressof(mail_in.c), C.addressof(mail_out.c),
                                   xr.ctypes.data, xk.ctypes.data, cap, C.byref(xn))
        assert rc == 0, f"host-compiled vote half: error {rc}"
        return xr[:xn.value].copy(), xk[:xn.value].copy()

    def step_dense_acks(self, acks):
        """jg_step_dense_acks: the ack-only leader tick from a host [R, G] array"""
        assert not self._pending
        a = np.ascontiguousarray(acks, np.uint64)
        assert a.shape == (self.R, self.G)
        rc = self.lib.hc_dense_acks(self._h, a.ctypes.data, self._fast_leader(), self._us())
        assert rc == 0, f"host-compiled ack-only tick: error {rc}"

    def step_dense_acks_n(self, acks):
        """jg_step_dense_acks_device_n: T consecutive ticks from a host [T, R, G] array, state read and written once"""
        assert not self._pending
        a = np.ascontiguousarray(acks, np.uint64)
        assert a.ndim == 3 and a.shape[1:] == (self.R, self.G)
        rc = self.lib.hc_dense_acks_n(self._h, a.ctypes.data, a.shape[0], C.cast(_fast.hf_leader_ticks, C.c_void_p), self._us())
        assert rc == 0, f"host-compiled T-tick launch: error {rc}"

    def chain_compact(self, trees):
        trees = list(trees)
        off = np.zeros(len(trees) + 1, np.uint64)
        for i, (blocks, _) in enumerate(trees):
            off[i + 1] = off[i] + len(blocks)
        ids = np.ascontiguousarray(np.array([b[0] for t in trees for b in t[0]], dtype=np.uint64))
        nexts = np.ascontiguousarray(np.array([b[1] for t in trees for b in t[0]], dtype=np.uint64))
        commits = np.array([t[1] for t in trees], dtype=np.uint64)
        removed = np.zeros(max(int(off[-1]), 1), np.uint8)
        self.lib.hc_chain_compact(len(trees), off.ctypes.data, ids.ctypes.data if len(ids) else None, nexts.ctypes.data if len(nexts) else None,
                                  commits.ctypes.data, removed.ctypes.data)
        return [removed[int(off[i]):int(off[i + 1])].copy() for i in range(len(trees))]

    def step_dense_leader(self, now_ms=0, acks=None, hbr_has=None, hbr_commit=None, tick=True):
        assert not self._pending
        G, R = self.G, self.R
        ans = hbc = None
        if acks is not None or hbr_has is not None:
            a = np.full((R, G), capi.NO_ACK, np.uint64) if acks is None else np.asarray(acks, np.uint64).reshape(R, G)
            if acks is None:
                a[self.read("self_slot"), np.arange(G)] = 0
            hh = np.full((R, G), capi.HB_NONE, np.uint8) if hbr_has is None else np.asarray(hbr_has, np.uint8).reshape(R, G)
            hbc = np.ascontiguousarray(np.zeros((R, G), np.uint64) if hbr_commit is None else np.asarray(hbr_commit, np.uint64).reshape(R, G))
            ans = np.ascontiguousarray(capi.pack_answers(a, hh))
        beat = np.zeros((G, 2), np.uint64)
        ae = np.full((R, G), capi.NO_ACK, np.uint64)
        rc = self.lib.hc_leader_half(self._h, int(now_ms), None if ans is None else ans.ctypes.data, None if hbc is None else hbc.ctypes.data,
                                     beat.ctypes.data if tick else None, ae.ctypes.data if tick else None, self._fast_leader(), self._us())
        assert rc == 0, f"host-compiled leader half: error {rc}"
        if not tick:
            return None
        own = self.read("self_slot")
        ae[own, np.arange(G)] = np.uint64(capi.NO_ACK)  # (the own slot's row is nobody's mail)
        ae_from, ae_n = capi.unpack_ae(ae)
        return {"term": np.ascontiguousarray(beat[:, 0]), "hb_commit": np.ascontiguousarray(beat[:, 1]), "ae_from": ae_from, "ae_n": ae_n}

    def step_dense_follower(self, now_ms, term, hb_commit, ae_from, ae_n, leader=None, leader_id=0, tick=True):
        assert not self._pending
        G = self.G
        beat = np.ascontiguousarray(np.stack([np.asarray(term, np.uint64), np.asarray(hb_commit, np.uint64)], axis=1))
        ae = np.ascontiguousarray(capi.pack_ae(np.asarray(ae_from, np.uint64), np.asarray(ae_n, np.uint8)))
        ld = None if leader is None else np.ascontiguousarray(leader, np.uint32)
        ans = np.zeros(G, np.uint64)
        hbc = np.zeros(G, np.uint64)
        rc = self.lib.hc_follower_half(self._h, int(now_ms), beat.ctypes.data, ae.ctypes.data, None if ld is None else ld.ctypes.data, int(leader_id),
                                       1 if tick else 0, ans.ctypes.data, hbc.ctypes.data, self._fast_follower())
        assert rc == 0, f"host-compiled follower half: error {rc}"
        ack_head, hb_has = capi.unpack_answers(ans)
        return {"ack_head": ack_head, "hb_commit": np.where(hb_has != capi.HB_NONE, hbc, 0).astype(np.uint64), "hb_has": hb_has}

    def step_node(self, now_ms=0, leader=True, follower=True, tick=True, async_=False, between=None):
        """jg_step_node (synchronous) over the rows submitted since the last step"""
        parts, self._pending = self._pending, []
        G, R = self.G, self.R
        if parts:
            shift = 0
            for p in parts:
                ae = p["kind"] == capi.CMD_APPEND_ENTRIES
                p["id"] = np.where(ae, p["id"] + np.uint64(shift), p["id"])
                shift += len(p["blk_id"])
            cols = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        else:
            cols = dict(kind=np.zeros(0, np.uint8), group=np.zeros(0, np.uint32), from_=np.zeros(0, np.uint32), term=np.zeros(0, np.uint64),
                        id=np.zeros(0, np.uint64), aux=np.zeros(0, np.uint64), flag=np.zeros(0, np.uint8), blk_id=np.zeros(0, np.uint64), blk_next=np.zeros(0, np.uint64))
        n, nb = len(cols["kind"]), len(cols["blk_id"])
        pad = lambda v, dt: v if len(v) else np.zeros(1, dt)  # noqa: E731
        seen = (1 if (cols["kind"] == capi.CMD_APPEND_ENTRIES).any() else 0) | (2 if (cols["kind"] == capi.CMD_HEARTBEAT).any() else 0)
        flags = (capi.NODE_LEADER_HALF if leader else 0) | (capi.NODE_FOLLOWER_HALF if follower else 0) | (capi.NODE_TICK if tick else 0)
        slots = self.read("self_slot")
        us = int(slots[0]) if (slots == slots[0]).all() else -1
        beat = np.zeros((G, 2), np.uint64)
        ae = np.full((R, G), capi.NO_ACK, np.uint64)
        ans = np.full(G, capi.NO_ACK, np.uint64)
        hbc = np.zeros(G, np.uint64)
        ngen = C.c_uint64(0)
        a = {k: pad(cols[k], cols[k].dtype) for k in cols}
        rc = self.lib.hc_step_node(self._h, n, a["kind"].ctypes.data, a["group"].ctypes.data, a["from_"].ctypes.data, a["term"].ctypes.data,
                                   a["id"].ctypes.data, a["aux"].ctypes.data, a["flag"].ctypes.data, nb, a["blk_id"].ctypes.data, a["blk_next"].ctypes.data,
                                   int(now_ms), flags, us, seen, beat.ctypes.data, ae.ctypes.data, ans.ctypes.data, hbc.ctypes.data, C.byref(ngen),
                                   self._fast_leader(), self._fast_follower())
        assert rc == 0, f"host-compiled node step: error {rc}"
        if between is not None:
            between()
        if us >= 0:
            ae[us] = np.uint64(capi.NO_ACK)  # (the own slot's row is nobody's mail)
        has_l, has_f = leader and tick, follower
        return {"beat_term": np.ascontiguousarray(beat[:, 0]) if has_l else None, "beat_commit": np.ascontiguousarray(beat[:, 1]) if has_l else None,
                "ae": ae if has_l else None, "answer": ans if has_f else None, "hb_commit": hbc if has_f else None,
                "rows": n, "rows_general": int(ngen.value), "bytes_h2d": 0, "bytes_d2h": 0}


def host_any_leader_cluster(G, R, seed=3):
    """tests/dense_node.py::AnyLeaderCluster whose dense round is the DEVICE's own any-leader round on the host
    (hc_cluster_any_round: k_cluster_claim + the slow bodies' owner / offered branches) instead of the numpy statement"""
    from dense_node import AnyLeaderCluster

    class HostAnyLeaderCluster(AnyLeaderCluster):
        def __init__(self):
            super().__init__(HostCompiled, G, R, seed=seed)
            self.words = np.full((R, G), capi.NO_ACK, np.uint64)   # the cluster's answer words (acks + HeartbeatResponse code)
            self.hbc_col = np.zeros((R, G), np.uint64)
            self.beat = np.zeros((G, 2), np.uint64)
            self.beat[:, 1] = np.uint64(capi.NO_ACK)
            self.ae = np.full((R, G), capi.NO_ACK, np.uint64)
            self.owner_col = np.full((G + 3) // 4 * 4, 255, np.uint8)  # (whole words: k_cluster_claim writes four groups at a time)

        def dense_round(self, appends, dt_ms):
            self.now += dt_ms
            appends = np.broadcast_to(np.asarray(appends, dtype=np.uint64), (G,))
            offered = np.ascontiguousarray(capi.pack_answers(appends, np.full(G, capi.HB_NONE, np.uint8)))
            lib = self.nodes[0].lib
            hs = (C.c_void_p * R)(*[n._h for n in self.nodes])
            rc = lib.hc_cluster_any_round(hs, R, int(self.now), self.words.ctypes.data, self.hbc_col.ctypes.data, self.beat.ctypes.data,
                                          self.ae.ctypes.data, self.owner_col.ctypes.data, offered.ctypes.data,
                                          self.nodes[0]._fast_leader(), self.nodes[0]._fast_follower())
            assert rc == 0, f"host-compiled any-leader round: error {rc}"
            self.owner = self.owner_col[:G].copy()
            drained = []
            for n in self.nodes:  # (a node's two halves are two steps: the leader half's rows, then the follower half's)
                mark = int(lib.hc_msgs_mark(n._h))
                rows = n.drain_messages()
                drained.append([rows[:mark], rows[mark:]])
            self.rows.append(drained)
            return {}
    return HostAnyLeaderCluster()
