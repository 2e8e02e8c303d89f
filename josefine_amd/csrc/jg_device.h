// jg_device.h — device-side state layout and the per-group Raft state machine
// for gfx950.  Independent implementation (SoA columns, bitmasks, implicit chain
// run) of the reference functions cited inline; the CPU oracle under oracle/ is
// never included or linked here.  All citations are relative to /root/reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/josefine_gpu.h"

// ---- per-group flag word ------------------------------------------------------
//  bits 0-1   role (JG_ROLE_*)                                   mod.rs:417-425
//  bit  2     voted_for.is_some()                                mod.rs:279
//  bit  3     Follower.leader_id.is_some()                       follower.rs:20
//  bit  4     FAST chain: RUN (bit 7) and id_gen == head+1 -> the id_gen column is
//             implicit too (possibly stale in memory); what leaders built by append have
//  bit  7     RUN chain: run_hi == head and no extra segments -> the id set is exactly
//             [0, head] with next(id) = id-1; the run_hi column is implicit.  What
//             followers have (extend does not advance id_gen: chain.rs:178-192, Q8)
//  bit  5     the "commit" key has been persisted               chain.rs:198
//  bit  6     the run is EMPTY: Chain::compact removed block 0 (chain.rs:246); the id set is the
//             window segments only and run_hi means nothing
//  bits 8-15  bit r: progress of slot r is Replicate (else Probe) progress.rs:62-66
//  bits 16-23 sticky fault code (JG_FAULT_*)
//  bits 24-26 own replica slot
//  bits 28-31 number of chain segments besides the run (0..JG_CHAIN_WINDOW)
#define JGF_ROLE_MASK 0x3u
#define JGF_VOTED (1u << 2)
#define JGF_HAS_LEADER (1u << 3)
#define JGF_FAST (1u << 4)
#define JGF_COMMIT_KEY (1u << 5)
// (experiment: -DJG_GSM_WAVES=4 holds the general state machine's kernels to 128 VGPRs - the rest goes to scratch)
#ifdef JG_GSM_WAVES
#define JG_GSM_OCC __attribute__((amdgpu_waves_per_eu(JG_GSM_WAVES)))
#else
#define JG_GSM_OCC
#endif
#define JGF_NO_GENESIS (1u << 6)
#define JGF_RUN (1u << 7)
#define JGF_REPL_SHIFT 8
#define JGF_REPL_MASK (0xffu << JGF_REPL_SHIFT)
#define JGF_FAULT_SHIFT 16
#define JGF_FAULT_MASK (0xffu << JGF_FAULT_SHIFT)
#define JGF_SELF_SHIFT 24
#define JGF_SELF_MASK (0x7u << JGF_SELF_SHIFT)
#define JGF_WIN_SHIFT 28
#define JGF_WIN_MASK (0xfu << JGF_WIN_SHIFT)

// A message row outside the dense mailbox vocabulary, emitted by a dense node step: queued
// with its step sequence number and emission index, ordered at drain time.
struct JgXqRec {
  jg_msg_row row;
  uint32_t seq;
  uint32_t k;
};

struct JgFaultRec {
  uint32_t group;
  uint32_t code;
  uint32_t seq;  // step sequence number (orders rows of different steps at drain)
  uint32_t pad;
};

// Structure-of-arrays state in HBM; column c of group g is c[g], replica-major
// for the [R][G] / [W][G] arrays so that a wave reads contiguous lanes.
// What the elections and the timers of a group touch, in TWO 16-byte records (two arrays): the general state machine
// visits a group here and a group there, and every column it reads is a memory transaction of its own (a 64-byte
// sector per 4-byte value: the routed round's state machine launch fetched 1.5 KB per group visit, profiles/README.md
// round 3) - seven of them are two sectors this way.  The dense follower half reads each with one 16-byte load per
// lane, lanes side by side (as coalesced as the seven columns were), and a Heartbeat's timer store is one 16-byte
// store per lane into the timer array: whole sectors, nothing but the timer rewritten.
struct JgCold {
  // cold_t[g]: the election timer - what a Heartbeat rewrites (follower.rs:103-113), every other round in steady state
  uint64_t election_time;     // State.election_time (ms)               mod.rs:281
  uint32_t election_timeout;  // State.election_timeout (ms)            mod.rs:283
  uint32_t rng_draws;         // draws taken from the timeout RNG
  // cold_v[g]: what changes with leadership only
  uint32_t voted_for;         // State.voted_for                        mod.rs:279
  uint32_t leader_id;         // Follower.leader_id                     follower.rs:20
  uint32_t queued;            // queued_reqs.len()                      follower.rs:22
  uint32_t votes;             // Election.votes: seen | granted << 8    election.rs:8
                              //   bits 16-23 / 24-31: the same two masks for voters OUTSIDE the membership
};
struct JgColdCols {
  uint4* t;  // [G] {election_time lo, hi, election_timeout, rng_draws}
  uint4* v;  // [G] {voted_for, leader_id, queued, votes}
};
#define JG_COLD_T_ELECTION_TIME 0
#define JG_COLD_T_ELECTION_TIMEOUT 8
#define JG_COLD_V_VOTED_FOR 0
#define JG_COLD_V_LEADER_ID 4
#define JG_COLD_V_QUEUED 8
#define JG_COLD_V_VOTES 12
__device__ __forceinline__ JgCold jg_cold_load(const JgColdCols& p, uint32_t g) {
  const uint4 a = p.t[g], b = p.v[g];
  JgCold c;
  c.election_time = (uint64_t)a.x | (uint64_t)a.y << 32;
  c.election_timeout = a.z, c.rng_draws = a.w, c.voted_for = b.x, c.leader_id = b.y, c.queued = b.z, c.votes = b.w;
  return c;
}
__device__ __forceinline__ void jg_cold_store_timer(const JgColdCols& p, uint32_t g, const JgCold& c) {
  p.t[g] = make_uint4((uint32_t)c.election_time, (uint32_t)(c.election_time >> 32), c.election_timeout, c.rng_draws);
}
__device__ __forceinline__ void jg_cold_store_rest(const JgColdCols& p, uint32_t g, const JgCold& c) {
  p.v[g] = make_uint4(c.voted_for, c.leader_id, c.queued, c.votes);
}
__device__ __forceinline__ void jg_cold_store(const JgColdCols& p, uint32_t g, const JgCold& c) {
  jg_cold_store_timer(p, g, c);
  jg_cold_store_rest(p, g, c);
}
__device__ __forceinline__ JgCold jg_cold_of(uint64_t election_time, uint32_t voted_for, uint32_t leader_id, uint32_t election_timeout,
                                             uint32_t rng_draws, uint32_t queued, uint32_t votes) {
  JgCold c;
  c.election_time = election_time, c.election_timeout = election_timeout, c.rng_draws = rng_draws;
  c.voted_for = voted_for, c.leader_id = leader_id, c.queued = queued, c.votes = votes;
  return c;
}

struct JgDev {
  uint32_t G, R;
  uint32_t node_ids[JG_MAX_REPLICAS];
  uint32_t hb_timeout, el_min, el_max, cfg_flags;
  uint64_t seed, group_base;
  uint64_t* term;            // State.current_term                     mod.rs:277
  uint64_t* commit;          // Chain.commit (leaders: only where the packed lag escapes) chain.rs:102
  uint64_t* head;            // Chain.head                             chain.rs:103
  uint64_t* id_gen;          // Chain.id_gen (valid unless FAST)       chain.rs:101
  uint64_t* run_hi;          // segment 0: ids [0, run_hi], next = id-1 (valid unless FAST)
  uint64_t* mlag;            // [G] leaders: Progress.head of all R slots + Chain.commit, packed as lags below the chain head
  uint64_t* match_wide;      // [R][G] Progress.head of the slots whose lag field holds the escape value
  uint64_t* heartbeat_time;  // Leader.heartbeat_time (ms)             leader.rs:27
  uint64_t* win_lo;          // [W][G] chain segments besides the run: first id,
  uint64_t* win_hi;          // [W][G]   last id,
  uint64_t* win_next;        // [W][G]   parent pointer of the first id
  uint32_t* flags;
  JgColdCols cold;           // [G] x 2: election timer, timeout, RNG draws | vote, leader id, queue length, vote masks
  uint32_t* fvote_id;        // [JG_FOREIGN_VOTERS][G] their NodeIds (election.rs:33-35 counts whoever answers)
  uint64_t* blk_decisions;   // per-workgroup decision counters (no atomics on the hot path)
  JgFaultRec* fault_q;
  uint32_t* fault_q_n;
  uint32_t fault_q_cap;
  uint32_t* deferred_seen;   // the dense fast path met a leader whose chain is not in FAST form
  uint32_t* slow_list;       // [JG_SHARDS][ceil(G/JG_SHARDS)] deferred groups per shard
  uint32_t* slow_cnt;        // [JG_SHARDS]
  uint64_t* defer_bits;      // [ceil(G/64)] groups a dense leader kernel handed to k_dense_slow (bit g & 63 of word g >> 6)
  uint64_t* fdefer_bits;     // [2][ceil(G/64)] ... the dense follower half handed to k_follower_slow: [0] inputs and Tick, [1] the Tick only
  uint32_t* irregular_seen;  // set when a leader is stored with a chain that is not in FAST form
  uint32_t* cold_seen;       // set when the ack-only dense kernel took its in-kernel general path
  JgXqRec* xq;               // exceptional message rows of dense node steps (lazily allocated)
  uint32_t* xq_n;
  uint32_t xq_cap;
  uint32_t slow_cap;         // entries per shard of slow_list
  uint32_t* err;             // device error word (status block)
};

// splitmix64 finaliser: the counter-based RNG of DESIGN.md "Logical time and randomness"
__device__ __forceinline__ uint64_t jg_mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// ---- progress heads, delta-packed (progress.rs:124 `Progress.head`, one per replica) -----------
// A follower's acknowledged head trails the leader's chain head by a few blocks (at most
// MAX_INFLIGHT are in flight, progress.rs:117), and the leader's commit index trails it by the
// round trip, so a leader's R progress heads and its commit index are stored as R + 1 lags
// `head - value` of B = 64 / (R + 1) bits in ONE 64-bit word (R = 5: 10 bits, R = 3: 16):
// field r < R is slot r's progress head, field R the commit index.  The two largest field values
// are escapes (below): the absolute value then lives in match_wide[r][g] / commit[g] (a replica
// that is far behind, a forged ack above the head).  In steady state the lags do not change from tick to
// tick, so the dense kernel reads 8 bytes of progress + commit state per group and writes none —
// instead of reading and writing (R + 1) x 8.  Non-leaders keep the absolute commit column.
__host__ __device__ __forceinline__ uint32_t jg_lag_bits(uint32_t R) { return 64u / (R + 1u); }
__host__ __device__ __forceinline__ uint64_t jg_lag_esc(uint32_t R) { return (1ull << jg_lag_bits(R)) - 1ull; }
__host__ __device__ __forceinline__ uint64_t jg_lag_field(uint64_t w, uint32_t r, uint32_t R) {
  return (w >> (r * jg_lag_bits(R))) & jg_lag_esc(R);
}
__host__ __device__ __forceinline__ uint64_t jg_lag_with(uint64_t w, uint32_t r, uint32_t R, uint64_t field) {
  const uint32_t sh = r * jg_lag_bits(R);
  return (w & ~(jg_lag_esc(R) << sh)) | (field << sh);
}
// Two escape codes, both meaning "the absolute value is in the wide column": the all-ones field
// for a value ABOVE the base (a forged ack above the head), all-ones minus one for a value too far
// BEHIND it (a replica that is down: its lag grows by one block per append).  The dense kernels
// keep serving a group with BEHIND fields in lag space (such a slot sorts after every in-range
// lag; nothing about it changes until an ack arrives for it); an ABOVE field takes the general path.
__host__ __device__ __forceinline__ uint64_t jg_lag_behind(uint32_t R) { return jg_lag_esc(R) - 1ull; }
__host__ __device__ __forceinline__ bool jg_lag_wide(uint64_t field, uint32_t R) { return field >= jg_lag_behind(R); }
// lag field for absolute value v under chain head `base`; an escape code when it does not fit
__host__ __device__ __forceinline__ uint64_t jg_lag_encode(uint64_t v, uint64_t base, uint32_t R) {
  if (v > base) return jg_lag_esc(R);
  return base - v < jg_lag_behind(R) ? base - v : jg_lag_behind(R);
}

// Registers of one group while a lane walks its command segment.
struct JgLane {
  uint32_t g;
  uint64_t term, commit, head, id_gen, run_hi, election_time, heartbeat_time;
  uint64_t mword, mbase;  // leaders: packed progress heads (d.mlag) and the head they are relative to
  uint32_t flags, voted_for, leader_id, election_timeout, rng_draws, queued, votes;
  uint64_t now;
  uint32_t seq;
  uint32_t decisions;
  jg_msg_row* mp;
  jg_msg_row* mend;
  jg_fsm_row* fp;
  jg_fsm_row* fend;
  uint32_t overflow;  // an output row did not fit its bound (engine bug guard)
  uint32_t xq_on;     // 1: message rows go to the exceptional queue d.xq instead of [mp, mend);
                      // 2: the same, but AppendResponse / HeartbeatResponse are captured below;
                      // 3: the same, but a leader Tick's Heartbeat is captured (cap_hbc) and its AppendEntries
                      //    go straight into the outbox block cap_ae (k_dense_slow: no row buffer in scratch)
                      // 4: the same as 1, but the VoteResponses to one requester fold into cap_ack / cap_hbc (jg_votes.h)
  uint32_t xq_k;      // emission index within this step
  uint64_t cap_ack, cap_hbc;  // follower half of the dense node tick: outbox row of this group
  uint32_t cap_has;
  uint64_t* cap_ae;           // leader half (xq_on == 3): the [R][G] AppendEntries block of the outbox
};

__device__ __forceinline__ uint32_t jg_role(const JgLane& L) { return L.flags & JGF_ROLE_MASK; }
__device__ __forceinline__ void jg_set_role(JgLane& L, uint32_t r) { L.flags = (L.flags & ~JGF_ROLE_MASK) | r; }
__device__ __forceinline__ uint32_t jg_fault(const JgLane& L) { return (L.flags & JGF_FAULT_MASK) >> JGF_FAULT_SHIFT; }
__device__ __forceinline__ uint32_t jg_self(const JgLane& L) { return (L.flags & JGF_SELF_MASK) >> JGF_SELF_SHIFT; }
__device__ __forceinline__ uint32_t jg_wcnt(const JgLane& L) { return (L.flags & JGF_WIN_MASK) >> JGF_WIN_SHIFT; }
__device__ __forceinline__ uint32_t jg_self_id(const JgDev& d, const JgLane& L) { return d.node_ids[jg_self(L)]; }

__device__ inline void jg_push_fault(const JgDev& d, uint32_t g, uint32_t code, uint32_t seq) {
  uint32_t i = atomicAdd(d.fault_q_n, 1u);
  if (i < d.fault_q_cap) {
    JgFaultRec r;
    r.group = g;
    r.code = code;
    r.seq = seq;
    r.pad = 0;
    d.fault_q[i] = r;
  }
}
__device__ inline void jg_raise(const JgDev& d, JgLane& L, uint32_t code) {
  if (jg_fault(L)) return;
  L.flags |= code << JGF_FAULT_SHIFT;
  jg_push_fault(d, L.g, code, L.seq);
}

__device__ inline void jg_chain_normalize(const JgDev& d, JgLane& L);

// What the lags of a leader's packed word are relative to: the TOP of its chain where that is known from the flag
// word - the chain is the run [0, run_hi] (no window segments, genesis present) - else the head.  The two agree
// for a chain in RUN form; a restarted leader (jg_restart_groups / the durable-store restart: head = the commit
// index, the run above it still there) keeps its lags below run_hi, so every valid ack (chain.rs:197-202: a
// block the chain holds) is in range and the node tick serves the group in lag space (jg_dense_group).
// (for readers of the columns: run_hi[g] is not kept up to date while the chain is in RUN form, where it equals the head)
__host__ __device__ __forceinline__ bool jg_lag_base_is_run_hi(uint32_t flags) {
  return (flags & (JGF_WIN_MASK | JGF_NO_GENESIS | JGF_RUN)) == 0;
}
__device__ __forceinline__ uint64_t jg_lane_base(const JgLane& L) {  // (L.run_hi is the head in RUN form: jg_load)
  return (L.flags & (JGF_WIN_MASK | JGF_NO_GENESIS)) == 0 ? L.run_hi : L.head;
}
// Progress.head of slot r of the lane's group (leaders)
__device__ inline uint64_t jg_match_get(const JgDev& d, const JgLane& L, uint32_t r) {
  const uint64_t f = jg_lag_field(L.mword, r, d.R);
  return jg_lag_wide(f, d.R) ? d.match_wide[(size_t)r * d.G + L.g] : L.mbase - f;
}
__device__ inline void jg_match_set(const JgDev& d, JgLane& L, uint32_t r, uint64_t v) {
  const uint64_t f = jg_lag_encode(v, L.mbase, d.R);
  if (jg_lag_wide(f, d.R)) d.match_wide[(size_t)r * d.G + L.g] = v;
  L.mword = jg_lag_with(L.mword, r, d.R, f);
}
__device__ inline void jg_match_rebase(const JgDev& d, JgLane& L, uint64_t base) {
  uint64_t w = 0;
  for (uint32_t r = 0; r < d.R; r++) {
    const uint64_t v = jg_match_get(d, L, r);
    const uint64_t f = jg_lag_encode(v, base, d.R);
    if (jg_lag_wide(f, d.R)) d.match_wide[(size_t)r * d.G + L.g] = v;
    w = jg_lag_with(w, r, d.R, f);
  }
  L.mword = w;
  L.mbase = base;
}

__device__ inline void jg_load(const JgDev& d, JgLane& L, uint32_t g) {
  L.g = g;
  L.flags = d.flags[g];
  L.term = d.term[g];
  L.head = d.head[g];
  L.id_gen = (L.flags & JGF_FAST) ? L.head + 1 : d.id_gen[g];
  L.run_hi = (L.flags & JGF_RUN) ? L.head : d.run_hi[g];
  L.heartbeat_time = d.heartbeat_time[g];
  {
    const JgCold c = jg_cold_load(d.cold, g);
    L.election_time = c.election_time, L.voted_for = c.voted_for, L.leader_id = c.leader_id;
    L.election_timeout = c.election_timeout, L.rng_draws = c.rng_draws, L.queued = c.queued, L.votes = c.votes;
  }
  L.mword = 0;
  L.mbase = jg_lane_base(L);
  if ((L.flags & JGF_ROLE_MASK) == JG_ROLE_LEADER) {
    L.mword = d.mlag[g];
    const uint64_t fc = jg_lag_field(L.mword, d.R, d.R);
    L.commit = jg_lag_wide(fc, d.R) ? d.commit[g] : L.mbase - fc;
  } else {
    L.commit = d.commit[g];
  }
  L.decisions = 0;
  L.overflow = 0;
  L.xq_on = 0;
  L.xq_k = 0;
}
// CHAIN = false: the commands applied since jg_load cannot have touched the chain (jg_apply<JG_KINDS_ELECTION>): a
// chain that was normalised when it was stored is still normalised, the pass over its segments is not compiled in
template <bool CHAIN = true>
__device__ inline void jg_store(const JgDev& d, JgLane& L) {
  uint32_t g = L.g;
  if (CHAIN && jg_wcnt(L)) jg_chain_normalize(d, L);
  const bool run = (L.run_hi == L.head) && (jg_wcnt(L) == 0) && !(L.flags & JGF_NO_GENESIS);
  const bool fast = run && (L.id_gen == L.head + 1);
  L.flags = run ? (L.flags | JGF_RUN) : (L.flags & ~JGF_RUN);
  L.flags = fast ? (L.flags | JGF_FAST) : (L.flags & ~JGF_FAST);
  // the host then schedules the slow kernel behind the dense leader kernel
  if (!fast && jg_role(L) == JG_ROLE_LEADER && !jg_fault(L)) *d.irregular_seen = 1;
  if (jg_role(L) == JG_ROLE_LEADER) {
    const uint64_t base = jg_lane_base(L);
    if (L.mbase != base) jg_match_rebase(d, L, base);  // the chain grew: the lags are relative to its top
    const uint64_t fc = jg_lag_encode(L.commit, base, d.R);
    if (jg_lag_wide(fc, d.R)) d.commit[g] = L.commit;
    d.mlag[g] = jg_lag_with(L.mword, d.R, d.R, fc);
  } else {
    d.commit[g] = L.commit;
  }
  d.flags[g] = L.flags;
  d.term[g] = L.term;
  d.head[g] = L.head;
  d.id_gen[g] = L.id_gen;
  d.run_hi[g] = L.run_hi;
  d.heartbeat_time[g] = L.heartbeat_time;
  jg_cold_store(d.cold, g, jg_cold_of(L.election_time, L.voted_for, L.leader_id, L.election_timeout, L.rng_draws, L.queued, L.votes));
}

__device__ inline int jg_slot_of(const JgDev& d, uint32_t node_id);

// jg_store for a lane whose state as loaded is still at hand (`O`: a copy of the lane right after jg_load): only the
// columns whose value changed are written.  The general kernels touch a group here and a group there: every
// 4- or 8-byte store is a memory transaction of its own, and a step that flips a vote wrote all fourteen columns
// (the routed round's state machine launches and the slow kernels are bound by exactly these transactions).
// A role change switches the commit index's representation (packed into mlag for leaders): everything is written then.
template <bool CHAIN = true>
__device__ inline void jg_store_dirty(const JgDev& d, JgLane& L, const JgLane& O) {
  if (jg_role(L) != jg_role(O)) {
    jg_store<CHAIN>(d, L);
    return;
  }
  const uint32_t g = L.g;
  if (CHAIN && jg_wcnt(L)) jg_chain_normalize(d, L);
  const bool run = (L.run_hi == L.head) && (jg_wcnt(L) == 0) && !(L.flags & JGF_NO_GENESIS);
  const bool fast = run && (L.id_gen == L.head + 1);
  L.flags = run ? (L.flags | JGF_RUN) : (L.flags & ~JGF_RUN);
  L.flags = fast ? (L.flags | JGF_FAST) : (L.flags & ~JGF_FAST);
  if (!fast && jg_role(L) == JG_ROLE_LEADER && !jg_fault(L)) *d.irregular_seen = 1;
  if (jg_role(L) == JG_ROLE_LEADER) {
    const uint64_t base = jg_lane_base(L);
    if (L.mbase != base) jg_match_rebase(d, L, base);
    const uint64_t fc = jg_lag_encode(L.commit, base, d.R);
    if (jg_lag_wide(fc, d.R)) d.commit[g] = L.commit;
    const uint64_t w = jg_lag_with(L.mword, d.R, d.R, fc);
    if (w != O.mword) d.mlag[g] = w;  // (O.mword: the packed word as loaded, commit field included)
  } else if (L.commit != O.commit) {
    d.commit[g] = L.commit;
  }
  if (L.flags != O.flags) d.flags[g] = L.flags;
  if (L.term != O.term) d.term[g] = L.term;
  if (L.head != O.head) d.head[g] = L.head;
  // (these two are implicit while the chain is in FAST / RUN form - jg_load and jg_read_state do not look at the
  // columns then - so they are written when the lane LEAVES that form or changes them outside it)
  if (!fast && (L.id_gen != O.id_gen || (O.flags & JGF_FAST))) d.id_gen[g] = L.id_gen;
  if (!run && (L.run_hi != O.run_hi || (O.flags & JGF_RUN))) d.run_hi[g] = L.run_hi;
  if (L.heartbeat_time != O.heartbeat_time) d.heartbeat_time[g] = L.heartbeat_time;
  if (L.election_time != O.election_time || L.voted_for != O.voted_for || L.leader_id != O.leader_id ||
      L.election_timeout != O.election_timeout || L.rng_draws != O.rng_draws || L.queued != O.queued || L.votes != O.votes)
    jg_cold_store(d.cold, g, jg_cold_of(L.election_time, L.voted_for, L.leader_id, L.election_timeout, L.rng_draws, L.queued, L.votes));
}

// ---- output rows ------------------------------------------------------------------
__device__ inline void jg_emit_msg(const JgDev& d, JgLane& L, uint8_t kind, uint8_t to_kind, uint32_t to_id,
                                   uint8_t flag, uint64_t term, uint64_t id, uint64_t aux) {
  if (!L.xq_on && L.mp >= L.mend) {
    L.overflow = 1;
    return;
  }
  if (L.xq_on == 2) {  // dense mailbox vocabulary of the follower half
    if (kind == JG_CMD_APPEND_RESPONSE) {
      L.cap_ack = id;
      return;
    }
    if (kind == JG_CMD_HEARTBEAT_RESPONSE) {
      L.cap_hbc = id;
      L.cap_has = flag;
      return;
    }
  }
  if (L.xq_on == 3) {  // dense mailbox vocabulary of the leader half's Tick (leader.rs:234-245)
    if (kind == JG_CMD_HEARTBEAT) {
      L.cap_hbc = id;  // Heartbeat.commit
      return;
    }
    if (kind == JG_CMD_APPEND_ENTRIES) {  // to one peer: (range start key, number of blocks)
      const int r = jg_slot_of(d, to_id);
      if (r >= 0) L.cap_ae[(size_t)r * d.G + L.g] = JG_AE(id, aux);
      return;
    }
  }
  if (L.xq_on == 4 && kind == JG_CMD_VOTE_RESPONSE && to_kind == JG_TO_PEER && id == 0 && aux == 0 && L.xq_k < 256u) {
    // the vote mail's receiving half (jg_votes.h): the VoteResponses this node gives to ONE requester, back to back in its
    // emission order, fold into its answer word - cap_ack: their term, cap_hbc: n | index of the first << 8 | first
    // answer << 19 | every further answer << 20 | addressee slot << 21 (jg_vote_actl with the step left out)
    const int to = jg_slot_of(d, to_id);
    const uint32_t n = (uint32_t)L.cap_hbc & 0xffu;
    if (to >= 0 && !n) {
      L.cap_ack = term;
      L.cap_hbc = 1u | L.xq_k << 8 | (uint32_t)(flag & 1u) << 19 | (uint32_t)to << 21;
      L.xq_k++;
      return;
    }
    if (to >= 0 && (uint32_t)to == (((uint32_t)L.cap_hbc >> 21) & 7u) && term == L.cap_ack && L.xq_k == (((uint32_t)L.cap_hbc >> 8) & 0xffu) + n && n < 255u &&
        (n == 1u || (uint32_t)(flag & 1u) == (((uint32_t)L.cap_hbc >> 20) & 1u))) {
      L.cap_hbc = (((uint32_t)L.cap_hbc & ~(1u << 20)) | (uint32_t)(flag & 1u) << 20) + 1u;
      L.xq_k++;
      return;
    }
  }
  jg_msg_row r;
  r.group = L.g;
  r.kind = kind;
  r.to_kind = to_kind;
  r.flag = flag;
  r.pad = 0;
  r.to_id = to_id;
  r.from = jg_self_id(d, L);  // Message::new(Address::Peer(self.id), ..)  mod.rs:390-400
  r.term = term;
  r.id = id;
  r.aux = aux;
  if (L.xq_on) {
    const uint32_t i = atomicAdd(d.xq_n, 1u);
    if (i < d.xq_cap) {
      JgXqRec q;
      q.row = r;
      q.seq = L.seq;
      q.k = L.xq_k++;
      d.xq[i] = q;
    }
    return;
  }
  *L.mp++ = r;
}
__device__ inline void jg_emit_fsm(JgLane& L, uint8_t kind, uint64_t a, uint64_t b) {
  if (L.fp >= L.fend) {
    L.overflow = 1;
    return;
  }
  jg_fsm_row r;
  r.group = L.g;
  r.kind = kind;
  r.pad[0] = r.pad[1] = r.pad[2] = 0;
  r.a = a;
  r.b = b;
  *L.fp++ = r;
}

// ---- Chain (src/raft/chain.rs:99-254) ------------------------------------------------
// The id set is a union of disjoint *segments* [lo, hi]: every id in a segment
// exists, next(lo) = lo_next and next(id) = id-1 for lo < id <= hi.  Segment 0 is
// the implicit run [0, run_hi] (genesis, next(0) = 0); up to JG_CHAIN_WINDOW more
// live in the win_* columns.  A chain built by in-order appends / extends is one
// run; each gap or fork costs one segment.
#define JG_SEG(col, w) d.col[(size_t)(w) * d.G + L.g]
__device__ inline int jg_seg_find(const JgDev& d, const JgLane& L, uint64_t id) {
  uint32_t n = jg_wcnt(L);
  for (uint32_t w = 0; w < n; w++)
    if (JG_SEG(win_lo, w) <= id && id <= JG_SEG(win_hi, w)) return (int)w;
  return -1;
}
__device__ __forceinline__ bool jg_in_run(const JgLane& L, uint64_t id) {
  return !(L.flags & JGF_NO_GENESIS) && id <= L.run_hi;
}
__device__ inline bool jg_chain_has(const JgDev& d, const JgLane& L, uint64_t id) {  // chain.rs:155-157
  if (jg_in_run(L, id)) return true;
  return jg_seg_find(d, L, id) >= 0;
}
__device__ inline uint32_t jg_seg_add(const JgDev& d, JgLane& L, uint64_t lo, uint64_t hi, uint64_t lo_next) {
  uint32_t n = jg_wcnt(L);
  if (n >= JG_CHAIN_WINDOW) return JG_FAULT_ENGINE_WINDOW_OVERFLOW;
  JG_SEG(win_lo, n) = lo;
  JG_SEG(win_hi, n) = hi;
  JG_SEG(win_next, n) = lo_next;
  L.flags = (L.flags & ~JGF_WIN_MASK) | ((n + 1) << JGF_WIN_SHIFT);
  return 0;
}
// make `id` (which exists) the first id of its segment
__device__ inline uint32_t jg_seg_split_at(const JgDev& d, JgLane& L, uint64_t id) {
  if (jg_in_run(L, id)) {
    if (id == 0) return 0;
    uint64_t hi = L.run_hi;
    L.run_hi = id - 1;
    return jg_seg_add(d, L, id, hi, id - 1);
  }
  int w = jg_seg_find(d, L, id);
  if (JG_SEG(win_lo, w) == id) return 0;
  uint64_t hi = JG_SEG(win_hi, w);
  JG_SEG(win_hi, w) = id - 1;
  return jg_seg_add(d, L, id, hi, id - 1);
}
// sled insert (upsert) of Block{id,next}; returns an engine fault or 0
__device__ inline uint32_t jg_chain_insert(const JgDev& d, JgLane& L, uint64_t id, uint64_t next) {
  if (jg_chain_has(d, L, id)) {
    uint64_t cur;
    if (jg_in_run(L, id)) {
      cur = id ? id - 1 : 0;
    } else {
      int w = jg_seg_find(d, L, id);
      cur = JG_SEG(win_lo, w) == id ? JG_SEG(win_next, w) : id - 1;
    }
    if (cur == next) return 0;  // same block again (e.g. a re-sent AppendEntries)
    // overwrite of an existing block's parent pointer: isolate [id, id]
    if (id == 0) return JG_FAULT_ENGINE_WINDOW_OVERFLOW;  // genesis parent is implicit
    uint32_t f = jg_seg_split_at(d, L, id);
    if (f) return f;
    int w = jg_seg_find(d, L, id);
    if (JG_SEG(win_hi, w) > id) {
      f = jg_seg_split_at(d, L, id + 1);
      if (f) return f;
    }
    JG_SEG(win_next, w) = next;
    return 0;
  }
  if (L.flags & JGF_NO_GENESIS) {
    if (id == 0 && next == 0) {  // genesis again (Chain::new on a tree without a commit key, chain.rs:132-153)
      L.flags &= ~JGF_NO_GENESIS;
      L.run_hi = 0;
      return 0;
    }
  } else if (id == L.run_hi + 1 && next == L.run_hi) {
    L.run_hi = id;
    return 0;
  }
  uint32_t n = jg_wcnt(L);
  for (uint32_t w = 0; w < n; w++)
    if (JG_SEG(win_hi, w) + 1 == id && next == JG_SEG(win_hi, w)) {
      JG_SEG(win_hi, w) = id;
      return 0;
    }
  return jg_seg_add(d, L, id, id, next);
}
// Canonical form: a segment that continues the run (first id run_hi+1 with parent run_hi — e.g. a
// gap that was filled later) is absorbed into it, repeatedly.  Then "the id set is [0, head] with
// every parent id-1" holds exactly when run_hi == head and no segment is left, which is what the
// RUN / FAST flags and the dense mailbox vocabulary are defined on.
__device__ inline void jg_chain_normalize(const JgDev& d, JgLane& L) {
  uint32_t n = jg_wcnt(L);
  bool merged = n != 0 && !(L.flags & JGF_NO_GENESIS);
  while (merged) {
    merged = false;
    for (uint32_t w = 0; w < n; w++) {
      if (JG_SEG(win_lo, w) == L.run_hi + 1 && JG_SEG(win_next, w) == L.run_hi) {
        L.run_hi = JG_SEG(win_hi, w);
        n--;
        if (w != n) {
          JG_SEG(win_lo, w) = JG_SEG(win_lo, n);
          JG_SEG(win_hi, w) = JG_SEG(win_hi, n);
          JG_SEG(win_next, w) = JG_SEG(win_next, n);
        }
        merged = n != 0;
        break;
      }
    }
  }
  L.flags = (L.flags & ~JGF_WIN_MASK) | (n << JGF_WIN_SHIFT);
}
// number of block keys >= from, saturated at `cap` (unbounded range(from..), leader.rs:135,152-157)
__device__ inline uint32_t jg_chain_blocks_from(const JgDev& d, const JgLane& L, uint64_t from, uint32_t cap) {
  uint64_t k = 0;
  if (jg_in_run(L, from)) {
    k = L.run_hi - from + 1;
    if (k >= cap) return cap;
  }
  uint32_t n = jg_wcnt(L);
  for (uint32_t w = 0; w < n; w++) {
    uint64_t lo = JG_SEG(win_lo, w), hi = JG_SEG(win_hi, w);
    if (hi >= from) {
      uint64_t c = hi - (lo > from ? lo : from) + 1;
      k = (k + c < k) ? UINT64_MAX : k + c;
    }
  }
  return k >= cap ? cap : (uint32_t)k;
}
// Chain::append, chain.rs:160-175
__device__ inline uint32_t jg_chain_append(const JgDev& d, JgLane& L, uint64_t* out) {
  uint64_t id = L.id_gen++;                                      // :161
  if (!(id > L.head)) return JG_FAULT_APPEND_ID_NOT_ABOVE_HEAD;  // :163
  uint32_t f = jg_chain_insert(d, L, id, L.head);                // :164-172
  if (f) return f;
  L.head = id;                                                   // :173
  *out = id;
  return 0;
}
// Chain::extend, chain.rs:178-192
__device__ inline uint32_t jg_chain_extend(const JgDev& d, JgLane& L, uint64_t id, uint64_t next) {
  if (!jg_chain_has(d, L, next)) return JG_FAULT_EXTEND_MISSING_PARENT;  // :180-185
  uint32_t f = jg_chain_insert(d, L, id, next);                          // :187-189
  if (f) return f;
  L.head = id;                                                           // :190
  return 0;
}
// Chain::commit, chain.rs:195-205
__device__ inline uint32_t jg_chain_commit(const JgDev& d, JgLane& L, uint64_t id) {
  if (!jg_chain_has(d, L, id)) return JG_FAULT_COMMIT_MISSING_BLOCK;  // :200-202
  L.flags |= JGF_COMMIT_KEY;                                          // :198
  L.commit = id;                                                      // :199
  return 0;
}
// Chain::new on the persisted tree, chain.rs:117-137 (+ init 139-153)
__device__ inline uint32_t jg_chain_reopen(const JgDev& d, JgLane& L) {
  uint64_t c = (L.flags & JGF_COMMIT_KEY) ? L.commit : 0;
  L.id_gen = c;
  L.commit = c;
  L.head = c;
  if (c == 0) {
    L.id_gen = 1;                        // id_gen.next() -> 0
    return jg_chain_insert(d, L, 0, 0);  // genesis re-inserted
  }
  return 0;
}

// ---- timers ---------------------------------------------------------------------------
__device__ inline void jg_set_election_timeout(const JgDev& d, JgLane& L) {  // follower.rs:103-113
  uint32_t span = d.el_max - d.el_min;
  uint64_t key = d.group_base + L.g;
  uint64_t r = jg_mix64(d.seed ^ jg_mix64(key * 0xd1342543de82ef95ull + L.rng_draws));
  L.rng_draws++;
  L.election_timeout = d.el_min + (span ? (uint32_t)(r % span) : 0u);
  L.election_time = L.now;
}
__device__ inline bool jg_needs_election(const JgLane& L) {  // mod.rs:352-357
  return (L.now - L.election_time) > (uint64_t)L.election_timeout;
}

// ---- progress (src/raft/progress.rs) ----------------------------------------------------
__device__ inline int jg_slot_of(const JgDev& d, uint32_t node_id) {
  for (uint32_t r = 0; r < d.R; r++)
    if (d.node_ids[r] == node_id) return (int)r;
  return -1;
}
// ReplicationProgress::committed_index, progress.rs:48-60: element R/2 of the heads
// sorted descending == the head whose rank (number of strictly-greater heads, ties
// broken by slot) is R/2.
__device__ inline uint64_t jg_committed_index(const JgDev& d, const JgLane& L) {
  uint64_t v[JG_MAX_REPLICAS];
#pragma unroll
  for (uint32_t r = 0; r < JG_MAX_REPLICAS; r++)
    v[r] = r < d.R ? jg_match_get(d, L, r) : 0;
  uint32_t k = d.R / 2;
  uint64_t q = 0;
#pragma unroll
  for (uint32_t j = 0; j < JG_MAX_REPLICAS; j++) {
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t i = 0; i < JG_MAX_REPLICAS; i++)
      cnt += (i < d.R && (v[i] > v[j] || (v[i] == v[j] && i < j))) ? 1u : 0u;
    if (j < d.R && cnt == k) q = v[j];
  }
  return q;
}

// ---- role transitions -----------------------------------------------------------------
__device__ inline void jg_drop_queue(const JgDev& d, JgLane& L) {
  if (L.queued) jg_emit_msg(d, L, JG_CMD_CLIENT_REQUEST, JG_TO_QUEUE, 0, JG_QUEUE_DROP, 0, 0, L.queued);
  L.queued = 0;
}
__device__ inline void jg_clear_leader(JgLane& L) {
  L.flags &= ~JGF_HAS_LEADER;
  L.leader_id = 0;
}
// Raft::term (mod.rs:360-365) + Role::term; returns a fault for the Leader role
__device__ inline uint32_t jg_set_term(JgLane& L, uint64_t t) {
  L.flags &= ~JGF_VOTED;
  L.voted_for = 0;
  L.term = t;
  switch (jg_role(L)) {
    case JG_ROLE_FOLLOWER: jg_clear_leader(L); return 0;              // follower.rs:27-29
    case JG_ROLE_CANDIDATE: L.votes = 0; return 0;                    // candidate.rs:161-163
    default: return JG_FAULT_LEADER_TERM_UNIMPLEMENTED;               // leader.rs:33-35
  }
}
__device__ inline void jg_vote_for(JgLane& L, uint32_t id) {
  L.flags |= JGF_VOTED;
  L.voted_for = id;
}
__device__ inline void jg_become_candidate(const JgDev& d, JgLane& L) {  // follower.rs:285-304
  L.votes = 0;
  jg_set_role(L, JG_ROLE_CANDIDATE);
  jg_drop_queue(d, L);  // Candidate{queued_reqs: Vec::new()} (follower.rs:295)
  jg_clear_leader(L);
}
__device__ inline void jg_follower_from_candidate(JgLane& L) {  // candidate.rs:198-214
  jg_set_role(L, JG_ROLE_FOLLOWER);
  jg_clear_leader(L);
  L.votes = 0;
}
__device__ inline void jg_follower_from_leader(JgLane& L) {  // leader.rs:268-284
  jg_set_role(L, JG_ROLE_FOLLOWER);
  jg_clear_leader(L);
  L.queued = 0;
  L.flags &= ~JGF_REPL_MASK;
}
__device__ inline void jg_become_leader(const JgDev& d, JgLane& L) {  // candidate.rs:216-238
  L.mword = 0;
  L.mbase = jg_lane_base(L);
  for (uint32_t r = 0; r < d.R; r++) jg_match_set(d, L, r, 0);            // progress.rs:155-162
  L.flags &= ~JGF_REPL_MASK;
  L.heartbeat_time = L.now;
  jg_set_role(L, JG_ROLE_LEADER);
  jg_drop_queue(d, L);
  L.votes = 0;
}
__device__ inline void jg_send_heartbeat(const JgDev& d, JgLane& L) {  // leader.rs:44-51
  jg_emit_msg(d, L, JG_CMD_HEARTBEAT, JG_TO_PEERS, 0, 0, L.term, L.commit, 0);
}

// ---- Leader (src/raft/leader.rs) -------------------------------------------------------
__device__ inline uint32_t jg_leader_commit(const JgDev& d, JgLane& L) {  // leader.rs:87-99
  L.decisions++;
  uint64_t q = jg_committed_index(d, L);
  if (q > L.commit) {
    uint64_t prev = L.commit;
    uint32_t f = jg_chain_commit(d, L, q);
    if (f) return f;
    jg_emit_fsm(L, JG_FSM_APPLY_LEADER, prev, q);  // :93
  }
  return 0;
}
__device__ inline uint32_t jg_leader_append_response(const JgDev& d, JgLane& L, uint32_t node, uint64_t head) {
  // leader.rs:211-219 -> progress.rs:42-46,76-94,133-140
  int s = jg_slot_of(d, node);
  if (s < 0) return JG_FAULT_PROGRESS_UNKNOWN_NODE;  // progress.rs:43
  const bool inc = jg_match_get(d, L, (uint32_t)s) < head;
  if (inc) jg_match_set(d, L, (uint32_t)s, head);
  uint32_t bit = 1u << (JGF_REPL_SHIFT + s);
  L.flags = inc ? (L.flags | bit) : (L.flags & ~bit);
  return jg_leader_commit(d, L);
}
__device__ inline uint32_t jg_leader_client_request(const JgDev& d, JgLane& L, uint64_t token) {  // leader.rs:177-197
  uint64_t bid = 0;
  uint32_t f = jg_chain_append(d, L, &bid);
  if (f) return f;
  jg_emit_fsm(L, JG_FSM_NOTIFY, bid, token);                               // :184-188
  return jg_leader_append_response(d, L, jg_self_id(d, L), L.head);        // :190-196
}
__device__ inline uint32_t jg_leader_replicate(const JgDev& d, JgLane& L) {  // leader.rs:124-174
  uint32_t self = jg_self(L);
  bool key_in_range = (L.flags & JGF_COMMIT_KEY) && !(d.cfg_flags & JG_CFG_SEPARATE_COMMIT_KEY);
  for (uint32_t r = 0; r < d.R; r++) {  // config.nodes: the other slots in ascending order
    if (r == self) continue;
    bool repl = (L.flags >> (JGF_REPL_SHIFT + r)) & 1u;
    uint32_t want = repl ? JG_MAX_INFLIGHT + 1 : 2;  // items consumed by skip(1).take(5) / nth(1)
    uint64_t from = jg_match_get(d, L, r);
    uint32_t k = jg_chain_blocks_from(d, L, from, want);
    if (k < want && key_in_range) return JG_FAULT_RANGE_HIT_COMMIT_KEY;  // chain.rs:219-226 (Q9)
    uint32_t n_blocks = k ? k - 1 : 0;
    jg_emit_msg(d, L, JG_CMD_APPEND_ENTRIES, JG_TO_PEER, d.node_ids[r], 0, L.term, from, n_blocks);
  }
  return 0;
}
__device__ inline uint32_t jg_leader_tick(const JgDev& d, JgLane& L) {  // leader.rs:234-245
  if ((L.now - L.heartbeat_time) > (uint64_t)d.hb_timeout) {            // :78-80
    jg_send_heartbeat(d, L);
    L.heartbeat_time = L.now;                                           // :82-84
  }
  return jg_leader_replicate(d, L);
}

// ---- Candidate (src/raft/candidate.rs) + Election (src/raft/election.rs) -----------------
// 0 = Voting, 1 = Elected, 2 = Defeated (election.rs:37-57, quorum_size 66-73).  The fold runs over
// every entry of the votes map: members (bits 0-7 / 8-15) and foreign voters (bits 16-23 / 24-31).
__device__ inline uint32_t jg_election_status(const JgDev& d, const JgLane& L) {
  uint32_t seen = (L.votes & 0xffu) | ((L.votes >> 8) & 0xff00u), granted = ((L.votes >> 8) & 0xffu) | ((L.votes >> 16) & 0xff00u);
  uint32_t yes = __popc(granted & seen), total = __popc(seen);
  uint32_t quorum = d.R == 1 ? 0u : d.R / 2 + 1;
  if (yes >= quorum) return 1;
  if (total - yes == quorum) return 2;
  return 0;
}
__device__ inline uint32_t jg_candidate_vote_response(const JgDev& d, JgLane& L, bool granted, uint32_t from) {
  // candidate.rs:91-98
  int s = jg_slot_of(d, from);
  if (s < 0) {
    // votes.insert(id, vote) does not ask who `id` is (election.rs:33-35): a voter outside the
    // membership takes (or re-uses: a later vote overwrites) one of JG_FOREIGN_VOTERS table entries
    const uint32_t fseen = (L.votes >> 16) & 0xffu;
    int k = -1;
    for (uint32_t i = 0; i < JG_FOREIGN_VOTERS; i++)
      if (((fseen >> i) & 1u) && d.fvote_id[(size_t)i * d.G + L.g] == from) k = (int)i;
    if (k < 0) {
      const uint32_t free_mask = ~fseen & ((1u << JG_FOREIGN_VOTERS) - 1u);
      if (!free_mask) return JG_FAULT_ENGINE_FOREIGN_VOTER;  // a 9th distinct stranger: the table is full
      k = __ffs(free_mask) - 1;
      d.fvote_id[(size_t)k * d.G + L.g] = from;
    }
    s = 16 + k;
  }
  const uint32_t bit = 1u << s;
  L.votes |= bit;                                                         // election.rs:33-35
  L.votes = granted ? (L.votes | (bit << 8)) : (L.votes & ~(bit << 8));
  L.decisions++;
  switch (jg_election_status(d, L)) {
    case 1:  // elect(): candidate.rs:108-113
      jg_become_leader(d, L);
      jg_send_heartbeat(d, L);
      return 0;
    case 0: return 0;
    default:  // defeat(): candidate.rs:101-105
      L.flags &= ~JGF_VOTED;
      L.voted_for = 0;
      jg_follower_from_candidate(L);
      return 0;
  }
}
__device__ inline uint32_t jg_seek_election(const JgDev& d, JgLane& L) {  // candidate.rs:24-45
  jg_vote_for(L, jg_self_id(d, L));                                       // :25
  L.term += 1;                                                            // :26
  for (uint32_t i = 0; i + 1 < d.R; i++)                                  // :30-37, one broadcast per peer
    jg_emit_msg(d, L, JG_CMD_VOTE_REQUEST, JG_TO_PEERS, 0, 0, L.term, L.head, L.term);
  return jg_candidate_vote_response(d, L, true, jg_self_id(d, L));        // :40-44
}
__device__ inline uint32_t jg_follower_timeout(const JgDev& d, JgLane& L) {  // follower.rs:248-256
  if (!(L.flags & JGF_VOTED)) {
    jg_set_election_timeout(d, L);
    jg_become_candidate(d, L);
    return jg_seek_election(d, L);
  }
  return 0;
}
__device__ inline uint32_t jg_candidate_tick(const JgDev& d, JgLane& L) {  // candidate.rs:48-68
  if (jg_needs_election(L)) {
    if (jg_election_status(d, L) == 1) return JG_FAULT_CANDIDATE_TICK_ELECTED;  // :64
    L.flags &= ~JGF_VOTED;  // :53 / :59
    L.voted_for = 0;
    jg_follower_from_candidate(L);
    return jg_follower_timeout(d, L);
  }
  return 0;
}

// ---- Follower (src/raft/follower.rs) -------------------------------------------------
struct JgCmd {
  uint32_t kind, from, flag;
  uint64_t term, id, aux;
};

__device__ inline uint32_t jg_follower_append_entries(const JgDev& d, JgLane& L, const JgCmd& c,
                                                      const uint64_t* blk_id, const uint64_t* blk_next) {
  // follower.rs:130-176
  if (!(L.flags & JGF_VOTED) && c.term >= L.term) {  // :137
    jg_set_term(L, c.term);                          // :138
    L.election_time = L.now;                         // :141
    L.flags |= JGF_HAS_LEADER;                       // :142
    L.leader_id = c.from;
    jg_vote_for(L, c.from);                          // :143
  }
  if ((L.flags & JGF_VOTED) && L.voted_for != c.from && c.term < L.term)  // :147-154
    return JG_FAULT_FOLLOWER_STALE_LEADER;
  if (c.aux) {  // :157
    for (uint64_t k = 0; k < c.aux; k++) {
      // no side arrays (dense mailbox): the blocks are ids c.id+1 .. c.id+aux, next = id-1 each
      const uint64_t bid = blk_id ? blk_id[c.id + k] : c.id + 1 + k;
      const uint64_t bnext = blk_id ? blk_next[c.id + k] : c.id + k;
      uint32_t f = jg_chain_extend(d, L, bid, bnext);  // :159
      if (f) return f;
    }
    jg_emit_msg(d, L, JG_CMD_APPEND_RESPONSE, JG_TO_PEER, c.from, 1, L.term, L.head, 0);  // :163-172
  }
  return 0;
}
__device__ inline uint32_t jg_follower_heartbeat(const JgDev& d, JgLane& L, uint32_t leader, uint64_t term,
                                                 uint64_t commit) {
  // follower.rs:178-217
  jg_set_election_timeout(d, L);  // :184
  jg_set_term(L, term);           // :185 (unconditional)
  L.flags |= JGF_HAS_LEADER;      // :186
  L.leader_id = leader;
  jg_vote_for(L, leader);         // :187
  if (L.queued) {                 // :190-197
    jg_emit_msg(d, L, JG_CMD_CLIENT_REQUEST, JG_TO_PEER, leader, JG_QUEUE_FLUSH, 0, 0, L.queued);
    L.queued = 0;
  }
  bool has_committed = jg_chain_has(d, L, commit);  // :200
  if (has_committed && commit > L.commit) {         // :201
    uint64_t prev = L.commit;
    uint32_t f = jg_chain_commit(d, L, commit);     // :203
    if (f) return f;
    jg_emit_fsm(L, JG_FSM_APPLY_FOLLOWER, prev, commit);  // :204-206, half-open range(prev..commit)
  }
  jg_emit_msg(d, L, JG_CMD_HEARTBEAT_RESPONSE, JG_TO_PEER, leader, has_committed ? 1 : 0, 0, L.commit, 0);
  return 0;
}
__device__ inline uint32_t jg_follower_vote_request(const JgDev& d, JgLane& L, uint32_t cand, uint64_t last_term,
                                                    uint64_t head) {
  // follower.rs:219-246 with can_vote 97-101
  // (the vote mail's LEAN visit, jg_votes.h jg_vote_half_group, loads only what this function and jg_vote_for touch of a
  // healthy follower - flags, term, commit, the {voted_for, leader_id, queued, votes} record - and zero-fills the rest of
  // the lane: anything else read here must be loaded there too; tests/test_vote_half.py runs both against the oracle)
  bool can = !((L.flags & JGF_VOTED) || L.term > last_term || L.commit > head);
  jg_emit_msg(d, L, JG_CMD_VOTE_RESPONSE, JG_TO_PEER, cand, can ? 1 : 0, L.term, 0, 0);
  if (can) jg_vote_for(L, cand);  // :234
  return 0;
}
__device__ inline void jg_enqueue(const JgDev& d, JgLane& L, uint64_t token) {
  jg_emit_msg(d, L, JG_CMD_CLIENT_REQUEST, JG_TO_QUEUE, 0, 0, 0, token, 0);
  L.queued++;
}

// KINDS (here and below): the command kinds the caller's batch can hold, as a bit mask over JG_CMD_* - a compile-time
// promise (the caller has a census of its batch).  Code for a kind outside the mask is not compiled in: a launch that
// only carries an election's traffic (JG_KINDS_ELECTION: VoteRequest, VoteResponse, Timeout) has no chain code at
// all, half the registers and twice the waves per SIMD.  A row of a kind outside the mask is a no-op, never undefined.
#define JG_KINDS_ALL 0xffffffffu
#define JG_KINDS_ELECTION ((1u << JG_CMD_VOTE_REQUEST) | (1u << JG_CMD_VOTE_RESPONSE) | (1u << JG_CMD_TIMEOUT))
#define JG_KIND_IN(K) ((KINDS >> (K)) & 1u)
template <uint32_t KINDS = JG_KINDS_ALL>
__device__ inline uint32_t jg_follower_apply(const JgDev& d, JgLane& L, const JgCmd& c, const uint64_t* blk_id,
                                             const uint64_t* blk_next) {  // follower.rs:36-64
  switch (c.kind) {
    case JG_CMD_TICK: if (!JG_KIND_IN(JG_CMD_TICK)) return 0; return jg_needs_election(L) ? jg_follower_timeout(d, L) : 0;  // :121-128
    case JG_CMD_APPEND_ENTRIES: if (!JG_KIND_IN(JG_CMD_APPEND_ENTRIES)) return 0; return jg_follower_append_entries(d, L, c, blk_id, blk_next);
    case JG_CMD_HEARTBEAT: if (!JG_KIND_IN(JG_CMD_HEARTBEAT)) return 0; return jg_follower_heartbeat(d, L, c.from, c.term, c.id);
    case JG_CMD_VOTE_REQUEST: if (!JG_KIND_IN(JG_CMD_VOTE_REQUEST)) return 0; return jg_follower_vote_request(d, L, c.from, c.aux, c.id);
    case JG_CMD_TIMEOUT: if (!JG_KIND_IN(JG_CMD_TIMEOUT)) return 0; return jg_follower_timeout(d, L);
    case JG_CMD_CLIENT_REQUEST:  // :258-270
      if (!JG_KIND_IN(JG_CMD_CLIENT_REQUEST)) return 0;
      if (L.flags & JGF_HAS_LEADER)
        jg_emit_msg(d, L, JG_CMD_CLIENT_REQUEST, JG_TO_PEER, L.leader_id, 0, 0, c.id, 0);
      else
        jg_enqueue(d, L, c.id);
      return 0;
    case JG_CMD_CLIENT_RESPONSE:  // :272-282
      if (!JG_KIND_IN(JG_CMD_CLIENT_RESPONSE)) return 0;
      jg_emit_msg(d, L, JG_CMD_CLIENT_RESPONSE, JG_TO_CLIENT, 0, 0, 0, c.id, 0);
      return 0;
    default: return 0;  // apply_self
  }
}
template <uint32_t KINDS = JG_KINDS_ALL>
__device__ inline uint32_t jg_candidate_apply(const JgDev& d, JgLane& L, const JgCmd& c) {  // candidate.rs:170-196
  switch (c.kind) {
    case JG_CMD_TICK: if (!JG_KIND_IN(JG_CMD_TICK)) return 0; return jg_candidate_tick(d, L);
    case JG_CMD_VOTE_REQUEST:  // :71-88
      if (!JG_KIND_IN(JG_CMD_VOTE_REQUEST)) return 0;
      if (c.term > L.term) {
        jg_set_term(L, c.term);
        jg_follower_from_candidate(L);
        return 0;
      }
      jg_emit_msg(d, L, JG_CMD_VOTE_RESPONSE, JG_TO_PEER, c.from, 0, L.term, 0, 0);
      return 0;
    case JG_CMD_VOTE_RESPONSE: if (!JG_KIND_IN(JG_CMD_VOTE_RESPONSE)) return 0; return jg_candidate_vote_response(d, L, c.flag != 0, c.from);
    case JG_CMD_APPEND_ENTRIES:  // :116-134
      if (!JG_KIND_IN(JG_CMD_APPEND_ENTRIES)) return 0;
      if (c.term >= L.term) jg_follower_from_candidate(L);
      return 0;
    case JG_CMD_HEARTBEAT: {  // :137-157
      if (!JG_KIND_IN(JG_CMD_HEARTBEAT)) return 0;
      bool has_committed = jg_chain_has(d, L, c.id);
      uint64_t own = L.commit;
      jg_set_term(L, c.term);
      jg_vote_for(L, c.from);
      jg_follower_from_candidate(L);
      jg_emit_msg(d, L, JG_CMD_HEARTBEAT_RESPONSE, JG_TO_PEER, c.from, has_committed ? 1 : 0, 0, own, 0);
      return 0;
    }
    case JG_CMD_CLIENT_REQUEST:  // :190-193
      if (!JG_KIND_IN(JG_CMD_CLIENT_REQUEST)) return 0;
      jg_enqueue(d, L, c.id);
      return 0;
    default: return 0;
  }
}
template <uint32_t KINDS = JG_KINDS_ALL>
__device__ inline uint32_t jg_leader_apply(const JgDev& d, JgLane& L, const JgCmd& c) {  // leader.rs:248-266
  switch (c.kind) {
    case JG_CMD_TICK: if (!JG_KIND_IN(JG_CMD_TICK)) return 0; return jg_leader_tick(d, L);
    case JG_CMD_HEARTBEAT_RESPONSE:  // :222-231
      if (!JG_KIND_IN(JG_CMD_HEARTBEAT_RESPONSE)) return 0;
      return (!c.flag && c.id > 0) ? jg_leader_replicate(d, L) : 0;
    case JG_CMD_APPEND_RESPONSE: if (!JG_KIND_IN(JG_CMD_APPEND_RESPONSE)) return 0; return jg_leader_append_response(d, L, c.from, c.id);
    case JG_CMD_APPEND_ENTRIES:  // :200-208
      if (!JG_KIND_IN(JG_CMD_APPEND_ENTRIES)) return 0;
      if (c.term > L.term) {
        uint32_t f = jg_set_term(L, c.term);  // unimplemented!() (Q3)
        if (f) return f;
        jg_follower_from_leader(L);
      }
      return 0;
    case JG_CMD_CLIENT_REQUEST: if (!JG_KIND_IN(JG_CMD_CLIENT_REQUEST)) return 0; return jg_leader_client_request(d, L, c.id);
    default: return 0;
  }
}

// process restart: Raft::<Follower>::new (follower.rs:68-95) on the persisted chain - or (`empty_store`: JG_CMD_RECREATE)
// on an EMPTY data directory: the replica of a partition that was re-created starts over from Chain::new's genesis
// (chain.rs:117-153: no blocks, no "commit" key)
__device__ inline void jg_restart(const JgDev& d, JgLane& L, bool empty_store = false) {
  if (empty_store) {
    L.flags &= ~(JGF_COMMIT_KEY | JGF_WIN_MASK | JGF_NO_GENESIS);  // the id set is {0}: the run [0, 0], no segments
    L.run_hi = 0;
    L.commit = 0;
  }
  L.flags &= ~(JGF_FAULT_MASK | JGF_ROLE_MASK | JGF_VOTED | JGF_HAS_LEADER | JGF_REPL_MASK);
  uint32_t f = jg_chain_reopen(d, L);
  L.term = 0;
  L.voted_for = 0;
  L.leader_id = 0;
  L.queued = 0;
  L.votes = 0;
  L.heartbeat_time = 0;
  jg_set_election_timeout(d, L);
  if (f) jg_raise(d, L, f);
}

// RaftHandle::apply, mod.rs:471-479
template <uint32_t KINDS = JG_KINDS_ALL>
__device__ inline void jg_apply(const JgDev& d, JgLane& L, const JgCmd& c, const uint64_t* blk_id,
                                const uint64_t* blk_next) {
  if (JG_KIND_IN(JG_CMD_RESTART) && (c.kind == JG_CMD_RESTART || c.kind == JG_CMD_RECREATE)) {  // (one mask bit for the two)
    jg_restart(d, L, c.kind == JG_CMD_RECREATE);
    return;
  }
  if (jg_fault(L)) return;  // the reference process is gone
  uint32_t f;
  switch (jg_role(L)) {
    case JG_ROLE_FOLLOWER: f = jg_follower_apply<KINDS>(d, L, c, blk_id, blk_next); break;
    case JG_ROLE_CANDIDATE: f = jg_candidate_apply<KINDS>(d, L, c); break;
    default: f = jg_leader_apply<KINDS>(d, L, c); break;
  }
  if (f) jg_raise(d, L, f);
}
// whether a census of command kinds (bit k: some row of kind k) stays inside a mask
__host__ __device__ inline bool jg_kinds_within(uint32_t census, uint32_t mask) { return (census & ~mask) == 0; }
