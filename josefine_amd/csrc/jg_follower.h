// jg_follower.h — follower half of the dense node tick (jg_step_dense_follower).
//
// Per group, in this order: Heartbeat (follower.rs:178-217), AppendEntries (follower.rs:130-176),
// Tick (follower.rs:121-128) from one dense inbox row, answers into one dense outbox row.  The
// HBM-bound kernel serves the steady-state case — a healthy follower whose chain is the run
// [0, head] and that has no queued client requests — in registers:
//
//   Heartbeat      set_election_timeout (one draw of the counter RNG), term := hb.term, leader /
//                  vote := sender, has_committed = commit <= head, commit advance, response row
//   AppendEntries  term / vote adoption (follower.rs:137-144), the stale-leader assert (:147-154),
//                  Chain::extend of ids from+1 .. from+n (each next = id-1): parent check once,
//                  head := from + n (extend sets head = block.id unconditionally, chain.rs:190, so a
//                  re-sent window can move it backwards: the run then splits and the group leaves
//                  run form), ack row
//   Tick           election timer; a follower that has voted never campaigns (follower.rs:249, Q4)
//
// Everything else (candidates whose timer fires, leaders with input, input for irregular chains or
// with queued requests, a timer that fires on a follower that has not voted) is deferred (jg_defer_mark_in) to k_follower_slow,
// which runs the general state machine and maps AppendResponse / HeartbeatResponse rows back to
// the outbox columns; rows outside the mailbox vocabulary go to the exceptional queue.
//
// Bytes per follower-step at steady state: read 4 (flags) + 8+4+4 (term, voted_for, leader_id) +
// 8+8 (head, commit) + 4 (queued) + 4+8+4 (rng_draws, election timer) + inbox 16+8 = 80;
// write head 8 + outbox 8, on a heartbeat also outbox 8 + commit 8 + timer 8+4+4: 16 / 48.
#pragma once
#include "jg_dense.h"

struct JgFollowerArgs {
  const uint32_t* leader;  // [G] or null
  uint32_t leader_id;
  const jg_leader_beat* beat;  // [G] {term, Heartbeat.commit or JG_NO_ACK}
  const uint64_t* ae;          // [G] JG_AE(from, n) or JG_NO_ACK
  const uint64_t* aec;         // [G] a jg_dense_cluster's common AppendEntries word (JgLeaderNode::o_aec) or null
  uint64_t* o_answer;          // [G] JG_ANSWER(AppendResponse.head, HeartbeatResponse code)
  uint64_t* o_hbc;             // [G] HeartbeatResponse.commit, where there is one
  uint64_t now;
  uint32_t seq;
  uint32_t tick;
  const JgClock* clock;  // non-null: `now` and `seq` come from here (slot clock_slot): a replayed round (see JgClock)
  uint32_t clock_slot, pad_;
  // jg_step_node: what the step pushed on fsm_tx, one word per group (jg_dense.h JG_FSM_*_BIT); null otherwise
  uint32_t* fsm_delta;
  uint64_t* fsm_prev;
  // a cluster with per-partition leadership (JgLeaderNode::owner): the sender of a group's mail is its owner; there is
  // mail only where somebody else owns the group (the columns keep last round's words where nobody wrote)
  const uint8_t* owner;
  uint32_t self_slot;
  uint32_t seq_off;  // added to the clock's step number (the follower half is the node's second step of a round there)
  const uint64_t* sparse_bits;  // jg_step_node, JG_NODE_ASYNC: see JgLeaderNode
  uint32_t sparse_mode, pad3_;
};
__device__ __forceinline__ uint32_t jg_member_id(const JgDev& d, uint32_t slot) {
  uint32_t id = 0;
#pragma unroll
  for (uint32_t r = 0; r < JG_MAX_REPLICAS; r++) id = r == slot ? d.node_ids[r] : id;
  return id;
}
__device__ __forceinline__ void jg_follower_fsm_note(const JgFollowerArgs& a, uint32_t g, uint64_t commit0, uint64_t commit1) {
  if (!a.fsm_delta || commit1 == commit0) return;  // follower.rs:201-207: one Apply range per Heartbeat that advances
  const uint64_t adv = commit1 - commit0;
  if (adv <= JG_FSM_ADV_MASK) {
    a.fsm_delta[g] = JG_FSM_FOLLOWER_BIT | (uint32_t)adv;
  } else {
    a.fsm_prev[g] = commit0;
    a.fsm_delta[g] = JG_FSM_FOLLOWER_BIT | JG_FSM_WIDE_BIT;
  }
}

template <bool ANY = false>
__device__ __forceinline__ void jg_follower_fast_body(const JgDev& d, JgFollowerArgs a) {
  if (a.clock) jg_clock_read(a.clock, a.clock_slot, a.now, a.seq), a.seq += a.seq_off;
  const uint32_t G = d.G;
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < G; g += gridDim.x * JG_BLOCK) {
    if (a.sparse_bits && jg_sparse_skip(a.sparse_bits, a.sparse_mode, g)) continue;
    // every load is independent of the others
    const uint32_t f = d.flags[g];
    uint32_t own = 0;
    if (ANY) {  // a wave whose 64 groups this node all leads (and owns) has nothing to do here, and its word of the answers is nobody's
      own = a.owner[g];
      if (__ballot(!((f & JGF_ROLE_MASK) == JG_ROLE_LEADER && own == a.self_slot)) == 0) continue;
    }
    const bool mail = !ANY || (own != JG_OWNER_NONE && own != a.self_slot);
    jg_leader_beat beat = a.beat[g];  // one 16-byte load
    uint64_t in_ae = jg_ae_word_for(a.aec, a.ae, g);
    if (ANY && !mail) beat = jg_leader_beat{0, JG_NO_ACK}, in_ae = JG_NO_ACK;
    const uint64_t in_term = beat.term, in_hbc = beat.hb_commit;
    const uint64_t in_from = in_ae >> 8;
    const uint32_t in_n = (uint32_t)in_ae & 0xffu;
    const uint32_t lead = ANY ? jg_member_id(d, own) : (a.leader ? a.leader[g] : a.leader_id);
    uint64_t term = d.term[g], head = d.head[g], commit = d.commit[g];
    // the vote, the leader id, the queue length and the election timer (a heartbeat redraws it: rng_draws; a Tick
    // without one reads it): two 16-byte records, one 16-byte load each
    const JgCold cold0 = jg_cold_load(d.cold, g);
    uint32_t voted_for = cold0.voted_for, leader_id = cold0.leader_id;
    const uint32_t queued = cold0.queued;
    uint32_t draws = cold0.rng_draws;
    uint64_t et = cold0.election_time;
    uint32_t eto = cold0.election_timeout;
    const bool has_hb = in_hbc != JG_NO_ACK, has_ae = in_n != JG_AE_NONE;

    const uint32_t role = f & JGF_ROLE_MASK;
    const bool dead = (f & JGF_FAULT_MASK) != 0;
    // leaders ignore Heartbeat (leader.rs:263) and are ticked by the leader half
    const bool idle_leader = role == JG_ROLE_LEADER && !has_ae;
    // a Tick alone changes nothing unless the election timer has fired - and then only for a candidate
    // or a follower that has not voted (follower.rs:121-128,248-256; candidate.rs:46-66): whatever the
    // chain looks like (a restarted replica's is not in run form), such a group needs no general path
    const bool fired = (a.now - et) > (uint64_t)eto;
    const bool quiet = !has_hb && !has_ae &&
                       ((role == JG_ROLE_FOLLOWER && (!fired || (f & JGF_VOTED))) || (role == JG_ROLE_CANDIDATE && !fired));
    const bool nothing = (!has_hb && !has_ae && !a.tick) || quiet;
    const bool fast = role == JG_ROLE_FOLLOWER && (f & JGF_RUN) && queued == 0;
    const bool defer = !dead && !idle_leader && !nothing && !fast;
    jg_defer_mark_in(d.fdefer_bits, d, g, defer);
    if (dead || idle_leader || nothing || defer) {
      // nothing; the slow kernel overwrites the words of its groups (ANY: the own slot's word of a group this node owns is nobody's)
      if (!(ANY && idle_leader && own == a.self_slot)) a.o_answer[g] = JG_NO_ACK;
      continue;
    }
    uint64_t o_ack = JG_MAILBOX_NONE;
    uint32_t o_has = JG_HB_NONE;

    uint32_t nf = f;
    const uint64_t term0 = term, head0 = head, commit0 = commit;
    const uint32_t vf0 = voted_for, lid0 = leader_id;
    bool timer_dirty = false;
    uint32_t fault = 0;

    if (has_hb) {  // ---- follower.rs:178-217
      // set_election_timeout (follower.rs:103-113): one draw, election_time = now
      const uint32_t span = d.el_max - d.el_min;
      const uint64_t r = jg_mix64(d.seed ^ jg_mix64((d.group_base + g) * 0xd1342543de82ef95ull + draws));
      eto = d.el_min + (span ? (uint32_t)(r % span) : 0u);
      et = a.now;
      draws += 1;
      timer_dirty = true;
      term = in_term;                     // :185, unconditional (Q6)
      nf |= JGF_HAS_LEADER | JGF_VOTED;   // :186-187
      leader_id = lead;
      voted_for = lead;
      const bool has = in_hbc <= head;    // :200, run form: the id set is [0, head]
      if (has && in_hbc > commit) {       // :201-207
        commit = in_hbc;
        nf |= JGF_COMMIT_KEY;
      }
      a.o_hbc[g] = commit;                // :209-215
      o_has = has ? 1 : 0;
    }
    if (has_ae) {  // ---- follower.rs:130-176
      if (!(nf & JGF_VOTED) && in_term >= term) {  // :137-144
        term = in_term;                            // Raft::term clears voted_for and leader_id first
        et = a.now;
        timer_dirty = true;
        nf |= JGF_HAS_LEADER | JGF_VOTED;
        leader_id = lead;
        voted_for = lead;
      }
      if ((nf & JGF_VOTED) && voted_for != lead && in_term < term) {  // :147-154
        fault = JG_FAULT_FOLLOWER_STALE_LEADER;
      } else if (in_n) {                                              // :157-172
        if (in_from > head) {
          fault = JG_FAULT_EXTEND_MISSING_PARENT;                     // chain.rs:180-185 on the first block
        } else {
          const uint64_t new_head = in_from + in_n;
          const uint64_t run_hi = new_head > head ? new_head : head;  // ids <= head exist with the same parent
          if (nf & JGF_FAST) {             // id_gen was implicit (head+1) and does not follow extend (Q8)
            d.id_gen[g] = head + 1;
            nf &= ~JGF_FAST;
          }
          head = new_head;                 // chain.rs:190, unconditionally the last block's id
          if (head != run_hi) {            // moved backwards: [0, run_hi] stays stored, run form is lost
            d.run_hi[g] = run_hi;
            nf &= ~JGF_RUN;
          }
          if (head >= JG_MAILBOX_NONE) fault = JG_FAULT_ENGINE_MAILBOX_RANGE;  // 56-bit ids in mailbox words
          else o_ack = head;               // follower.rs:163-172
        }
      }
    }
    bool tick_defer = false;
    if (a.tick && !fault) {  // ---- follower.rs:121-128 -> 248-256
      // a follower that has voted does nothing on Timeout (Q4); otherwise it becomes a candidate
      tick_defer = (a.now - et) > (uint64_t)eto && !(nf & JGF_VOTED);
    }
    if (fault) {
      nf |= fault << JGF_FAULT_SHIFT;
      jg_push_fault(d, g, fault, a.seq);
    }
    a.o_answer[g] = JG_ANSWER(o_ack, o_has);
    if (term != term0) d.term[g] = term;
    if (head != head0) d.head[g] = head;
    if (commit != commit0) d.commit[g] = commit;
    jg_follower_fsm_note(a, g, commit0, commit);
    {  // (a heartbeat rewrites the timer half of the record; the other half only when the vote or the leader changed)
      const JgCold c = jg_cold_of(et, voted_for, leader_id, eto, draws, queued, cold0.votes);
      if (timer_dirty) jg_cold_store_timer(d.cold, g, c);
      if (voted_for != vf0 || leader_id != lid0) jg_cold_store_rest(d.cold, g, c);
    }
    if (nf != f) d.flags[g] = nf;
    // (divergent use of the wave-aggregated mark is fine: the ballot covers the active lanes)
    jg_defer_mark_in(d.fdefer_bits + (G + 63u) / 64u, d, g, tick_defer);
  }
}

__global__ __launch_bounds__(JG_BLOCK) void k_follower_tick_dense(JgDev d, JgFollowerArgs a) { jg_follower_fast_body<false>(d, a); }

// The follower halves of several nodes that share a device in ONE launch (blockIdx.y = node): what a
// replayed cluster round uses - a kernel costs ~4.6 us before it does anything, and a round had four of these
// and four slow kernels behind them.  The jobs live in device memory (written when the round is captured).
struct JgFollowerJob {
  JgDev d;
  JgFollowerArgs a;
};
__global__ __launch_bounds__(JG_BLOCK) JG_FOLLOWER_OCC void k_follower_tick_dense_multi(const JgFollowerJob* __restrict__ jobs) {
  const JgFollowerJob& j = jobs[blockIdx.y];
  jg_follower_fast_body<false>(j.d, j.a);
}
__global__ __launch_bounds__(JG_BLOCK) void k_follower_tick_dense_any(const JgFollowerJob* __restrict__ jobs) {  // (per-partition leadership)
  const JgFollowerJob& j = jobs[blockIdx.y];
  jg_follower_fast_body<true>(j.d, j.a);
}

// The deferred groups through the general state machine.  AppendResponse / HeartbeatResponse
// rows are captured into the outbox columns, everything else goes to the exceptional queue.
__device__ __forceinline__ void jg_follower_slow_body(const JgDev& d, JgFollowerArgs a) {
  if (a.clock) jg_clock_read(a.clock, a.clock_slot, a.now, a.seq), a.seq += a.seq_off;
  uint32_t dec = 0;
  // this workgroup's shard of the two deferral bitmaps (jg_defer_mark_in: a wave of the dense half or-s its ballot in
  // and goes on - appending to a list made every wave with a deferred group wait for its slot) -> its list, the
  // words cleared for the next launch; the order within the list is immaterial (see k_dense_slow)
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  uint32_t* list = d.slow_list + (size_t)blockIdx.x * d.slow_cap;
  {
    const uint32_t n_words = (d.G + 63u) / 64u;
    const uint32_t wpb = (n_words + gridDim.x - 1) / gridDim.x;
    const uint32_t w0 = blockIdx.x * wpb, w1 = w0 + wpb < n_words ? w0 + wpb : n_words;
    for (uint32_t w = w0 + threadIdx.x; w < w1; w += JG_BLOCK) {
      uint64_t m = d.fdefer_bits[w], mt = d.fdefer_bits[n_words + w];
      if (!(m | mt)) continue;
      if (m) d.fdefer_bits[w] = 0;
      if (mt) d.fdefer_bits[n_words + w] = 0;
      uint32_t at = atomicAdd(&s_n, (uint32_t)(__popcll(m) + __popcll(mt)));
      for (int pass = 0; pass < 2; pass++) {
        uint64_t b = pass ? mt : m;
        for (; b; b &= b - 1, at++) {
          const uint32_t g = w * 64u + (uint32_t)__ffsll((long long)b) - 1u;
          if (at < d.slow_cap) list[at] = g | (pass ? JG_DEFER_TICK_ONLY : 0u);
          else *d.err = 4;
        }
      }
    }
  }
  __syncthreads();
  const uint32_t n = s_n < d.slow_cap ? s_n : d.slow_cap;
  for (uint32_t i = threadIdx.x; i < n; i += JG_BLOCK) {
    const uint32_t entry = list[i];
    const uint32_t g = entry & ~JG_DEFER_TICK_ONLY;
    const bool tick_only = (entry & JG_DEFER_TICK_ONLY) != 0;
    JgLane L;
    jg_load(d, L, g);
    const uint64_t fsm_commit0 = L.commit;
    // the Tick of a group the leader half ticks is not this half's: a leader that steps down on its way through the
    // inputs (leader.rs:200-208) has had its Tick for this round (the role at entry decides, not the role after)
    const bool was_leader = jg_role(L) == JG_ROLE_LEADER;
    L.now = a.now;
    L.seq = a.seq;
    L.mp = L.mend = nullptr;
    jg_fsm_row sink[2];
    L.xq_on = 2;  // capture mode: mailbox rows -> L.cap_*, the rest -> exceptional queue
    L.cap_ack = JG_NO_ACK;
    L.cap_hbc = 0;
    L.cap_has = JG_HB_NONE;
    const uint32_t own = a.owner ? a.owner[g] : 0u;
    const bool mail = !a.owner || (own != JG_OWNER_NONE && own != a.self_slot);  // (per-partition leadership: see JgFollowerArgs)
    const uint32_t lead = a.owner ? jg_member_id(d, own) : (a.leader ? a.leader[g] : a.leader_id);
    JgCmd c;
    c.from = lead;
    c.flag = 0;
    c.term = a.beat[g].term;
    c.aux = 0;
    if (!tick_only) {
      const uint64_t hbc = mail ? a.beat[g].hb_commit : JG_NO_ACK;
      const uint64_t ae = mail ? jg_ae_word_for(a.aec, a.ae, g) : JG_NO_ACK;
      const uint32_t n_blk = (uint32_t)ae & 0xffu;
      if (hbc != JG_NO_ACK) {
        c.kind = JG_CMD_HEARTBEAT;
        c.id = hbc;
        L.fp = sink;
        L.fend = sink + 2;
        jg_apply(d, L, c, nullptr, nullptr);
      }
      if (n_blk != JG_AE_NONE) {
        c.kind = JG_CMD_APPEND_ENTRIES;
        c.id = ae >> 8;  // implicit blocks: ids id+1 .. id+aux, next = id-1 each
        c.aux = n_blk;
        L.fp = sink;
        L.fend = sink + 2;
        jg_apply(d, L, c, nullptr, nullptr);
        if (L.cap_ack != JG_NO_ACK && L.cap_ack >= JG_MAILBOX_NONE && !jg_fault(L)) {  // 56-bit ids in mailbox words:
          jg_raise(d, L, JG_FAULT_ENGINE_MAILBOX_RANGE);                                // raised where the answer is produced
          L.cap_ack = JG_NO_ACK;
        }
      }
    }
    if (a.tick && !was_leader && jg_role(L) != JG_ROLE_LEADER) {
      c.kind = JG_CMD_TICK;
      c.from = 0;
      c.term = c.id = c.aux = 0;
      L.fp = sink;
      L.fend = sink + 2;
      jg_apply(d, L, c, nullptr, nullptr);
    }
    if (!tick_only) {
      a.o_answer[g] = JG_ANSWER(L.cap_ack == JG_NO_ACK ? JG_MAILBOX_NONE : L.cap_ack, L.cap_has);
      if (L.cap_has != JG_HB_NONE) a.o_hbc[g] = L.cap_hbc;
    }
    // (the only fsm_tx output of this half is a follower's Apply range: a candidate or a leader that takes a
    // Heartbeat does not move its commit index, candidate.rs:137-157, leader.rs:263)
    jg_follower_fsm_note(a, g, fsm_commit0, L.commit);
    dec += L.decisions;
    jg_store(d, L);
  }
  jg_block_count(d.blk_decisions, dec);
}
__global__ __launch_bounds__(JG_BLOCK) void k_follower_slow(JgDev d, JgFollowerArgs a) { jg_follower_slow_body(d, a); }
// The slow kernels' jobs travel as KERNEL ARGUMENTS (an array in the kernarg segment, indexed by blockIdx.y):
// the general state machine reads JgDev's pointers around every store; through a reference into global memory
// they were re-loaded each time, a by-value copy of the job went to scratch - 1 ms per launch either way as soon
// as the lists were not empty (the routed round of configs[4]).  7 jobs: 4 KB of kernel arguments is the limit.
#define JG_FOLLOWER_MULTI 7
struct JgFollowerJobs {
  JgFollowerJob j[JG_FOLLOWER_MULTI];
};
__global__ __launch_bounds__(JG_BLOCK) void k_follower_slow_multi(JgFollowerJobs jobs) {
  jg_follower_slow_body(jobs.j[blockIdx.y].d, jobs.j[blockIdx.y].a);
}
