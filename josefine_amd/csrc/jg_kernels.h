// jg_kernels.h — gfx950 kernels of the batched Chained-Raft engine.
//
//   k_leader_tick_dense<R>  the HBM-roofline kernel: AppendEntries-ack tally,
//                           majority test and commit-index advance for steady-state
//                           leaders over SoA columns (progress.rs:42-60,133-140;
//                           leader.rs:87-99,177-197,211-219)
//   k_leader_node_tick<R>   the same with HeartbeatResponses in and the Tick's outbox out
//   k_follower_tick_dense   Heartbeat + AppendEntries + Tick for followers over dense mailboxes
//   k_dense_slow / k_follower_slow  the same ticks for deferred groups (general state machine)
//   k_apply_rows            the full state machine over a group-sorted command batch
//                           (every role, every Command; mod.rs:471-479)
//   k_chain_compact         Chain::compact parent-pointer walk (chain.rs:239-253)
//   k_synth_acks            synthetic ack-stream generator (bench / parity input)
//   k_gather_rows           drain-time compaction of the per-group output regions
#pragma once
#include "jg_device.h"

#include "jg_dense.h"  // k_leader_tick_dense / _n, jg_block_count, JG_BLOCK

// ---- groups the dense leader kernel deferred ----------------------------------------------------
// Healthy leaders whose chain is not in FAST form, and (node tick, T-tick kernel) leaders whose
// tick does not stay in lag space or that received a HeartbeatResponse without the commit: the
// same tick(s) through the general state machine.  Workgroup s owns shard s of the deferral
// bitmap (jg_defer_mark), turns it into its list and clears it for the next launch.  Message rows
// outside the dense mailbox vocabulary go to the exceptional queue.
// NODE: behind a node tick (HeartbeatResponses in, the Tick's outbox out).  A template parameter so that the
// instance behind the ack-only kernels carries neither the Tick's local row buffer (scratch) nor its code.
template <bool NODE>
__device__ __forceinline__ void jg_dense_slow_body(const JgDev& d, const uint64_t* __restrict__ acks, uint32_t n_ticks,
                                                   size_t tick_stride, uint32_t seq0, JgLeaderNode nd, bool advance_clock) {
  if (nd.clock) jg_clock_read(nd.clock, nd.clock_slot, nd.now, seq0);
  uint32_t dec = 0;
  // this workgroup's shard of the deferral bitmap (jg_defer_mark) -> its list; the words are
  // cleared for the next launch.  (Order within the list is immaterial: groups are independent,
  // fault and exceptional rows are ordered at drain time.)
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  uint32_t* list = d.slow_list + (size_t)blockIdx.x * d.slow_cap;
  {
    const uint32_t n_words = (d.G + 63u) / 64u;
    const uint32_t wpb = (n_words + gridDim.x - 1) / gridDim.x;
    const uint32_t w0 = blockIdx.x * wpb, w1 = w0 + wpb < n_words ? w0 + wpb : n_words;
    for (uint32_t w = w0 + threadIdx.x; w < w1; w += JG_BLOCK) {
      uint64_t m = d.defer_bits[w];
      if (!m) continue;
      d.defer_bits[w] = 0;
      uint32_t at = atomicAdd(&s_n, (uint32_t)__popcll(m));
      while (m) {
        const uint32_t b = (uint32_t)__ffsll((long long)m) - 1u;
        if (at < d.slow_cap) list[at] = w * 64u + b;
        else *d.err = 4;
        at++;
        m &= m - 1;
      }
    }
  }
  __syncthreads();
  const uint32_t n = s_n < d.slow_cap ? s_n : d.slow_cap;
  for (uint32_t i = threadIdx.x; i < n; i += JG_BLOCK) {
    const uint32_t g = list[i];
    JgLane L;
    jg_load(d, L, g);
    const uint64_t fsm_head0 = L.head, fsm_commit0 = L.commit;  // (jg_step_node: what this tick pushes on fsm_tx)
    L.now = nd.now;
    L.mp = L.mend = nullptr;
    L.xq_on = d.xq != nullptr;  // plain jg_step_dense_acks: a leader's appends / acks emit no messages
    jg_fsm_row sink[2];
    const uint32_t s = jg_self(L);
    JgCmd c;
    c.from = 0;
    c.flag = 0;
    c.term = c.id = c.aux = 0;
    // own slot outside its domain (JG_MAX_DENSE_APPENDS; a leader only: non-leaders never get here):
    // nothing of the tick is applied.  Checked again per tick below (T-tick launches).
    // (node tick: the block holds answer words, JG_ANSWER(head, HeartbeatResponse code))
    const bool packed = NODE && nd.packed != 0;
    auto ack_of = [=](uint64_t w) { return packed ? jg_answer_ack(w) : w; };
    // per-partition leadership (JgLeaderNode::owner): the group's mailboxes are its owner's; a leader that is not the
    // owner has no inbox, appends nothing and its Tick travels as rows; the own slot's word comes from `offered`
    const bool mine = !(NODE && nd.owner) || nd.owner[g] == s;
    const uint64_t* const ga = mine ? acks : nullptr;  // this group's inbox
    auto own_word = [=](const uint64_t* A) { return (NODE && nd.offered) ? nd.offered[g] : A[(size_t)s * d.G + g]; };
    // (jg_step_node: the word holds 0 or 1 and, above JG_NODE_PRE_SHIFT, the arrival bits)
    if (!(NODE && nd.arr) && ga && n_ticks && jg_role(L) == JG_ROLE_LEADER && !jg_fault(L) &&
        ack_of(own_word(ga)) >= JG_MAX_DENSE_APPENDS) {
      L.seq = seq0;
      jg_raise(d, L, JG_FAULT_ENGINE_DENSE_APPENDS);
    }
    uint64_t fsm_mid = fsm_commit0;  // (jg_step_node: the commit index at the moment the ClientRequest was applied)
    if (NODE && nd.arr && packed && ga) {
      // jg_step_node: the group's commands one at a time IN THE ORDER THEY ARRIVED (server.rs:120-161) - the inbox
      // entries sorted by the arrival index k_node_classify left with each (selection: at most 2R entries); a slot
      // that spoke a column comes after the rows, HeartbeatResponse then AppendResponse
      L.seq = seq0;
      uint32_t last = 0;
      for (;;) {
        uint32_t best = 0xffffffffu, be = 0;
        for (uint32_t e2 = 0; e2 < 2u * d.R; e2++) {
          const bool is_ack = e2 < d.R;
          const uint32_t r = is_ack ? e2 : e2 - d.R;
          const uint64_t w = ga[(size_t)r * d.G + g];
          bool present;
          if (r == s) present = is_ack && (uint32_t)(w >> 8) != 0;  // the ClientRequest
          else present = is_ack ? (w >> 8) != JG_MAILBOX_NONE : jg_answer_hb(w) != JG_HB_NONE;
          if (!present) continue;
          const uint32_t key = (r != s && ((nd.col_mask >> r) & 1u)) ? 0x80000000u + 2u * r + (is_ack ? 1u : 0u)
                                                                     : nd.arr[(size_t)e2 * d.G + g];
          if (key > last && key < best) best = key, be = e2;
        }
        if (best == 0xffffffffu || jg_fault(L)) break;
        last = best;
        const bool is_ack = be < d.R;
        const uint32_t r = is_ack ? be : be - d.R;
        const uint64_t w = ga[(size_t)r * d.G + g];
        c.from = d.node_ids[r];
        c.term = c.aux = 0;
        if (r == s) {
          if ((uint32_t)(w >> 8) > 1u) *d.err = 1;  // (jg_step_node offers at most one: one Notify per step)
          fsm_mid = L.commit;
          c.kind = JG_CMD_CLIENT_REQUEST, c.from = 0, c.flag = 0, c.id = 0;
        } else if (is_ack) {
          c.kind = JG_CMD_APPEND_RESPONSE, c.flag = 1, c.id = w >> 8;
        } else {
          const uint32_t has = jg_answer_hb(w);
          c.kind = JG_CMD_HEARTBEAT_RESPONSE, c.flag = (uint8_t)has, c.id = has ? 0 : nd.hbr_commit[(size_t)r * d.G + g];
        }
        L.fp = sink;
        L.fend = sink + 2;
        jg_apply(d, L, c, nullptr, nullptr);
      }
      c.from = 0;
      c.id = 0;
    } else {
      if (NODE && packed && ga && !jg_fault(L)) {  // 1. HeartbeatResponses, ascending slot (leader.rs:222-231)
        L.seq = seq0;
        c.kind = JG_CMD_HEARTBEAT_RESPONSE;
        for (uint32_t r = 0; r < d.R && !jg_fault(L); r++) {
          if (r == s) continue;
          const uint32_t has = jg_answer_hb(ga[(size_t)r * d.G + g]);
          if (has == JG_HB_NONE) continue;
          c.from = d.node_ids[r];
          c.flag = has;
          c.id = has ? 0 : nd.hbr_commit[(size_t)r * d.G + g];
          jg_apply(d, L, c, nullptr, nullptr);
        }
        c.from = 0;
        c.id = 0;
      }
      for (uint32_t t = 0; ga && t < n_ticks && !jg_fault(L); t++) {  // 2. appends, then ga
        const uint64_t* A = ga + (size_t)t * tick_stride;
        L.seq = seq0 + t;
        uint64_t n_app = ack_of(own_word(A));
        if (n_app >= JG_MAX_DENSE_APPENDS) {
          jg_raise(d, L, JG_FAULT_ENGINE_DENSE_APPENDS);
          break;
        }
        if (NODE && nd.fsm_delta && n_app > 1) *d.err = 1;  // (jg_step_node offers at most one: the word below holds one Notify)
        c.kind = JG_CMD_CLIENT_REQUEST;
        c.flag = 0;
        for (uint64_t k = 0; k < n_app && !jg_fault(L); k++) {
          L.fp = sink;
          L.fend = sink + 2;
          jg_apply(d, L, c, nullptr, nullptr);
        }
        c.kind = JG_CMD_APPEND_RESPONSE;
        c.flag = 1;
        for (uint32_t r = 0; r < d.R && !jg_fault(L); r++) {
          if (r == s) continue;
          uint64_t h = ack_of(A[(size_t)r * d.G + L.g]);
          if (h == JG_NO_ACK) continue;
          c.from = d.node_ids[r];
          c.id = h;
          L.fp = sink;
          L.fend = sink + 2;
          jg_apply(d, L, c, nullptr, nullptr);
        }
      }
    }
    if (NODE && nd.o_beat && !jg_fault(L)) {  // 3. Command::Tick (leader.rs:234-245)
      L.seq = seq0;
      c.kind = JG_CMD_TICK;
      c.from = 0;
      c.flag = 0;
      c.id = 0;
      if (jg_wcnt(L)) jg_chain_normalize(d, L);
      // the Tick fits the columns when the id set is a run [0, run_hi] with every parent id - 1 - whatever head and id_gen
      // are: a re-elected leader that was restarted has its head at its commit index, below the top of what sled kept
      // (chain.rs:117-137), and replicates out of that run (leader.rs:135,152-157 range over the stored keys)
      bool fast = jg_wcnt(L) == 0 && !(L.flags & JGF_NO_GENESIS);
      // ... and every AppendEntries word must be able to hold its range start key (a progress head forged
      // up to 2^56 - 1 or beyond does not fit the 56-bit field): otherwise the Tick travels as rows
      for (uint32_t r = 0; r < d.R; r++) fast = fast && (r == s || jg_match_get(d, L, r) < JG_MAILBOX_NONE);
      fast = fast && mine;  // (the columns hold the owner's Tick)
      if (!fast) {
        jg_apply(d, L, c, nullptr, nullptr);  // rows: the blocks are not id-consecutive
      } else if (L.run_hi >= JG_MAILBOX_NONE) {  // 56-bit block ids in mailbox words (the top of the run: what a word may name)
        jg_raise(d, L, JG_FAULT_ENGINE_MAILBOX_RANGE);
      } else {                                // columns: the Tick's rows are captured as they are emitted
        L.xq_on = 3;
        L.cap_hbc = JG_NO_ACK;
        L.cap_ae = nd.o_ae;
        if (nd.o_aec) {  // a cluster's mailboxes: this group's words are in the rows (the general state machine writes them one by one)
          for (uint32_t r = 0; r < d.R; r++) nd.o_ae[(size_t)r * d.G + g] = JG_NO_ACK;
          nd.o_aec[g] = JG_AEC_INDIVIDUAL;
        }
        jg_apply(d, L, c, nullptr, nullptr);
        nd.o_beat[g] = jg_leader_beat{L.term, L.cap_hbc};
      }
    }
    if (NODE && nd.fsm_delta) {
      // the rows the general state machine pushed on fsm_tx for this tick are Notify (if it appended) and
      // the Apply ranges of its commit advances, which concatenate: (commit before, commit after]
      uint32_t w = L.head != fsm_head0 ? JG_FSM_APPENDED_BIT : 0u;
      if (L.commit != fsm_commit0) {
        w |= JG_FSM_WIDE_BIT;
        nd.fsm_prev[g] = fsm_commit0;
        nd.fsm_mid[g] = fsm_mid;
      }
      nd.fsm_delta[g] = w;
    }
    dec += L.decisions;
    jg_store(d, L);
  }
  __syncthreads();
  if (threadIdx.x == 0) d.slow_cnt[blockIdx.x] = 0;
  if (NODE && nd.clock && advance_clock && blockIdx.x == 0 && threadIdx.x == 0) {  // the next round's clock (see JgClock)
    JgClock* c = nd.clock;
    const uint32_t b = c->idx_rest & 1u;
    JgClockVal nv = c->v[b];
    nv.now += c->dt;
    for (uint32_t r = 0; r < c->n_nodes; r++) nv.seq[r] += c->seq_step;  // every node takes one step per round (two: both halves)
    c->v[b ^ 1u] = nv;
    c->idx_lead = b ^ 1u;
  }
  jg_block_count(d.blk_decisions, dec);
}
template <bool NODE>
__global__ __launch_bounds__(JG_BLOCK) JG_GSM_OCC void k_dense_slow(JgDev d, const uint64_t* __restrict__ acks, uint32_t n_ticks,
                                                          size_t tick_stride, uint32_t seq0, JgLeaderNode nd) {
  jg_dense_slow_body<NODE>(d, acks, n_ticks, tick_stride, seq0, nd, true);
}
// The leader slow kernels of all nodes of a cluster with per-partition leadership in ONE launch (blockIdx.y = node);
// the jobs travel as kernel arguments for the reason k_follower_slow_multi's do.
#define JG_LEADER_MULTI 6
struct JgLeaderSlowJob {
  JgDev d;
  const uint64_t* acks;
  uint32_t seq0, pad;
  JgLeaderNode nd;
};
struct JgLeaderSlowJobs {
  JgLeaderSlowJob j[JG_LEADER_MULTI];
};
__global__ __launch_bounds__(JG_BLOCK) JG_GSM_OCC void k_dense_slow_multi(JgLeaderSlowJobs jobs) {
  const JgLeaderSlowJob& j = jobs.j[blockIdx.y];
  jg_dense_slow_body<true>(j.d, j.acks, 1, 0, j.seq0, j.nd, blockIdx.y == 0);
}

#include "jg_sparse.h"  // k_apply_rows, k_gather_rows

// ---- Chain::compact (chain.rs:239-253) -------------------------------------------------------
// One lane per tree: walk the ids below `commit` in descending key order; keep the
// first, remove b when b.id != next_id, and set next_id = b.next even for a
// removed block (Q7).  Duplicate ids follow sled's upsert: the last entry wins.
__global__ void k_chain_compact(size_t n_trees, const uint64_t* __restrict__ off, const uint64_t* __restrict__ ids,
                                const uint64_t* __restrict__ nexts, const uint64_t* __restrict__ commits,
                                uint8_t* __restrict__ removed) {
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n_trees; t += (size_t)gridDim.x * blockDim.x) {
    const uint64_t lo = off[t], hi = off[t + 1];
    uint64_t bound = commits[t];
    bool have_next = false;
    uint64_t next_id = 0;
    for (;;) {
      bool found = false;
      uint64_t best = 0, best_i = 0;
      for (uint64_t i = lo; i < hi; i++) {
        uint64_t id = ids[i];
        if (id < bound && (!found || id >= best)) {
          found = true;
          best = id;
          best_i = i;
        }
      }
      if (!found) break;
      if (have_next && best != next_id) removed[best_i] = 1;  // :244-247
      have_next = true;
      next_id = nexts[best_i];                                // :249
      bound = best;
    }
  }
}

// ---- the clock of a replayed closed loop (JgClock) ----------------------------------------------
__global__ void k_clock_set(JgClock* c, JgClock init) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c = init;
}

// ---- Chain::compact on the engine's own chains (chain.rs:239-253) ------------------------------
// The walk visits the block keys below `commit` in descending order; the first is kept, every
// later block b is removed when b.id != next_id, and next_id = b.next either way (Q7: also for a
// removed block).  On the segment representation (jg_device.h "Chain") that is a walk over the
// segments in descending order: inside a segment every id is the parent pointer of the id above
// it, so only the TOP id of a segment can ever be removed (when it is not the parent the segment
// above ended on) - and then its own parent pointer (top-1) keeps the rest of the dead branch
// alive, exactly as the reference's walk does.  One lane per group; the (at most 9) segments of a
// lane are staged and sorted in LDS (dynamic indexing without scratch memory).
struct JgCompactRow {
  uint32_t group, pad;
  uint64_t id;
};
#define JG_COMPACT_SEGS (JG_CHAIN_WINDOW + 1)
__global__ __launch_bounds__(JG_BLOCK) void k_compact_resident(JgDev d, JgCompactRow* __restrict__ out,
                                                                uint32_t* __restrict__ out_n, uint32_t out_cap,
                                                                uint32_t seq) {
  __shared__ uint64_t s_lo[JG_COMPACT_SEGS][JG_BLOCK], s_hi[JG_COMPACT_SEGS][JG_BLOCK], s_nx[JG_COMPACT_SEGS][JG_BLOCK];
  __shared__ uint8_t s_ix[JG_COMPACT_SEGS][JG_BLOCK];  // which stored segment (0 = the run, w+1 = window w)
  const uint32_t t = threadIdx.x;
  const uint32_t g_end = ((d.G + JG_BLOCK - 1) / JG_BLOCK) * JG_BLOCK;  // whole waves take every iteration together
  for (uint32_t g0 = blockIdx.x * JG_BLOCK + t; g0 < g_end; g0 += gridDim.x * JG_BLOCK) {
    const bool in_range = g0 < d.G;
    const uint32_t g = in_range ? g0 : 0;
    JgLane L;
    jg_load(d, L, g);
    L.now = 0;
    L.seq = seq;
    L.mp = L.mend = nullptr;
    L.fp = L.fend = nullptr;
    const bool live = in_range && !jg_fault(L);  // (the reference process of a faulted group is gone)
    const uint64_t commit = L.commit;
    const uint32_t nw = live ? jg_wcnt(L) : 0;
    // stage: segment 0 = the run [0, run_hi] (genesis: next(0) = 0), then the window segments
    uint32_t n = 0;
    if (live && !(L.flags & JGF_NO_GENESIS)) {
      s_lo[0][t] = 0, s_hi[0][t] = L.run_hi, s_nx[0][t] = 0, s_ix[0][t] = 0;
      n = 1;
    }
    for (uint32_t w = 0; w < nw; w++) {
      s_lo[n][t] = JG_SEG(win_lo, w), s_hi[n][t] = JG_SEG(win_hi, w), s_nx[n][t] = JG_SEG(win_next, w);
      s_ix[n][t] = (uint8_t)(w + 1);
      n++;
    }
    // insertion sort, descending by first id (segments are disjoint intervals)
    for (uint32_t i = 1; i < n; i++) {
      const uint64_t lo = s_lo[i][t], hi = s_hi[i][t], nx = s_nx[i][t];
      const uint8_t ix = s_ix[i][t];
      uint32_t j = i;
      while (j > 0 && s_lo[j - 1][t] < lo) {
        s_lo[j][t] = s_lo[j - 1][t], s_hi[j][t] = s_hi[j - 1][t], s_nx[j][t] = s_nx[j - 1][t], s_ix[j][t] = s_ix[j - 1][t];
        j--;
      }
      s_lo[j][t] = lo, s_hi[j][t] = hi, s_nx[j][t] = nx, s_ix[j][t] = ix;
    }
    // the walk (the same number of iterations for every lane of the wave: the removed-block rows
    // are appended wave by wave, one atomic per wave and iteration instead of one per block)
    bool have_next = false, changed = false;
    uint64_t next_id = 0;
    uint32_t drop_mask = 0;  // window segments that disappear (their only id was removed)
    for (uint32_t i = 0; i < JG_COMPACT_SEGS; i++) {
      const bool visit = live && i < n && s_lo[i < n ? i : 0][t] < commit;  // range(0..commit)
      const uint64_t lo = visit ? s_lo[i][t] : 0, hi = visit ? s_hi[i][t] : 0;
      const uint64_t top = hi < commit ? hi : commit - 1;  // (a segment that straddles commit starts the walk inside)
      const bool remove = visit && have_next && top != next_id;  // chain.rs:244-247 — top == hi here
      const uint64_t m = __ballot(remove);
      if (m) {
        const uint32_t lane = t & 63u;
        const int first = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if ((int)lane == first) base = atomicAdd(out_n, (uint32_t)__popcll(m));
        base = __shfl(base, first, 64);
        if (remove) {
          const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          if (at < out_cap) out[at] = JgCompactRow{g, i, top};  // pad = position in the walk (orders the rows of a group)
          changed = true;
          const uint32_t ix = s_ix[i][t];
          if (ix == 0) {
            if (top == 0) L.flags |= JGF_NO_GENESIS;  // block 0 itself: the run is empty from now on
            else L.run_hi = top - 1;
          } else if (hi == lo) {
            drop_mask |= 1u << (ix - 1);
          } else {
            d.win_hi[(size_t)(ix - 1) * d.G + g] = hi - 1;
          }
        }
      }
      if (visit) {
        have_next = true;
        next_id = s_nx[i][t];  // = next(lo): everything between top and lo is the parent of the id above it
      }
    }
    if (!changed) continue;
    if (drop_mask) {  // close the gaps in the window columns
      uint32_t k = 0;
      for (uint32_t w = 0; w < nw; w++) {
        if ((drop_mask >> w) & 1u) continue;
        if (k != w) {
          JG_SEG(win_lo, k) = JG_SEG(win_lo, w);
          JG_SEG(win_hi, k) = JG_SEG(win_hi, w);
          JG_SEG(win_next, k) = JG_SEG(win_next, w);
        }
        k++;
      }
      L.flags = (L.flags & ~JGF_WIN_MASK) | (k << JGF_WIN_SHIFT);
    }
    jg_store(d, L);
  }
}

// ---- synthetic AppendEntries-ack stream (DESIGN.md "Synthetic traces") -------------------------
__device__ __forceinline__ uint64_t jg_synth_hash(uint64_t seed, uint64_t tick, uint64_t gg, uint32_t r) {
  return jg_mix64(jg_mix64(seed + tick * 0x9e3779b97f4a7c15ull) ^ (gg * 8 + r));
}
__global__ void k_synth_acks(JgDev d, uint32_t mode, uint64_t tick, uint64_t* __restrict__ sim,
                             uint64_t* __restrict__ acks) {
  const uint32_t G = d.G, R = d.R;
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x) {
    const uint32_t s = (d.flags[g] & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
    const uint64_t gg = d.group_base + g;
    const uint64_t lead = sim[(size_t)s * G + g];
    const uint64_t n_app = mode == 0 ? 1 : jg_synth_hash(d.seed, tick, gg, s) % 3;
    acks[(size_t)s * G + g] = n_app;
    sim[(size_t)s * G + g] = lead + n_app;
    for (uint32_t r = 0; r < R; r++) {
      if (r == s) continue;
      const size_t k = (size_t)r * G + g;
      if (mode == 0) {
        acks[k] = lead;
        sim[k] = lead;
      } else {
        const uint64_t u = jg_synth_hash(d.seed, tick, gg, r);
        const uint32_t p = (uint32_t)(u % 100);
        if (p < 5) {
          acks[k] = JG_NO_ACK;
        } else if (p < 10) {
          acks[k] = sim[k];
        } else {
          const uint64_t adv = (u >> 32) % (JG_MAX_INFLIGHT + 1);
          uint64_t v = sim[k] + adv;
          v = v < lead ? v : lead;
          acks[k] = v;
          sim[k] = v;
        }
      }
    }
  }
}

// ---- stream calibration (jg_calibrate_stream) ----------------------------------------------------
// The byte profile of k_leader_tick_dense<R> without any of its logic: R non-temporal 8-byte reads
// per group from a streamed block, two resident 8-byte columns and a 4-byte column read, one
// 8-byte column written in place.  What a launch of this shape costs on the machine at hand.
template <int R>
__global__ __launch_bounds__(JG_BLOCK) void k_stream_calib(const uint64_t* __restrict__ rot,
                                                            const uint64_t* __restrict__ a8, uint64_t* b8,
                                                            const uint32_t* __restrict__ c4, uint32_t G) {
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < G; g += gridDim.x * JG_BLOCK) {
    uint64_t v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = __builtin_nontemporal_load(&rot[(size_t)r * G + g]);
    uint64_t s = a8[g] ^ b8[g] ^ c4[g];
#pragma unroll
    for (int r = 0; r < R; r++) s += v[r];
    b8[g] = s;
  }
}

// ---- engine init: RaftHandle::new for every group (mod.rs:428-435, follower.rs:68-95) ----------
__global__ void k_init_groups(JgDev d, const uint8_t* __restrict__ self_slots) {
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < d.G; g += gridDim.x * blockDim.x) {
    JgLane L;
    L.g = g;
    L.now = 0;
    L.seq = 0;
    L.mp = L.mend = nullptr;
    L.fp = L.fend = nullptr;
    L.overflow = 0;
    L.xq_on = 0;
    L.xq_k = 0;
    L.decisions = 0;
    L.term = 0;
    L.mword = 0;
    L.mbase = 0;
    L.commit = L.head = 0;   // Chain::new on an empty tree: genesis block 0
    L.id_gen = 1;
    L.run_hi = 0;
    L.heartbeat_time = 0;
    L.voted_for = L.leader_id = L.queued = L.votes = 0;
    L.rng_draws = 0;
    uint32_t s = self_slots ? self_slots[g] : 0;
    L.flags = JG_ROLE_FOLLOWER | (s << JGF_SELF_SHIFT);
    jg_set_election_timeout(d, L);  // follower.rs:93-95 at now = 0
    d.mlag[g] = 0;  // every progress head 0 = lag 0 below head 0
    jg_store(d, L);
  }
}
