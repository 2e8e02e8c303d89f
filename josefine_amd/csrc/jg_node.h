// jg_node.h — a node's whole tick from HOST ROWS: the dense kernels behind the Apply surface
// (jg_step_node; SURVEY.md §8(f) rank 2: server::event_loop for many partitions).
//
// The reference's one caller, event_loop (src/raft/server.rs:103-165), hands Raft<T> one Command
// at a time: a Tick every 100 ms, whatever tcp_rx / client_rx delivered in between.  For a node
// that hosts G partitions almost all of that traffic is the steady state: AppendResponse and
// HeartbeatResponse rows for the partitions it leads (leader.rs:211-231), ClientRequests
// (leader.rs:177-197), Heartbeat and AppendEntries rows for the ones it follows
// (follower.rs:130-217) — exactly the vocabulary of the dense mailbox columns (josefine_gpu.h,
// "dense node tick").  The kernels here turn an UNSORTED batch of command rows into those
// columns on the device:
//
//   k_node_prefill   the inbox columns start as "nothing from anybody" (own slot: zero appends)
//   k_node_classify  one atomicOr per row into the group's class word: which mailbox entries the
//                    step's rows fill, and whether the group can be served in column form at all
//                    (a kind outside the vocabulary, two rows for one mailbox entry, a value a
//                    mailbox word cannot hold, a ClientRequest for a group this node does not
//                    lead: the whole group takes the general path, so that its rows keep their
//                    stream order)
//   k_node_route     rows of column-form groups are scattered into the inbox columns; the others
//                    are flagged for the general path (k_apply_rows, before the dense halves)
//   k_node_fsm_build the fsm_tx rows of the dense halves (Instruction::Notify / Apply, fsm.rs:20-29),
//                    from the per-group deltas the tick kernels leave behind
//
// Nothing here interprets Raft: the arithmetic stays in jg_dense.h / jg_follower.h / jg_device.h.
#pragma once
#include "jg_device.h"
#include "jg_sparse.h"

// class word of a group for one node step
#define JGN_ACK_SHIFT 0            // bits 0-7:  an AppendResponse from slot r is in the batch
#define JGN_HBR_SHIFT 8            // bits 8-15: a HeartbeatResponse from slot r
#define JGN_HB (1u << 16)          // a Heartbeat
#define JGN_AE (1u << 17)          // an AppendEntries
#define JGN_CR (1u << 18)          // a ClientRequest
#define JGN_SPARSE (1u << 31)      // the group's rows take the general path (k_apply_rows)

// fsm delta word a dense half leaves per group (k_node_fsm_build turns it into rows)
#define JGN_FSM_APPENDED JG_FSM_APPENDED_BIT  // leader: one block was appended (Notify)
#define JGN_FSM_WIDE JG_FSM_WIDE_BIT          // the commit index before the step is in fsm_prev[g] (else: commit_after - low bits)
#define JGN_FSM_FOLLOWER JG_FSM_FOLLOWER_BIT  // the range is a follower's: range(prev..commit), follower.rs:204
#define JGN_FSM_ADV_MASK (JG_FSM_FOLLOWER_BIT - 1u)

struct JgNodeCols {  // device scratch of the node step (engine-owned, grow-only)
  // leader half inbox
  uint64_t* answers;     // [R][G] JG_ANSWER words (own slot: number of appends)
  uint64_t* hbr_commit;  // [R][G] HeartbeatResponse.commit where has_committed == 0
  uint64_t* token;       // [G] request token of the group's ClientRequest (Notify.id)
  // follower half inbox
  jg_leader_beat* f_beat;  // [G]
  uint64_t* f_ae;          // [G]
  uint32_t* f_leader;      // [G] sender NodeId
  // classification
  uint32_t* cls;           // [G]
  uint64_t *lt_max, *lt_min;  // [G] max / min term over the group's Heartbeat + AppendEntries rows
  uint32_t *lf_max, *lf_min;  // [G] ... and sender
  // fsm deltas of the dense halves
  uint32_t* fsm_delta;   // [G]
  uint64_t* fsm_prev;    // [G]
};

struct JgNodeRows {  // the step's command rows in device memory, unsorted (stream order)
  uint32_t n;
  const uint32_t* group;
  const uint8_t* kind;
  const uint32_t* from;  // null: all zeros (as are term, aux, flag)
  const uint64_t* term;
  const uint64_t* id;
  const uint64_t* aux;
  const uint8_t* flag;
  const uint64_t* blk_id;
  const uint64_t* blk_next;
  uint64_t n_blocks;
  __device__ __forceinline__ uint32_t from_of(uint32_t i) const { return from ? from[i] : 0u; }
  __device__ __forceinline__ uint64_t term_of(uint32_t i) const { return term ? term[i] : 0ull; }
  __device__ __forceinline__ uint64_t aux_of(uint32_t i) const { return aux ? aux[i] : 0ull; }
  __device__ __forceinline__ uint32_t flag_of(uint32_t i) const { return flag ? flag[i] : 0u; }
};

// `col_mask`: member slots whose answers arrived as a column (jg_node_inbox_columns: already copied into
// c.answers): left alone - except where the slot is the group's own, whose word always carries the append count
__global__ __launch_bounds__(JG_BLOCK) void k_node_prefill(JgDev d, JgNodeCols c, int us, uint32_t leader_half,
                                                           uint32_t follower_half, uint32_t both_beats, uint32_t col_mask) {
  const uint32_t G = d.G;
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < G; g += gridDim.x * JG_BLOCK) {
    c.cls[g] = 0;
    c.fsm_delta[g] = 0;
    if (leader_half) {
      const uint32_t s = us >= 0 ? (uint32_t)us : (d.flags[g] & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
      for (uint32_t r = 0; r < d.R; r++) {
        if (r == s) c.answers[(size_t)r * G + g] = JG_ANSWER(0, JG_HB_NONE);
        else if (!((col_mask >> r) & 1u)) c.answers[(size_t)r * G + g] = JG_NO_ACK;
      }
    }
    if (follower_half) {
      c.f_beat[g] = jg_leader_beat{0, JG_NO_ACK};
      c.f_ae[g] = JG_NO_ACK;
      c.f_leader[g] = 0;
      if (both_beats) {  // (only a batch that holds Heartbeat AND AppendEntries rows needs the consistency columns)
        c.lt_max[g] = 0, c.lt_min[g] = ~0ull;
        c.lf_max[g] = 0, c.lf_min[g] = ~0u;
      }
    }
  }
}

__device__ __forceinline__ int jg_node_slot_of(const JgDev& d, uint32_t id) {
  int s = -1;
#pragma unroll
  for (uint32_t r = 0; r < JG_MAX_REPLICAS; r++) s = (r < d.R && d.node_ids[r] == id) ? (int)r : s;
  return s;
}

// Is AppendEntries row (first index, n) of the side arrays the run (from, from + n]: ids consecutive,
// every block's parent its predecessor - what one JG_AE word stands for?
__device__ __forceinline__ bool jg_node_ae_run(const JgNodeRows& a, uint64_t first, uint64_t n, uint64_t* from) {
  if (n > 0xfeu) return false;  // the count is one byte of the word (JG_AE_NONE = 0xff)
  if (n == 0) {
    *from = 0;
    return true;
  }
  if (n > a.n_blocks || first > a.n_blocks - n) return false;  // (jg_submit has rejected this already)
  const uint64_t id0 = a.blk_id[first];
  if (id0 == 0 || id0 - 1 + n >= JG_MAILBOX_NONE) return false;  // 56-bit ids in mailbox words; block 0 is genesis
  for (uint64_t k = 0; k < n; k++)
    if (a.blk_id[first + k] != id0 + k || a.blk_next[first + k] != id0 + k - 1) return false;
  *from = id0 - 1;
  return true;
}

__global__ __launch_bounds__(JG_BLOCK) void k_node_classify(JgDev d, JgNodeCols c, JgNodeRows a, int us, uint32_t halves,
                                                            uint32_t both_beats, uint32_t col_mask) {
  for (uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x; i < a.n; i += gridDim.x * JG_BLOCK) {
    const uint32_t g = a.group[i];
    const uint32_t kind = a.kind[i];
    uint32_t bit = 0;
    bool sparse = false;
    switch (kind) {
      case JG_CMD_APPEND_RESPONSE:
      case JG_CMD_HEARTBEAT_RESPONSE: {
        const uint32_t f = d.flags[g];
        const uint32_t self = us >= 0 ? (uint32_t)us : (f & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
        const int s = jg_node_slot_of(d, a.from_of(i));
        if (s >= 0 && ((col_mask >> s) & 1u)) *d.err = 6;  // this sender's answers arrived as a column: rows AND a column in one tick
        // a sender outside the membership (progress.rs:43 panics on it), the own id (the own slot of the
        // inbox block carries the number of appends), a head a mailbox word cannot hold: general path
        sparse = !(halves & 1u) || s < 0 || (uint32_t)s == self ||
                 (kind == JG_CMD_APPEND_RESPONSE && a.id[i] >= JG_MAILBOX_NONE);
        bit = s < 0 ? 0u : 1u << ((kind == JG_CMD_APPEND_RESPONSE ? JGN_ACK_SHIFT : JGN_HBR_SHIFT) + (uint32_t)s);
        break;
      }
      case JG_CMD_CLIENT_REQUEST: {
        // only a healthy leader appends (leader.rs:177-197); everybody else forwards or queues the
        // request (follower.rs:258-270, candidate.rs:190-193): rows, the general path
        const uint32_t f = d.flags[g];
        sparse = !(halves & 1u) || (f & JGF_ROLE_MASK) != JG_ROLE_LEADER;
        bit = JGN_CR;
        break;
      }
      case JG_CMD_HEARTBEAT:
        sparse = !(halves & 2u) || a.id[i] == JG_NO_ACK || a.from_of(i) == 0;  // (JG_NO_ACK in the beat means "no heartbeat")
        bit = JGN_HB;
        break;
      case JG_CMD_APPEND_ENTRIES: {
        uint64_t from;
        sparse = !(halves & 2u) || a.from_of(i) == 0 || !jg_node_ae_run(a, a.id[i], a.aux_of(i), &from);
        bit = JGN_AE;
        break;
      }
      default: sparse = true;  // votes, Timeout, Restart, explicit Tick rows, ...: the general state machine
    }
    if (both_beats && (kind == JG_CMD_HEARTBEAT || kind == JG_CMD_APPEND_ENTRIES)) {
      // one beat word carries the term and the sender of both: they must agree (decided in k_node_route)
      atomicMax((unsigned long long*)&c.lt_max[g], (unsigned long long)a.term_of(i));
      atomicMin((unsigned long long*)&c.lt_min[g], (unsigned long long)a.term_of(i));
      atomicMax(&c.lf_max[g], a.from_of(i));
      atomicMin(&c.lf_min[g], a.from_of(i));
    }
    const uint32_t old = atomicOr(&c.cls[g], bit | (sparse ? JGN_SPARSE : 0u));
    if ((old & bit) && !sparse) atomicOr(&c.cls[g], JGN_SPARSE);  // a second row for the same mailbox entry
  }
}

// final verdict on a group (every row of the group evaluates the same data: no ordering between rows)
__device__ __forceinline__ bool jg_node_group_sparse(const JgNodeCols& c, uint32_t g, uint32_t w, uint32_t both_beats) {
  if (w & JGN_SPARSE) return true;
  if (both_beats && (w & (JGN_HB | JGN_AE)) == (JGN_HB | JGN_AE))
    return c.lt_max[g] != c.lt_min[g] || c.lf_max[g] != c.lf_min[g];
  return false;
}

// Scatter the rows of column-form groups into the inbox columns; flag the others (keep[i] = 1) and
// count them: *n_sparse, one atomic per workgroup.
__global__ __launch_bounds__(JG_BLOCK) void k_node_route(JgDev d, JgNodeCols c, JgNodeRows a, int us, uint32_t both_beats,
                                                         uint8_t* __restrict__ keep, uint32_t* __restrict__ n_sparse) {
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const uint32_t G = d.G;
  uint32_t mine = 0;
  for (uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x; i < a.n; i += gridDim.x * JG_BLOCK) {
    const uint32_t g = a.group[i];
    const uint32_t w = c.cls[g];
    const bool sparse = jg_node_group_sparse(c, g, w, both_beats);
    keep[i] = sparse ? 1 : 0;
    mine += sparse;
    if (sparse) continue;
    const uint32_t kind = a.kind[i];
    switch (kind) {
      case JG_CMD_APPEND_RESPONSE: {  // bits 63..8 of the sender's answer word (all ones before)
        const int s = jg_node_slot_of(d, a.from_of(i));
        (void)__hip_atomic_fetch_and(&c.answers[(size_t)s * G + g], (a.id[i] << 8) | 0xffull, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      case JG_CMD_HEARTBEAT_RESPONSE: {  // low byte of the same word
        const int s = jg_node_slot_of(d, a.from_of(i));
        const uint64_t has = a.flag_of(i) ? 1 : 0;
        (void)__hip_atomic_fetch_and(&c.answers[(size_t)s * G + g], ~0xffull | has, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
        if (!has) c.hbr_commit[(size_t)s * G + g] = a.id[i];
        break;
      }
      case JG_CMD_CLIENT_REQUEST: {
        const uint32_t self = us >= 0 ? (uint32_t)us : (d.flags[g] & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
        c.answers[(size_t)self * G + g] = JG_ANSWER(1, JG_HB_NONE);
        c.token[g] = a.id[i];
        break;
      }
      case JG_CMD_HEARTBEAT:
        c.f_beat[g] = jg_leader_beat{a.term_of(i), a.id[i]};
        c.f_leader[g] = a.from_of(i);
        break;
      default: {  // JG_CMD_APPEND_ENTRIES
        uint64_t from = 0;
        (void)jg_node_ae_run(a, a.id[i], a.aux_of(i), &from);
        c.f_ae[g] = JG_AE(from, a.aux_of(i));
        if (!(w & JGN_HB)) {  // (with a Heartbeat in the batch: the same term and sender, written by its row)
          c.f_beat[g].term = a.term_of(i);
          c.f_leader[g] = a.from_of(i);
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off; off >>= 1) mine += __shfl_down(mine, off, 64);
  if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(&s_cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) atomicAdd(n_sparse, s_cnt);
}

// keep-flagged rows, already compacted (stream order) and sorted by group (stable): index list -> the
// command columns k_apply_rows consumes
struct JgNodeSorted {
  uint32_t* group;
  uint8_t* kind;
  uint32_t* from;
  uint64_t* term;
  uint64_t* id;
  uint64_t* aux;
  uint8_t* flag;
};
__global__ __launch_bounds__(JG_BLOCK) void k_node_gather_rows(uint32_t n, const uint32_t* __restrict__ order, JgNodeRows a,
                                                               JgNodeSorted o) {
  const uint32_t p = blockIdx.x * JG_BLOCK + threadIdx.x;
  if (p >= n) return;
  const uint32_t i = order[p];
  o.group[p] = a.group[i];
  o.kind[p] = a.kind[i];
  o.from[p] = a.from_of(i);
  o.term[p] = a.term_of(i);
  o.id[p] = a.id[i];
  o.aux[p] = a.aux_of(i);
  o.flag[p] = a.flag_of(i);
}
__global__ __launch_bounds__(JG_BLOCK) void k_node_keys(uint32_t n, const uint32_t* __restrict__ idx,
                                                        const uint32_t* __restrict__ group, uint32_t* __restrict__ keys) {
  const uint32_t p = blockIdx.x * JG_BLOCK + threadIdx.x;
  if (p < n) keys[p] = group[idx[p]];
}

// ---- fsm_tx rows of the dense halves -------------------------------------------------------------
// A dense half leaves one word per group (JGN_FSM_*): what the reference pushed on fsm_tx while the
// group's tick was applied is fully determined by it and the state after the step:
//   leader    Notify{block_id = head_after, id = token}           if a block was appended (leader.rs:184-188)
//             Apply for range(commit_before..=commit_after).skip(1)   if the commit index moved (leader.rs:93;
//             consecutive ranges of one tick concatenate exactly: match[] only grows)
//   follower  Apply for range(commit_before..commit_after)        if the commit index moved (follower.rs:204)
// Rows go to a [G][2] region with per-group counts and the drain's tile sums: from there on the
// ordinary drain machinery (scan + gather) delivers them, in step order with everything else.
__device__ __forceinline__ uint64_t jg_node_commit_of(const JgDev& d, uint32_t g, uint32_t f, uint64_t head) {
  if ((f & JGF_ROLE_MASK) != JG_ROLE_LEADER) return d.commit[g];
  const uint64_t fc = jg_lag_field(d.mlag[g], d.R, d.R);
  return jg_lag_wide(fc, d.R) ? d.commit[g] : head - fc;
}
__global__ __launch_bounds__(JG_BLOCK) void k_node_fsm_build(JgDev d, JgNodeCols c, jg_fsm_row* __restrict__ out,
                                                             uint32_t* __restrict__ cnt, uint64_t* __restrict__ bsum) {
  const uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x;  // one tile of JG_SCAN_TILE (= JG_BLOCK) groups per workgroup
  uint32_t n = 0;
  if (g < d.G) {
    const uint32_t w = c.fsm_delta[g];
    if (w) {
      const uint32_t f = d.flags[g];
      const uint64_t head = d.head[g];
      const uint64_t commit1 = jg_node_commit_of(d, g, f, head);
      const uint64_t commit0 = (w & JGN_FSM_WIDE) ? c.fsm_prev[g] : commit1 - (w & JGN_FSM_ADV_MASK);
      jg_fsm_row* r = out + (size_t)g * 2;
      if (w & JGN_FSM_APPENDED) {
        r[n].group = g, r[n].kind = JG_FSM_NOTIFY, r[n].pad[0] = r[n].pad[1] = r[n].pad[2] = 0;
        r[n].a = head, r[n].b = c.token[g];
        n++;
      }
      if (commit1 != commit0) {
        r[n].group = g, r[n].kind = (w & JGN_FSM_FOLLOWER) ? JG_FSM_APPLY_FOLLOWER : JG_FSM_APPLY_LEADER;
        r[n].pad[0] = r[n].pad[1] = r[n].pad[2] = 0;
        r[n].a = commit0, r[n].b = commit1;
        n++;
      }
    }
    cnt[g] = n;
  }
  uint32_t tot;
  (void)jg_block_exclusive_scan(n, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
