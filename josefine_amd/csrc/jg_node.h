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
//                    lead, a Heartbeat / AppendEntries for one it does lead, an AppendEntries that
//                    arrived before the Heartbeat: the whole group takes the general path, whose
//                    rows keep their stream order); every row also leaves its ARRIVAL INDEX with the
//                    mailbox entry it fills
//   k_node_route     rows of column-form groups are scattered into the inbox columns; the others
//                    are flagged for the general path (k_apply_rows, before the dense halves)
//
// Arrival order (server.rs:120-161: the event loop applies what its channels deliver one command at a
// time).  The dense halves must be indistinguishable from that, so what a mailbox column forgets - the
// order of the rows - is kept where it can matter:
//   * an AppendResponse that arrived BEFORE the group's ClientRequest met the chain head before the
//     append: the own slot's word carries one bit per slot for them (JGN_PRE_SHIFT); the lag-space tick
//     evaluates the majority once over those acknowledgements alone (the commit index the Notify is
//     preceded by on fsm_tx, and the bound chain.rs:197-202 holds them to) and once over all;
//   * everything the lag-space tick does not serve (a HeartbeatResponse without the commit: replicate()
//     on the progress as it is THEN, leader.rs:222-231; escaped fields; forged acks) is replayed by
//     k_dense_slow one command at a time in arrival order, from the indices in `arr`;
//   * a follower's answer word holds HeartbeatResponse, AppendResponse in that order: a group whose
//     AppendEntries arrived first is a general-path group.
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
// own slot's answer word of a node step: JG_ANSWER(#ClientRequests (0 or 1) | pre << 32, JG_HB_NONE), pre = one bit per
// slot whose AppendResponse arrived before the ClientRequest
#define JGN_PRE_SHIFT (8 + JG_NODE_PRE_SHIFT)

// fsm delta word a dense half leaves per group (k_node_fsm_build turns it into rows)
#define JGN_FSM_APPENDED JG_FSM_APPENDED_BIT  // leader: one block was appended (Notify)
#define JGN_FSM_WIDE JG_FSM_WIDE_BIT          // the commit index before the step is in fsm_prev[g] (else: commit_after - low bits)
#define JGN_FSM_FOLLOWER JG_FSM_FOLLOWER_BIT  // the range is a follower's: range(prev..commit), follower.rs:204
#define JGN_FSM_ADV_MASK JG_FSM_ADV_MASK       // bits 0-13: how far the commit index moved in the step; bits 14-27: ... before the Notify

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
  // arrival index + 1 of the row that filled a mailbox entry (stream order of the step's batch)
  uint32_t* arr;         // [2R][G]  [r]: AppendResponse of slot r (own slot: the ClientRequest), [R + r]: HeartbeatResponse of slot r
  uint32_t* fo;          // [2][G]   the Heartbeat, the AppendEntries
  // bit g: group g's rows take the general path (the verdict of k_node_route, one word per 64 groups): what the dense
  // halves of an ASYNCHRONOUS step skip - and come back for, once the host has seen that the general path is not empty
  uint64_t* sparse_bits;  // [ceil(G / 64)]
  // fsm deltas of the dense halves
  uint32_t* fsm_delta;   // [G]
  uint64_t* fsm_prev;    // [G] JGN_FSM_WIDE: the commit index before the step
  uint64_t* fsm_mid;     // [G] JGN_FSM_WIDE, leader: ... and when the ClientRequest was applied
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
  // JG_COL_PACKED_KIND: kind[i] = kind | sender slot << 4 | flag << 7 (no from / flag columns); ids = jg_config.node_ids
  uint32_t packed;
  uint32_t id32;  // JG_COL_ID32: the id column holds 32-bit values (zero-extended here)
  uint32_t ids[JG_MAX_REPLICAS];
  __device__ __forceinline__ uint64_t id_of(uint32_t i) const { return id32 ? (uint64_t)((const uint32_t*)id)[i] : id[i]; }
  __device__ __forceinline__ uint32_t kind_of(uint32_t i) const { return packed ? kind[i] & 15u : kind[i]; }
  __device__ __forceinline__ uint32_t from_of(uint32_t i) const {
    if (!packed) return from ? from[i] : 0u;
    const uint32_t b = kind[i];  // (VoteRequest ... HeartbeatResponse carry a sender: kinds 2-7)
    if (!((0xfcu >> (b & 15u)) & 1u)) return 0u;
    const uint32_t slot = (b >> 4) & 7u;
    uint32_t id = 0;  // (a select per slot, not an indexed read of a kernel argument: that would go through scratch)
#pragma unroll
    for (uint32_t r = 0; r < JG_MAX_REPLICAS; r++) id = slot == r ? ids[r] : id;
    return id;
  }
  __device__ __forceinline__ uint64_t term_of(uint32_t i) const { return term ? term[i] : 0ull; }
  __device__ __forceinline__ uint64_t aux_of(uint32_t i) const { return aux ? aux[i] : 0ull; }
  __device__ __forceinline__ uint32_t flag_of(uint32_t i) const { return packed ? kind[i] >> 7 : (flag ? flag[i] : 0u); }
};

// `col_mask`: member slots whose answers arrived as a column (jg_node_inbox_columns: already copied into
// c.answers): left alone - except where the slot is the group's own, whose word always carries the append count
__global__ __launch_bounds__(JG_BLOCK) void k_node_prefill(JgDev d, JgNodeCols c, int us, uint32_t leader_half,
                                                           uint32_t follower_half, uint32_t both_beats, uint32_t col_mask) {
  const uint32_t G = d.G;
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < G; g += gridDim.x * JG_BLOCK) {
    c.cls[g] = 0;
    c.fsm_delta[g] = 0;
    if ((g & 63u) == 0) c.sparse_bits[g >> 6] = 0;
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
    const uint32_t kind = a.kind_of(i);
    if (g >= d.G || kind >= JG_CMD__COUNT) {  // (rows committed with JG_COL_UNCHECKED are validated here: not applied, JG_EINVAL)
      *d.err = 7;
      continue;
    }
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
                 (kind == JG_CMD_APPEND_RESPONSE && a.id_of(i) >= JG_MAILBOX_NONE);
        bit = s < 0 ? 0u : 1u << ((kind == JG_CMD_APPEND_RESPONSE ? JGN_ACK_SHIFT : JGN_HBR_SHIFT) + (uint32_t)s);
        if (!sparse) c.arr[(size_t)((kind == JG_CMD_APPEND_RESPONSE ? 0u : d.R) + (uint32_t)s) * d.G + g] = i + 1u;
        break;
      }
      case JG_CMD_CLIENT_REQUEST: {
        // only a healthy leader appends (leader.rs:177-197); everybody else forwards or queues the
        // request (follower.rs:258-270, candidate.rs:190-193): rows, the general path
        const uint32_t f = d.flags[g];
        sparse = !(halves & 1u) || (f & JGF_ROLE_MASK) != JG_ROLE_LEADER;
        bit = JGN_CR;
        if (!sparse) c.arr[(size_t)(us >= 0 ? (uint32_t)us : (f & JGF_SELF_MASK) >> JGF_SELF_SHIFT) * d.G + g] = i + 1u;
        break;
      }
      // (a leader's answer to a Heartbeat / AppendEntries is a role change or nothing, leader.rs:200-208,263 - never an
      //  answer word, and its Tick must come AFTER the row: the general path)
      case JG_CMD_HEARTBEAT:
        sparse = !(halves & 2u) || a.id_of(i) == JG_NO_ACK || a.from_of(i) == 0 ||  // (JG_NO_ACK in the beat means "no heartbeat")
                 (d.flags[g] & JGF_ROLE_MASK) == JG_ROLE_LEADER;
        bit = JGN_HB;
        c.fo[g] = i + 1u;
        break;
      case JG_CMD_APPEND_ENTRIES: {
        uint64_t from;
        sparse = !(halves & 2u) || a.from_of(i) == 0 || !jg_node_ae_run(a, a.id_of(i), a.aux_of(i), &from) ||
                 (d.flags[g] & JGF_ROLE_MASK) == JG_ROLE_LEADER;
        bit = JGN_AE;
        c.fo[d.G + g] = i + 1u;
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
__device__ __forceinline__ bool jg_node_group_sparse(const JgNodeCols& c, uint32_t G, uint32_t g, uint32_t w, uint32_t both_beats) {
  if (w & JGN_SPARSE) return true;
  // one beat word carries the term and the sender of both rows, one answer word the two responses in the order
  // HeartbeatResponse, AppendResponse: an AppendEntries that arrived BEFORE the Heartbeat is answered the other way round
  if (both_beats && (w & (JGN_HB | JGN_AE)) == (JGN_HB | JGN_AE))
    return c.lt_max[g] != c.lt_min[g] || c.lf_max[g] != c.lf_min[g] || c.fo[G + g] < c.fo[g];
  return false;
}

// Scatter the rows of column-form groups into the inbox columns; list the others for the general path.
// General-path rows are appended to a list as (group << 32 | arrival index, arrival index) pairs - in no particular
// order: the key says where a row belongs (group-major, a group's rows in the order they arrived) and the bucket
// ordering of jg_route.h puts it there; *n_sparse counts them.
__global__ __launch_bounds__(JG_BLOCK) void k_node_route(JgDev d, JgNodeCols c, JgNodeRows a, int us, uint32_t both_beats,
                                                         uint64_t* __restrict__ sp_key, uint32_t* __restrict__ sp_idx,
                                                         uint32_t* __restrict__ n_sparse) {
  const uint32_t G = d.G;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n_up = (a.n + 63u) & ~63u;  // (whole waves take every iteration together: the appends are wave-aggregated)
  for (uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x; i < n_up; i += gridDim.x * JG_BLOCK) {
    const bool in = i < a.n;
    const uint32_t g = in ? a.group[i] : 0u;
    const bool valid = in && g < G && a.kind_of(i) < JG_CMD__COUNT;  // (an invalid row: reported by k_node_classify, not applied)
    const uint32_t w = valid ? c.cls[g] : 0u;
    const bool sparse = valid && jg_node_group_sparse(c, G, g, w, both_beats);
    const uint64_t m = __ballot(sparse);
    if (m) {
      uint32_t base = 0;
      if (lane == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(n_sparse, (uint32_t)__popcll(m));
      base = __shfl(base, __ffsll((long long)m) - 1, 64);
      if (sparse) {
        const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        sp_key[at] = (uint64_t)g << 32 | i;
        sp_idx[at] = i;
        (void)__hip_atomic_fetch_or(&c.sparse_bits[g >> 6], 1ull << (g & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (!valid || sparse) continue;
    const uint32_t kind = a.kind_of(i);
    switch (kind) {
      case JG_CMD_APPEND_RESPONSE: {  // bits 63..8 of the sender's answer word (all ones before)
        const int s = jg_node_slot_of(d, a.from_of(i));
        (void)__hip_atomic_fetch_and(&c.answers[(size_t)s * G + g], (a.id_of(i) << 8) | 0xffull, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
        if (w & JGN_CR) {  // did it arrive before the group's ClientRequest?  (it met the head before the append)
          const uint32_t self = us >= 0 ? (uint32_t)us : (d.flags[g] & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
          if (i + 1u < c.arr[(size_t)self * G + g])
            (void)__hip_atomic_fetch_or(&c.answers[(size_t)self * G + g], 1ull << (JGN_PRE_SHIFT + (uint32_t)s), __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
        }
        break;
      }
      case JG_CMD_HEARTBEAT_RESPONSE: {  // low byte of the same word
        const int s = jg_node_slot_of(d, a.from_of(i));
        const uint64_t has = a.flag_of(i) ? 1 : 0;
        (void)__hip_atomic_fetch_and(&c.answers[(size_t)s * G + g], ~0xffull | has, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
        if (!has) c.hbr_commit[(size_t)s * G + g] = a.id_of(i);
        break;
      }
      case JG_CMD_CLIENT_REQUEST: {
        const uint32_t self = us >= 0 ? (uint32_t)us : (d.flags[g] & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
        // (an atomic: the AppendResponse rows that arrived before this one set their bits in the same word)
        (void)__hip_atomic_fetch_or(&c.answers[(size_t)self * G + g], 1ull << 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.token[g] = a.id_of(i);
        break;
      }
      case JG_CMD_HEARTBEAT:
        c.f_beat[g] = jg_leader_beat{a.term_of(i), a.id_of(i)};
        c.f_leader[g] = a.from_of(i);
        break;
      default: {  // JG_CMD_APPEND_ENTRIES
        uint64_t from = 0;
        (void)jg_node_ae_run(a, a.id_of(i), a.aux_of(i), &from);
        c.f_ae[g] = JG_AE(from, a.aux_of(i));
        if (!(w & JGN_HB)) {  // (with a Heartbeat in the batch: the same term and sender, written by its row)
          c.f_beat[g].term = a.term_of(i);
          c.f_leader[g] = a.from_of(i);
        }
      }
    }
  }
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
  o.kind[p] = (uint8_t)a.kind_of(i);
  o.from[p] = a.from_of(i);
  o.term[p] = a.term_of(i);
  o.id[p] = a.id_of(i);
  o.aux[p] = a.aux_of(i);
  o.flag[p] = a.flag_of(i);
}

// ---- fsm_tx rows of the dense halves -------------------------------------------------------------
// A dense half leaves one word per group (JGN_FSM_*): what the reference pushed on fsm_tx while the
// group's rows and Tick were applied is fully determined by it and the state after the step:
//   leader    Apply for range(commit_before..=commit_mid).skip(1)    what the AppendResponses that arrived BEFORE the
//                                                                    ClientRequest committed (leader.rs:93)
//             Notify{block_id = head_after, id = token}              if a block was appended (leader.rs:184-188)
//             Apply for range(commit_mid..=commit_after).skip(1)     the self-ack and the AppendResponses after it
//             (consecutive ranges concatenate exactly: match[] only grows)
//   follower  Apply for range(commit_before..commit_after)           if the commit index moved (follower.rs:204)
// Rows go to a [G][3] region with per-group counts and the drain's tile sums: from there on the
// ordinary drain machinery (scan + gather) delivers them, in step order with everything else.
#define JGN_FSM_ROWS 3
__device__ __forceinline__ uint64_t jg_node_commit_of(const JgDev& d, uint32_t g, uint32_t f, uint64_t head) {
  if ((f & JGF_ROLE_MASK) != JG_ROLE_LEADER) return d.commit[g];
  const uint64_t fc = jg_lag_field(d.mlag[g], d.R, d.R);
  if (jg_lag_wide(fc, d.R)) return d.commit[g];
  return (jg_lag_base_is_run_hi(f) ? d.run_hi[g] : head) - fc;  // (a restarted leader: lags below the top of its run)
}
__device__ __forceinline__ void jg_node_fsm_row(jg_fsm_row& r, uint32_t g, uint32_t kind, uint64_t a, uint64_t b) {
  r.group = g, r.kind = (uint8_t)kind, r.pad[0] = r.pad[1] = r.pad[2] = 0;
  r.a = a, r.b = b;
}
// `fused` (JG_NODE_FSM_FUSED): a leader's rows of a step that appended, within 255 of the new block, are ONE
// JG_FSM_LEADER_STEP row (josefine_gpu.h): the steady state pays 24 bytes per partition on the bus, not 48.
__global__ __launch_bounds__(JG_BLOCK) void k_node_fsm_build(JgDev d, JgNodeCols c, jg_fsm_row* __restrict__ out,
                                                             uint32_t* __restrict__ cnt, uint64_t* __restrict__ bsum, uint32_t fused) {
  const uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x;  // one tile of JG_SCAN_TILE (= JG_BLOCK) groups per workgroup
  uint32_t n = 0;
  if (g < d.G) {
    const uint32_t w = c.fsm_delta[g];
    if (w) {
      const uint32_t f = d.flags[g];
      const uint64_t head = d.head[g];
      const uint64_t commit1 = jg_node_commit_of(d, g, f, head);
      const bool fol = (w & JGN_FSM_FOLLOWER) != 0;
      uint64_t commit0, mid;
      if (w & JGN_FSM_WIDE) {
        commit0 = c.fsm_prev[g];
        mid = fol ? commit0 : c.fsm_mid[g];
      } else {
        commit0 = commit1 - (w & JGN_FSM_ADV_MASK);
        mid = commit0 + ((w >> JG_FSM_PRE_SHIFT) & JGN_FSM_ADV_MASK);
      }
      jg_fsm_row* r = out + (size_t)g * JGN_FSM_ROWS;
      if (fol) {
        if (commit1 != commit0) jg_node_fsm_row(r[n++], g, JG_FSM_APPLY_FOLLOWER, commit0, commit1);
      } else if (fused && (w & JGN_FSM_APPENDED) && head >= commit1 && head - commit0 <= 255u && commit0 <= mid && mid <= commit1) {
        jg_fsm_row& o = r[n++];
        o.group = g, o.kind = (uint8_t)JG_FSM_LEADER_STEP;
        o.pad[0] = (uint8_t)(head - commit0), o.pad[1] = (uint8_t)(head - mid), o.pad[2] = (uint8_t)(head - commit1);
        o.a = head, o.b = c.token[g];
      } else {
        if (mid != commit0) jg_node_fsm_row(r[n++], g, JG_FSM_APPLY_LEADER, commit0, mid);
        if (w & JGN_FSM_APPENDED) jg_node_fsm_row(r[n++], g, JG_FSM_NOTIFY, head, c.token[g]);
        if (commit1 != mid) jg_node_fsm_row(r[n++], g, JG_FSM_APPLY_LEADER, mid, commit1);
      }
    }
    cnt[g] = n;
  }
  uint32_t tot;
  (void)jg_block_exclusive_scan(n, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// JG_NODE_COMMON_AE: how many partitions' AppendEntries words differ by addressee (their rows of `ae` are wanted on the host)
__global__ __launch_bounds__(JG_BLOCK) void k_node_count_individual(const uint64_t* __restrict__ aec, uint32_t G, uint32_t* __restrict__ count) {
  uint32_t n = 0;
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < G; g += gridDim.x * JG_BLOCK) n += aec[g] == JG_AEC_INDIVIDUAL;
  uint32_t tot;
  (void)jg_block_exclusive_scan(n, &tot);
  if (threadIdx.x == 0 && tot) atomicAdd(count, tot);
}
