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

// ---- the same two passes, TILED (round 6) ---------------------------------------------------------------------------
// k_node_classify and k_node_route touch, per 9-byte row, the partition's class word, its arrival-index entry and its
// answer word: three random 4-8-byte accesses into 40 MB columns per row and pass - 126-134 bytes of HBM traffic per row
// (PMC, profiles/r05/event_loop_row_order_and_traffic.txt), 339 + 288 us for the 7 M shuffled rows of a 1 M x 5 tick, 78 %
// of the event loop's kernel time.  Tiled: the rows are first BINNED by tile of JGN_TILE partitions (a counting pass, a
// scan, a scattering pass - per workgroup an LDS table of cursors, no global atomic), then ONE workgroup per tile keeps the
// tile's columns in LDS - the class words, the R answer words, the 2 R arrival indices, the follower half's inbox - applies
// the tile's rows to them (classification, a barrier, the scatter: LDS atomics), and writes the columns back in whole
// lines.  What k_node_prefill wrote is the tile's initial value: that launch is gone too.  Results are the flat passes',
// bit for bit (the binning carries every row's ARRIVAL INDEX: nothing depends on the order of a tile's rows) - the flat
// kernels stay as the statement of it (JG_NODE_FLAT=1; tests/test_node_step.py::test_tiled_row_pass_equals_the_flat_one).
#if JG_BLOCK == 256  // (the kernels: 256-thread workgroups; the one-lane and one-wave host builds of the tests run the flat passes)
#define JGN_TILE_BITS 8u
#define JGN_TILE (1u << JGN_TILE_BITS)  // partitions per tile: one per thread of the tile's workgroup
static_assert(JGN_TILE == JG_BLOCK, "a partition per thread");
#define JGN_BIN_WGS 512u   // row chunks: workgroups of the counting and the scattering pass
#define JGN_BIN_SEGS 8u    // the scan: a thread per (tile, segment of JGN_BIN_WGS / JGN_BIN_SEGS chunks)
// A binned row: what every row has, as ONE 16-byte record - one store per row in the scattering pass, one load per row and
// pass in the tile's kernel (as four columns the pass issued four scattered stores per row: 411 us per 7 M rows, more than
// the flat passes it replaces - profiles/r06/tiled_row_pass.txt).  The optional columns (from, term, aux, flag, the upper
// half of a 64-bit id) stay columns of their own, written only where the step has them: the compact bus has none.
struct JgNodeBinRec {
  uint32_t group, idx, id_lo, kind;  // kind: the row's kind byte as it came (JG_COL_PACKED_KIND: kind | sender slot << 4 | flag << 7)
};
struct JgNodeBin {
  uint32_t n, n_tiles, n_wg, chunk;  // rows; tiles (bin n_tiles: rows whose group is out of range); chunks; rows per chunk
  uint32_t* cnt;       // [n_wg][n_tiles + 1] rows of chunk w in tile t; after the scan: where chunk w's rows of tile t begin WITHIN the tile
  uint32_t* tile_off;  // [n_tiles + 2] first row of tile t in the binned arrays (after the scan); before: the tiles' totals
  uint32_t* done;      // the scan's ticket (zero before a step)
  JgNodeBinRec* rec;   // [n] binned
  uint32_t* id_hi;     // [n] (a 64-bit id column) or null
  JgNodeRows rows;     // the binned copies of the optional columns the step has (from / term / aux / flag), the block side arrays and the formats' switches
};
template <uint32_t CAP>  // CAP >= n_tiles + 1: the LDS table
__global__ __launch_bounds__(JG_BLOCK) void k_node_bin_count(JgNodeRows a, JgNodeBin b, uint32_t G) {
  __shared__ uint32_t s_cnt[CAP];
  const uint32_t nt1 = b.n_tiles + 1u;
  for (uint32_t t = threadIdx.x; t < nt1; t += JG_BLOCK) s_cnt[t] = 0;
  __syncthreads();
  const uint32_t lo = blockIdx.x * b.chunk, hi = min(lo + b.chunk, a.n);  // (lo: a multiple of JG_BLOCK)
  // four rows per thread and trip, one 16-byte load (a 4-byte load per trip left the loop waiting for one round trip to HBM
  // per 256 rows: 30 us for a pass that reads 28 MB)
  for (uint32_t i = lo + threadIdx.x * 4u; i < hi; i += JG_BLOCK * 4u) {
    uint32_t g[4];
    if (i + 4u <= hi) {
      const uint4 v = *(const uint4*)(a.group + i);
      g[0] = v.x, g[1] = v.y, g[2] = v.z, g[3] = v.w;
    } else {
      for (uint32_t k = 0; k < 4u; k++) g[k] = i + k < hi ? a.group[i + k] : 0xffffffffu;
    }
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++)
      if (i + k < hi) atomicAdd(&s_cnt[g[k] < G ? g[k] >> JGN_TILE_BITS : b.n_tiles], 1u);
  }
  __syncthreads();
  uint32_t* out = b.cnt + (size_t)blockIdx.x * nt1;
  for (uint32_t t = threadIdx.x; t < nt1; t += JG_BLOCK) out[t] = s_cnt[t];
}
// counts -> places: thread (tile, segment) sums its chunks' counts of the tile, the segments' sums are scanned in LDS, the
// thread rewrites its chunks' counts as running offsets within the tile; the last workgroup to finish scans the tiles' totals
__global__ __launch_bounds__(JG_BLOCK) void k_node_bin_scan(JgNodeBin b) {
  constexpr uint32_t TL = JG_BLOCK / JGN_BIN_SEGS;  // tiles per workgroup: 32 neighbours = a 128-byte line per chunk
  static_assert(JG_BLOCK % JGN_BIN_SEGS == 0 && JGN_BIN_WGS % JGN_BIN_SEGS == 0, "scan geometry");
  __shared__ uint32_t s_part[JGN_BIN_SEGS][TL];
  const uint32_t nt1 = b.n_tiles + 1u;
  const uint32_t tl = threadIdx.x % TL, seg = threadIdx.x / TL;
  const uint32_t t = blockIdx.x * TL + tl;
  const uint32_t per = (b.n_wg + JGN_BIN_SEGS - 1u) / JGN_BIN_SEGS, w0 = min(seg * per, b.n_wg), w1 = min(w0 + per, b.n_wg);
  // (eight loads in flight per thread: one at a time the two loops were 2 x 64 round trips one behind the other, 39 us)
  constexpr uint32_t U = 8;
  uint32_t sum = 0;
  if (t < nt1)
    for (uint32_t w = w0; w < w1; w += U) {
      uint32_t v[U];
#pragma unroll
      for (uint32_t k = 0; k < U; k++) v[k] = w + k < w1 ? b.cnt[(size_t)(w + k) * nt1 + t] : 0u;
#pragma unroll
      for (uint32_t k = 0; k < U; k++) sum += v[k];
    }
  s_part[seg][tl] = sum;
  __syncthreads();
  uint32_t pre = 0, tot = 0;
  for (uint32_t q = 0; q < JGN_BIN_SEGS; q++) {
    const uint32_t v = s_part[q][tl];
    pre += q < seg ? v : 0u;
    tot += v;
  }
  if (t < nt1) {
    uint32_t run = pre;
    for (uint32_t w = w0; w < w1; w += U) {
      uint32_t v[U];
#pragma unroll
      for (uint32_t k = 0; k < U; k++) v[k] = w + k < w1 ? b.cnt[(size_t)(w + k) * nt1 + t] : 0u;
#pragma unroll
      for (uint32_t k = 0; k < U; k++) {
        if (w + k < w1) b.cnt[(size_t)(w + k) * nt1 + t] = run;
        run += v[k];
      }
    }
    if (seg == 0) (void)atomicExch(&b.tile_off[t], tot);
  }
  __shared__ uint32_t last_s;
  __syncthreads();
  if (threadIdx.x == 0) last_s = __hip_atomic_fetch_add(b.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
  __syncthreads();
  if (!last_s) return;
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base <= nt1; base += JG_BLOCK) {  // exclusive scan of the nt1 totals; entry nt1: all rows
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nt1 ? atomicAdd(&b.tile_off[i], 0u) : 0u;
    uint32_t tt;
    const uint32_t e = carry_s + jg_block_exclusive_scan(v, &tt);
    __syncthreads();
    if (i <= nt1) b.tile_off[i] = e;
    if (threadIdx.x == 0) carry_s += tt;
    __syncthreads();
  }
}
template <uint32_t CAP>
__global__ __launch_bounds__(JG_BLOCK) void k_node_bin_scatter(JgDev d, JgNodeRows a, JgNodeBin b) {
  __shared__ uint32_t s_cur[CAP];
  const uint32_t nt1 = b.n_tiles + 1u;
  const uint32_t* mine = b.cnt + (size_t)blockIdx.x * nt1;
  for (uint32_t t = threadIdx.x; t < nt1; t += JG_BLOCK) s_cur[t] = b.tile_off[t] + mine[t];
  __syncthreads();
  const uint32_t lo = blockIdx.x * b.chunk, hi = min(lo + b.chunk, a.n);
  const JgNodeRows& o = b.rows;
  constexpr uint32_t U = 4;  // rows per thread and trip: their loads first, back to back
  for (uint32_t i0 = lo + threadIdx.x; i0 < hi; i0 += JG_BLOCK * U) {
    uint32_t g[U], kind[U], idl[U];
#pragma unroll
    for (uint32_t k = 0; k < U; k++) {
      const uint32_t i = i0 + k * JG_BLOCK;
      const bool in = i < hi;
      g[k] = in ? a.group[i] : 0xffffffffu;
      kind[k] = in ? a.kind[i] : 0u;
      idl[k] = in ? (a.id32 ? ((const uint32_t*)a.id)[i] : (uint32_t)a.id[i]) : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < U; k++) {
      const uint32_t i = i0 + k * JG_BLOCK;
      if (i >= hi) continue;
      if (g[k] >= d.G) {  // (rows committed with JG_COL_UNCHECKED are validated on the device: not applied, JG_EINVAL)
        *d.err = 7;
        continue;
      }
      const uint32_t at = atomicAdd(&s_cur[g[k] >> JGN_TILE_BITS], 1u);
      *(uint4*)&b.rec[at] = make_uint4(g[k], i, idl[k], kind[k]);
      if (b.id_hi) b.id_hi[at] = (uint32_t)(a.id[i] >> 32);
      if (a.from) ((uint32_t*)o.from)[at] = a.from[i];
      if (a.term) ((uint64_t*)o.term)[at] = a.term[i];
      if (a.aux) ((uint64_t*)o.aux)[at] = a.aux[i];
      if (a.flag) ((uint8_t*)o.flag)[at] = a.flag[i];
    }
  }
}
// a binned row, decoded: what JgNodeRows' accessors say of row k of the binned arrays
struct JgNodeBinRow {
  uint32_t group, idx, kind, from, flag;
  uint64_t id, term, aux;
};
__device__ __forceinline__ JgNodeBinRow jg_node_bin_row(const JgNodeBin& b, uint32_t k) {
  const uint4 v = *(const uint4*)&b.rec[k];
  const JgNodeRows& a = b.rows;
  JgNodeBinRow r;
  r.group = v.x, r.idx = v.y;
  r.id = (uint64_t)v.z | (b.id_hi ? (uint64_t)b.id_hi[k] << 32 : 0ull);
  const uint32_t kb = v.w;
  r.kind = a.packed ? kb & 15u : kb;
  if (a.packed) {
    const uint32_t slot = (kb >> 4) & 7u;
    uint32_t id = 0;  // (a select per slot, not an indexed read of a kernel argument: JgNodeRows::from_of)
#pragma unroll
    for (uint32_t q = 0; q < JG_MAX_REPLICAS; q++) id = slot == q ? a.ids[q] : id;
    r.from = ((0xfcu >> (kb & 15u)) & 1u) ? id : 0u;
    r.flag = (kb >> 7) & 1u;
  } else {
    r.from = a.from ? a.from[k] : 0u;
    r.flag = a.flag ? a.flag[k] : 0u;
  }
  r.term = a.term ? a.term[k] : 0ull;
  r.aux = a.aux ? a.aux[k] : 0ull;
  return r;
}
// the tile's columns in LDS: sized by R (R = 5, both halves: 40 KB; the leader half alone: 24 KB)
template <int R>
struct JgNodeTile {
  uint32_t cls[JGN_TILE], flags[JGN_TILE];
  uint64_t answers[R][JGN_TILE];
  uint32_t arr[2 * R][JGN_TILE];
  uint64_t token[JGN_TILE];
  uint64_t sparse[JGN_TILE / 64];
};
struct JgNodeTileF {  // ... and the follower half's inbox with its consistency columns (16 KB)
  jg_leader_beat f_beat[JGN_TILE];
  uint64_t f_ae[JGN_TILE];
  uint32_t f_leader[JGN_TILE];
  uint32_t fo[2][JGN_TILE];
  uint64_t lt_max[JGN_TILE], lt_min[JGN_TILE];
  uint32_t lf_max[JGN_TILE], lf_min[JGN_TILE];
};
template <int R, bool FOLLOWER>
__global__ __launch_bounds__(JG_BLOCK) void k_node_tile(JgDev d, JgNodeCols c, JgNodeBin b, int us, uint32_t halves, uint32_t both_beats, uint32_t col_mask,
                                                        uint64_t* __restrict__ sp_key, uint32_t* __restrict__ sp_idx, uint32_t* __restrict__ n_sparse) {
  __shared__ JgNodeTile<R> T;
  __shared__ JgNodeTileF F[1];  // (the leader-only instance never touches it: the compiler drops it)
  const uint32_t G = d.G;
  const uint32_t g0 = blockIdx.x << JGN_TILE_BITS, p = threadIdx.x, g = g0 + p;
  const bool live = g < G;
  const bool leader_half = halves & 1u, follower_half = FOLLOWER && (halves & 2u);
  // -- the tile as k_node_prefill leaves a partition: nothing from anybody, the own slot's word = zero appends
  const uint32_t f = live ? d.flags[g] : 0u;
  const uint32_t self = us >= 0 ? (uint32_t)us : (f & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
  T.cls[p] = 0, T.flags[p] = f;
#pragma unroll
  for (int r = 0; r < R; r++) T.answers[r][p] = (uint32_t)r == self ? JG_ANSWER(0, JG_HB_NONE) : JG_NO_ACK;
#pragma unroll
  for (int r = 0; r < 2 * R; r++) T.arr[r][p] = 0;
  T.token[p] = 0;
  if (FOLLOWER) {
    F[0].fo[0][p] = F[0].fo[1][p] = 0;
    F[0].f_beat[p] = jg_leader_beat{0, JG_NO_ACK}, F[0].f_ae[p] = JG_NO_ACK, F[0].f_leader[p] = 0;
    F[0].lt_max[p] = 0, F[0].lt_min[p] = ~0ull, F[0].lf_max[p] = 0, F[0].lf_min[p] = ~0u;
  }
  if (p < JGN_TILE / 64) T.sparse[p] = 0;
  __syncthreads();
  const JgNodeRows& a = b.rows;
  const uint32_t lo = b.tile_off[blockIdx.x], hi = b.tile_off[blockIdx.x + 1];
  // -- k_node_classify over the tile's rows (any order: every row carries its arrival index)
  for (uint32_t k = lo + p; k < hi; k += JG_BLOCK) {
    const JgNodeBinRow row = jg_node_bin_row(b, k);
    const uint32_t q = row.group - g0, kind = row.kind, i = row.idx;
    if (kind >= JG_CMD__COUNT) {
      *d.err = 7;
      continue;
    }
    uint32_t bit = 0;
    bool sparse = false;
    const uint32_t fq = T.flags[q];
    const uint32_t selfq = us >= 0 ? (uint32_t)us : (fq & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
    switch (kind) {
      case JG_CMD_APPEND_RESPONSE:
      case JG_CMD_HEARTBEAT_RESPONSE: {
        const int s = jg_node_slot_of(d, row.from);
        if (s >= 0 && ((col_mask >> s) & 1u)) *d.err = 6;  // this sender's answers arrived as a column: rows AND a column in one tick
        sparse = !leader_half || s < 0 || (uint32_t)s == selfq || (kind == JG_CMD_APPEND_RESPONSE && row.id >= JG_MAILBOX_NONE);
        bit = s < 0 ? 0u : 1u << ((kind == JG_CMD_APPEND_RESPONSE ? JGN_ACK_SHIFT : JGN_HBR_SHIFT) + (uint32_t)s);
        if (!sparse) T.arr[(kind == JG_CMD_APPEND_RESPONSE ? 0 : R) + s][q] = i + 1u;
        break;
      }
      case JG_CMD_CLIENT_REQUEST:
        sparse = !leader_half || (fq & JGF_ROLE_MASK) != JG_ROLE_LEADER;
        bit = JGN_CR;
        if (!sparse) T.arr[selfq][q] = i + 1u;
        break;
      case JG_CMD_HEARTBEAT:
        sparse = !follower_half || row.id == JG_NO_ACK || row.from == 0 || (fq & JGF_ROLE_MASK) == JG_ROLE_LEADER;
        bit = JGN_HB;
        if (FOLLOWER) F[0].fo[0][q] = i + 1u;
        break;
      case JG_CMD_APPEND_ENTRIES: {
        uint64_t from;
        sparse = !follower_half || row.from == 0 || !jg_node_ae_run(a, row.id, row.aux, &from) || (fq & JGF_ROLE_MASK) == JG_ROLE_LEADER;
        bit = JGN_AE;
        if (FOLLOWER) F[0].fo[1][q] = i + 1u;
        break;
      }
      default: sparse = true;
    }
    if (FOLLOWER && both_beats && (kind == JG_CMD_HEARTBEAT || kind == JG_CMD_APPEND_ENTRIES)) {
      atomicMax((unsigned long long*)&F[0].lt_max[q], (unsigned long long)row.term);
      atomicMin((unsigned long long*)&F[0].lt_min[q], (unsigned long long)row.term);
      atomicMax(&F[0].lf_max[q], row.from);
      atomicMin(&F[0].lf_min[q], row.from);
    }
    const uint32_t old = atomicOr(&T.cls[q], bit | (sparse ? JGN_SPARSE : 0u));
    if ((old & bit) && !sparse) atomicOr(&T.cls[q], JGN_SPARSE);  // a second row for the same mailbox entry
  }
  __syncthreads();
  // -- k_node_route
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t span = hi - lo, span_up = (span + 63u) & ~63u;  // (whole waves take every iteration together: the appends are wave-aggregated)
  for (uint32_t j = p; j < span_up; j += JG_BLOCK) {
    const uint32_t k = lo + j;
    const bool in = j < span;
    JgNodeBinRow row{};
    if (in) row = jg_node_bin_row(b, k);
    const uint32_t q = in ? row.group - g0 : 0u;
    const bool valid = in && row.kind < JG_CMD__COUNT;
    const uint32_t w = valid ? T.cls[q] : 0u;
    bool sparse = valid && (w & JGN_SPARSE);
    if (FOLLOWER && valid && !sparse && both_beats && (w & (JGN_HB | JGN_AE)) == (JGN_HB | JGN_AE))
      sparse = F[0].lt_max[q] != F[0].lt_min[q] || F[0].lf_max[q] != F[0].lf_min[q] || F[0].fo[1][q] < F[0].fo[0][q];
    const uint32_t i = row.idx;
    const uint64_t m = __ballot(sparse);
    if (m) {
      uint32_t base = 0;
      if (lane == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(n_sparse, (uint32_t)__popcll(m));
      base = __shfl(base, __ffsll((long long)m) - 1, 64);
      if (sparse) {
        const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        sp_key[at] = (uint64_t)(g0 + q) << 32 | i;
        sp_idx[at] = i;
        atomicOr((unsigned long long*)&T.sparse[q >> 6], 1ull << (q & 63u));
      }
    }
    if (!valid || sparse) continue;
    const uint32_t kind = row.kind;
    const uint32_t fq = T.flags[q];
    const uint32_t selfq = us >= 0 ? (uint32_t)us : (fq & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
    switch (kind) {
      case JG_CMD_APPEND_RESPONSE: {
        const int s = jg_node_slot_of(d, row.from);
        atomicAnd((unsigned long long*)&T.answers[s][q], (unsigned long long)((row.id << 8) | 0xffull));
        if ((w & JGN_CR) && i + 1u < T.arr[selfq][q])  // it arrived before the group's ClientRequest: it met the head before the append
          atomicOr((unsigned long long*)&T.answers[selfq][q], 1ull << (JGN_PRE_SHIFT + (uint32_t)s));
        break;
      }
      case JG_CMD_HEARTBEAT_RESPONSE: {
        const int s = jg_node_slot_of(d, row.from);
        const uint64_t has = row.flag ? 1 : 0;
        atomicAnd((unsigned long long*)&T.answers[s][q], (unsigned long long)(~0xffull | has));
        if (!has) c.hbr_commit[(size_t)s * G + g0 + q] = row.id;
        break;
      }
      case JG_CMD_CLIENT_REQUEST:
        atomicOr((unsigned long long*)&T.answers[selfq][q], 1ull << 8);
        T.token[q] = row.id;
        break;
      case JG_CMD_HEARTBEAT:
        if (FOLLOWER) F[0].f_beat[q] = jg_leader_beat{row.term, row.id}, F[0].f_leader[q] = row.from;
        break;
      default: {  // JG_CMD_APPEND_ENTRIES
        uint64_t from = 0;
        (void)jg_node_ae_run(a, row.id, row.aux, &from);
        if (FOLLOWER) {
          F[0].f_ae[q] = JG_AE(from, row.aux);
          if (!(w & JGN_HB)) {  // (with a Heartbeat in the batch: the same term and sender, written by its row)
            F[0].f_beat[q].term = row.term;
            F[0].f_leader[q] = row.from;
          }
        }
      }
    }
  }
  __syncthreads();
  // -- the columns go home, a line per wave and column
  if (live) {
    c.fsm_delta[g] = 0;
    if (leader_half) {
#pragma unroll
      for (int r = 0; r < R; r++)
        if ((uint32_t)r == self || !((col_mask >> r) & 1u)) c.answers[(size_t)r * G + g] = T.answers[r][p];
#pragma unroll
      for (int r = 0; r < 2 * R; r++) c.arr[(size_t)r * G + g] = T.arr[r][p];
      c.token[g] = T.token[p];
    }
    if (follower_half) c.f_beat[g] = F[0].f_beat[p], c.f_ae[g] = F[0].f_ae[p], c.f_leader[g] = F[0].f_leader[p];
  }
  if (p < JGN_TILE / 64 && g0 + p * 64u < G) c.sparse_bits[(g0 >> 6) + p] = T.sparse[p];
}
#endif  // JG_BLOCK == 256

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
