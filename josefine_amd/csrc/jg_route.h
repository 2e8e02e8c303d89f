// jg_route.h — the transport of a dense cluster for everything outside the mailbox vocabulary.
//
// The nodes of a jg_dense_cluster exchange their steady-state traffic as dense columns; what
// else a node emits (VoteRequest / VoteResponse of an election, a Heartbeat from a leader whose
// chain left run form, …) is queued as ordinary jg_msg_row rows — in the reference those rows go
// out on rpc_tx and come back through the peers' event loops as Commands (src/raft/server.rs:
// 127-137, tcp.rs:139-170).  These kernels are that path for nodes that share a device: they
// take the rows addressed to cluster members out of the senders' undrained output (the slots of
// their sparse steps and their exceptional-row queues) and turn them into the addressees' next
// command batch, per group in the order (phase of the round, emission index, sender slot): what every sender emitted
// first in a phase travels before anybody's second row of it, the senders interleaved - each sender's own stream in its
// own order, which is all a network promises (tcp.rs: one connection per peer pair).  The phases of a routed round are
// the steps every node takes in lockstep: 1 = the rows delivered by the last round (and the vote mail's receiving half),
// 2 = the rows injected for this round, 3 = the leader half, 4 = the follower half.  (Until round 6 the order was
// sender-major - each sender's rows back to back - under which an election of more than three nodes cannot be won: a
// candidate broadcasts its VoteRequest once per peer (candidate.rs:30-37), a voter grants the first copy and refuses the
// rest, and the later answer of a voter overwrites the earlier, election.rs:33-35.)  Nothing here
// interprets a row beyond its address.
//
// Not delivered (they stay queued for the host): AppendEntries rows — the payload is the
// sender's block store — and ClientRequest rows, which are instructions to the host adapter
// about its request mirror (josefine_gpu.h, "client request queue rows").
#pragma once
#include "jg_device.h"
#include "jg_sparse.h"
#include "jg_votes.h"

#define JG_ROUTE_ORD_BITS 26u       // widest emission-index field: the key then has 35 + bits(G) <= 64 bits (G <= 2^29)
#define JG_ROUTE_STEP_BITS 3u       // a node takes up to 4 steps per routed round (delivered rows, injected rows, leader half, follower half)
#define JG_ROUTE_ORD_BITS_FAST 12u  // what a round's keys are built with first: 5 sort passes instead of 7 at 1 M groups
// ordering key of a delivered row, most significant first: destination member (3 bits, right above
// the group's bits), group, phase of the round (3), emission index within the group's step (ord_bits: the pass
// reports an index that does not fit and is repeated with the wide field), sender slot (3)
struct JgRouteTable {
  uint32_t R, src;                      // members, the sending member's index
  uint32_t member_id[JG_MAX_REPLICAS];  // NodeId of member n
  uint32_t src_id, pad;                 // = member_id[src] (a dynamic index into a by-value copy of the table would put it in scratch)
  uint32_t group_bits;                  // bits of a group index
  uint32_t ord_bits;                    // bits of the emission-index field
  uint32_t cap;                         // entries of the staging below
  uint32_t seg_cap, seg_mask;           // the staging is n_seg = seg_mask + 1 segments of seg_cap entries, each with its own cursor
  uint64_t* key;                        // staging shared by all destinations (the sort separates them)
  uint32_t* idx;
  jg_msg_row* row;
  uint32_t* cursor;                     // [JG_ROUTE_SEGS] staging entries reserved so far, per segment
  uint32_t* count;                      // [R+4] of this sender: rows per destination, then JG_ROUTE_*
  uint32_t* kinds;                      // [R] per destination: the command kinds delivered to it this round (bit k: JG_CMD_* k)
};
enum { JG_ROUTE_KEPT = 0, JG_ROUTE_FSM = 1, JG_ROUTE_OVERFLOW = 2, JG_ROUTE_KEPT_XQ = 3 };  // count[R + …]

// the members a row is delivered to, as a bit mask
__device__ __forceinline__ uint32_t jg_route_dests(const jg_msg_row& r, const JgRouteTable& t) {
  if (r.kind == JG_CMD_APPEND_ENTRIES || r.kind == JG_CMD_CLIENT_REQUEST) return 0;
  const uint32_t all = ((1u << t.R) - 1u) & ~(1u << t.src);
  if (r.to_kind == JG_TO_PEERS) return all;
  if (r.to_kind != JG_TO_PEER) return 0;
  uint32_t m = 0;
#pragma unroll
  for (uint32_t n = 0; n < JG_MAX_REPLICAS; n++)
    if (n < t.R && t.member_id[n] == r.to_id) m |= 1u << n;
  return m & all;
}
// JG_ROUTE_VOTE_WORDS (WORDS): the members a row is delivered to AS A ROW - a campaign's broadcast stays away from the
// addressees whose partition takes this round's mail in words (jg_votes.h; `k`: the row's emission index)
template <bool WORDS>
__device__ __forceinline__ uint32_t jg_route_dests_rows(const jg_msg_row& r, const JgRouteTable& t, uint32_t k, const JgVoteMail& vm) {
  uint32_t m = jg_route_dests(r, t);
  if (WORDS && m && jg_vote_row_is_request_copy(r, t.src_id, k))
    for (uint32_t b = m; b; b &= b - 1)
      if (!jg_votes_as_rows(vm, (uint32_t)__ffs(b) - 1u, r.group, t.R - 1u)) m &= ~(b & (~b + 1u));
  return m;
}
__device__ __forceinline__ uint64_t jg_route_key(const JgRouteTable& t, uint32_t dest, uint32_t group, uint32_t step,
                                                 uint32_t ord) {
  if (ord >> t.ord_bits) t.count[t.R + JG_ROUTE_OVERFLOW] = 1;
  return ((((uint64_t)dest << t.group_bits | group) << JG_ROUTE_STEP_BITS | step) << t.ord_bits | ord) << 3 | t.src;
}
// A node's steps of a routed round -> the round's phases (JG_ROUTE_PHASE_*): which of its steps is which depends on what
// the node had to do (no delivered rows: its first step of the round is the injected rows' or a dense half), so the host
// hands the map along - 3 bits per step of the round, step i (1 .. 7) at bits [3 i, 3 i + 3).
enum { JG_ROUTE_PHASE_DELIVERED = 1, JG_ROUTE_PHASE_INJECTED = 2, JG_ROUTE_PHASE_LEADER = 3, JG_ROUTE_PHASE_FOLLOWER = 4 };
__host__ __device__ __forceinline__ uint32_t jg_route_phase(uint32_t phases, uint32_t step) { return (phases >> (3u * (step & 7u))) & 7u; }
// One staging reservation per workgroup and tile (every wave of a launch reserving for itself made the
// one cursor the bottleneck: a returning atomic on a single address retires every ~18 ns, 85 us for the
// 4.7 k waves of a 300 k-slot step) - and, since ~3 000 workgroup reservations of a round's delivering pass were
// still 54 us of that queue, on one of JG_ROUTE_SEGS cursors: workgroup x takes segment x mod n_seg of the staging
// (neighbouring workgroups run on different XCDs, so a cursor is mostly one XCD's).  The staged entries need no
// particular place: the ordering keys are unique and the bucket pass reads every segment.  A position at or beyond
// the end of the segment is not written: the host sees the cursor above seg_cap, grows the staging and repeats the
// pass (segmenting turned out not to be what bounded the pass - one cursor measured the same - and stays
// as the cheaper bound on that queue).
#define JG_ROUTE_SEGS 8u
struct JgRouteSpot {
  uint32_t pos, lim;   // the calling thread's first position; the end of the segment
  uint32_t base, tot;  // the workgroup's first position and its number of entries (whole: base + tot <= lim, or `whole` is false)
  uint32_t excl;       // the calling thread's offset within the workgroup's entries
  bool whole;
};
__device__ __forceinline__ JgRouteSpot jg_route_reserve(const JgRouteTable& t, uint32_t c) {
  __shared__ uint32_t base_s;
  JgRouteSpot s;
  s.excl = jg_block_exclusive_scan(c, &s.tot);
  const uint32_t seg = blockIdx.x & t.seg_mask;
  if (threadIdx.x == 0) base_s = s.tot ? atomicAdd(t.cursor + seg, s.tot) : 0u;
  __syncthreads();
  const uint32_t b = base_s;
  s.lim = seg * t.seg_cap + t.seg_cap;
  s.base = seg * t.seg_cap + min(b, t.seg_cap);
  s.whole = b <= t.seg_cap && s.tot <= t.seg_cap - b;
  s.pos = seg * t.seg_cap + min(b + s.excl, t.seg_cap);  // (saturating: beyond the segment is never a valid position)
  __syncthreads();  // (base_s is reused by the next tile)
  return s;
}
// The workgroup's entries are one contiguous range of the staging: they are collected in LDS and leave in whole
// lines, 8 bytes per lane and lanes side by side (written from the lanes that found them - a 40-byte row per lane,
// five partial stores per line - the delivering pass put 104 MB on the bus for 34 MB of entries).
#define JG_ROUTE_ITEMS 2  // slots per thread of the sparse steps' pass: a workgroup serves a tile of JG_BLOCK * JG_ROUTE_ITEMS
template <uint32_t CAP>
struct JgRouteStage {
  uint64_t key[CAP];
  uint64_t row[CAP * 5];
};
static_assert(sizeof(jg_msg_row) == 40, "a staged row is five 8-byte words");
template <uint32_t CAP>
__device__ __forceinline__ void jg_stage_put(JgRouteStage<CAP>& st, uint32_t at, uint64_t key, const jg_msg_row& r) {
  st.key[at] = key;
  uint64_t w[5];
  __builtin_memcpy(w, &r, sizeof(w));
#pragma unroll
  for (int k = 0; k < 5; k++) st.row[at * 5 + k] = w[k];
}
template <uint32_t CAP>
__device__ __forceinline__ void jg_stage_flush(const JgRouteStage<CAP>& st, const JgRouteTable& t, uint32_t base, uint32_t tot) {
  __syncthreads();
  for (uint32_t w = threadIdx.x; w < tot; w += JG_BLOCK) t.key[base + w] = st.key[w], t.idx[base + w] = base + w;
  uint64_t* rows = (uint64_t*)(t.row + base);
  for (uint32_t w = threadIdx.x; w < tot * 5; w += JG_BLOCK) rows[w] = st.row[w];
  __syncthreads();
}
// The launch's tallies -> the sender's count words, one global atomic per workgroup and word (per wave
// they queued up behind each other on a handful of addresses).  `pd_*`: rows per destination, 16 bits
// each (destinations 0-3 in lo, 4-7 in hi); call once, at the end of the kernel, from every thread.
// `kd_*`: the kinds of the rows noted per destination, 16 bits each (JG_CMD__COUNT <= 16), or-ed into t.kinds: the
// census that lets the addressee's next step pick a kernel without the code for kinds that are not there.
__device__ __forceinline__ void jg_route_tally(const JgRouteTable& t, uint64_t pd_lo, uint64_t pd_hi, uint32_t kept,
                                               uint32_t kept_word, uint32_t fsm, uint64_t kd_lo, uint64_t kd_hi) {
  static_assert(JG_CMD__COUNT <= 16, "a destination's kinds field is 16 bits");
  __shared__ uint32_t tally_s[2 * JG_MAX_REPLICAS + 2];
  if (threadIdx.x < 2 * JG_MAX_REPLICAS + 2) tally_s[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int off = 32; off; off >>= 1) {
    kd_lo |= __shfl_xor(kd_lo, off, 64);
    kd_hi |= __shfl_xor(kd_hi, off, 64);
  }
  if ((threadIdx.x & 63u) == 0)
    for (uint32_t n = 0; n < t.R; n++) {
      const uint32_t v = (uint32_t)((n < 4 ? kd_lo : kd_hi) >> (16 * (n & 3u))) & 0xffffu;
      if (v) atomicOr(&tally_s[JG_MAX_REPLICAS + 2 + n], v);
    }
  // (the 16-bit fields are PER LANE - a lane never counts 65 536 rows for one destination in a launch - and are
  // widened before they are added up: summed across a wave in their packed form, a wave total of 65 536 rows
  // to one destination used to carry into its neighbour's field)
  for (uint32_t n = 0; n < t.R; n++) {
    const uint32_t v = (uint32_t)((n < 4 ? pd_lo : pd_hi) >> (16 * (n & 3u))) & 0xffffu;
    if (v) atomicAdd(&tally_s[n], v);
  }
#pragma unroll
  for (int off = 32; off; off >>= 1) {
    kept += __shfl_down(kept, off, 64);
    fsm += __shfl_down(fsm, off, 64);
  }
  if ((threadIdx.x & 63u) == 0) {
    if (kept) atomicAdd(&tally_s[JG_MAX_REPLICAS], kept);
    if (fsm) atomicAdd(&tally_s[JG_MAX_REPLICAS + 1], fsm);
  }
  __syncthreads();
  if (threadIdx.x < t.R && tally_s[threadIdx.x]) atomicAdd(&t.count[threadIdx.x], tally_s[threadIdx.x]);
  if (threadIdx.x == JG_MAX_REPLICAS && tally_s[JG_MAX_REPLICAS]) atomicAdd(&t.count[t.R + kept_word], tally_s[JG_MAX_REPLICAS]);
  if (threadIdx.x == JG_MAX_REPLICAS + 1 && tally_s[JG_MAX_REPLICAS + 1])
    atomicAdd(&t.count[t.R + JG_ROUTE_FSM], tally_s[JG_MAX_REPLICAS + 1]);
  if (threadIdx.x >= 64 && threadIdx.x - 64 < t.R && tally_s[JG_MAX_REPLICAS + 2 + threadIdx.x - 64])
    atomicOr(&t.kinds[threadIdx.x - 64], tally_s[JG_MAX_REPLICAS + 2 + threadIdx.x - 64]);
}
__device__ __forceinline__ void jg_route_note(uint64_t& pd_lo, uint64_t& pd_hi, uint32_t dest) {
  if (dest < 4) pd_lo += 1ull << (16 * dest);
  else pd_hi += 1ull << (16 * (dest - 4));
}
__device__ __forceinline__ void jg_route_note_kind(uint64_t& kd_lo, uint64_t& kd_hi, uint32_t dest, uint32_t kind) {
  const uint64_t bit = 1ull << (kind & 15u);
  if (dest < 4) kd_lo |= bit << (16 * dest);
  else kd_hi |= bit << (16 * (dest - 4));
}

// The slots of one sparse step: every deliverable row goes to the staging (nothing is modified: the
// pass can be repeated with a larger staging); rows that stay and FSM rows are counted.
#define JG_ROUTE_REC_CAP (JG_BLOCK * JG_ROUTE_ITEMS)  // (one delivered row per slot is the usual yield: 24 KB of LDS)
template <bool WORDS = false>
__device__ __forceinline__ void jg_route_rec_body(JgRouteStage<JG_ROUTE_REC_CAP>& st, const JgRouteTable& t, uint32_t n, uint32_t per_row, uint32_t step,
                                                  const uint32_t* __restrict__ msg_cnt, const jg_msg_row* __restrict__ msg,
                                                  const uint32_t* __restrict__ fsm_cnt, const JgVoteMail& vm = JgVoteMail{}) {
  if (blockIdx.x * (JG_BLOCK * JG_ROUTE_ITEMS) >= n) return;  // (a multi launch is as wide as its largest job)
  const uint32_t tile0 = blockIdx.x * (JG_BLOCK * JG_ROUTE_ITEMS) + threadIdx.x;
  uint32_t cnt[JG_ROUTE_ITEMS];
  uint32_t c = 0, kept = 0, f = 0;
  uint64_t pd_lo = 0, pd_hi = 0, kd_lo = 0, kd_hi = 0;
#pragma unroll
  for (int k = 0; k < JG_ROUTE_ITEMS; k++) {
    const uint32_t i = tile0 + k * JG_BLOCK;
    cnt[k] = i < n ? msg_cnt[i] : 0u;
    f += i < n ? fsm_cnt[i] : 0u;
  }
#pragma unroll
  for (int k = 0; k < JG_ROUTE_ITEMS; k++) {
    const jg_msg_row* mine = msg + (size_t)(tile0 + k * JG_BLOCK) * per_row;
    for (uint32_t j = 0; j < cnt[k]; j++) {
      const uint32_t m = jg_route_dests_rows<WORDS>(mine[j], t, j, vm);
      c += __popc(m);
      kept += WORDS ? !jg_route_dests(mine[j], t) : !m;  // (what a word carries does not stay either)
      for (uint32_t b = m; b; b &= b - 1) {
        jg_route_note(pd_lo, pd_hi, (uint32_t)__ffs(b) - 1u);
        jg_route_note_kind(kd_lo, kd_hi, (uint32_t)__ffs(b) - 1u, mine[j].kind);
      }
    }
  }
  constexpr uint32_t CAP = JG_ROUTE_REC_CAP;
  const JgRouteSpot sp = jg_route_reserve(t, c);
  const bool staged = sp.whole && sp.tot <= CAP;  // (workgroup-uniform)
  uint32_t pos = sp.pos, at = sp.excl;
  const uint32_t lim = sp.lim;
  if (c) {
#pragma unroll
    for (int k = 0; k < JG_ROUTE_ITEMS; k++) {
      const uint32_t i = tile0 + k * JG_BLOCK;
      const jg_msg_row* mine = msg + (size_t)i * per_row;
      for (uint32_t j = 0; j < cnt[k]; j++) {
        const jg_msg_row r = mine[j];
        for (uint32_t b = jg_route_dests_rows<WORDS>(r, t, j, vm); b; b &= b - 1, pos++, at++) {
          if (staged) {
            jg_stage_put(st, at, jg_route_key(t, (uint32_t)__ffs(b) - 1u, r.group, step, j), r);
            continue;
          }
          if (pos >= lim) continue;  // (the host sees the cursor above the segment, grows the staging and repeats the pass)
          // (a run's rows lie back to back from its first slot: j is the emission index within the group's step)
          t.key[pos] = jg_route_key(t, (uint32_t)__ffs(b) - 1u, r.group, step, j);
          t.idx[pos] = pos;
          t.row[pos] = r;
        }
      }
    }
  }
  if (staged) jg_stage_flush(st, t, sp.base, sp.tot);
  jg_route_tally(t, pd_lo, pd_hi, kept, JG_ROUTE_KEPT, f, kd_lo, kd_hi);
}
// every (sender, step) of a round in ONE launch: blockIdx.y = job (7-8 launches of ~20 us before)
struct JgRouteRecJob {
  JgRouteTable t;
  uint32_t n, per_row, step, pad;  // step: the PHASE of the round this sparse step is (JG_ROUTE_PHASE_*)
  const uint32_t* msg_cnt;
  const jg_msg_row* msg;
  const uint32_t* fsm_cnt;
};
// the census of the same slots (jg_votes.h), before anything is delivered
__device__ __forceinline__ void jg_votes_census_rec_body(const JgRouteRecJob& j, const JgVoteMail& vm, uint32_t need = 0) {  // (the job through the reference: a by-value copy went to scratch, 168 B per lane)
  const uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x;
  if (i >= j.n) return;
  const uint32_t cnt = j.msg_cnt[i];
  const jg_msg_row* mine = j.msg + (size_t)i * j.per_row;
  for (uint32_t k = 0; k < cnt; k++) jg_votes_census_row(vm, j.t.src, j.t.member_id[j.t.src], mine[k], j.step, k, jg_route_dests(mine[k], j.t), need);
}
// second pass, only for a step that keeps rows for the host: the delivered rows leave their slots
__global__ __launch_bounds__(JG_BLOCK) void k_route_rec_compact(JgRouteTable t, uint32_t n, uint32_t per_row,
                                                                uint32_t* __restrict__ msg_cnt, jg_msg_row* __restrict__ msg) {
  const uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t cnt = msg_cnt[i];
  jg_msg_row* mine = msg + (size_t)i * per_row;
  uint32_t kept = 0;
  for (uint32_t j = 0; j < cnt; j++) {
    const jg_msg_row r = mine[j];
    if (jg_route_dests(r, t)) continue;
    if (kept != j) mine[kept] = r;
    kept++;
  }
  if (kept != cnt) msg_cnt[i] = kept;
}

// The exceptional-row queue of the dense steps.  COMPACT = false: deliver + count (nothing modified);
// COMPACT = true: the rows that stay are appended to `keep` (the queue is unordered; its rows carry
// their own step and emission index).
#define JG_ROUTE_XQ_CAP (JG_BLOCK * 4)  // (a tile of JG_BLOCK queue entries yields up to JG_BLOCK * (R - 1) staged rows: a campaign's VoteRequest goes to every peer)
template <bool COMPACT, bool WORDS = false>
__device__ __forceinline__ void jg_route_xq_body(JgRouteStage<JG_ROUTE_XQ_CAP>* stp, const JgRouteTable& t, const JgXqRec* __restrict__ xq,
                                                 const uint32_t* __restrict__ xq_n, uint32_t xq_cap, uint32_t seq_base, uint32_t phases,
                                                 JgXqRec* __restrict__ keep, uint32_t* __restrict__ keep_n, const JgVoteMail& vm = JgVoteMail{}) {
  const uint32_t n = min(*xq_n, xq_cap);
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t pd_lo = 0, pd_hi = 0, kd_lo = 0, kd_hi = 0;
  uint32_t stays = 0;
  for (uint32_t i0 = blockIdx.x * JG_BLOCK; i0 < n; i0 += gridDim.x * JG_BLOCK) {  // (block-uniform trip count)
    const uint32_t i = i0 + threadIdx.x;
    uint32_t mask = 0;
    JgXqRec q{};
    if (i < n) {
      q = xq[i];
      // (rows of steps before this round - left undrained by the caller - are not this round's mail: they stay)
      mask = (q.seq - seq_base - 1u < 7u) ? jg_route_dests(q.row, t) : 0u;  // (steps 1..7 of the round: JG_ROUTE_STEP_BITS)
    }
    const bool stay = i < n && !mask;
    if (WORDS && mask) mask = jg_route_dests_rows<true>(q.row, t, q.k, vm);  // (a word's copies do not stay, and are not staged)
    if (COMPACT) {
      const uint64_t b = __ballot(stay);
      if (b) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(keep_n, (uint32_t)__popcll(b));
        base = __shfl(base, 0, 64);
        if (stay) keep[base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = q;
      }
      continue;
    }
    const uint32_t step = jg_route_phase(phases, q.seq - seq_base);
    constexpr uint32_t CAP = JG_ROUTE_XQ_CAP;
    JgRouteStage<CAP>& st = *stp;  // (COMPACT: never touched - null)
    const JgRouteSpot sp = jg_route_reserve(t, __popc(mask));
    const bool staged = sp.whole && sp.tot <= CAP;  // (workgroup-uniform)
    uint32_t pos = sp.pos, at = sp.excl;
    stays += stay;
    for (uint32_t b = mask; b; b &= b - 1, pos++, at++) {
      jg_route_note(pd_lo, pd_hi, (uint32_t)__ffs(b) - 1u);
      jg_route_note_kind(kd_lo, kd_hi, (uint32_t)__ffs(b) - 1u, q.row.kind);
      if (staged) {
        jg_stage_put(st, at, jg_route_key(t, (uint32_t)__ffs(b) - 1u, q.row.group, step, q.k), q.row);
        continue;
      }
      if (pos >= sp.lim) continue;
      t.key[pos] = jg_route_key(t, (uint32_t)__ffs(b) - 1u, q.row.group, step, q.k);
      t.idx[pos] = pos;
      t.row[pos] = q.row;
    }
    if (staged) jg_stage_flush(st, t, sp.base, sp.tot);
  }
  if (!COMPACT) jg_route_tally(t, pd_lo, pd_hi, stays, JG_ROUTE_KEPT_XQ, 0, kd_lo, kd_hi);
}

template <bool COMPACT>
__global__ __launch_bounds__(JG_BLOCK) void k_route_xq(JgRouteTable t, const JgXqRec* __restrict__ xq,
                                                       const uint32_t* __restrict__ xq_n, uint32_t xq_cap,
                                                       uint32_t seq_base, uint32_t phases, JgXqRec* __restrict__ keep,
                                                       uint32_t* __restrict__ keep_n) {
  jg_route_xq_body<COMPACT>(nullptr, t, xq, xq_n, xq_cap, seq_base, phases, keep, keep_n);
}
struct JgRouteXqJob {  // the delivering pass over every sender's exceptional-row queue in one launch
  JgRouteTable t;
  const JgXqRec* xq;
  const uint32_t* xq_n;
  uint32_t xq_cap, seq_base;
  uint32_t phases, pad;  // the sender's steps of the round as phases (jg_route_phase)
};
// the census of the same queues (jg_votes.h), before anything is delivered
__device__ __forceinline__ void jg_votes_census_xq_body(const JgRouteXqJob& j, const JgVoteMail& vm, uint32_t need = 0) {
  const uint32_t n = min(*j.xq_n, j.xq_cap);
  for (uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x; i < n; i += gridDim.x * JG_BLOCK) {
    const JgXqRec q = j.xq[i];
    if (q.seq - j.seq_base - 1u >= 7u) continue;  // (not this round's mail)
    jg_votes_census_row(vm, j.t.src, j.t.member_id[j.t.src], q.row, jg_route_phase(j.phases, q.seq - j.seq_base), q.k, jg_route_dests(q.row, j.t), need);
  }
}
// ONE launch for the census of everything the round emitted: blockIdx.y < n_rec - a sparse step's slots (gridDim.x covers the
// widest), the rest - the senders' exceptional queues (their workgroups stride).  (Two launches until round 6: 17 + 7 us.)
// `need` = R - 1: the validation of the copies' counts rides along (jg_votes_census_row); 0: it is a pass of its own
__global__ __launch_bounds__(JG_BLOCK) void k_votes_census_multi(const JgRouteRecJob* __restrict__ rjobs, uint32_t n_rec, const JgRouteXqJob* __restrict__ xjobs,
                                                                 JgVoteMail vm, uint32_t need = 0) {
  if (blockIdx.y < n_rec) jg_votes_census_rec_body(rjobs[blockIdx.y], vm, need);
  else jg_votes_census_xq_body(xjobs[blockIdx.y - n_rec], vm, need);
}
#if JG_BLOCK % 64 == 0
// the answer words whose addressee's partition takes rows after all: staged as the rows they stand for (blockIdx.y = the
// sender; its table comes with its queue's job).  Rare - an answer word has to be rows only where its addressee's
// partition has BOTH kinds of mail - so the pass runs over the bitmaps: a workgroup takes JG_VOTE_CHUNK words of
// OR_d (wordmail[d] & rowmail[d]), a chunk without a bit costs nothing more (no reservation: the pass took 50 us of a
// round that had nothing to expand), and the lanes take the set bits (JgBitChunk, jg_votes.h).
__device__ __forceinline__ void jg_votes_expand_body(const JgRouteTable& t, const JgVoteMail& vm) {
  __shared__ JgBitChunk s;
  uint64_t pd_lo = 0, pd_hi = 0, kd_lo = 0, kd_hi = 0;
  const uint32_t n_chunks = (vm.words + JG_VOTE_CHUNK - 1) / JG_VOTE_CHUNK;
  for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {  // (block-uniform trip counts)
    const uint32_t w = c * JG_VOTE_CHUNK + threadIdx.x;
    uint64_t both = 0;
    if (threadIdx.x < JG_VOTE_CHUNK && w < vm.words)
      for (uint32_t d = 0; d < vm.R; d++) both |= vm.wordmail[(size_t)d * vm.words + w] & vm.rowmail[(size_t)d * vm.words + w];
    const uint32_t total = jg_chunk_scan(s, both);
    for (uint32_t t0 = 0; t0 < total; t0 += JG_BLOCK) {  // (the reservation is the workgroup's)
      const uint32_t i = t0 + threadIdx.x;
      uint32_t g = 0, n = 0, to = 0, step = 0, k0 = 0;
      if (i < total) {
        g = c * JG_VOTE_CHUNK * 64u + jg_chunk_pick(s, i);
        n = jg_votes_expand_count(vm, t.src, g, t.R - 1u, &to, &step, &k0);
      }
      const JgRouteSpot sp = jg_route_reserve(t, n);
      for (uint32_t j = 0, pos = sp.pos; j < n; j++, pos++) {
        jg_route_note(pd_lo, pd_hi, to);
        jg_route_note_kind(kd_lo, kd_hi, to, JG_CMD_VOTE_RESPONSE);
        if (pos >= sp.lim) continue;  // (the host sees the cursor above the segment, grows the staging and repeats the pass)
        t.key[pos] = jg_route_key(t, to, g, step, k0 + j);
        t.idx[pos] = pos;
        t.row[pos] = jg_votes_expand_row(vm, t.member_id, t.src, g, j);
      }
    }
  }
  jg_route_tally(t, pd_lo, pd_hi, 0, JG_ROUTE_KEPT, 0, kd_lo, kd_hi);
}
#endif
// ONE launch for the delivering pass (three until round 6: 29 + 16 + 14 us one behind the other, each a few dozen busy
// workgroups): blockIdx.y < n_rec - a sparse step's slots (gridDim.x covers the widest job: a workgroup beyond its job's end
// returns at once); < n_rec + n_xq - a sender's exceptional queue (strides); WORDS: the rest - a sender's answer words that
// must be rows after all (strides over the bitmaps).  The passes only append to the staging and add to the tallies: any
// order.  The two stagings share their LDS.
union JgRouteStageAny {
  JgRouteStage<JG_ROUTE_REC_CAP> rec;
  JgRouteStage<JG_ROUTE_XQ_CAP> xq;
};
template <bool WORDS>
__device__ __forceinline__ void jg_route_deliver_body(const JgRouteRecJob* __restrict__ rjobs, uint32_t n_rec, const JgRouteXqJob* __restrict__ xjobs, uint32_t n_xq,
                                                      const JgVoteMail& vm) {
  __shared__ JgRouteStageAny st;
  if (blockIdx.y < n_rec) {
    const JgRouteRecJob j = rjobs[blockIdx.y];
    jg_route_rec_body<WORDS>(st.rec, j.t, j.n, j.per_row, j.step, j.msg_cnt, j.msg, j.fsm_cnt, vm);
  } else if (blockIdx.y < n_rec + n_xq) {
    const JgRouteXqJob j = xjobs[blockIdx.y - n_rec];
    jg_route_xq_body<false, WORDS>(&st.xq, j.t, j.xq, j.xq_n, j.xq_cap, j.seq_base, j.phases, nullptr, nullptr, vm);
  }
#if JG_BLOCK % 64 == 0
  else if (WORDS) {
    const JgRouteTable t = xjobs[blockIdx.y - n_rec - n_xq].t;
    jg_votes_expand_body(t, vm);
  }
#endif
}
__global__ __launch_bounds__(JG_BLOCK) void k_route_deliver_multi(const JgRouteRecJob* __restrict__ rjobs, uint32_t n_rec, const JgRouteXqJob* __restrict__ xjobs, uint32_t n_xq) {
  jg_route_deliver_body<false>(rjobs, n_rec, xjobs, n_xq, JgVoteMail{});
}
__global__ __launch_bounds__(JG_BLOCK) void k_route_deliver_multi_words(const JgRouteRecJob* __restrict__ rjobs, uint32_t n_rec, const JgRouteXqJob* __restrict__ xjobs, uint32_t n_xq,
                                                                         JgVoteMail vm) {
  jg_route_deliver_body<true>(rjobs, n_rec, xjobs, n_xq, vm);
}

// sorted staging -> the command columns k_apply_rows consumes
struct JgRouteCols {
  uint8_t *kind, *flag;
  uint32_t *group, *from;
  uint64_t *term, *id, *aux;
};

// ---- ordering the staged rows without a library sort ------------------------------------------------
// The addressees apply their batch per group, so the staging has to come out ordered by (destination,
// group, phase, emission index, sender slot) - the 64-bit key above.  Round 2 did that with one rocPRIM
// radix sort over all staged pairs: 5-7 passes, ~15 launches, 240 us per 1.4 M rows, in the timed
// region of a BASELINE config.  The rows are nearly ordered already (every sender's output is
// group-major), and what the addressee needs is only *group order with a fixed order inside a group*:
//   k_route_hist     rows per bucket, a bucket = (destination, tile of 2^JG_ROUTE_TILE_BITS groups)
//   k_route_scan     exclusive scan of the bucket counts (one workgroup; buckets are destination-major,
//                    so a destination's rows end up contiguous: its batch is a slice)
//   k_route_scatter  every staged (key, index) pair into its bucket (order inside a bucket: arbitrary)
//   k_route_sort_build  one workgroup per bucket: its pairs sorted by key in LDS (bitonic), and the
//                    command columns of the addressees' next step written straight from the sorted
//                    order (the old k_route_build fused in)
// A bucket holds a few hundred rows at configs[4]'s rates (860 k of a round's 1.3 M rows go to the one
// candidate node: 220 per tile of 256 groups); one that outgrows the LDS tile is ranked in global memory by
// its workgroup (slow, exact).
#define JG_ROUTE_TILE_BITS 8u
#define JG_ROUTE_SORT_CAP 1024u  // pairs a workgroup sorts in LDS (12 KB: a dozen workgroups per CU)
#define JG_ROUTE_SCAN_TILE 1024u  // buckets one workgroup of k_route_scan scans (4 per thread, one 16-byte access)
struct JgRouteBuckets {
  uint32_t n_buckets;
  uint32_t shift;   // key >> shift = bucket id
  uint32_t* hist;   // [n_buckets rounded up to the scan tile] counts, then (after the scan) offsets within the scan tile
  uint32_t* cur;    // [n_buckets] scatter cursors (zeroed with hist)
  uint32_t* done;   // [1] scan tiles finished (zeroed with hist): the last workgroup of k_route_scan scans the tiles' totals
  uint32_t* tile;   // [n_tiles + 1] rows before each scan tile; entry n_tiles: all rows
  // first staging position of bucket i (i == n_buckets: the number of staged rows)
  __device__ __forceinline__ uint32_t off(uint32_t i) const {
    return i >= n_buckets ? tile[(n_buckets + JG_ROUTE_SCAN_TILE - 1u) / JG_ROUTE_SCAN_TILE] : tile[i / JG_ROUTE_SCAN_TILE] + hist[i];
  }
};
// A workgroup's 256 staged pairs sit in a handful of buckets (a sender's output is group-major; a row is
// staged once per addressee, so neighbouring entries alternate between the destinations' regions): the
// workgroup tallies them in a small LDS table first and touches global memory once per (workgroup, bucket) -
// per row the two passes took 184 + 197 us per 1.3 M rows (a few thousand hot addresses).
#define JG_ROUTE_TABLE 512u  // LDS table slots (open addressing; >= 2 x the pairs of a tile)
struct JgBucketTable {
  uint32_t key[JG_ROUTE_TABLE];  // bucket id + 1, 0 = free
  uint32_t cnt[JG_ROUTE_TABLE];
  uint32_t base[JG_ROUTE_TABLE];
};
__device__ __forceinline__ void jg_table_clear(JgBucketTable& t) {
  for (uint32_t i = threadIdx.x; i < JG_ROUTE_TABLE; i += JG_BLOCK) t.key[i] = 0, t.cnt[i] = 0;
  __syncthreads();
}
// the slot of bucket `bk` (inserted if new) and this entry's rank among the workgroup's entries of that bucket
__device__ __forceinline__ uint32_t jg_table_add(JgBucketTable& t, uint32_t bk, uint32_t& rank) {
  uint32_t s = (bk * 2654435761u) >> 23;  // 9 bits
  for (;;) {
    const uint32_t old = atomicCAS(&t.key[s], 0u, bk + 1u);
    if (old == 0u || old == bk + 1u) break;
    s = (s + 1u) & (JG_ROUTE_TABLE - 1u);
  }
  rank = atomicAdd(&t.cnt[s], 1u);
  return s;
}
// (blockIdx.y = staging segment: its entries are [y * seg_cap, y * seg_cap + seg_n[y]))
__global__ __launch_bounds__(JG_BLOCK) void k_route_hist(const uint32_t* __restrict__ seg_n, uint32_t seg_cap, const uint64_t* __restrict__ key,
                                                         JgRouteBuckets b) {
  __shared__ JgBucketTable t;
  const uint32_t n = min(seg_n[blockIdx.y], seg_cap);
  key += (size_t)blockIdx.y * seg_cap;
  for (uint32_t base = blockIdx.x * JG_BLOCK; base < n; base += gridDim.x * JG_BLOCK) {  // (workgroup-uniform trips)
    jg_table_clear(t);
    const uint32_t i = base + threadIdx.x;
    uint32_t rank;
    if (i < n) (void)jg_table_add(t, (uint32_t)(key[i] >> b.shift), rank);
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < JG_ROUTE_TABLE; s += JG_BLOCK)
      if (t.key[s]) (void)__hip_atomic_fetch_add(&b.hist[t.key[s] - 1u], t.cnt[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
}
// exclusive scan of the bucket counts: every workgroup scans its own tile of 1024 buckets (one 16-byte access per thread; a
// single workgroup walking all 20 k buckets took 33-56 us), and the LAST workgroup to finish scans the tile totals (a few
// dozen) - a launch of their own until round 6 (4.4 us of a round that is a chain of launches).  The same last workgroup
// empties the exceptional queues the delivering pass - complete by now - has delivered whole (k_route_clear_words' job in the
// rounds that have rows to order): a queue that kept nothing for the host, in a pass that does not have to be repeated (no
// segment ran over, no emission index too wide: the host sees the same words and repeats it, with the queue intact).
struct JgRouteXqDone {
  uint32_t R, route_words, n_seg, seg_cap;
  const uint32_t* count;   // [R][route_words] the senders' tallies (jg_route_tally)
  const uint32_t* cursor;  // [n_seg]
  uint32_t* xq_n[JG_MAX_REPLICAS];  // the senders' queue lengths (null: not this launch's business)
};
__global__ __launch_bounds__(JG_BLOCK) void k_route_scan_all(JgRouteBuckets b, JgRouteXqDone xd) {
  static_assert(JG_ROUTE_SCAN_TILE == 4 * JG_BLOCK, "4 buckets per thread");
  uint4* p = (uint4*)(b.hist + (size_t)blockIdx.x * JG_ROUTE_SCAN_TILE) + threadIdx.x;
  const uint4 v = *p;  // (the array is padded to whole tiles and zeroed)
  uint32_t tot;
  const uint32_t ex = jg_block_exclusive_scan(v.x + v.y + v.z + v.w, &tot);
  *p = uint4{ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z};
  __shared__ uint32_t last_s;
  if (threadIdx.x == 0) {
    // (the total goes out and is read back through L2 atomics; the ticket's release / acquire orders it for the last workgroup)
    (void)atomicExch(&b.tile[blockIdx.x], tot);
    last_s = __hip_atomic_fetch_add(b.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
  }
  __syncthreads();
  if (!last_s) return;
  const uint32_t n_tiles = (b.n_buckets + JG_ROUTE_SCAN_TILE - 1u) / JG_ROUTE_SCAN_TILE;  // (= gridDim.x)
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base <= n_tiles; base += JG_BLOCK) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t t = i < n_tiles ? atomicAdd(&b.tile[i], 0u) : 0u;  // (an L2 read: what the other workgroups' exchanges left)
    uint32_t tt;
    const uint32_t e = carry_s + jg_block_exclusive_scan(t, &tt);
    __syncthreads();
    if (i <= n_tiles) b.tile[i] = e;
    if (threadIdx.x == 0) carry_s += tt;
    __syncthreads();
  }
  if (threadIdx.x == 0 && xd.count) {
    bool again = false;
    for (uint32_t k = 0; k < xd.n_seg; k++) again = again || xd.cursor[k] > xd.seg_cap;
    for (uint32_t s = 0; s < xd.R; s++) again = again || xd.count[(size_t)s * xd.route_words + xd.R + JG_ROUTE_OVERFLOW];
    for (uint32_t s = 0; s < xd.R && !again; s++)
      if (xd.xq_n[s] && !xd.count[(size_t)s * xd.route_words + xd.R + JG_ROUTE_KEPT_XQ]) *xd.xq_n[s] = 0;
  }
}
// (the two-launch form: the fault records' and the node step's general-path ordering - off the routed round - keep it)
__global__ __launch_bounds__(JG_BLOCK) void k_route_scan(JgRouteBuckets b) {
  static_assert(JG_ROUTE_SCAN_TILE == 4 * JG_BLOCK, "4 buckets per thread");
  uint4* p = (uint4*)(b.hist + (size_t)blockIdx.x * JG_ROUTE_SCAN_TILE) + threadIdx.x;
  const uint4 v = *p;  // (the array is padded to whole tiles and zeroed)
  uint32_t tot;
  const uint32_t ex = jg_block_exclusive_scan(v.x + v.y + v.z + v.w, &tot);
  *p = uint4{ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z};
  if (threadIdx.x == 0) b.tile[blockIdx.x] = tot;
}
__global__ __launch_bounds__(JG_BLOCK) void k_route_scan_tiles(JgRouteBuckets b) {
  const uint32_t n_tiles = (b.n_buckets + JG_ROUTE_SCAN_TILE - 1u) / JG_ROUTE_SCAN_TILE;
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base <= n_tiles; base += JG_BLOCK) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n_tiles ? b.tile[i] : 0u;
    uint32_t tot;
    const uint32_t ex = carry_s + jg_block_exclusive_scan(v, &tot);
    __syncthreads();
    if (i <= n_tiles) b.tile[i] = ex;
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
}
__global__ __launch_bounds__(JG_BLOCK) void k_route_scatter(const uint32_t* __restrict__ seg_n, uint32_t seg_cap, const uint64_t* __restrict__ key,
                                                            const uint32_t* __restrict__ idx, JgRouteBuckets b, uint64_t* __restrict__ key_out,
                                                            uint32_t* __restrict__ idx_out) {
  __shared__ JgBucketTable t;
  const uint32_t n = min(seg_n[blockIdx.y], seg_cap);
  key += (size_t)blockIdx.y * seg_cap, idx += (size_t)blockIdx.y * seg_cap;
  for (uint32_t base = blockIdx.x * JG_BLOCK; base < n; base += gridDim.x * JG_BLOCK) {
    jg_table_clear(t);
    const uint32_t i = base + threadIdx.x;
    const bool live = i < n;
    const uint64_t k = live ? key[i] : 0ull;
    uint32_t rank = 0, slot = 0;
    if (live) slot = jg_table_add(t, (uint32_t)(k >> b.shift), rank);
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < JG_ROUTE_TABLE; s += JG_BLOCK)  // one reservation per (workgroup, bucket)
      if (t.key[s]) t.base[s] = b.off(t.key[s] - 1u) + atomicAdd(&b.cur[t.key[s] - 1u], t.cnt[s]);
    __syncthreads();
    if (live) {
      const uint32_t at = t.base[slot] + rank;
      key_out[at] = k;
      idx_out[at] = idx[i];
    }
    __syncthreads();
  }
}
__device__ __forceinline__ void jg_route_sort_bucket(const JgRouteBuckets& b, uint32_t bucket, uint64_t (&s_key)[JG_ROUTE_SORT_CAP], uint32_t (&s_idx)[JG_ROUTE_SORT_CAP],
                                                     uint64_t* __restrict__ key, uint32_t* __restrict__ idx, const jg_msg_row* __restrict__ rows, const JgRouteCols& c) {
  const uint32_t lo = b.off(bucket), n = b.off(bucket + 1) - lo;  // (workgroup-uniform)
  if (!n) return;
  if (n <= JG_BLOCK) {
    // the usual bucket (a tile of 256 groups holds ~64 rows of a round): every key's rank by counting the smaller
    // ones - the keys are unique - with all JG_BLOCK threads on it: `per` neighbouring lanes share a key and a
    // stride of the scan (n = 64: 16 LDS reads per thread and no barrier but one, against the 21 compare-exchange
    // stages and barriers of the bitonic network below: this kernel was 68 us of a round)
    uint32_t m = 1;
    while (m < n) m <<= 1;
    const uint32_t per = JG_BLOCK / m;  // 1 .. 256, a power of two; lanes [k * per, (k + 1) * per) work on key k
    if (threadIdx.x < n) s_key[threadIdx.x] = key[lo + threadIdx.x], s_idx[threadIdx.x] = idx[lo + threadIdx.x];
    __syncthreads();
    const uint32_t ki = threadIdx.x / per, part = threadIdx.x % per;
    const uint64_t k = ki < n ? s_key[ki] : 0ull;
    uint32_t rank = 0;
    if (ki < n)
      for (uint32_t j = part; j < n; j += per) rank += s_key[j] < k;
    if (per > 64) {  // (n <= 2: the lanes of a key span waves)
      __shared__ uint32_t s_rank[2];
      if (threadIdx.x < 2) s_rank[threadIdx.x] = 0;
      __syncthreads();
      if (ki < n && rank) atomicAdd(&s_rank[ki], rank);
      __syncthreads();
      rank = ki < n ? s_rank[ki] : 0u;
    } else {
      for (uint32_t off = per >> 1; off; off >>= 1) rank += __shfl_xor(rank, (int)off, 64);
    }
    if (ki < n && part == 0) {
      const jg_msg_row r = rows[s_idx[ki]];
      const uint32_t p = lo + rank;
      c.kind[p] = r.kind, c.flag[p] = r.flag, c.group[p] = r.group, c.from[p] = r.from;
      c.term[p] = r.term, c.id[p] = r.id, c.aux[p] = r.aux;
    }
    return;
  }
  if (n <= JG_ROUTE_SORT_CAP) {
    uint32_t m = 1;
    while (m < n) m <<= 1;
    for (uint32_t i = threadIdx.x; i < m; i += JG_BLOCK) {
      s_key[i] = i < n ? key[lo + i] : ~0ull;  // (padding sorts last; real keys are < 2^64 - 1: destination < 8)
      s_idx[i] = i < n ? idx[lo + i] : 0u;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t i = threadIdx.x; i < m; i += JG_BLOCK) {
          const uint32_t p = i ^ j;
          if (p > i) {
            const bool up = (i & k) == 0;
            const uint64_t a = s_key[i], q = s_key[p];
            if ((a > q) == up) {
              s_key[i] = q, s_key[p] = a;
              const uint32_t t = s_idx[i];
              s_idx[i] = s_idx[p], s_idx[p] = t;
            }
          }
        }
        __syncthreads();
      }
    for (uint32_t i = threadIdx.x; i < n; i += JG_BLOCK) {
      const jg_msg_row r = rows[s_idx[i]];
      const uint32_t p = lo + i;
      c.kind[p] = r.kind, c.flag[p] = r.flag, c.group[p] = r.group, c.from[p] = r.from;
      c.term[p] = r.term, c.id[p] = r.id, c.aux[p] = r.aux;
    }
    return;
  }
  // a bucket larger than the LDS tile: every pair's rank by counting the smaller keys (keys are unique:
  // destination, group, phase, emission index and sender name one row)
  for (uint32_t i = threadIdx.x; i < n; i += JG_BLOCK) {
    const uint64_t k = key[lo + i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; j++) rank += key[lo + j] < k;
    const jg_msg_row r = rows[idx[lo + i]];
    const uint32_t p = lo + rank;
    c.kind[p] = r.kind, c.flag[p] = r.flag, c.group[p] = r.group, c.from[p] = r.from;
    c.term[p] = r.term, c.id[p] = r.id, c.aux[p] = r.aux;
  }
}
// A workgroup takes `per_wg` consecutive buckets, one after the other (a round's rows sit in a few percent of the
// R x G / 256 buckets - 19.5 k at 1 M x 5 - and a launch of one workgroup per bucket spends most of its 20 us dispatching
// empty ones; but a workgroup's buckets are served one BEHIND the other, each a chain of dependent loads: 8 per workgroup
// took 27 us - profiles/r06/ab_sort_buckets.txt)
#define JG_ROUTE_SORT_BUCKETS 1u
__global__ __launch_bounds__(JG_BLOCK) void k_route_sort_build(JgRouteBuckets b, uint64_t* __restrict__ key, uint32_t* __restrict__ idx,
                                                               const jg_msg_row* __restrict__ rows, JgRouteCols c, uint32_t per_wg) {
  __shared__ uint64_t s_key[JG_ROUTE_SORT_CAP];
  __shared__ uint32_t s_idx[JG_ROUTE_SORT_CAP];
  for (uint32_t k = 0; k < per_wg; k++) {
    const uint32_t bucket = blockIdx.x * per_wg + k;
    if (bucket >= b.n_buckets) return;
    jg_route_sort_bucket(b, bucket, s_key, s_idx, key, idx, rows, c);
    __syncthreads();  // (the LDS tile is the next bucket's)
  }
}

// The same ordering for any list of (64-bit key, 32-bit value) pairs with UNIQUE keys (k_route_hist / _scan /
// _scan_tiles / _scatter as above, then this instead of k_route_sort_build): one workgroup per bucket ranks its
// pairs and writes the values in key order.  jg_step_node's general path orders its rows with it (key = group << 32 |
// arrival index: group-major, a group's rows in the order they arrived) - no library sort on that path either.
__global__ __launch_bounds__(JG_BLOCK) void k_bucket_order(JgRouteBuckets b, const uint64_t* __restrict__ key, const uint32_t* __restrict__ val,
                                                           uint32_t* __restrict__ val_out) {
  __shared__ uint64_t s_key[JG_ROUTE_SORT_CAP];
  __shared__ uint32_t s_val[JG_ROUTE_SORT_CAP];
  const uint32_t lo = b.off(blockIdx.x), n = b.off(blockIdx.x + 1) - lo;
  if (!n) return;
  if (n <= JG_ROUTE_SORT_CAP) {
    for (uint32_t i = threadIdx.x; i < n; i += JG_BLOCK) s_key[i] = key[lo + i], s_val[i] = val[lo + i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += JG_BLOCK) {  // rank by counting the smaller keys (unique): n <= 1024
      const uint64_t k = s_key[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n; j++) rank += s_key[j] < k;
      val_out[lo + rank] = s_val[i];
    }
    return;
  }
  for (uint32_t i = threadIdx.x; i < n; i += JG_BLOCK) {  // a bucket larger than the LDS tile: ranked in global memory
    const uint64_t k = key[lo + i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; j++) rank += key[lo + j] < k;
    val_out[lo + rank] = val[lo + i];
  }
}

// The round's counters back to zero in ONE launch each (a hipMemsetAsync is up to three fill kernels of ~4 us, and a
// round had seven of them): two word ranges; and up to JG_MAX_REPLICAS single words named by pointer.
__global__ __launch_bounds__(JG_BLOCK) void k_route_clear(uint32_t* __restrict__ a, uint32_t na, uint32_t* __restrict__ b, uint32_t nb) {
  for (uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x; i < na + nb; i += gridDim.x * JG_BLOCK) {
    if (i < na) a[i] = 0;
    else b[i - na] = 0;
  }
}
// the round's job tables from the pinned host staging to their device copy, by a kernel: a hipMemcpyAsync of these 64 KB
// took the copy engine 25-30 us to get going, with the stream idle behind it at the head of every round (rocprofv3
// --kernel-trace: profiles/r06/routed_round_trace_before.txt); `n`: 8-byte words
__global__ __launch_bounds__(JG_BLOCK) void k_copy_words(uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, uint32_t n) {
  for (uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x; i < n; i += gridDim.x * JG_BLOCK) dst[i] = src[i];
}
struct JgWordList {
  uint32_t* p[JG_MAX_REPLICAS];
  uint32_t n;
};
__global__ void k_route_clear_words(JgWordList w) {
  if (threadIdx.x < w.n) *w.p[threadIdx.x] = 0;
}

// jg_dense_cluster_mailboxes: the common AppendEntries word of a group (JgLeaderNode::o_aec) into the rows of the block -
// every slot's but the sender's (lead, or with per-partition leadership the group's owner: none where nobody owns it)
__global__ void k_aec_expand(uint32_t G, uint32_t R, uint32_t lead, const uint8_t* __restrict__ owner, const uint64_t* __restrict__ aec,
                             uint64_t* __restrict__ ae) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const uint64_t c = aec[g];
  const uint32_t from = owner ? owner[g] : lead;
  if (from == JG_OWNER_NONE) {  // (nobody owns the group this round: nobody's mail - not last round's words)
    for (uint32_t r = 0; r < R; r++) ae[(size_t)r * G + g] = JG_NO_ACK;
    return;
  }
  if (c == JG_AEC_INDIVIDUAL) return;
  for (uint32_t r = 0; r < R; r++)
    if (r != from) ae[(size_t)r * G + g] = c;
}
// jg_dense_cluster_offer_appends / _withdraw_appends: `per_round` ClientRequests (0: no more) for the listed groups
__global__ void k_offer_appends(uint32_t n, const uint32_t* __restrict__ groups, uint32_t G, uint64_t per_round, uint64_t* __restrict__ offered,
                                uint64_t* __restrict__ own_col) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = groups[i];
  if (g >= G) return;
  offered[g] = JG_ANSWER(per_round, JG_HB_NONE);
  if (own_col) own_col[g] = JG_ANSWER(per_round, JG_HB_NONE);
}
