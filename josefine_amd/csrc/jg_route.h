// jg_route.h — the transport of a dense cluster for everything outside the mailbox vocabulary.
//
// The nodes of a jg_dense_cluster exchange their steady-state traffic as dense columns; what
// else a node emits (VoteRequest / VoteResponse of an election, a Heartbeat from a leader whose
// chain left run form, …) is queued as ordinary jg_msg_row rows — in the reference those rows go
// out on rpc_tx and come back through the peers' event loops as Commands (src/raft/server.rs:
// 127-137, tcp.rs:139-170).  These kernels are that path for nodes that share a device: they
// take the rows addressed to cluster members out of the senders' undrained output (the slots of
// their sparse steps and their exceptional-row queues) and turn them into the addressees' next
// command batch, per group in the order (sender slot, step, emission order).  Nothing here
// interprets a row beyond its address.
//
// Not delivered (they stay queued for the host): AppendEntries rows — the payload is the
// sender's block store — and ClientRequest rows, which are instructions to the host adapter
// about its request mirror (josefine_gpu.h, "client request queue rows").
#pragma once
#include "jg_device.h"

#define JG_ROUTE_ORD_BITS 27u  // key: group << 32 | sender slot << 29 | step of the round << 27 | emission index
#define JG_ROUTE_INJECT_SRC 7u

struct JgRouteTable {
  uint32_t R, src;                        // members, the sending member's index
  uint32_t member_id[JG_MAX_REPLICAS];    // NodeId of member n
  uint64_t* key[JG_MAX_REPLICAS];         // staging of destination n (scatter pass)
  jg_msg_row* row[JG_MAX_REPLICAS];
  uint32_t cap[JG_MAX_REPLICAS];
  uint32_t* cursor;                       // [R]   scatter positions
  uint32_t* count;                        // [R+3] count pass: rows per destination, kept rows, fsm rows, overflow
};
enum { JG_ROUTE_KEPT = 0, JG_ROUTE_FSM = 1, JG_ROUTE_OVERFLOW = 2 };  // count[R + …]

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

// one wave-aggregated reservation per destination and iteration
__device__ __forceinline__ void jg_route_emit(const JgRouteTable& t, uint32_t mask, const jg_msg_row& r, uint64_t key,
                                              bool scatter) {
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t n = 0; n < t.R; n++) {
    const uint64_t b = __ballot((mask >> n) & 1u);
    if (!b) continue;
    const uint32_t first = (uint32_t)__ffsll((long long)b) - 1u;
    uint32_t base = 0;
    if (lane == first) base = atomicAdd(scatter ? &t.cursor[n] : &t.count[n], (uint32_t)__popcll(b));
    base = __shfl(base, (int)first, 64);
    if (scatter && ((mask >> n) & 1u)) {
      const uint32_t pos = base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
      if (pos < t.cap[n]) {
        t.key[n][pos] = key;
        t.row[n][pos] = r;
      } else {
        t.count[t.R + JG_ROUTE_OVERFLOW] = 1;
      }
    }
  }
}

// The slots of one sparse step.  SCATTER = false: count only.  SCATTER = true: deliver, and
// compact the rows that stay to the front of their slot (msg_cnt rewritten).
template <bool SCATTER>
__global__ __launch_bounds__(JG_BLOCK) void k_route_rec(JgRouteTable t, uint32_t n, uint32_t per_row, uint32_t step,
                                                        uint32_t* __restrict__ msg_cnt, jg_msg_row* __restrict__ msg,
                                                        const uint32_t* __restrict__ fsm_cnt) {
  const uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x;
  const uint32_t cnt = i < n ? msg_cnt[i] : 0u;
  uint32_t kept = 0;
  for (uint32_t j = 0; __any(j < cnt); j++) {
    uint32_t mask = 0;
    jg_msg_row r{};
    if (j < cnt) {
      r = msg[(size_t)i * per_row + j];
      mask = jg_route_dests(r, t);
      if (!mask) {
        if (SCATTER && kept != j) msg[(size_t)i * per_row + kept] = r;
        kept++;
      }
    }
    const uint64_t ord = (uint64_t)i * per_row + j;
    const uint64_t key = (uint64_t)r.group << 32 | (uint64_t)t.src << 29 | (uint64_t)step << JG_ROUTE_ORD_BITS | ord;
    jg_route_emit(t, mask, r, key, SCATTER);
  }
  if (SCATTER) {
    if (i < n && kept != cnt) msg_cnt[i] = kept;
  } else {
    uint32_t f = i < n ? fsm_cnt[i] : 0u;
    for (int off = 32; off; off >>= 1) {
      kept += __shfl_down(kept, off, 64);
      f += __shfl_down(f, off, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
      if (kept) atomicAdd(&t.count[t.R + JG_ROUTE_KEPT], kept);
      if (f) atomicAdd(&t.count[t.R + JG_ROUTE_FSM], f);
    }
  }
}

// The exceptional-row queue of the dense steps.  SCATTER: the rows that stay are appended to
// `keep` (the queue is unordered; its rows carry their own step and emission index).
template <bool SCATTER>
__global__ __launch_bounds__(JG_BLOCK) void k_route_xq(JgRouteTable t, const JgXqRec* __restrict__ xq,
                                                       const uint32_t* __restrict__ xq_n, uint32_t xq_cap,
                                                       uint32_t seq_base, JgXqRec* __restrict__ keep,
                                                       uint32_t* __restrict__ keep_n) {
  const uint32_t n = min(*xq_n, xq_cap);
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t i0 = (blockIdx.x * JG_BLOCK + threadIdx.x) & ~63u; i0 < n; i0 += gridDim.x * JG_BLOCK) {
    const uint32_t i = i0 + lane;
    uint32_t mask = 0;
    JgXqRec q{};
    bool stay = false;
    if (i < n) {
      q = xq[i];
      mask = jg_route_dests(q.row, t);
      stay = !mask;
    }
    const uint32_t step = q.seq - seq_base;
    if (mask && (step > 3u || q.k >> JG_ROUTE_ORD_BITS)) t.count[t.R + JG_ROUTE_OVERFLOW] = 1;
    const uint64_t key = (uint64_t)q.row.group << 32 | (uint64_t)t.src << 29 | (uint64_t)(step & 3u) << JG_ROUTE_ORD_BITS | q.k;
    jg_route_emit(t, mask, q.row, key, SCATTER);
    const uint64_t b = __ballot(stay);
    if (b) {
      const uint32_t first = (uint32_t)__ffsll((long long)b) - 1u;
      uint32_t base = 0;
      if (lane == first) base = atomicAdd(SCATTER ? keep_n : &t.count[t.R + JG_ROUTE_KEPT], (uint32_t)__popcll(b));
      base = __shfl(base, (int)first, 64);
      if (SCATTER && stay) keep[base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = q;
    }
  }
}

// sorted staging -> the command columns k_apply_rows consumes
struct JgRouteCols {
  uint8_t *kind, *flag;
  uint32_t *group, *from;
  uint64_t *term, *id, *aux;
};
__global__ __launch_bounds__(JG_BLOCK) void k_route_build(uint32_t n, const uint32_t* __restrict__ order,
                                                          const jg_msg_row* __restrict__ rows, JgRouteCols c) {
  const uint32_t p = blockIdx.x * JG_BLOCK + threadIdx.x;
  if (p >= n) return;
  const jg_msg_row r = rows[order[p]];
  c.kind[p] = r.kind;
  c.flag[p] = r.flag;
  c.group[p] = r.group;
  c.from[p] = r.from;
  c.term[p] = r.term;
  c.id[p] = r.id;
  c.aux[p] = r.aux;
}
__global__ void k_route_iota(uint32_t n, uint32_t* __restrict__ v) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) v[p] = p;
}

// ClientRequests are offered only where the lead node leads (at a leaderless replica the reference
// queues them, follower.rs:258-270 — not expressible in the dense append column)
__global__ void k_route_mask_appends(uint32_t G, const uint32_t* __restrict__ flags, const uint64_t* __restrict__ offered,
                                     uint64_t* __restrict__ own_col) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) own_col[g] = (flags[g] & JGF_ROLE_MASK) == JG_ROLE_LEADER ? offered[g] : 0ull;
}
