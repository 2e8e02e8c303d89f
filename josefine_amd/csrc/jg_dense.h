// jg_dense.h — the HBM-roofline kernel: dense steady-state leader tick.
//
// AppendEntries-ack tally (progress.rs:42-46,76-94,133-140), majority test
// (progress.rs:48-60) and commit-index advance (leader.rs:87-99) for every leader
// group at once, plus the tick's own appends with their self-acks
// (leader.rs:177-197, chain.rs:160-175), over SoA columns.
//
// Per group-step the kernel reads R ack heads, R match heads, commit, head (8 B
// each) and the 4-B flag word and writes back what changed: B(R) = 24R + 36
// algorithmic bytes; at steady state exactly that.
//
// Exactness of the fusion: the reference evaluates Leader::commit after every
// ack.  match[] is monotone, hence so is committed_index(), and the guard
// `q > commit` makes the final commit max(commit, q_final) — provided
// chain.commit(q) never panics on the way, which in FAST form (id set == [0, head])
// means q <= head at each evaluation.  If every old match head and every ack is
// <= the head before this tick's appends that cannot happen and the tick is one
// majority evaluation; otherwise the lane replays appends and acks one by one and
// faults exactly where the reference would panic (chain.rs:197-202).
#pragma once
#include "jg_device.h"

#ifndef JG_BLOCK
#define JG_BLOCK 256
#endif

typedef unsigned long long jg_u64x2 __attribute__((ext_vector_type(2)));  // one 16-B access

// ---- wave64 / workgroup reduction of the per-lane decision counts --------------------
// One plain read-modify-write per workgroup into its own slot: kernels on the
// engine stream are serialised, so no atomics are needed (a single hot atomic
// would cost ~12 ns x #waves, more than the tick itself).
__device__ __forceinline__ void jg_block_count(uint64_t* slots, uint32_t v) {
  __shared__ uint32_t wave_sum[JG_BLOCK / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) wave_sum[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < JG_BLOCK / 64; w++) s += wave_sum[w];
    if (s) slots[blockIdx.x] += s;
  }
}

// element R/2 of the heads sorted descending (progress.rs:48-60) by rank counting
template <int R>
__device__ __forceinline__ uint64_t jg_kth(const uint64_t (&v)[R]) {
  constexpr int K = R / 2;
  uint64_t q = 0;
#pragma unroll
  for (int j = 0; j < R; j++) {
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < R; i++) cnt += (v[i] > v[j] || (v[i] == v[j] && i < j)) ? 1 : 0;
    q = (cnt == K) ? v[j] : q;
  }
  return q;
}

// What a lane must do with one group after looking at its flag word.
enum { JG_DENSE_SKIP = 0, JG_DENSE_RUN = 1 };

// Classify the group; handles the rare non-RUN outcomes itself.
template <int R>
__device__ __forceinline__ int jg_dense_classify(const JgDev& d, uint32_t g, uint32_t f, const uint64_t (&a)[R],
                                                 uint32_t seq, uint32_t* s_out, uint64_t* n_app_out) {
  if (f & JGF_FAULT_MASK) return JG_DENSE_SKIP;  // the reference process is gone
  const uint32_t s = (f & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
  uint64_t n_app = 0;
#pragma unroll
  for (int r = 0; r < R; r++) n_app = (r == (int)s) ? a[r] : n_app;
  *s_out = s;
  *n_app_out = n_app;
  if ((f & JGF_ROLE_MASK) != JG_ROLE_LEADER) {
    // acks are ignored by followers / candidates (follower.rs:62, candidate.rs:194)
    if (n_app) {
      d.flags[g] = f | (JG_FAULT_ENGINE_DENSE_NONLEADER << JGF_FAULT_SHIFT);
      jg_push_fault(d, g, JG_FAULT_ENGINE_DENSE_NONLEADER, seq);
    }
    return JG_DENSE_SKIP;
  }
  if (!(f & JGF_FAST)) {
    // irregular chain: k_dense_slow, launched right behind this kernel, finds the group by
    // the same test on its flag word (no list, no atomics on this path) and replays the tick
    // through the general state machine.  *deferred_seen lets the host verify it was scheduled.
    *d.deferred_seen = 1;
    return JG_DENSE_SKIP;
  }
  return JG_DENSE_RUN;
}

// The tick of one FAST leader group, entirely in registers.  Updates m[], commit,
// head, flag word; returns the number of quorum decisions taken.
template <int R>
__device__ __forceinline__ uint32_t jg_dense_core(const JgDev& d, uint32_t g, uint32_t seq, uint32_t s,
                                                  uint64_t n_app, const uint64_t (&a)[R], uint64_t (&m)[R],
                                                  uint64_t& commit, uint64_t& head, uint32_t& nf) {
  const uint64_t head0 = head, commit0 = commit;
  uint32_t dec = 0;
  uint64_t hi = 0;  // max over old match heads and follower acks
#pragma unroll
  for (int r = 0; r < R; r++) {
    hi = m[r] > hi ? m[r] : hi;
    bool is_ack = (r != (int)s) && (a[r] != JG_NO_ACK);
    hi = (is_ack && a[r] > hi) ? a[r] : hi;
  }
  if (hi <= head0) {
    // ---- fused path ---------------------------------------------------------------
    head = head0 + n_app;  // n appends: ids head0+1 .. head0+n (chain.rs:160-175, FAST form)
#pragma unroll
    for (int r = 0; r < R; r++) {
      uint32_t bit = 1u << (JGF_REPL_SHIFT + r);
      if (r == (int)s) {
        if (n_app) {  // n self-acks; the last increment decides Probe/Replicate
          bool inc = m[r] < head;
          m[r] = inc ? head : m[r];
          nf = inc ? (nf | bit) : (nf & ~bit);
          dec += (uint32_t)n_app;
        }
      } else if (a[r] != JG_NO_ACK) {  // progress.rs:76-94,133-140
        bool inc = m[r] < a[r];
        m[r] = inc ? a[r] : m[r];
        nf = inc ? (nf | bit) : (nf & ~bit);
        dec += 1;
      }
    }
    uint64_t q = jg_kth<R>(m);         // progress.rs:48-60
    commit = q > commit ? q : commit;  // leader.rs:89-92
  } else {
    // ---- exact replay: one Leader::commit per append / ack ---------------------------
    uint32_t fault = 0;
    const uint32_t sbit = 1u << (JGF_REPL_SHIFT + s);
    for (uint64_t i = 0; i < n_app && !fault; i++) {
      head += 1;
      bool inc = false;
#pragma unroll
      for (int r = 0; r < R; r++)
        if (r == (int)s) {
          inc = m[r] < head;
          m[r] = inc ? head : m[r];
        }
      nf = inc ? (nf | sbit) : (nf & ~sbit);
      dec += 1;
      uint64_t q = jg_kth<R>(m);
      if (q > commit) {
        if (q <= head) commit = q;
        else fault = JG_FAULT_COMMIT_MISSING_BLOCK;  // chain.rs:197-202
      }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (r == (int)s || a[r] == JG_NO_ACK || fault) continue;
      uint32_t bit = 1u << (JGF_REPL_SHIFT + r);
      bool inc = m[r] < a[r];
      m[r] = inc ? a[r] : m[r];
      nf = inc ? (nf | bit) : (nf & ~bit);
      dec += 1;
      uint64_t q = jg_kth<R>(m);
      if (q > commit) {
        if (q <= head) commit = q;
        else fault = JG_FAULT_COMMIT_MISSING_BLOCK;
      }
    }
    if (fault) {
      nf |= fault << JGF_FAULT_SHIFT;
      jg_push_fault(d, g, fault, seq);
    }
  }
  if (commit != commit0) nf |= JGF_COMMIT_KEY;  // chain.rs:198
  return dec;
}

// ---- variant 1: one group per lane, 8-B accesses ------------------------------------------
template <int R>
__global__ __launch_bounds__(JG_BLOCK) void k_leader_tick_dense(JgDev d, const uint64_t* __restrict__ acks,
                                                                 uint32_t seq) {
  const uint32_t G = d.G;
  uint32_t dec = 0;
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < G; g += gridDim.x * JG_BLOCK) {
    const uint32_t f = d.flags[g];
    uint64_t a[R], m[R];
#pragma unroll
    for (int r = 0; r < R; r++) a[r] = __builtin_nontemporal_load(&acks[(size_t)r * G + g]);
#pragma unroll
    for (int r = 0; r < R; r++) m[r] = d.match[(size_t)r * G + g];
    const uint64_t commit0 = d.commit[g], head0 = d.head[g];
    uint32_t s;
    uint64_t n_app;
    if (jg_dense_classify<R>(d, g, f, a, seq, &s, &n_app) != JG_DENSE_RUN) continue;
    uint64_t commit = commit0, head = head0;
    uint32_t nf = f;
    // a match head changes only through an increment, i.e. exactly when its ack (or the
    // self-ack) is above the old value: remember that instead of keeping the old heads
    uint32_t chg = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      const bool self = r == (int)s;
      const bool up = self ? (n_app != 0 && m[r] < head0 + n_app) : (a[r] != JG_NO_ACK && m[r] < a[r]);
      chg |= up ? (1u << r) : 0u;
    }
    dec += jg_dense_core<R>(d, g, seq, s, n_app, a, m, commit, head, nf);
#pragma unroll
    for (int r = 0; r < R; r++)
      if (chg & (1u << r)) d.match[(size_t)r * G + g] = m[r];
    if (commit != commit0) d.commit[g] = commit;
    if (head != head0) d.head[g] = head;
    if (nf != f) d.flags[g] = nf;
  }
  jg_block_count(d.blk_decisions, dec);
}

// ---- variant 2: two adjacent groups per lane, 16-B accesses (G even) ------------------------
// 16 B per lane is the coalescing sweet spot on gfx950 (1 KiB per wave instruction);
// every u64 column is read and written as one 16-B vector, the flag column as uint2.
template <int R>
__global__ __launch_bounds__(JG_BLOCK) void k_leader_tick_dense_x2(JgDev d, const uint64_t* __restrict__ acks,
                                                                    uint32_t seq) {
  const uint32_t G = d.G, P = G >> 1;  // pairs
  uint32_t dec = 0;
  for (uint32_t p = blockIdx.x * JG_BLOCK + threadIdx.x; p < P; p += gridDim.x * JG_BLOCK) {
    const uint32_t g0 = p << 1;
    const uint2 f2 = *reinterpret_cast<const uint2*>(d.flags + g0);
    jg_u64x2 a2[R], m2[R];
#pragma unroll
    for (int r = 0; r < R; r++)
      a2[r] = __builtin_nontemporal_load(reinterpret_cast<const jg_u64x2*>(acks + (size_t)r * G + g0));
#pragma unroll
    for (int r = 0; r < R; r++) m2[r] = *reinterpret_cast<const jg_u64x2*>(d.match + (size_t)r * G + g0);
    const jg_u64x2 c2 = *reinterpret_cast<const jg_u64x2*>(d.commit + g0);
    const jg_u64x2 h2 = *reinterpret_cast<const jg_u64x2*>(d.head + g0);

    uint64_t ax[R], ay[R], mx[R], my[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      ax[r] = a2[r].x;
      ay[r] = a2[r].y;
      mx[r] = m2[r].x;
      my[r] = m2[r].y;
    }
    uint64_t cx = c2.x, cy = c2.y, hx = h2.x, hy = h2.y;
    uint32_t fx = f2.x, fy = f2.y;
    uint32_t s;
    uint64_t n_app;
    if (jg_dense_classify<R>(d, g0, f2.x, ax, seq, &s, &n_app) == JG_DENSE_RUN)
      dec += jg_dense_core<R>(d, g0, seq, s, n_app, ax, mx, cx, hx, fx);
    if (jg_dense_classify<R>(d, g0 + 1, f2.y, ay, seq, &s, &n_app) == JG_DENSE_RUN)
      dec += jg_dense_core<R>(d, g0 + 1, seq, s, n_app, ay, my, cy, hy, fy);

#pragma unroll
    for (int r = 0; r < R; r++)
      if (mx[r] != m2[r].x || my[r] != m2[r].y)
        *reinterpret_cast<jg_u64x2*>(d.match + (size_t)r * G + g0) = jg_u64x2{mx[r], my[r]};
    if (cx != c2.x || cy != c2.y) *reinterpret_cast<jg_u64x2*>(d.commit + g0) = jg_u64x2{cx, cy};
    if (hx != h2.x || hy != h2.y) *reinterpret_cast<jg_u64x2*>(d.head + g0) = jg_u64x2{hx, hy};
    // a skipped group's flag word may have been rewritten by jg_dense_classify (fault):
    // only store the words this lane changed itself.
    if (fx != f2.x) d.flags[g0] = fx;
    if (fy != f2.y) d.flags[g0 + 1] = fy;
  }
  jg_block_count(d.blk_decisions, dec);
}

// ---- variant N: T consecutive ticks per launch (temporal fusion) -----------------------------
// When the caller already holds the ack blocks of several ticks (a batched event loop, the
// pre-generated bench stream), the group's state stays in registers across them: it is read
// once and written once per launch, so a group-step costs 8R (acks) + (16R+36)/T bytes of
// traffic instead of 24R+36.  Semantically identical to T calls of the single-tick kernel:
// tick t reads acks + t*tick_stride and carries sequence number seq0 + t.
template <int R>
__global__ __launch_bounds__(JG_BLOCK) void k_leader_tick_dense_n(JgDev d, const uint64_t* __restrict__ acks,
                                                                   uint32_t n_ticks, size_t tick_stride,
                                                                   uint32_t seq0) {
  const uint32_t G = d.G;
  uint32_t dec = 0;
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < G; g += gridDim.x * JG_BLOCK) {
    const uint32_t f = d.flags[g];
    uint64_t a[R], an[R], m[R], m0[R];
#pragma unroll
    for (int r = 0; r < R; r++) a[r] = __builtin_nontemporal_load(&acks[(size_t)r * G + g]);
#pragma unroll
    for (int r = 0; r < R; r++) m0[r] = m[r] = d.match[(size_t)r * G + g];
    const uint64_t commit0 = d.commit[g], head0 = d.head[g];
    if (f & JGF_FAULT_MASK) continue;  // the reference process is gone
    const uint32_t s = (f & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
    const bool leader = (f & JGF_ROLE_MASK) == JG_ROLE_LEADER;
    if (leader && !(f & JGF_FAST)) {  // irregular chain: k_dense_slow replays all ticks
      *d.deferred_seen = 1;
      continue;
    }
    uint64_t commit = commit0, head = head0;
    uint32_t nf = f;
    for (uint32_t t = 0; t < n_ticks; t++) {
      const bool more = t + 1 < n_ticks;
      if (more) {  // software prefetch of the next tick's acks
        const uint64_t* nx = acks + (size_t)(t + 1) * tick_stride;
#pragma unroll
        for (int r = 0; r < R; r++) an[r] = __builtin_nontemporal_load(&nx[(size_t)r * G + g]);
      }
      uint64_t n_app = 0;
#pragma unroll
      for (int r = 0; r < R; r++) n_app = (r == (int)s) ? a[r] : n_app;
      if (!leader) {
        // acks are ignored by followers / candidates (follower.rs:62, candidate.rs:194)
        if (n_app) {
          nf = f | (JG_FAULT_ENGINE_DENSE_NONLEADER << JGF_FAULT_SHIFT);
          jg_push_fault(d, g, JG_FAULT_ENGINE_DENSE_NONLEADER, seq0 + t);
          break;
        }
      } else {
        dec += jg_dense_core<R>(d, g, seq0 + t, s, n_app, a, m, commit, head, nf);
        if (nf & JGF_FAULT_MASK) break;
      }
      if (more) {
#pragma unroll
        for (int r = 0; r < R; r++) a[r] = an[r];
      }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
      if (m[r] != m0[r]) d.match[(size_t)r * G + g] = m[r];
    if (commit != commit0) d.commit[g] = commit;
    if (head != head0) d.head[g] = head;
    if (nf != f) d.flags[g] = nf;
  }
  jg_block_count(d.blk_decisions, dec);
}
