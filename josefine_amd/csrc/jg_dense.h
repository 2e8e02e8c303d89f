// jg_dense.h — the HBM-roofline kernel: dense steady-state leader tick.
//
// AppendEntries-ack tally (progress.rs:42-46,76-94,133-140), majority test
// (progress.rs:48-60) and commit-index advance (leader.rs:87-99) for every leader
// group at once, plus the tick's own appends with their self-acks
// (leader.rs:177-197, chain.rs:160-175), over SoA columns.
//
// Algorithmic bytes per group-step: B(R) = 24R + 36 — R ack heads,
// R match heads, commit, head, term read (8 B each) + the 4-B flag word, R match heads
// + commit written.  The kernel moves less: the leader's own match head equals the
// chain head after every self-ack (SELF-SYNC, jg_device.h) and is then implicit, so at
// steady state it reads 16R + 12 and writes 8R + 8 bytes = 24R + 20 (140 B at R = 5).
//
// Exactness of the fusion: the reference evaluates Leader::commit after every
// ack.  match[] is monotone, hence so is committed_index(), and the guard
// `q > commit` makes the final commit max(commit, q_final) — provided
// chain.commit(q) never panics on the way, which in FAST form (id set == [0, head])
// means q <= head at each evaluation.  If every old match head and every ack is
// <= the head before this tick's appends that cannot happen and the tick is one
// majority evaluation; otherwise the lane replays appends and acks one by one and
// faults exactly where the reference would panic (chain.rs:197-202).
//
// Register form: the group's own slot s and the R-1 other slots are kept apart
// ("self + others": other k is slot k + (k >= s), ascending), so that with an
// engine-uniform own slot every load except the rarely needed own match head has an
// address that does not depend on the flag word: one round trip to HBM per group.
#pragma once
#include "jg_device.h"

#ifndef JG_BLOCK
#define JG_BLOCK 256
#endif

// ---- wave64 / workgroup reduction of the per-lane decision counts --------------------
// One fire-and-forget atomic per workgroup into the workgroup's own slot (no contention,
// no return value: the wave does not wait for it; a load/add/store would keep the
// workgroup resident for one more HBM round trip; a single hot counter would cost
// ~12 ns x #waves, more than the tick itself).
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
    if (s)
      (void)__hip_atomic_fetch_add(&slots[blockIdx.x], (uint64_t)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One group's tick in registers.  O = max(R-1, 1) other slots.
template <int R>
struct JgDenseRegs {
  static constexpr int O = R > 1 ? R - 1 : 1;
  uint64_t ao[O];  // acks of the other slots (JG_NO_ACK = none)
  uint64_t mo[O];  // their match heads
  uint64_t ms;     // own match head
  uint64_t n_app;  // ClientRequests to append this tick (the own slot of the ack block)
  uint64_t commit, head;
  uint32_t nf;     // flag word being rebuilt
};

// element R/2 of the heads sorted descending (progress.rs:48-60) by rank counting.  Ties are
// broken by position only to make the ranks distinct; the selected value does not depend on it.
template <int R>
__device__ __forceinline__ uint64_t jg_kth(const JgDenseRegs<R>& x) {
  constexpr int K = R / 2;
  uint64_t v[R];
  v[0] = x.ms;
#pragma unroll
  for (int k = 0; k + 1 < R; k++) v[k + 1] = x.mo[k];
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

// slot of other k for own slot s
__device__ __forceinline__ uint32_t jg_other_slot(uint32_t k, uint32_t s) { return k + (k >= s ? 1u : 0u); }

// The tick of one FAST leader group, entirely in registers.  Updates the match heads, commit,
// head and flag word in `x`; returns the number of quorum decisions taken.
template <int R>
__device__ __forceinline__ uint32_t jg_dense_core(const JgDev& d, uint32_t g, uint32_t seq, uint32_t s,
                                                  JgDenseRegs<R>& x) {
  const uint64_t head0 = x.head, commit0 = x.commit;
  const uint32_t sbit = 1u << (JGF_REPL_SHIFT + s);
  uint32_t dec = 0;
  uint64_t hi = x.ms;  // max over old match heads and follower acks
#pragma unroll
  for (int k = 0; k + 1 < R; k++) {
    hi = x.mo[k] > hi ? x.mo[k] : hi;
    hi = (x.ao[k] != JG_NO_ACK && x.ao[k] > hi) ? x.ao[k] : hi;
  }
  if (hi <= head0) {
    // ---- fused path ---------------------------------------------------------------
    x.head = head0 + x.n_app;  // n appends: ids head0+1 .. head0+n (chain.rs:160-175, FAST form)
    if (x.n_app) {             // n self-acks; the last increment decides Probe/Replicate
      bool inc = x.ms < x.head;
      x.ms = inc ? x.head : x.ms;
      x.nf = inc ? (x.nf | sbit) : (x.nf & ~sbit);
      dec += (uint32_t)x.n_app;
    }
#pragma unroll
    for (int k = 0; k + 1 < R; k++) {
      if (x.ao[k] != JG_NO_ACK) {  // progress.rs:76-94,133-140
        const uint32_t bit = 1u << (JGF_REPL_SHIFT + jg_other_slot(k, s));
        bool inc = x.mo[k] < x.ao[k];
        x.mo[k] = inc ? x.ao[k] : x.mo[k];
        x.nf = inc ? (x.nf | bit) : (x.nf & ~bit);
        dec += 1;
      }
    }
    uint64_t q = jg_kth<R>(x);               // progress.rs:48-60
    x.commit = q > x.commit ? q : x.commit;  // leader.rs:89-92
  } else {
    // ---- exact replay: one Leader::commit per append / ack ---------------------------
    uint32_t fault = 0;
    for (uint64_t i = 0; i < x.n_app && !fault; i++) {
      x.head += 1;
      bool inc = x.ms < x.head;
      x.ms = inc ? x.head : x.ms;
      x.nf = inc ? (x.nf | sbit) : (x.nf & ~sbit);
      dec += 1;
      uint64_t q = jg_kth<R>(x);
      if (q > x.commit) {
        if (q <= x.head) x.commit = q;
        else fault = JG_FAULT_COMMIT_MISSING_BLOCK;  // chain.rs:197-202
      }
    }
#pragma unroll
    for (int k = 0; k + 1 < R; k++) {  // ascending k = ascending slot
      if (x.ao[k] == JG_NO_ACK || fault) continue;
      const uint32_t bit = 1u << (JGF_REPL_SHIFT + jg_other_slot(k, s));
      bool inc = x.mo[k] < x.ao[k];
      x.mo[k] = inc ? x.ao[k] : x.mo[k];
      x.nf = inc ? (x.nf | bit) : (x.nf & ~bit);
      dec += 1;
      uint64_t q = jg_kth<R>(x);
      if (q > x.commit) {
        if (q <= x.head) x.commit = q;
        else fault = JG_FAULT_COMMIT_MISSING_BLOCK;
      }
    }
    if (fault) {
      x.nf |= fault << JGF_FAULT_SHIFT;
      jg_push_fault(d, g, fault, seq);
    }
  }
  if (x.commit != commit0) x.nf |= JGF_COMMIT_KEY;  // chain.rs:198
  return dec;
}

// Issue every load of one group whose address is known without the flag word.
template <int R>
__device__ __forceinline__ void jg_dense_load_acks(const uint64_t* __restrict__ acks, uint32_t G, uint32_t g,
                                                   uint32_t s, uint64_t& n_app, uint64_t (&ao)[JgDenseRegs<R>::O]) {
  n_app = __builtin_nontemporal_load(&acks[(size_t)s * G + g]);
#pragma unroll
  for (int k = 0; k + 1 < R; k++) ao[k] = __builtin_nontemporal_load(&acks[(size_t)jg_other_slot(k, s) * G + g]);
}
template <int R>
__device__ __forceinline__ void jg_dense_load(const JgDev& d, const uint64_t* __restrict__ acks, uint32_t g,
                                              uint32_t s, JgDenseRegs<R>& x) {
  const uint32_t G = d.G;
  jg_dense_load_acks<R>(acks, G, g, s, x.n_app, x.ao);
#pragma unroll
  for (int k = 0; k + 1 < R; k++) x.mo[k] = d.match[(size_t)jg_other_slot(k, s) * G + g];
  x.commit = d.commit[g];
  x.head = d.head[g];
}

// Store what changed.  `chg`: bit k+1 set = match head of other k may differ from what was
// loaded, bit 0 = the own one.  The own slot stays implicit while it equals the chain head.
template <int R>
__device__ __forceinline__ void jg_dense_store(const JgDev& d, uint32_t g, uint32_t s, uint32_t f, uint32_t chg,
                                               const JgDenseRegs<R>& x, uint64_t commit0, uint64_t head0) {
  const uint32_t G = d.G;
#pragma unroll
  for (int k = 0; k + 1 < R; k++)
    if ((chg >> (k + 1)) & 1u) d.match[(size_t)jg_other_slot(k, s) * G + g] = x.mo[k];
  const bool sync1 = x.ms == x.head;
  if (!sync1 && ((chg & 1u) || (f & JGF_SELF_SYNC))) d.match[(size_t)s * G + g] = x.ms;
  const uint32_t nf = sync1 ? (x.nf | JGF_SELF_SYNC) : (x.nf & ~JGF_SELF_SYNC);
  if (x.commit != commit0) d.commit[g] = x.commit;
  if (x.head != head0) d.head[g] = x.head;
  if (nf != f) d.flags[g] = nf;
}

// What a lane must do with one group after looking at its flag word.
enum { JG_DENSE_SKIP = 0, JG_DENSE_RUN = 1 };

// Classify the group; handles the rare non-RUN outcomes itself.
__device__ __forceinline__ int jg_dense_classify(const JgDev& d, uint32_t g, uint32_t f, uint64_t n_app,
                                                 uint32_t seq) {
  if (f & JGF_FAULT_MASK) return JG_DENSE_SKIP;  // the reference process is gone
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

// ---- one tick per launch ------------------------------------------------------------------------
// UNIFORM: every group of the engine has own slot `us` (the normal case: a node has one
// NodeId).  Otherwise the own slot comes from the flag word and the other loads wait for it.
// (A two-groups-per-lane variant with 16-B accesses measured no faster — the kernel is
// bandwidth-, not issue-bound: profiles/README.md round 1 — and was dropped.)
template <int R, bool UNIFORM>
__device__ __forceinline__ uint32_t jg_dense_tick_body(const JgDev& d, const uint64_t* __restrict__ acks,
                                                       uint32_t seq, uint32_t us) {
  const uint32_t G = d.G;
  uint32_t dec = 0;
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < G; g += gridDim.x * JG_BLOCK) {
    const uint32_t f = d.flags[g];
    const uint32_t s = UNIFORM ? us : (f & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
    JgDenseRegs<R> x;
    jg_dense_load<R>(d, acks, g, s, x);
    if (jg_dense_classify(d, g, f, x.n_app, seq) != JG_DENSE_RUN) continue;
    const uint64_t commit0 = x.commit, head0 = x.head;
    x.ms = head0;  // SELF-SYNC: implicit
    if (!(f & JGF_SELF_SYNC)) x.ms = d.match[(size_t)s * G + g];
    x.nf = f;
    // a match head changes only through an increment, i.e. exactly when its ack (or the
    // self-ack) is above the old value: remember that instead of keeping the old heads
    uint32_t chg = (x.n_app != 0 && x.ms < head0 + x.n_app) ? 1u : 0u;
#pragma unroll
    for (int k = 0; k + 1 < R; k++) chg |= (x.ao[k] != JG_NO_ACK && x.mo[k] < x.ao[k]) ? (2u << k) : 0u;
    dec += jg_dense_core<R>(d, g, seq, s, x);
    jg_dense_store<R>(d, g, s, f, chg, x, commit0, head0);
  }
  return dec;
}

template <int R>
__global__ __launch_bounds__(JG_BLOCK) void k_leader_tick_dense(JgDev d, const uint64_t* __restrict__ acks,
                                                                 uint32_t seq, int us) {
  uint32_t dec;
  if (us >= 0) dec = jg_dense_tick_body<R, true>(d, acks, seq, (uint32_t)us);
  else dec = jg_dense_tick_body<R, false>(d, acks, seq, 0);
  jg_block_count(d.blk_decisions, dec);
}

// ---- T consecutive ticks per launch (temporal fusion) ----------------------------------------
// When the caller already holds the ack blocks of several ticks (a batched event loop, the
// pre-generated bench stream), the group's state stays in registers across them: it is read
// once and written once per launch, so a group-step costs 8R (acks) + (16R+20)/T bytes of
// traffic instead of 24R+36.  Semantically identical to T calls of the single-tick kernel:
// tick t reads acks + t*tick_stride and carries sequence number seq0 + t.
template <int R, bool UNIFORM>
__device__ __forceinline__ uint32_t jg_dense_ticks_body(const JgDev& d, const uint64_t* __restrict__ acks,
                                                        uint32_t n_ticks, size_t tick_stride, uint32_t seq0,
                                                        uint32_t us) {
  const uint32_t G = d.G;
  uint32_t dec = 0;
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < G; g += gridDim.x * JG_BLOCK) {
    const uint32_t f = d.flags[g];
    const uint32_t s = UNIFORM ? us : (f & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
    JgDenseRegs<R> x;
    jg_dense_load<R>(d, acks, g, s, x);
    if (f & JGF_FAULT_MASK) continue;  // the reference process is gone
    const bool leader = (f & JGF_ROLE_MASK) == JG_ROLE_LEADER;
    if (leader && !(f & JGF_FAST)) {  // irregular chain: k_dense_slow replays all ticks
      *d.deferred_seen = 1;
      continue;
    }
    const uint64_t commit0 = x.commit, head0 = x.head;
    x.ms = head0;
    if (leader && !(f & JGF_SELF_SYNC)) x.ms = d.match[(size_t)s * G + g];
    const uint64_t ms0 = x.ms;
    uint64_t mo0[JgDenseRegs<R>::O];
#pragma unroll
    for (int k = 0; k + 1 < R; k++) mo0[k] = x.mo[k];
    x.nf = f;
    for (uint32_t t = 0; t < n_ticks; t++) {
      const bool more = t + 1 < n_ticks;
      uint64_t n_app_n = 0, an[JgDenseRegs<R>::O];
      if (more)  // software prefetch of the next tick's acks
        jg_dense_load_acks<R>(acks + (size_t)(t + 1) * tick_stride, G, g, s, n_app_n, an);
      if (!leader) {
        // acks are ignored by followers / candidates (follower.rs:62, candidate.rs:194)
        if (x.n_app) {
          x.nf = f | (JG_FAULT_ENGINE_DENSE_NONLEADER << JGF_FAULT_SHIFT);
          jg_push_fault(d, g, JG_FAULT_ENGINE_DENSE_NONLEADER, seq0 + t);
          break;
        }
      } else {
        dec += jg_dense_core<R>(d, g, seq0 + t, s, x);
        if (x.nf & JGF_FAULT_MASK) break;
      }
      if (more) {
        x.n_app = n_app_n;
#pragma unroll
        for (int k = 0; k + 1 < R; k++) x.ao[k] = an[k];
      }
    }
    if (leader) {
      uint32_t chg = x.ms != ms0 ? 1u : 0u;
#pragma unroll
      for (int k = 0; k + 1 < R; k++) chg |= (x.mo[k] != mo0[k]) ? (2u << k) : 0u;
      jg_dense_store<R>(d, g, s, f, chg, x, commit0, head0);
    } else if (x.nf != f) {
      d.flags[g] = x.nf;
    }
  }
  return dec;
}

template <int R>
__global__ __launch_bounds__(JG_BLOCK) void k_leader_tick_dense_n(JgDev d, const uint64_t* __restrict__ acks,
                                                                   uint32_t n_ticks, size_t tick_stride,
                                                                   uint32_t seq0, int us) {
  uint32_t dec;
  if (us >= 0) dec = jg_dense_ticks_body<R, true>(d, acks, n_ticks, tick_stride, seq0, (uint32_t)us);
  else dec = jg_dense_ticks_body<R, false>(d, acks, n_ticks, tick_stride, seq0, 0);
  jg_block_count(d.blk_decisions, dec);
}
