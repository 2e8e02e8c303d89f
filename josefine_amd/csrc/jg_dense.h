// jg_dense.h — the HBM-roofline kernel: dense steady-state leader tick.
//
// AppendEntries-ack tally (progress.rs:42-46,76-94,133-140), majority test
// (progress.rs:48-60) and commit-index advance (leader.rs:87-99) for every leader
// group at once, plus the tick's own appends with their self-acks
// (leader.rs:177-197, chain.rs:160-175), over SoA columns.
//
// Algorithmic bytes per group-step: B(R) = 24R + 36 — R ack heads,
// R match heads, commit, head, term read (8 B each) + the 4-B flag word, R match heads
// + commit written.  The kernel moves less: the R match heads and the commit index are
// delta-packed into one 64-bit word of lags below the chain head (jg_device.h), which in
// steady state does not even change from tick to tick; it reads 8R + 20 and writes 8 bytes
// (68 B at R = 5).
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
// Three paths, the same results (tests/test_gpu_parity.py, tests/test_dense_node.py vs the oracle):
//   hot      jg_lag_tick: the whole tick as 32-bit arithmetic on the lags of the packed word, for
//            a healthy FAST leader whose fields are un-escaped and whose acks are at or below the
//            head — all loads of the group (flag word, R acks, packed word, head) in one round trip;
//   general  k_leader_tick_dense: jg_dense_cold_lds, rolled loops over a per-lane LDS column of
//            absolute heads (few registers: the kernel's occupancy is the hot path's);
//            k_leader_node_tick / k_leader_tick_dense_n: the group is handed to k_dense_slow (the
//            general state machine), which is always launched behind them;
//   skip     dead groups, non-leaders, irregular chains: decided from the flag word.
#pragma once
#include "jg_device.h"

#ifndef JG_BLOCK
#define JG_BLOCK 256
#endif

// The columns the hot path of the ack-only kernel touches, as kernel arguments of their own; the
// rest of JgDev reaches that kernel as a pointer to a device-resident copy and is dereferenced on
// the general path only (kernel arguments are loop-invariant loads the compiler hoists into the
// prologue of every wave: ~40 pointers cost SGPR spills and an occupancy step).
struct JgDenseHot {
  uint32_t* flags;
  uint64_t* mlag;
  uint64_t* head;
  uint64_t* blk_decisions;
  uint32_t G;
  // node tick only (leader.rs:234-245): term and heartbeat timer columns, config
  uint32_t hb_timeout, cfg_flags;
  uint64_t* term;
  uint64_t* heartbeat_time;
};
__host__ __device__ __forceinline__ JgDenseHot jg_dense_hot_of(const JgDev& d) {
  return JgDenseHot{d.flags, d.mlag, d.head, d.blk_decisions, d.G, d.hb_timeout, d.cfg_flags, d.term, d.heartbeat_time};
}


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

// The dense leader kernels count without a barrier and without a reduction in the common case: a
// wave whose lanes all took the same number of decisions in a group-step (the steady state: one
// append + R-1 acks each) adds count x popcount(lanes) to the workgroup's slot with ONE
// fire-and-forget atomic, issued before the step's stores so that its latency overlaps theirs (as
// the last instruction of the wave it held the wave's slot for a full trip to L2: 0.8 us per
// 1 M-group launch).  Uneven steps go through a per-lane counter that is reduced across the wave
// at the end - if any lane used it at all.
struct JgDecCount {
  uint32_t lane = 0;  // per-lane decisions of uneven steps
};
__device__ __forceinline__ void jg_count_add(uint64_t* slots, uint32_t total) {
  (void)__hip_atomic_fetch_add(&slots[blockIdx.x], (uint64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// `on` marks the lanes that took n decisions in this step
__device__ __forceinline__ void jg_count_step(uint64_t* slots, JgDecCount& c, bool on, uint32_t n) {
  const uint64_t m = __ballot(on);
  if (!m) return;
  const int first = __ffsll((long long)m) - 1;
  const uint32_t n0 = __builtin_amdgcn_readlane(n, first);
  if (__ballot(on && n != n0) == 0) {
    if ((int)(threadIdx.x & 63u) == first && n0) jg_count_add(slots, n0 * (uint32_t)__popcll(m));
  } else {
    c.lane += on ? n : 0u;
  }
}
__device__ __forceinline__ void jg_wave_count(uint64_t* slots, const JgDecCount& c) {
  uint32_t v = c.lane;
  if (__ballot(v != 0) == 0) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63u) == 0 && v) jg_count_add(slots, v);
}

// Issue every load of one group whose address is known without the flag word: the R slots of
// the ack block by constant index (the own slot carries the number of appends), the packed
// progress / commit word and the chain head.
template <int R>
__device__ __forceinline__ void jg_dense_load_acks(const uint64_t* __restrict__ acks, uint32_t G, uint32_t g,
                                                   uint64_t (&a)[R]) {
#pragma unroll
  for (int r = 0; r < R; r++) a[r] = __builtin_nontemporal_load(&acks[(size_t)r * G + g]);
}
template <int R, bool MAYBE_NO_ACKS = true>
__device__ __forceinline__ void jg_dense_load(const JgDenseHot& d, const uint64_t* __restrict__ acks, uint32_t g,
                                              uint64_t (&a)[R], uint64_t& mword, uint64_t& head) {
  if (!MAYBE_NO_ACKS || acks) {  // (the ack-only kernel always has an ack block: no branch in its loop)
    jg_dense_load_acks<R>(acks, d.G, g, a);
  } else {  // node tick without acks: nothing to append, no AppendResponse
#pragma unroll
    for (int r = 0; r < R; r++) a[r] = JG_NO_ACK;
  }
  mword = d.mlag[g];
  head = d.head[g];
}
// ---- the tick in lag space -----------------------------------------------------------------------
// The common case never leaves the packed representation.  With every field of the progress
// word un-escaped, every ack at or below the chain head and within 2^32 of it, and fewer than
// 2^20 appends, the whole tick is 32-bit arithmetic on lags below the head:
//   n appends + self-acks                     <=>  every lag += n, own lag = 0     leader.rs:177-197
//   match[r] = max(match[r], ack[r])          <=>  lag[r] = min(lag[r], head1 - ack[r])   (head1 = head0 + n)
//   increment() returned true (-> Replicate)  <=>  head1 - ack[r] < lag[r]         progress.rs:133-140
//   element R/2 of the heads sorted desc.     <=>  element R/2 of the lags sorted ascending
//   commit = max(commit, q)                   <=>  clag = min(clag + n, qlag)      leader.rs:89-92
// Old heads <= head0 and acks <= head0 + n (the appends are applied first, so that is the head an
// ack meets) is exactly the precondition of the fused evaluation above (no chain.commit panic
// possible), so it is exact; anything else (escaped field,
// forged ack above the head, a lag that no longer fits its field) returns false with nothing
// modified and the caller takes the general path.
template <int R>
struct JgLagTick {
  uint64_t w1, head1;  // new packed word, new chain head
  uint32_t nf;         // new flag word
  uint32_t l[R + 1];   // new lags below head1 (field R: commit)
  uint32_t adv;        // how far the commit index moved (leader.rs:89-92); meaningless when cwide
  bool cwide;          // the commit index BEFORE the tick was in the wide column (far behind: quorum had been lost)
};

// element K of v[0..R) sorted ascending
template <int R>
__device__ __forceinline__ uint32_t jg_kth_lag(const uint32_t (&v)[R + 1]) {
  constexpr int K = R / 2;
  if (R == 3) {  // median of 3
    const uint32_t lo = min(v[0], v[1]), hi = max(v[0], v[1]);
    return max(lo, min(hi, v[2]));
  } else if (R == 5) {  // median of 5: med3(e, max(min(a,b), min(c,d)), min(max(a,b), max(c,d)))
    const uint32_t x = max(min(v[0], v[1]), min(v[2], v[3]));
    const uint32_t y = min(max(v[0], v[1]), max(v[2], v[3]));
    const uint32_t lo = min(x, y), hi = max(x, y);
    return max(lo, min(hi, v[4]));
  } else {
    uint32_t q = 0;
#pragma unroll
    for (int j = 0; j < R; j++) {
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < R; i++) cnt += (v[i] < v[j] || (v[i] == v[j] && i < j)) ? 1 : 0;
      q = (cnt == K) ? v[j] : q;
    }
    return q;
  }
}

template <int R>
__device__ __forceinline__ bool jg_lag_tick(uint32_t s, uint32_t f, uint64_t w0, uint64_t head0, uint64_t n_app,
                                            const uint64_t (&a)[R], JgLagTick<R>& o, uint32_t& dec) {
  constexpr uint32_t B = 64u / (R + 1u);
  if (B > 21) return false;  // R = 1: 32-bit fields, take the general path
  constexpr uint32_t ESC = (uint32_t)((1ull << (B > 21 ? 21 : B)) - 1ull);
  constexpr uint32_t BEHIND = ESC - 1u, INF = 0xffffffffu;  // jg_lag_behind(); how a BEHIND slot sorts
  const uint32_t n = (uint32_t)n_app;
  bool bad = n_app >= JG_MAX_DENSE_APPENDS;
  const uint64_t head1 = head0 + n;  // the appends come first (leader.rs:177-197): acks meet the new head
  uint32_t incm = 0, somem = 0;  // per slot: increment() returned true / an ack arrived
#pragma unroll
  for (int r = 0; r < R; r++) {
    const uint32_t fr = (uint32_t)(w0 >> (r * B)) & ESC;
    const uint64_t ar = a[r];
    const uint64_t dk = head1 - ar;
    const bool self = (uint32_t)r == s;  // engine-uniform in the normal case: scalar
    const bool some = !self && ar != JG_NO_ACK;
    // A replica that is too far behind for its field (a follower that is down) stays where it
    // is until an ack arrives for it: its lag is larger than every in-range lag, which is all the
    // majority needs to know.  An ack for it, or the own slot in that state: exact compare, general path.
    const bool behind = fr == BEHIND;
    // lag of the ack below the NEW head; acks further than 2^32 behind are just "stale"
    uint32_t dl = (uint32_t)(dk >> 32) ? 0xffffffffu : (uint32_t)dk;
    dl = some ? dl : 0xffffffffu;
    const uint32_t frn = fr + n;  // the old progress head seen from the new chain head
    const bool inc = dl < frn;    // progress.rs:133-140: increment() returned true
    incm |= inc ? (1u << r) : 0u;
    somem |= some ? (1u << r) : 0u;
    // own slot: n self-acks leave its head at the new chain head
    uint32_t lo = (self && n) ? 0u : min(frn, dl);
    bad |= fr == ESC;                      // a head above the chain head (forged ack)
    bad |= behind ? some : lo >= BEHIND;   // an ack for a BEHIND slot / a lag leaving its field (wide column)
    bad |= some && ar > head1;             // an ack above the head it meets: replay, the reference may panic
    lo = (behind && !(self && n)) ? INF : lo;  // (an own BEHIND slot: its self-ack lands on the head all the same)
    o.l[r] = lo;
  }
  const uint32_t fc = (uint32_t)(w0 >> (R * B)) & ESC;
  const uint32_t lc = fc == BEHIND ? INF : fc + n;  // the commit index, possibly far behind (quorum was lost)
  const uint32_t ql = jg_kth_lag<R>(o.l);  // progress.rs:48-60; INF: the majority slot is a BEHIND one, q < commit
  const uint32_t nl = min(lc, ql);         // leader.rs:89-92
  o.l[R] = nl;
  bad |= fc == ESC;
  bad |= nl >= BEHIND;  // (both unknown, or the commit lag leaving its field)
  if (bad) return false;
  // Probe / Replicate bits: cleared where an ack did not advance, set where one did; the own
  // slot's last self-ack always advances (-> Replicate)
  uint32_t nf = (f & ~(somem << JGF_REPL_SHIFT)) | (incm << JGF_REPL_SHIFT);
  nf |= n ? (1u << (JGF_REPL_SHIFT + s)) : 0u;
  nf |= ql < lc ? JGF_COMMIT_KEY : 0u;  // chain.rs:198
  uint64_t w = 0;
#pragma unroll
  for (int r = 0; r <= R; r++) w |= (uint64_t)min(o.l[r], BEHIND) << (r * B);
  o.w1 = w;
  o.head1 = head1;
  o.nf = nf;
  o.cwide = fc == BEHIND;
  o.adv = lc - nl;  // (commit_after - commit_before = (head1 - nl) - (head1 - lc))
  dec += n + (uint32_t)__popc(somem);
  return true;
}

// jg_step_node: the acknowledgements that arrived BEFORE the tick's ClientRequest (`pre`: one bit per slot) met
// the chain head before the append.  The reference evaluated Leader::commit after each of them: the commit
// index they reached is what fsm_tx saw before the Notify (adv_pre), and chain.rs:197-202 holds them to the OLD
// head - one above it is the slow kernel's (replay in arrival order, a fault where the reference panics).
template <int R>
__device__ __forceinline__ bool jg_lag_pre(uint32_t s, uint64_t w0, uint64_t head0, const uint64_t (&a)[R], uint32_t pre,
                                           uint32_t& adv_pre) {
  constexpr uint32_t B = 64u / (R + 1u);
  constexpr uint32_t ESC = (uint32_t)((1ull << (B > 21 ? 21 : B)) - 1ull);
  constexpr uint32_t BEHIND = ESC - 1u, INF = 0xffffffffu;
  bool bad = false;
  uint32_t l[R + 1];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const uint32_t fr = (uint32_t)(w0 >> (r * B)) & ESC;
    const uint64_t ar = a[r];
    const bool some = (uint32_t)r != s && ((pre >> r) & 1u) && ar != JG_NO_ACK;
    const uint64_t dk = head0 - ar;
    uint32_t dl = (uint32_t)(dk >> 32) ? INF : (uint32_t)dk;
    dl = some ? dl : INF;
    bad |= some && ar > head0;
    l[r] = fr == BEHIND ? INF : min(fr, dl);  // (an ack for a BEHIND slot, an escaped field: jg_lag_tick refuses those)
  }
  const uint32_t fc = (uint32_t)(w0 >> (R * B)) & ESC;
  const uint32_t lc = fc == BEHIND ? INF : fc;
  const uint32_t ql = jg_kth_lag<R>(l);
  adv_pre = lc - min(lc, ql);  // (meaningless when the old commit index is in the wide column: not the hot path's)
  return !bad;
}

// The same tick on lags that are already unpacked (the T-tick kernel carries them in registers from
// tick to tick and packs once per launch): l[r] may exceed its field between ticks, only 2^30 is a
// hard bound here; the caller checks the field width when it packs.
template <int R>
__device__ __forceinline__ bool jg_lag_tick_regs(uint32_t s, uint32_t& nf_io, uint32_t (&l)[R + 1], uint64_t& head_io,
                                                 uint64_t n_app, const uint64_t (&a)[R], uint32_t& dec) {
  const uint64_t head0 = head_io;
  constexpr uint32_t INF = 0xffffffffu;  // a BEHIND slot (see jg_lag_tick)
  const uint32_t n = (uint32_t)n_app;
  bool bad = n_app >= JG_MAX_DENSE_APPENDS;
  const uint64_t head1 = head0 + n;
  uint32_t incm = 0, somem = 0;
  uint32_t nl[R + 1];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const uint32_t fr = l[r];
    const uint64_t ar = a[r];
    const uint64_t dk = head1 - ar;
    const bool self = (uint32_t)r == s;
    const bool some = !self && ar != JG_NO_ACK;
    const bool behind = fr == INF;
    uint32_t dl = (uint32_t)(dk >> 32) ? 0xffffffffu : (uint32_t)dk;
    dl = some ? dl : 0xffffffffu;
    const uint32_t frn = behind ? INF : fr + n;
    const bool inc = dl < frn;  // progress.rs:133-140
    incm |= inc ? (1u << r) : 0u;
    somem |= some ? (1u << r) : 0u;
    uint32_t lo = (self && n) ? 0u : min(frn, dl);
    bad |= behind ? some : lo >= (1u << 30);
    bad |= some && ar > head1;  // an ack above the head it meets: replay, the reference may panic
    nl[r] = (behind && !(self && n)) ? INF : lo;
  }
  const uint32_t lc = l[R] == INF ? INF : l[R] + n;
  const uint32_t ql = jg_kth_lag<R>(nl);  // progress.rs:48-60
  nl[R] = min(lc, ql);                    // leader.rs:89-92
  bad |= nl[R] >= (1u << 30);
  if (bad) return false;
  uint32_t nf = (nf_io & ~(somem << JGF_REPL_SHIFT)) | (incm << JGF_REPL_SHIFT);
  nf |= n ? (1u << (JGF_REPL_SHIFT + s)) : 0u;
  nf |= ql < lc ? JGF_COMMIT_KEY : 0u;  // chain.rs:198
  nf_io = nf;
#pragma unroll
  for (int r = 0; r <= R; r++) l[r] = nl[r];
  head_io = head1;
  dec += n + (uint32_t)__popc(somem);
  return true;
}

// ---- deferral to the slow kernel: wave-aggregated append to sharded lists -----------------------
// Groups a dense kernel cannot serve in registers (irregular chains, exceptional inputs) are
// handed to a slow kernel launched right behind it.  The lanes of a wave that defer are
// counted with one __ballot / __popcll, the first of them reserves the wave's slots with ONE
// atomic on the counter of the workgroup's shard (JG_SHARDS counters: ~16 workgroups each at
// 1 M groups, so no hot address), and every deferring lane writes its group at base + rank.
// Costs nothing when nobody defers (the ballot is wave-uniform).  `tag` is or-ed into the entry.
#define JG_SHARDS 256
#define JG_DEFER_TICK_ONLY 0x80000000u  // follower half: inputs already applied, Tick pending
__device__ __forceinline__ void jg_defer_push(const JgDev& d, uint32_t g, bool want, uint32_t tag = 0) {
  const uint64_t mask = __ballot(want);
  if (!mask) return;
  const uint32_t shard = blockIdx.x & (JG_SHARDS - 1);
  const uint32_t lane = threadIdx.x & 63u;
  const int first = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if ((int)lane == first) base = atomicAdd(&d.slow_cnt[shard], (uint32_t)__popcll(mask));
  base = __shfl(base, first, 64);
  if (want) {
    const uint32_t i = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (i < d.slow_cap) d.slow_list[(size_t)shard * d.slow_cap + i] = g | tag;
    else *d.err = 4;
    *d.deferred_seen = 1;  // lets the host verify that the slow kernel was scheduled
  }
}

// The dense LEADER kernels mark deferred groups in a bitmap instead: a wave serves 64 consecutive
// groups, so its __ballot is the word of the bitmap, or-ed in by one lane with a fire-and-forget
// atomic - nothing comes back, the wave does not wait.  (The list append above returns the slot
// base to the wave: at 1 % deferred groups per tick half the waves of a launch sat out a trip to
// L2 for it and the configs[4] tick of this kernel took 39 us instead of 11.)  k_dense_slow turns
// its shard of the bitmap into its list.
__device__ __forceinline__ void jg_defer_mark_in(uint64_t* bits, const JgDev& d, uint32_t g, bool want) {
  const uint64_t mask = __ballot(want);
  if (!mask) return;
  const int first = __ffsll((long long)mask) - 1;
  if ((int)(threadIdx.x & 63u) == first) {
    (void)__hip_atomic_fetch_or(&bits[g >> 6], mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *d.deferred_seen = 1;  // lets the host verify that the slow kernel was scheduled
  }
}
__device__ __forceinline__ void jg_defer_mark(const JgDev& d, uint32_t g, bool want) { jg_defer_mark_in(d.defer_bits, d, g, want); }

// Node-tick extras of the leader kernel (jg_step_dense_leader): HeartbeatResponse input and the
// Tick's outbox (leader.rs:234-245).  All pointers may be null (= plain jg_step_dense_acks).
// Logical time and step numbers of a closed loop that is replayed as a hipGraph
// (jg_dense_cluster_rounds): kernel arguments are frozen in a graph, so the clock lives in device
// memory and the graph's first node advances it.
// The clock advances itself - no kernel of its own in the replayed round (an empty launch costs what a
// tenth of a follower half costs).  Two values, two index words, each index written only by a kernel
// during which nobody reads it:
//   the leader half (first kernel of a round)  reads v[idx_lead]; its first thread sets idx_rest = idx_lead
//   the leader's slow kernel (second)          reads v[idx_rest]; its first thread, when the kernel's own
//                                              work is done, writes v[idx_rest ^ 1] = this round + dt
//                                              (every node one step further) and points idx_lead at it
//   every later kernel of the round            reads v[idx_rest]
struct JgClockVal {
  uint64_t now;
  uint32_t seq[JG_MAX_REPLICAS];  // per node of the cluster
};
struct JgClock {
  uint32_t idx_lead, idx_rest;
  uint64_t dt;
  uint32_t n_nodes, seq_step;  // seq_step: steps a node takes per round (1; 2 with per-partition leadership: both halves)
  JgClockVal v[2];
};
__device__ __forceinline__ void jg_clock_read(const JgClock* c, uint32_t slot, uint64_t& now, uint32_t& seq) {
  const JgClockVal& v = c->v[c->idx_rest & 1u];
  now = v.now, seq = v.seq[slot];
}
#define JG_AEC_INDIVIDUAL 0xfffffffffffffffeull  // JgLeaderNode::o_aec: "see the rows" (no JG_AE word: a range start key stays below JG_MAILBOX_NONE; not JG_NO_ACK: "nothing for anybody")
struct JgLeaderNode {
  JgClock* clock;              // non-null: `now` and the step number come from here (slot clock_slot)
  uint32_t clock_slot;
  uint32_t mask_offers;        // 1: the own slot's append count is an OFFER, good only where this node leads the group (a routed
                               //    round's ClientRequests: at a leaderless replica the reference queues them, follower.rs:258-270 -
                               //    not a dense append); elsewhere it is not looked at.  0: asking a non-leader to append is
                               //    JG_FAULT_ENGINE_DENSE_NONLEADER (jg_step_dense_acks)
  uint32_t ack_stride;         // 1, or 0: nothing came in (the `acks` argument points at an all-ones word)
  uint32_t packed;             // 1: `acks` holds jg_leader_inbox answer words (JG_ANSWER), not bare heads
  const uint64_t* hbr_commit;  // [R][G] (slow kernel only)
  jg_leader_beat* o_beat;      // [G]     null: no Tick
  uint64_t* o_ae;              // [R][G]
  // a jg_dense_cluster's mailboxes only (null otherwise): ONE AppendEntries word per group where every follower is sent
  // the same one - the steady state: all of them in Replicate at the same head, or nobody sent anything - and
  // JG_AEC_INDIVIDUAL where they differ (then, and only then, the rows of o_ae hold this round's words).  The leader half
  // wrote 16 + 8 (R - 1) + 8 bytes per group, 59 MB of its 137 MB at 1 M x 5; now 32, and the R - 1 follower halves read one
  // shared line instead of R - 1 private ones.
  uint64_t* o_aec;             // [G]
  uint64_t now;
  // jg_step_node: what the step pushed on fsm_tx, as one word per group (jg_node.h JGN_FSM_*); null otherwise
  uint32_t* fsm_delta;         // [G]
  uint64_t* fsm_prev;          // [G] the commit index before the step where the word says JGN_FSM_WIDE
  uint64_t* fsm_mid;           // [G] ... and at the moment the ClientRequest was applied
  // jg_step_node: the arrival index + 1 of the row behind every inbox entry ([2R][G], jg_node.h JgNodeCols::arr): the
  // slow kernel replays a group's commands in that order; slots in `col_mask` spoke a column (after the rows)
  const uint32_t* arr;
  uint32_t col_mask, pad2_;
  // a cluster with per-partition leadership (jg_dense_cluster_create, JG_CLUSTER_ANY_LEADER): the mailboxes are the
  // CLUSTER's, indexed by group - whoever leads a group reads its inbox and writes its outbox.  owner[g] = the slot
  // whose Tick travels in the columns this round (k_cluster_claim: the lowest slot that leads g, 0xff: nobody);
  // a leader that is not the owner (two terms' leaders in one round) has no inbox and sends its Tick as rows.
  // offered[g] = the own slot's word (JG_ANSWER(#ClientRequests, none)) for whoever owns g.
  const uint8_t* owner;
  const uint64_t* offered;
  // jg_step_node, JG_NODE_ASYNC: bit g of sparse_bits = group g's rows take the general path.  sparse_mode 1: the half
  // leaves those groups alone (their rows have not been applied yet); 2: it serves ONLY those (the catch-up pass)
  const uint64_t* sparse_bits;
  uint32_t sparse_mode, pad3_;
};
#define JG_OWNER_NONE 0xffu
// the follower's side: its AppendEntries word of group g (aec: the cluster's common column or null; ae: its row of the block)
__device__ __forceinline__ uint64_t jg_ae_word_for(const uint64_t* __restrict__ aec, const uint64_t* __restrict__ ae, uint32_t g) {
  if (!aec) return __builtin_nontemporal_load(&ae[g]);
  const uint64_t c = aec[g];  // (every follower node reads this line: not a streaming load)
  return __builtin_expect(c == JG_AEC_INDIVIDUAL, 0) ? ae[g] : c;
}
__device__ __forceinline__ bool jg_sparse_skip(const uint64_t* bits, uint32_t mode, uint32_t g) {
  const bool sp = (bits[g >> 6] >> (g & 63u)) & 1ull;
  return mode == 1u ? sp : !sp;
}
#define JG_FSM_APPENDED_BIT (1u << 31)
#define JG_FSM_WIDE_BIT (1u << 30)
#define JG_FSM_FOLLOWER_BIT (1u << 29)
#define JG_FSM_PRE_SHIFT 14            // bits 14-27: the commit advance BEFORE the Notify; bits 0-13: of the whole step
#define JG_FSM_ADV_MASK 0x3fffu
// jg_step_node: the own slot's append count (0 or 1) carries, from bit JG_NODE_PRE_SHIFT up, one bit per slot whose
// AppendResponse arrived BEFORE the ClientRequest (k_node_route)
#define JG_NODE_PRE_SHIFT 32

// jg_leader_inbox answer word -> AppendResponse head (JG_NO_ACK: none) / HeartbeatResponse code
__device__ __forceinline__ uint64_t jg_answer_ack(uint64_t w) {
  const uint64_t v = w >> 8;
  return v == JG_MAILBOX_NONE ? JG_NO_ACK : v;
}
__device__ __forceinline__ uint32_t jg_answer_hb(uint64_t w) { return (uint32_t)w & 0xffu; }

// What a lane must do with one group after looking at its flag word.
enum { JG_DENSE_SKIP = 0, JG_DENSE_RUN = 1, JG_DENSE_DEFER = 2 };

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
  if (n_app >= JG_MAX_DENSE_APPENDS) {  // own slot outside its domain (JG_NO_ACK included): nothing is applied
    d.flags[g] = f | (JG_FAULT_ENGINE_DENSE_APPENDS << JGF_FAULT_SHIFT);
    jg_push_fault(d, g, JG_FAULT_ENGINE_DENSE_APPENDS, seq);
    return JG_DENSE_SKIP;
  }
  // irregular chain: k_dense_slow, launched right behind this kernel, replays the tick
  // through the general state machine
  if (!(f & JGF_FAST)) return JG_DENSE_DEFER;
  return JG_DENSE_RUN;
}

// ---- one tick per launch ------------------------------------------------------------------------
// UNIFORM: every group of the engine has own slot `us` (the normal case: a node has one
// NodeId).  Otherwise the own slot comes from the flag word and the other loads wait for it.
// (A two-groups-per-lane variant with 16-B accesses measured no faster — the kernel is
// bandwidth-, not issue-bound: profiles/README.md round 1 — and was dropped.)
// SKIP_OWN (here and below): every group of the engine has own slot `s`: row s of the `ae` block - nobody's mail,
// JG_NO_ACK by definition - is not written at all (8 of the node tick's 64 written bytes per group; the block's
// owner fills the row once).
template <int R, bool SKIP_OWN>
__device__ __forceinline__ void jg_dense_outbox_none(uint32_t G, const JgLeaderNode& nd, uint32_t g, uint32_t s) {
  nd.o_beat[g] = jg_leader_beat{0, JG_NO_ACK};
  if (nd.o_aec) {  // (a cluster's mailboxes: nothing for anybody is one word; k_dense_slow fills the rows of what it serves in columns)
    nd.o_aec[g] = JG_NO_ACK;
    return;
  }
#pragma unroll
  for (int r = 0; r < R; r++)
    if (!(SKIP_OWN && (uint32_t)r == s)) nd.o_ae[(size_t)r * G + g] = JG_NO_ACK;
}
// Command::Tick of a FAST leader into the outbox columns (leader.rs:234-245): heartbeat() if due,
// then replicate() — per other slot the range start key (= its progress head) and the number
// of blocks after it (Probe: nth(1) -> 1, Replicate: skip(1).take(5), leader.rs:135,152-157).
// `mo_of(r)` = progress head of slot r; returns the flag word (with the Q9 fault if it was raised).
template <int R, bool SKIP_OWN, class MoOf>
__device__ __forceinline__ uint32_t jg_dense_leader_tick(const JgDenseHot& h, const JgDev* dp, const JgLeaderNode& nd,
                                                         uint32_t g, uint32_t seq, uint32_t s, uint64_t term,
                                                         uint64_t hbt, uint64_t head, uint64_t commit, uint32_t nf,
                                                         MoOf mo_of) {
  const uint32_t G = h.G;
  if (head >= JG_MAILBOX_NONE) {  // 56-bit block ids in mailbox words: loud, never wrong
    jg_dense_outbox_none<R, SKIP_OWN>(G, nd, g, s);
    jg_push_fault(*dp, g, JG_FAULT_ENGINE_MAILBOX_RANGE, seq);
    return nf | (JG_FAULT_ENGINE_MAILBOX_RANGE << JGF_FAULT_SHIFT);
  }
  // First every word, then every store.  A progress head in the wide column (BEHIND) is a load under a
  // branch, and the wait the compiler puts behind that branch counts the stores issued so far as well
  // (one counter for loads and stores here): with the stores interleaved, every follower's word waited
  // for the previous follower's store to be acknowledged - five store round trips in a row per wave.
  const bool due = (nd.now - hbt) > (uint64_t)h.hb_timeout;  // leader.rs:78-84
  const uint64_t hb = due ? commit : JG_NO_ACK;              // leader.rs:44-51
  const bool key_in_range = (nf & JGF_COMMIT_KEY) && !(h.cfg_flags & JG_CFG_SEPARATE_COMMIT_KEY);
  bool dead = false;
  uint64_t word[R];
#pragma unroll
  for (int r = 0; r < R; r++) {  // ascending slot = the order of replicate()'s loop
    uint32_t n = JG_AE_NONE;
    uint64_t from = 0;
    if ((uint32_t)r != s && !dead) {
      const bool repl = (nf >> (JGF_REPL_SHIFT + r)) & 1u;
      const uint32_t want = repl ? JG_MAX_INFLIGHT + 1 : 2;  // items consumed by the range iterator
      from = mo_of(r);
      const uint64_t avail = from <= head ? head - from + 1 : 0;  // block keys >= from (FAST form)
      const uint32_t cnt = avail >= want ? want : (uint32_t)avail;
      if (cnt < want && key_in_range) {  // the iterator runs into the "commit" key: chain.rs:219-226 (Q9)
        dead = true;
        from = 0;
        nf |= JG_FAULT_RANGE_HIT_COMMIT_KEY << JGF_FAULT_SHIFT;
      } else {
        n = cnt ? cnt - 1 : 0;
      }
    }
    word[r] = n == JG_AE_NONE ? JG_NO_ACK : JG_AE(from, n);
  }
  if (dead) jg_push_fault(*dp, g, JG_FAULT_RANGE_HIT_COMMIT_KEY, seq);
  if (due) h.heartbeat_time[g] = nd.now;
  nd.o_beat[g] = jg_leader_beat{term, hb};  // one 16-byte store
  if (nd.o_aec) {  // a cluster's mailboxes: the followers' words are one word where they agree (JgLeaderNode::o_aec)
    const uint64_t first = s == 0 ? word[R > 1 ? 1 : 0] : word[0];
    bool same = true;
#pragma unroll
    for (int r = 0; r < R; r++) same = same && ((uint32_t)r == s || word[r] == first);
    nd.o_aec[g] = same ? first : JG_AEC_INDIVIDUAL;
    if (__builtin_expect(same, 1)) return nf;
  }
#pragma unroll
  for (int r = 0; r < R; r++)
    if (!(SKIP_OWN && (uint32_t)r == s)) nd.o_ae[(size_t)r * G + g] = word[r];
  return nf;
}

// ---- one tick per launch ------------------------------------------------------------------------
// UNIFORM: every group of the engine has own slot `us` (the normal case: a node has one
// NodeId).  Otherwise the own slot comes from the flag word and the other loads wait for it.
// NODE: jg_step_dense_leader — HeartbeatResponse input and / or the Tick's outbox.
// (A two-groups-per-lane variant with 16-B accesses measured no faster — the kernel is
// bandwidth-, not issue-bound: profiles/README.md round 1 — and was dropped.)
// Everything one group-step reads, as issued loads (no use of the values: the loads of the next
// group can be in flight while this one is evaluated).
template <int R>
struct JgDenseIn {
  uint32_t f;
  uint64_t a[R], w, head;
  uint64_t term, hbt;  // NODE
  uint32_t own;        // ANY: owner[g]
};
template <int R, bool UNIFORM, bool NODE, bool ANY = false>
__device__ __forceinline__ void jg_dense_issue(const JgDenseHot& h, const JgDev* dp,
                                               const uint64_t* __restrict__ acks, uint32_t us,
                                               const JgLeaderNode& nd, bool emit, uint32_t g, JgDenseIn<R>& in) {
  in.f = h.flags[g];
  in.term = in.hbt = 0;
  if (!NODE) {
    jg_dense_load<R, false>(h, acks, g, in.a, in.w, in.head);
  } else {
    // Every load of the group unconditionally and back to back - ONE round trip.  An absent input
    // (no ack block, no HeartbeatResponses) is not a branch around its loads (the compiler waits for
    // the loads issued so far at every branch: the node tick took eight dependent trips per group,
    // 38 us per 1 M groups) but a stride of 0 into an all-ones word: "nothing" for everybody.  One word
    // per slot carries both answers of that follower (JG_ANSWER): R loads, not 2R + R byte loads - the
    // kernel's time follows the number of its memory instructions (profiles/README.md).
#pragma unroll
    for (int r = 0; r < R; r++)
      in.a[r] = (ANY && (uint32_t)r == us) ? nd.offered[g] : __builtin_nontemporal_load(&acks[((size_t)r * h.G + g) * nd.ack_stride]);
    if (ANY) in.own = nd.owner[g];
    in.w = h.mlag[g];
    in.head = h.head[g];
    in.term = h.term[g];
    in.hbt = h.heartbeat_time[g];
  }
  // keep the flag load up here, in the same round trip as the others: without a use the
  // compiler sinks it behind the hot-path test (a second, dependent trip to HBM per group)
  asm volatile("" ::"v"(in.f));
  if (NODE) asm volatile("" ::"v"(in.term), "v"(in.hbt));
  __builtin_amdgcn_sched_barrier(0);
}

// ---- the general path of the ack-only kernel, in memory form ------------------------------------
// The tick on absolute 64-bit progress heads (fused when no chain.commit panic is possible, else
// one Leader::commit per append / ack with a fault exactly where the reference would panic),
// written as rolled loops over a per-lane LDS column of the R progress
// heads: a handful of registers instead of ~90, because the register allocation of a kernel is the
// maximum over all its paths and this one is taken by almost no group (escaped lag fields, acks
// above the head).  Its inputs come from the registers the hot path loaded; the ack block is staged in LDS too.
template <int R>
__device__ __forceinline__ uint64_t jg_lds_kth(const uint64_t (*sm)[JG_BLOCK]) {
  // element R/2 of the heads sorted descending (progress.rs:48-60) by rank counting
  const uint32_t t = threadIdx.x;
  uint64_t q = 0;
#pragma clang loop unroll(disable)
  for (int j = 0; j < R; j++) {
    const uint64_t vj = sm[j][t];
    int cnt = 0;
#pragma clang loop unroll(disable)
    for (int i = 0; i < R; i++) {
      const uint64_t vi = sm[i][t];
      cnt += (vi > vj || (vi == vj && i < j)) ? 1 : 0;
    }
    q = (cnt == R / 2) ? vj : q;
  }
  return q;
}

template <int R, bool UNIFORM>
__device__ __forceinline__ void jg_dense_cold_lds(const JgDev& d, const uint64_t* __restrict__ acks, uint32_t seq,
                                               uint32_t s, uint32_t f, uint64_t n_app, uint32_t g,
                                               const uint64_t (&a_in)[R], uint64_t w0, uint64_t head0, JgDecCount& dec,
                                               uint64_t (*sm)[JG_BLOCK]) {
  const uint32_t G = d.G, t = threadIdx.x;
  const uint32_t B = jg_lag_bits(R);
  const uint64_t esc = jg_lag_esc(R);
  uint64_t(*sa)[JG_BLOCK] = sm + R;  // the ack block of this lane, staged from the registers the hot
#pragma unroll                        // path loaded it into (no second trip to HBM per rolled iteration)
  for (int r = 0; r < R; r++) sa[r][t] = ((uint32_t)r == s || !acks) ? JG_NO_ACK : a_in[r];
  // packed lags -> absolute progress heads (escaped fields: the wide column)
  uint64_t hi = 0;
#pragma clang loop unroll(disable)
  for (int r = 0; r < R; r++) {
    const uint64_t fl = (w0 >> (r * B)) & esc;
    const uint64_t v = jg_lag_wide(fl, R) ? d.match_wide[(size_t)r * G + g] : head0 - fl;
    sm[r][t] = v;
    hi = v > hi ? v : hi;
    const uint64_t a = sa[r][t];
    hi = (a != JG_NO_ACK && a > hi) ? a : hi;
  }
  const uint64_t fc = (w0 >> (R * B)) & esc;
  const uint64_t commit0 = jg_lag_wide(fc, R) ? d.commit[g] : head0 - fc;
  uint64_t commit = commit0, head = head0;
  uint32_t nf = f, fault = 0, dc = 0;
  const bool fused = hi <= head0;  // no chain.commit panic possible: one majority evaluation
  // appends with their self-acks (leader.rs:177-197); replayed one at a time unless fused
  if (fused) {
    head = head0 + n_app;
    if (n_app) {
      const bool inc = sm[s][t] < head;
      if (inc) sm[s][t] = head;
      nf = inc ? (nf | (1u << (JGF_REPL_SHIFT + s))) : (nf & ~(1u << (JGF_REPL_SHIFT + s)));
      dc += (uint32_t)n_app;
    }
  } else {
    for (uint64_t i = 0; i < n_app && !fault; i++) {
      head += 1;
      const bool inc = sm[s][t] < head;
      if (inc) sm[s][t] = head;
      nf = inc ? (nf | (1u << (JGF_REPL_SHIFT + s))) : (nf & ~(1u << (JGF_REPL_SHIFT + s)));
      dc += 1;
      const uint64_t q = jg_lds_kth<R>(sm);
      if (q > commit) {
        if (q <= head) commit = q;
        else fault = JG_FAULT_COMMIT_MISSING_BLOCK;  // chain.rs:197-202
      }
    }
  }
  // the other slots' acks, ascending (progress.rs:76-94,133-140); one Leader::commit each unless fused
#pragma clang loop unroll(disable)
  for (int r = 0; r < R; r++) {
    if (fault) continue;
    const uint64_t a = sa[r][t];  // (JG_NO_ACK in the own slot)
    if (a == JG_NO_ACK) continue;
    const bool inc = sm[r][t] < a;
    if (inc) sm[r][t] = a;
    nf = inc ? (nf | (1u << (JGF_REPL_SHIFT + r))) : (nf & ~(1u << (JGF_REPL_SHIFT + r)));
    dc += 1;
    if (!fused) {
      const uint64_t q = jg_lds_kth<R>(sm);
      if (q > commit) {
        if (q <= head) commit = q;
        else fault = JG_FAULT_COMMIT_MISSING_BLOCK;
      }
    }
  }
  if (fused) {
    const uint64_t q = jg_lds_kth<R>(sm);  // progress.rs:48-60
    commit = q > commit ? q : commit;      // leader.rs:89-92
  }
  if (fault) {
    nf |= fault << JGF_FAULT_SHIFT;
    jg_push_fault(d, g, fault, seq);
  }
  if (commit != commit0) nf |= JGF_COMMIT_KEY;  // chain.rs:198
  dec.lane += dc;
  // store what changed: the lags re-packed against the new head
  uint64_t w = 0;
#pragma clang loop unroll(disable)
  for (int r = 0; r < R; r++) {
    const uint64_t v = sm[r][t];
    const uint64_t fl = jg_lag_encode(v, head, R);
    if (jg_lag_wide(fl, R)) d.match_wide[(size_t)r * G + g] = v;
    w |= fl << (r * B);
  }
  const uint64_t fl = jg_lag_encode(commit, head, R);
  if (jg_lag_wide(fl, R) && (commit != commit0 || !jg_lag_wide(fc, R))) d.commit[g] = commit;
  w |= fl << (R * B);
  if (w != w0) d.mlag[g] = w;
  if (head != head0) d.head[g] = head;
  if (nf != f) d.flags[g] = nf;
}

// DEFER: the host has k_dense_slow scheduled behind this launch: everything that is not served in
// lag space goes there.  A compile-time switch on purpose: as a run-time flag it cost the hot path
// of the 1 M x 5 launch 0.5 us (the compiler prepared general-path operands ahead of the branch).
template <int R, bool UNIFORM, bool NODE, bool DEFER, bool FSM = false, bool ANY = false>
__device__ __forceinline__ void jg_dense_group(const JgDenseHot& h, const JgDev* dp,
                                               const uint64_t* __restrict__ acks, uint32_t seq, uint32_t us, const JgLeaderNode& nd, bool emit, uint32_t g,
                                               const JgDenseIn<R>& in, JgDecCount& dec, uint64_t (*sm)[JG_BLOCK]) {
  const uint32_t f = in.f;
  const uint32_t s = UNIFORM ? us : (f & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
  const uint64_t mword0 = in.w, head0 = in.head, term = in.term, hbt = in.hbt;
  bool hbr_trigger = false;
  uint64_t a[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    a[r] = NODE ? jg_answer_ack(in.a[r]) : in.a[r];
    // leader.rs:222-231: a response without the commit makes the leader replicate again
    if (NODE) hbr_trigger |= (uint32_t)r != s && jg_answer_hb(in.a[r]) == 0;
  }
  uint64_t n_app = a[0];
#pragma unroll
  for (int r = 1; r < R; r++) n_app = (uint32_t)r == s ? a[r] : n_app;
  if (NODE && !nd.ack_stride) n_app = 0;
  // per-partition leadership: the group's mailboxes belong to its owner; a leader that is not (another term's leader
  // in the same round) has no inbox, appends nothing and sends its Tick as rows (k_dense_slow)
  const bool mine = !ANY || in.own == s;
  if (ANY && !mine) {
#pragma unroll
    for (int r = 0; r < R; r++) a[r] = JG_NO_ACK;
    n_app = 0;
    hbr_trigger = false;
  }
  uint32_t pre = 0;  // jg_step_node: slots whose AppendResponse arrived before the ClientRequest
  if (FSM) {
    pre = (uint32_t)(n_app >> JG_NODE_PRE_SHIFT) & 0xffu;
    n_app &= (1ull << JG_NODE_PRE_SHIFT) - 1ull;
  }
  // ---- hot path: a healthy leader in FAST form whose tick stays in lag space --------------------
  // straight-line 32-bit arithmetic, evaluated for every lane; everything else is behind one
  // (normally wave-uniform, not taken) branch
  JgLagTick<R> lt;
  lt.cwide = false, lt.adv = 0;
  uint32_t dl = 0;
  bool hot = (f & (JGF_FAULT_MASK | JGF_ROLE_MASK | JGF_FAST)) == (JG_ROLE_LEADER | JGF_FAST);
  // A healthy leader whose chain is not in FAST form but still the run [0, top] - a restarted leader, head at its
  // commit index below the top of what the store kept, or id_gen off the head (Q8: it cannot append any more, but it
  // counts acknowledgements, commits and ticks for as long as it leads): its lags are relative to that top
  // (jg_lag_base_is_run_hi), every valid ack is at or below it, and the tick is the same arithmetic.  An append
  // there is the slow kernel's (the Q8 fault).  One more load, behind a branch only such a group's wave takes.
  uint64_t base0 = head0;
  if (NODE) {
    const bool runx = (f & (JGF_FAULT_MASK | JGF_ROLE_MASK | JGF_FAST | JGF_WIN_MASK | JGF_NO_GENESIS)) == JG_ROLE_LEADER && n_app == 0;
    if (__builtin_expect(runx, 0)) {
      if (!(f & JGF_RUN)) base0 = dp->run_hi[g];
      hot = true;
    }
  }
  if (NODE) hot = hot && !hbr_trigger;
  if (ANY) hot = hot && mine;
  hot = jg_lag_tick<R>(s, f, mword0, base0, n_app, a, lt, dl) && hot;
  // fsm rows come from the (appended?, commit advance) word: one Notify at most, and a commit index whose old
  // value is in the packed word - anything else is the general state machine's (k_dense_slow)
  if (FSM) hot = hot && n_app <= 1 && !lt.cwide;
  uint32_t adv_pre = 0;
  if (FSM && __builtin_expect(pre != 0, 0)) hot = jg_lag_pre<R>(s, mword0, base0, a, pre, adv_pre) && hot;
  // (the node tick counts behind its stores: the counter's atomic would otherwise be one more thing the
  // waits inside the Tick's emission wait for)
  if (!NODE) jg_count_step(h.blk_decisions, dec, hot, dl);
  if (__builtin_expect(hot, 1)) {
    if (emit)  // Command::Tick into the outbox; may raise the Q9 fault
      lt.nf = jg_dense_leader_tick<R, UNIFORM>(h, dp, nd, g, seq, s, term, hbt, lt.head1, lt.head1 - lt.l[R], lt.nf, [&](int r) {
        return lt.l[r] == 0xffffffffu ? dp->match_wide[(size_t)r * h.G + g] : lt.head1 - lt.l[r];  // BEHIND: wide column
      });
    if (lt.w1 != mword0) h.mlag[g] = lt.w1;
    if (lt.head1 != base0) h.head[g] = lt.head1;  // (an append: FAST form, base0 = the head)
    if (lt.nf != f) h.flags[g] = lt.nf;
    if (FSM) {  // (a Q9 fault raised by the Tick comes after the appends and acks: their rows stand)
      uint32_t w = (n_app ? JG_FSM_APPENDED_BIT : 0u) | lt.adv | (adv_pre << JG_FSM_PRE_SHIFT);
      if (__builtin_expect(lt.adv > JG_FSM_ADV_MASK, 0)) {  // (16-bit lag fields at R = 3: a commit index that jumps by 2^14 blocks)
        const uint64_t commit0 = lt.head1 - lt.l[R] - lt.adv;
        nd.fsm_prev[g] = commit0;
        nd.fsm_mid[g] = commit0 + adv_pre;
        w = (n_app ? JG_FSM_APPENDED_BIT : 0u) | JG_FSM_WIDE_BIT;
      }
      if (w) nd.fsm_delta[g] = w;  // (the column was zeroed by the step's prefill)
    }
    if (NODE) jg_count_step(h.blk_decisions, dec, true, dl);  // (the ballots see the lanes of this branch: the hot ones)
    return;
  }
  const JgDev& d = *dp;  // (the ack-only kernel: loads from the device copy, general path only)
  // ---- everything else ---------------------------------------------------------------------------
  // dead groups, non-leaders and irregular chains are decided from the flag word in registers
  // (cold: a non-leader's append count under mask_offers is nobody's request - what classify does with a non-leader that
  // was asked nothing: no fault, no deferral, "nothing" in the outbox)
  if (NODE && !ANY && nd.mask_offers && (f & JGF_ROLE_MASK) != JG_ROLE_LEADER) {
    if (emit) jg_dense_outbox_none<R, UNIFORM>(h.G, nd, g, s);
    return;
  }
  int cls = jg_dense_classify(d, g, f, n_app, seq);
  // node tick: whatever is not served in lag space (a HeartbeatResponse without the commit, an
  // escaped lag field, an ack above the head) goes to k_dense_slow, which is always launched
  // behind a node tick and runs HeartbeatResponses, appends, acks and the Tick of these groups
  // through the general state machine (columns for a FAST chain, rows otherwise)
  // the ack-only kernel does the same whenever the host has k_dense_slow scheduled behind it anyway
  // (DEFER): ONE lane of a wave on the rolled LDS path below keeps the whole wave for several
  // microseconds (1 % of the groups there tripled the launch: profiles/README.md round 2)
  if ((NODE || DEFER) && cls == JG_DENSE_RUN) cls = JG_DENSE_DEFER;
  jg_defer_mark(d, g, cls == JG_DENSE_DEFER);
  if (NODE) {
    if (emit && mine) jg_dense_outbox_none<R, UNIFORM>(h.G, nd, g, s);  // (ANY: another node's mail is not this one's to erase)
    return;
  }
  if (DEFER || cls != JG_DENSE_RUN) return;
  // the ack-only kernel: rolled loops over LDS, so that the kernel's register allocation
  // (= its occupancy) is the hot path's
  *d.cold_seen = 1;  // read back at the next synchronisation point: the host then schedules k_dense_slow
  jg_dense_cold_lds<R, UNIFORM>(d, acks, seq, s, f, n_app, g, a, mword0, head0, dec, sm);
}

// Grid-stride loop over the groups (a software prefetch of the next group's loads measured no gain
// and cost 13 VGPRs: profiles/README.md).
template <int R, bool UNIFORM, bool NODE, bool DEFER, bool FSM = false, bool ANY = false>
__device__ __forceinline__ JgDecCount jg_dense_tick_body(const JgDenseHot& h, const JgDev* dp,
                                                       const uint64_t* __restrict__ acks, uint32_t seq, uint32_t us,
                                                       const JgLeaderNode& nd, uint64_t (*sm)[JG_BLOCK]) {
  const uint32_t G = h.G, stride = gridDim.x * JG_BLOCK;
  const bool emit = NODE && nd.o_beat != nullptr;
  JgDecCount dec;
  uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x;
  for (; g < G; g += stride) {
    if (FSM && nd.sparse_bits && jg_sparse_skip(nd.sparse_bits, nd.sparse_mode, g)) continue;
    if (ANY) {  // the flag word first: a wave that leads none of its 64 groups loads nothing else (a node leads G / R of them)
      const uint32_t f0 = h.flags[g];
      if (__ballot((f0 & (JGF_ROLE_MASK | JGF_FAULT_MASK)) == JG_ROLE_LEADER) == 0) continue;
    }
    JgDenseIn<R> in;
    jg_dense_issue<R, UNIFORM, NODE, ANY>(h, dp, acks, us, nd, emit, g, in);
    jg_dense_group<R, UNIFORM, NODE, DEFER, FSM, ANY>(h, dp, acks, seq, us, nd, emit, g, in, dec, sm);
  }
  return dec;
}

template <int R, bool DEFER>
__global__ __launch_bounds__(JG_BLOCK) void k_leader_tick_dense(JgDenseHot h, const JgDev* __restrict__ dp,
                                                                               const uint64_t* __restrict__ acks,
                                                                               uint32_t seq, int us) {
  // progress heads + acks of the (rare) groups on the general path (none of it in the DEFER build)
  __shared__ uint64_t sm[DEFER ? 1 : 2 * R][JG_BLOCK];
  JgDecCount dec;
  JgLeaderNode nd{};
  if (us >= 0) dec = jg_dense_tick_body<R, true, false, DEFER>(h, dp, acks, seq, (uint32_t)us, nd, sm);
  else dec = jg_dense_tick_body<R, false, false, DEFER>(h, dp, acks, seq, 0, nd, sm);
  jg_wave_count(h.blk_decisions, dec);
}

// jg_step_dense_leader: the same tick with HeartbeatResponses in and / or the Tick's outbox out
// FSM (jg_step_node): the step's fsm_tx output is left behind as one word per group (nd.fsm_delta)
// (experiment: -DJG_LEADER_WAVES=n / -DJG_FOLLOWER_WAVES=n cap the dense halves' occupancy - what they would run at
// if they carried the general state machine's registers: profiles/r04/ab_dense_occupancy.txt)
#ifdef JG_LEADER_WAVES
#define JG_LEADER_OCC __attribute__((amdgpu_waves_per_eu(JG_LEADER_WAVES, JG_LEADER_WAVES)))
#else
#define JG_LEADER_OCC
#endif
#ifdef JG_FOLLOWER_WAVES
#define JG_FOLLOWER_OCC __attribute__((amdgpu_waves_per_eu(JG_FOLLOWER_WAVES, JG_FOLLOWER_WAVES)))
#else
#define JG_FOLLOWER_OCC
#endif
template <int R, bool FSM>
__global__ __launch_bounds__(JG_BLOCK) JG_LEADER_OCC void k_leader_node_tick(JgDenseHot h, const JgDev* __restrict__ dp,
                                                                const uint64_t* __restrict__ acks, uint32_t seq, int us,
                                                                JgLeaderNode nd) {
  if (nd.clock) {  // a replayed round: the first kernel of the round (see JgClock)
    const uint32_t a = nd.clock->idx_lead & 1u;
    nd.now = nd.clock->v[a].now, seq = nd.clock->v[a].seq[nd.clock_slot];
    if (blockIdx.x == 0 && threadIdx.x == 0) nd.clock->idx_rest = a;
  }
  JgDecCount dec;
  if (us >= 0) dec = jg_dense_tick_body<R, true, true, true, FSM>(h, dp, acks, seq, (uint32_t)us, nd, nullptr);
  else dec = jg_dense_tick_body<R, false, true, true, FSM>(h, dp, acks, seq, 0, nd, nullptr);
  jg_wave_count(h.blk_decisions, dec);
}

// Per-partition leadership (jg_dense_cluster, JG_CLUSTER_ANY_LEADER): the leader halves of ALL nodes of a cluster in one
// launch, blockIdx.y = node; every node's lanes serve the groups that node leads, out of the cluster's mailboxes.
struct JgLeaderJob {
  JgDenseHot h;
  const JgDev* dp;
  const uint64_t* acks;
  uint32_t seq;
  int us;
  JgLeaderNode nd;
};
template <int R>
__global__ __launch_bounds__(JG_BLOCK) void k_leader_node_tick_any(const JgLeaderJob* __restrict__ jobs) {
  const JgLeaderJob& j = jobs[blockIdx.y];
  JgLeaderNode nd = j.nd;
  uint32_t seq = j.seq;
  if (nd.clock) {  // a replayed round: the first kernel of the round that reads the clock (see JgClock)
    const uint32_t a = nd.clock->idx_lead & 1u;
    nd.now = nd.clock->v[a].now, seq = nd.clock->v[a].seq[nd.clock_slot];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) nd.clock->idx_rest = a;
  }
  const JgDecCount dec = jg_dense_tick_body<R, true, true, true, false, true>(j.h, j.dp, j.acks, seq, (uint32_t)j.us, nd, nullptr);
  jg_wave_count(j.h.blk_decisions, dec);
}

// owner[g] = the lowest slot whose node leads group g (healthy leaders only), JG_OWNER_NONE: nobody - four groups per lane.
// Where a group passes from one owner STRAIGHT to another (two terms' leaders in one round, the newer one in the lower
// slot), the old owner's row of the answers still holds what it said when it last FOLLOWED - it never writes its own
// row while it owns the group - and the new owner's leader half of this very round would read that as a fresh
// AppendResponse / HeartbeatResponse: the claim withdraws it (JG_NO_ACK: nothing from that slot).
struct JgClaimArgs {
  const uint32_t* flags[JG_MAX_REPLICAS];
  uint32_t R, G;
  uint8_t* owner;
  uint64_t* answers;  // [R][G] the cluster's inbox
};
__global__ __launch_bounds__(JG_BLOCK) void k_cluster_claim(JgClaimArgs a) {
  const uint32_t n4 = (a.G + 3u) / 4u;
  constexpr uint32_t M = JGF_ROLE_MASK | JGF_FAULT_MASK;
  for (uint32_t q = blockIdx.x * JG_BLOCK + threadIdx.x; q < n4; q += gridDim.x * JG_BLOCK) {
    uint32_t o[4] = {JG_OWNER_NONE, JG_OWNER_NONE, JG_OWNER_NONE, JG_OWNER_NONE};
    if (q * 4u + 3u < a.G) {  // one 16-byte load per node (the flag columns are 16-byte aligned)
      for (uint32_t r = a.R; r-- > 0;) {
        const uint4 f = ((const uint4*)a.flags[r])[q];
        o[0] = (f.x & M) == JG_ROLE_LEADER ? r : o[0];
        o[1] = (f.y & M) == JG_ROLE_LEADER ? r : o[1];
        o[2] = (f.z & M) == JG_ROLE_LEADER ? r : o[2];
        o[3] = (f.w & M) == JG_ROLE_LEADER ? r : o[3];
      }
    } else {
      for (uint32_t k = 0; k < 4; k++)
        for (uint32_t r = a.R; q * 4u + k < a.G && r-- > 0;) o[k] = (a.flags[r][q * 4u + k] & M) == JG_ROLE_LEADER ? r : o[k];
    }
    const uint32_t now = o[0] | o[1] << 8 | o[2] << 16 | o[3] << 24, was = ((const uint32_t*)a.owner)[q];  // (the column is allocated in whole words)
    if (now == was) continue;
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t p = (was >> (8 * k)) & 0xffu;
      if (p != JG_OWNER_NONE && p != o[k] && o[k] != JG_OWNER_NONE && q * 4u + k < a.G) a.answers[(size_t)p * a.G + q * 4u + k] = JG_NO_ACK;
    }
    ((uint32_t*)a.owner)[q] = now;
  }
}

// ---- T consecutive ticks per launch (temporal fusion) ----------------------------------------
// When the caller already holds the ack blocks of several ticks (a batched event loop, the
// pre-generated bench stream), the group's state stays in registers across them: it is read
// once and written once per launch, so a group-step costs 8R (acks) + 44/T bytes of
// traffic instead of 8R+44.  Semantically identical to T calls of the single-tick kernel:
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
    uint64_t a[R], mword0, head0;
    jg_dense_load<R, false>(jg_dense_hot_of(d), acks, g, a, mword0, head0);
    asm volatile("" ::"v"(f));  // (one round trip: see jg_dense_issue)
    __builtin_amdgcn_sched_barrier(0);
    const bool leader = (f & JGF_ROLE_MASK) == JG_ROLE_LEADER;
    const bool dead = (f & JGF_FAULT_MASK) != 0;                // the reference process is gone
    const bool defer = !dead && leader && !(f & JGF_FAST);      // irregular chain: k_dense_slow replays all ticks
    jg_defer_mark(d, g, defer);
    if (dead || defer) continue;
    // Leaders stay in lag space: jg_lag_tick on the packed word, carried from tick to tick in
    // registers.  Nothing is stored before the last tick, so a group with a tick that does not
    // fit (escaped field, ack above the head) is handed to k_dense_slow as it was, for all T ticks.
    constexpr uint32_t B = 64u / (R + 1u);
    constexpr uint32_t ESC = (uint32_t)((1ull << (B > 31 ? 31 : B)) - 1ull);
    uint64_t head = head0;
    uint32_t nf = f, gdec = 0;
    uint32_t l[R + 1];
    bool fits = B <= 21;  // (R = 1: 32-bit fields, general path)
#pragma unroll
    for (int r = 0; r <= R; r++) {
      l[r] = (uint32_t)(mword0 >> (r * B)) & ESC;
      fits = fits && l[r] != ESC;                  // a head above the chain head: general path
      l[r] = l[r] == ESC - 1u ? 0xffffffffu : l[r];  // BEHIND: stays as it is while no ack arrives
    }
    fits = fits || !leader;
    // software prefetch, two ticks deep: the acks of tick t + 2 are requested before tick t is
    // evaluated (one tick ahead left the memory system short of requests in flight)
    uint64_t a1[R];
    if (n_ticks > 1) {
      jg_dense_load_acks<R>(acks + tick_stride, G, g, a1);
    } else {
#pragma unroll
      for (int r = 0; r < R; r++) a1[r] = JG_NO_ACK;
    }
    for (uint32_t t = 0; fits && t < n_ticks; t++) {
      uint64_t a2[R];
      if (t + 2 < n_ticks) {
        jg_dense_load_acks<R>(acks + (size_t)(t + 2) * tick_stride, G, g, a2);
      } else {
#pragma unroll
        for (int r = 0; r < R; r++) a2[r] = JG_NO_ACK;  // (never looked at)
      }
      uint64_t n_app = a[0];
#pragma unroll
      for (int r = 1; r < R; r++) n_app = (uint32_t)r == s ? a[r] : n_app;
      if (!leader) {
        // acks are ignored by followers / candidates (follower.rs:62, candidate.rs:194)
        if (n_app) {
          nf = f | (JG_FAULT_ENGINE_DENSE_NONLEADER << JGF_FAULT_SHIFT);
          jg_push_fault(d, g, JG_FAULT_ENGINE_DENSE_NONLEADER, seq0 + t);
          break;
        }
      } else if (!jg_lag_tick_regs<R>(s, nf, l, head, n_app, a, gdec)) {
        fits = false;
      }
#pragma unroll
      for (int r = 0; r < R; r++) a[r] = a1[r], a1[r] = a2[r];
    }
    uint64_t w = mword0;
    if (leader && fits) {  // pack once per launch; a lag that left its field: general path
      w = 0;
#pragma unroll
      for (int r = 0; r <= R; r++) {
        fits = fits && (l[r] < ESC - 1u || l[r] == 0xffffffffu);
        w |= (uint64_t)min(l[r], ESC - 1u) << (r * B);
      }
    }
    jg_defer_mark(d, g, !fits);
    if (!fits) continue;
    dec += gdec;
    if (w != mword0) d.mlag[g] = w;
    if (head != head0) d.head[g] = head;
    if (nf != f) d.flags[g] = nf;
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
