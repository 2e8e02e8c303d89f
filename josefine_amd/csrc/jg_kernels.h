// jg_kernels.h — gfx950 kernels of the batched Chained-Raft engine.
//
//   k_leader_tick_dense<R>  the HBM-roofline kernel: AppendEntries-ack tally,
//                           majority test and commit-index advance for steady-state
//                           leaders over SoA columns (progress.rs:42-60,133-140;
//                           leader.rs:87-99,177-197,211-219)
//   k_dense_slow            same tick for groups whose chain is not in FAST form
//   k_apply_cmds            the full state machine over a CSR command batch
//                           (every role, every Command; mod.rs:471-479)
//   k_chain_compact         Chain::compact parent-pointer walk (chain.rs:239-253)
//   k_synth_acks            synthetic ack-stream generator (bench / parity input)
//   k_gather_rows           drain-time compaction of the per-group output regions
#pragma once
#include "jg_device.h"

#define JG_BLOCK 256

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

// ---- dense steady-state leader tick -----------------------------------------------------
// One lane per group.  Per group-step it reads R ack heads, R match heads,
// commit, head (8 B each) and the flag word, and writes back what changed:
// B(R) = 24R + 36 algorithmic bytes.
//
// Exactness: the reference evaluates Leader::commit after every ack.  match[] is
// monotone, hence so is committed_index(), and the commit guard `q > commit`
// makes the final commit max(commit, q_final) — provided chain.commit(q) never
// panics on the way, which in FAST form (id set == [0, head]) means q <= head at
// the time of each evaluation.  If every old match head and every ack is <= the
// head before this tick's appends that cannot happen and the tick is fused into
// one majority evaluation; otherwise the lane replays the acks one by one.
template <int R>
__global__ __launch_bounds__(JG_BLOCK) void k_leader_tick_dense(JgDev d, const uint64_t* __restrict__ acks,
                                                                 uint32_t seq) {
  const uint32_t G = d.G;
  uint32_t dec = 0;
  for (uint32_t g = blockIdx.x * JG_BLOCK + threadIdx.x; g < G; g += gridDim.x * JG_BLOCK) {
    uint32_t f = d.flags[g];
    uint64_t a[R];
#pragma unroll
    for (int r = 0; r < R; r++) a[r] = acks[(size_t)r * G + g];
    if (f & JGF_FAULT_MASK) continue;
    const uint32_t s = (f & JGF_SELF_MASK) >> JGF_SELF_SHIFT;
    uint64_t n_app = 0;
#pragma unroll
    for (int r = 0; r < R; r++) n_app = (r == (int)s) ? a[r] : n_app;
    if ((f & JGF_ROLE_MASK) != JG_ROLE_LEADER) {
      // acks are ignored by followers / candidates (follower.rs:62, candidate.rs:194)
      if (n_app) {
        d.flags[g] = f | (JG_FAULT_ENGINE_DENSE_NONLEADER << JGF_FAULT_SHIFT);
        jg_push_fault(d, g, JG_FAULT_ENGINE_DENSE_NONLEADER, seq);
      }
      continue;
    }
    if (!(f & JGF_FAST)) {  // irregular chain: exact general path in k_dense_slow
      uint32_t idx = atomicAdd(d.slow_n, 1u);
      if (idx < G) d.slow_list[idx] = g;
      continue;
    }
    uint64_t m[R], m0[R];
#pragma unroll
    for (int r = 0; r < R; r++) m0[r] = m[r] = d.match[(size_t)r * G + g];
    const uint64_t commit0 = d.commit[g];
    uint64_t commit = commit0;
    const uint64_t head0 = d.head[g];
    uint64_t head = head0;
    uint32_t nf = f;

    uint64_t hi = 0;  // max over old match heads and follower acks
#pragma unroll
    for (int r = 0; r < R; r++) {
      hi = m[r] > hi ? m[r] : hi;
      bool is_ack = (r != (int)s) && (a[r] != JG_NO_ACK);
      hi = (is_ack && a[r] > hi) ? a[r] : hi;
    }
    uint32_t fault = 0;
    if (hi <= head0) {
      // ---- fused path -------------------------------------------------------------
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
      uint64_t q = jg_kth<R>(m);          // progress.rs:48-60
      commit = q > commit ? q : commit;   // leader.rs:89-92
    } else {
      // ---- exact replay: one Leader::commit per append / ack -------------------------
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
#pragma unroll
    for (int r = 0; r < R; r++)
      if (m[r] != m0[r]) d.match[(size_t)r * G + g] = m[r];
    if (commit != commit0) d.commit[g] = commit;
    if (head != head0) d.head[g] = head;
    if (nf != f) d.flags[g] = nf;
  }
  jg_block_count(d.blk_decisions, dec);
}

// Same tick through the general state machine, for the groups the fast kernel
// deferred (chain not in FAST form).
__global__ __launch_bounds__(JG_BLOCK) void k_dense_slow(JgDev d, const uint64_t* __restrict__ acks, uint32_t seq) {
  uint32_t dec = 0;
  const uint32_t n = *d.slow_n;
  for (uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x; i < n; i += gridDim.x * JG_BLOCK) {
    JgLane L;
    jg_load(d, L, d.slow_list[i]);
    L.now = 0;
    L.seq = seq;
    L.mp = L.mend = nullptr;  // a leader's client requests / acks emit no messages
    jg_fsm_row sink[2];
    const uint32_t s = jg_self(L);
    uint64_t n_app = acks[(size_t)s * d.G + L.g];
    JgCmd c;
    c.kind = JG_CMD_CLIENT_REQUEST;
    c.from = 0;
    c.flag = 0;
    c.term = c.id = c.aux = 0;
    for (uint64_t k = 0; k < n_app && !jg_fault(L); k++) {
      L.fp = sink;
      L.fend = sink + 2;
      jg_apply(d, L, c, nullptr, nullptr);
    }
    c.kind = JG_CMD_APPEND_RESPONSE;
    c.flag = 1;
    for (uint32_t r = 0; r < d.R && !jg_fault(L); r++) {
      if (r == s) continue;
      uint64_t h = acks[(size_t)r * d.G + L.g];
      if (h == JG_NO_ACK) continue;
      c.from = d.node_ids[r];
      c.id = h;
      L.fp = sink;
      L.fend = sink + 2;
      jg_apply(d, L, c, nullptr, nullptr);
    }
    dec += L.decisions;
    jg_store(d, L);
  }
  jg_block_count(d.blk_decisions, dec);
}

// ---- general command kernel -----------------------------------------------------------------
struct JgStepArgs {
  uint32_t n_active;
  const uint32_t* seg_group;  // [n_active] group of each segment (ascending)
  const uint32_t* seg_off;    // [n_active+1] command range of each segment
  const uint8_t* kind;        // commands in CSR (group-major, stream) order
  const uint32_t* from;
  const uint64_t* term;
  const uint64_t* id;
  const uint64_t* aux;
  const uint8_t* flag;
  const uint64_t* blk_id;
  const uint64_t* blk_next;
  const uint32_t* msg_base;   // [n_active+1] output-region bounds (prefix sums)
  const uint32_t* fsm_base;
  jg_msg_row* msg_out;
  jg_fsm_row* fsm_out;
  uint32_t* msg_cnt;          // [n_active] rows actually produced
  uint32_t* fsm_cnt;
  uint32_t* err;              // set when a row did not fit its bound
  uint64_t now;
  uint32_t seq;
};

__global__ __launch_bounds__(JG_BLOCK) void k_apply_cmds(JgDev d, JgStepArgs a) {
  uint32_t dec = 0;
  for (uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x; i < a.n_active; i += gridDim.x * JG_BLOCK) {
    JgLane L;
    jg_load(d, L, a.seg_group[i]);
    L.now = a.now;
    L.seq = a.seq;
    jg_msg_row* m0 = a.msg_out + a.msg_base[i];
    jg_fsm_row* f0 = a.fsm_out + a.fsm_base[i];
    L.mp = m0;
    L.mend = a.msg_out + a.msg_base[i + 1];
    L.fp = f0;
    L.fend = a.fsm_out + a.fsm_base[i + 1];
    for (uint32_t k = a.seg_off[i]; k < a.seg_off[i + 1]; k++) {
      JgCmd c;
      c.kind = a.kind[k];
      c.from = a.from[k];
      c.flag = a.flag[k];
      c.term = a.term[k];
      c.id = a.id[k];
      c.aux = a.aux[k];
      jg_apply(d, L, c, a.blk_id, a.blk_next);
    }
    a.msg_cnt[i] = (uint32_t)(L.mp - m0);
    a.fsm_cnt[i] = (uint32_t)(L.fp - f0);
    if (L.overflow) *a.err = 1;
    dec += L.decisions;
    jg_store(d, L);
  }
  jg_block_count(d.blk_decisions, dec);
}

// drain-time compaction: copy each segment's rows to its final offset
template <typename Row>
__global__ void k_gather_rows(uint32_t n_seg, const uint32_t* __restrict__ src_base, const uint32_t* __restrict__ cnt,
                              const uint32_t* __restrict__ dst_off, const Row* __restrict__ src, Row* __restrict__ dst) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_seg; i += gridDim.x * blockDim.x) {
    uint32_t n = cnt[i];
    const Row* s = src + src_base[i];
    Row* t = dst + dst_off[i];
    for (uint32_t k = 0; k < n; k++) t[k] = s[k];
  }
}

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
    L.decisions = 0;
    L.term = 0;
    L.commit = L.head = 0;   // Chain::new on an empty tree: genesis block 0
    L.id_gen = 1;
    L.run_hi = 0;
    L.heartbeat_time = 0;
    L.voted_for = L.leader_id = L.queued = L.votes = 0;
    L.rng_draws = 0;
    uint32_t s = self_slots ? self_slots[g] : 0;
    L.flags = JG_ROLE_FOLLOWER | (s << JGF_SELF_SHIFT);
    jg_set_election_timeout(d, L);  // follower.rs:93-95 at now = 0
    for (uint32_t r = 0; r < d.R; r++) d.match[(size_t)r * d.G + g] = 0;
    jg_store(d, L);
  }
}
