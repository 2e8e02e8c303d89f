// jg_kernels.h — gfx950 kernels of the batched Chained-Raft engine.
//
//   k_leader_tick_dense<R>  the HBM-roofline kernel: AppendEntries-ack tally,
//                           majority test and commit-index advance for steady-state
//                           leaders over SoA columns (progress.rs:42-60,133-140;
//                           leader.rs:87-99,177-197,211-219)
//   k_dense_slow            same tick for groups whose chain is not in FAST form
//   k_apply_rows            the full state machine over a group-sorted command batch
//                           (every role, every Command; mod.rs:471-479)
//   k_chain_compact         Chain::compact parent-pointer walk (chain.rs:239-253)
//   k_synth_acks            synthetic ack-stream generator (bench / parity input)
//   k_gather_rows           drain-time compaction of the per-group output regions
#pragma once
#include "jg_device.h"

#include "jg_dense.h"  // k_leader_tick_dense / _n, jg_block_count, JG_BLOCK

// ---- groups the dense kernel deferred (healthy leaders whose chain is not in FAST form) ----
// Two kernels, no contended global atomics: k_collect_deferred compacts the deferred
// groups of each of JG_SHARDS contiguous group ranges into that shard's list (LDS counter);
// k_dense_slow then replays the ticks for them with densely packed lanes, so its fault-queue
// pushes coalesce per wave.  (Appending ~1 % of 1 M groups to one list from the dense kernel
// cost it +65 us per tick: ~8 ns per contended wave-level atomic request; scanning the flag
// column with one sparse lane per wave cost the slow kernel the same in fault pushes —
// profiles/README.md.)
#define JG_SHARDS 256
__global__ __launch_bounds__(JG_BLOCK) void k_collect_deferred(JgDev d) {
  __shared__ uint32_t n_s;
  if (threadIdx.x == 0) n_s = 0;
  __syncthreads();
  const uint32_t cap = (d.G + JG_SHARDS - 1) / JG_SHARDS;
  const uint32_t g0 = blockIdx.x * cap;
  const uint32_t g1 = g0 + cap < d.G ? g0 + cap : d.G;
  uint32_t* list = d.slow_list + (size_t)blockIdx.x * cap;
  for (uint32_t g = g0 + threadIdx.x; g < g1; g += JG_BLOCK) {
    const uint32_t f = d.flags[g];
    if (!(f & JGF_FAULT_MASK) && (f & JGF_ROLE_MASK) == JG_ROLE_LEADER && !(f & JGF_FAST))
      list[atomicAdd(&n_s, 1u)] = g;  // LDS atomic; order within a shard does not matter
  }
  __syncthreads();
  if (threadIdx.x == 0) d.slow_cnt[blockIdx.x] = n_s;
}

// Same ticks through the general state machine for the collected groups; workgroup s owns
// shard s.
__global__ __launch_bounds__(JG_BLOCK) void k_dense_slow(JgDev d, const uint64_t* __restrict__ acks, uint32_t n_ticks,
                                                          size_t tick_stride, uint32_t seq0) {
  uint32_t dec = 0;
  const uint32_t cap = (d.G + JG_SHARDS - 1) / JG_SHARDS;
  const uint32_t n = d.slow_cnt[blockIdx.x];
  const uint32_t* list = d.slow_list + (size_t)blockIdx.x * cap;
  for (uint32_t i = threadIdx.x; i < n; i += JG_BLOCK) {
    const uint32_t g = list[i];
    JgLane L;
    jg_load(d, L, g);
    L.now = 0;
    L.mp = L.mend = nullptr;  // a leader's client requests / acks emit no messages
    jg_fsm_row sink[2];
    const uint32_t s = jg_self(L);
    for (uint32_t t = 0; t < n_ticks && !jg_fault(L); t++) {
      const uint64_t* A = acks + (size_t)t * tick_stride;
      L.seq = seq0 + t;
      uint64_t n_app = A[(size_t)s * d.G + L.g];
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
        uint64_t h = A[(size_t)r * d.G + L.g];
        if (h == JG_NO_ACK) continue;
        c.from = d.node_ids[r];
        c.id = h;
        L.fp = sink;
        L.fend = sink + 2;
        jg_apply(d, L, c, nullptr, nullptr);
      }
    }
    dec += L.decisions;
    jg_store(d, L);
  }
  jg_block_count(d.blk_decisions, dec);
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
    L.self_match = 0;
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
