"""Whole KERNELS of the device source on the host, with a workgroup's semantics (CPU): a second stand-in for
<hip/hip_runtime.h> in which a workgroup is 256 fibers (ucontext) scheduled cooperatively - `__syncthreads()` is a barrier
of the workgroup's live threads, a shuffle / ballot a rendezvous of the wave's live lanes, `__shared__` is memory the
workgroup's threads share, atomics are the sequential ones - so that the code lanes run TOGETHER (the transport's LDS
staging and tallies, its bucket pass and in-LDS sort, the workgroup scans) executes as written, one workgroup after the
other.  tests/host_compiled.py runs one lane at a time and says what it cannot do; this runs what it cannot.

TEST INFRASTRUCTURE, and nothing else: built at test time into a temporary directory from josefine_amd/csrc AS IT STANDS
(JG_BLOCK = 256, no textual patch), nothing under josefine_amd/ can reach it.  It is not a model of the memory system or
of parallel interleavings: workgroups run one after the other and a fiber runs until its next rendezvous, so a data race
that needs two waves to interleave between rendezvous is not found here; and reconvergence behind a divergent branch is a
heuristic (the scheduler's comment), where the hardware has the compiler's post-dominators.  What it is for: the launch-level logic of
kernels that no GPU-minute was left for (the routed round with the election vocabulary as mailbox words, jg_votes.h) is
held to the numpy statement of the same pass before a device sees it."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

from josefine_amd import capi
import host_compiled
from host_compiled import CSRC, HOST_H, ROOT, VoteMail

WG_SHIM = r'''
#pragma once
// stand-in for <hip/hip_runtime.h>: a workgroup = JG_BLOCK fibers, scheduled round-robin; a fiber runs until it ends or
// reaches a rendezvous (tests/host_workgroups.py)
#include <cstdint>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <unordered_map>
#include <vector>
#include <ucontext.h>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
#define __shared__ static
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 { uint32_t x = 1, y = 1, z = 1; dim3() {} dim3(uint32_t a, uint32_t b = 1, uint32_t c = 1) : x(a), y(b), z(c) {} };

namespace wg {
enum Wait { RUN = 0, AT_BLOCK = 1, AT_WAVE = 2, DONE = 3 };
struct Fiber {
  ucontext_t ctx;
  void* sp = nullptr;  // (x86-64: the fiber's saved stack pointer - the switch is a dozen instructions, no system call)
  char* stack = nullptr;
  int state = DONE;
  const void* site = nullptr;  // AT_WAVE: which shuffle / ballot (lanes of a wave that diverged wait at different ones)
  const void* prev_site = nullptr;
  dim3 tid;
};
struct Wave {
  uint64_t slot[64];   // what each lane brought to the rendezvous
  uint64_t mask = 0;   // the lanes that took part in the last one
};
static const size_t STACK = 256 * 1024;
static std::vector<Fiber> fibers;
static std::vector<Wave> waves;
struct SiteStat { uint64_t whole = 0, part = 0; };  // releases of this rendezvous: with every live lane of the wave / with some
static std::unordered_map<const void*, SiteStat> sites;
static std::unordered_map<const void*, std::vector<const void*>> succ;  // rendezvous -> the ones a lane came to next
static bool reaches(const void* a, const void* b) {
  std::vector<const void*> todo{a};
  std::unordered_map<const void*, bool> seen;
  seen[a] = true;
  while (!todo.empty()) {
    const void* x = todo.back();
    todo.pop_back();
    auto it = succ.find(x);
    if (it == succ.end()) continue;
    for (const void* y : it->second) {
      if (y == b) return true;
      if (!seen[y]) seen[y] = true, todo.push_back(y);
    }
  }
  return false;
}
struct SiteDump {  // JG_WG_STATS=1: the rendezvous that were released with a part of a wave, at exit (addresses: addr2line -e <the library>)
  ~SiteDump() {
    if (!std::getenv("JG_WG_STATS")) return;
    for (const auto& kv : sites)
      if (kv.second.part) std::fprintf(stderr, "[wg] site %p whole %llu part %llu\n", kv.first, (unsigned long long)kv.second.whole, (unsigned long long)kv.second.part);
  }
};
static SiteDump site_dump;
static Fiber* cur = nullptr;
static ucontext_t sched;
static const std::function<void()>* body = nullptr;
static dim3 block_idx, grid_dim, block_dim;
#if defined(__x86_64__) && !defined(WG_UCONTEXT)
// swapcontext saves the signal mask with a system call on every switch - a launch is hundreds of thousands of fibers; this
// saves what the ABI says a call preserves and swaps the stack pointer
extern "C" void wg_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl wg_switch
.type wg_switch,@function
wg_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size wg_switch,.-wg_switch
)");
static void* sched_sp = nullptr;
static void to_sched() { wg_switch(&cur->sp, sched_sp); }
static void to_fiber(Fiber& f) { wg_switch(&sched_sp, f.sp); }
static void trampoline() {
  (*body)();
  cur->state = DONE;
  to_sched();
  std::abort();  // (a finished fiber is never resumed)
}
static void prepare(Fiber& f) {
  uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
  void** sp = (void**)top;
  *--sp = nullptr;              // (the return address trampoline never uses: the stack is aligned as after a call)
  *--sp = (void*)&trampoline;   // where the first switch returns to
  for (int r = 0; r < 6; r++) *--sp = nullptr;
  f.sp = sp;
}
#else
static void to_sched() { swapcontext(&cur->ctx, &sched); }
static void to_fiber(Fiber& f) { swapcontext(&sched, &f.ctx); }
static void trampoline() {
  (*body)();
  cur->state = DONE;
  to_sched();
}
static void prepare(Fiber& f) {
  getcontext(&f.ctx);
  f.ctx.uc_stack.ss_sp = f.stack, f.ctx.uc_stack.ss_size = STACK, f.ctx.uc_link = &sched;
  makecontext(&f.ctx, trampoline, 0);
}
#endif
static void wait(int what, const void* site = nullptr) {
  cur->state = what, cur->site = site;
  if (what == AT_WAVE) {
    if (cur->prev_site && cur->prev_site != site) {
      std::vector<const void*>& v = succ[cur->prev_site];
      if (std::find(v.begin(), v.end(), site) == v.end()) v.push_back(site);
    }
    cur->prev_site = site;
  }
  to_sched();
}
// one workgroup: every fiber to its end
static void run_block(uint32_t n_threads) {
  if (fibers.size() < n_threads) {
    fibers.resize(n_threads);
    for (Fiber& f : fibers)
      if (!f.stack) f.stack = (char*)malloc(STACK);
  }
  waves.assign((n_threads + 63) / 64, Wave{});
  for (uint32_t t = 0; t < n_threads; t++) {
    Fiber& f = fibers[t];
    prepare(f);
    f.state = RUN, f.tid = dim3(t), f.prev_site = nullptr;
  }
  for (;;) {
    bool ran = false, live = false;
    for (uint32_t t = 0; t < n_threads; t++) {
      Fiber& f = fibers[t];
      if (f.state != RUN) continue;
      cur = &f, ran = true;
      to_fiber(f);
    }
    // rendezvous: a wave's, when every live lane of it waits there; the workgroup's, when every live thread does
    bool all_block = true;
    for (uint32_t t = 0; t < n_threads; t++) {
      if (fibers[t].state == DONE) continue;
      live = true;
      all_block = all_block && fibers[t].state == AT_BLOCK;
    }
    if (!live) return;
    bool released = false;
    if (all_block) {
      for (uint32_t t = 0; t < n_threads; t++)
        if (fibers[t].state == AT_BLOCK) fibers[t].state = RUN;
      released = true;
    }
    // A wave's lanes that diverged wait at different rendezvous: the lanes at ONE of them go on together, with that set as
    // the active mask, as the hardware runs one side of a branch at a time.  WHICH one: the hardware brings the lanes back
    // together behind the branch (the compiler's post-dominator), so the lanes that skipped it wait there for the others -
    // here they have run ahead to their next rendezvous, and releasing them first would let them meet without the others.
    // The side that is still INSIDE the branch is recognised by its record: a rendezvous inside a divergent region is
    // rarely reached by a whole wave, one behind the join nearly always is (sites: share of the releases that were whole
    // waves; unknown sites first, ties to the lower address - code that comes earlier).
    for (uint32_t w = 0; w < waves.size() && !all_block; w++) {
      const void* site = nullptr;
      uint64_t live_m = 0;
      std::vector<const void*> cand;
      std::vector<uint32_t> lanes_at;
      for (uint32_t l = 0; l < 64 && w * 64 + l < n_threads; l++) {
        const Fiber& f = fibers[w * 64 + l];
        if (f.state != DONE) live_m |= 1ull << l;
        if (f.state != AT_WAVE) continue;
        size_t i = 0;
        while (i < cand.size() && cand[i] != f.site) i++;
        if (i == cand.size()) cand.push_back(f.site), lanes_at.push_back(0);
        lanes_at[i]++;
      }
      if (cand.size() == 1) site = cand[0];
      else if (cand.size() > 1) {
        // (1) a rendezvous from which lanes have been seen to come to another candidate, and not the other way round, is
        // the earlier one: its lanes are inside the branch the others skipped
        std::vector<bool> later(cand.size(), false);
        for (size_t i = 0; i < cand.size(); i++)
          for (size_t j = 0; j < cand.size(); j++)
            if (i != j && reaches(cand[j], cand[i]) && !reaches(cand[i], cand[j])) later[i] = true;
        double best = 2.0;
        uint32_t best_n = 0;
        for (size_t i = 0; i < cand.size(); i++) {
          if (later[i]) continue;
          const SiteStat& st = sites[cand[i]];
          // (2) the share of whole-wave releases, unknown sites first; (3) the higher address (the unlikely side of a branch is laid out last)
          const double share = st.whole + st.part ? (double)st.whole / (double)(st.whole + st.part) : -1.0;
          if (!site || share < best || (share == best && (cand[i] > site)))
            site = cand[i], best = share, best_n = lanes_at[i];
        }
        if (!site) site = cand[0];  // (a cycle of "later": cannot be, but nobody waits forever)
      }
      if (!site) continue;
      uint64_t m = 0;
      for (uint32_t l = 0; l < 64 && w * 64 + l < n_threads; l++)
        if (fibers[w * 64 + l].state == AT_WAVE && fibers[w * 64 + l].site == site) m |= 1ull << l;
      SiteStat& st = sites[site];
      (m == live_m ? st.whole : st.part)++;
      waves[w].mask = m;
      for (uint32_t l = 0; l < 64 && w * 64 + l < n_threads; l++)
        if ((m >> l) & 1ull) fibers[w * 64 + l].state = RUN;
      released = true;
    }
    if (!ran && !released) {
      std::fprintf(stderr, "host_workgroups: deadlock in workgroup (%u, %u): live threads wait at different rendezvous\n", block_idx.x, block_idx.y);
      std::abort();
    }
  }
}
template <class F>
static void launch(dim3 grid, uint32_t n_threads, F&& f) {
  const std::function<void()> fn = f;
  body = &fn;
  grid_dim = grid, block_dim = dim3(n_threads);
  for (uint32_t y = 0; y < grid.y; y++)
    for (uint32_t x = 0; x < grid.x; x++) {
      block_idx = dim3(x, y);
      run_block(n_threads);
    }
  body = nullptr;
}
static inline Wave& my_wave() { return waves[cur->tid.x >> 6]; }
// a rendezvous of the wave's live lanes around a 64-bit value per lane
static inline void bring(uint64_t v, const void* site) {
  my_wave().slot[cur->tid.x & 63u] = v;
  wait(AT_WAVE, site);
}
static inline void leave(const void* site) { wait(AT_WAVE, (const char*)site + 1); }  // (nobody's slot is overwritten before everybody has read)
}  // namespace wg
#define threadIdx (wg::cur->tid)
#define blockIdx (wg::block_idx)
#define gridDim (wg::grid_dim)
#define blockDim (wg::block_dim)
static inline void __syncthreads() { wg::wait(wg::AT_BLOCK); }
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T, class U> static inline T atomicOr(T* p, U v) { T o = *p; *p = o | (T)v; return o; }
template <class T, class U> static inline T atomicAnd(T* p, U v) { T o = *p; *p = o & (T)v; return o; }
template <class T, class U> static inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T> static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
template <class T, class U> static inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
static inline int __ffs(uint32_t v) { return __builtin_ffs((int)v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
template <class T> static inline uint64_t wg_bits(T v) { static_assert(sizeof(T) <= 8, "shuffles move up to 64 bits"); uint64_t b = 0; std::memcpy(&b, &v, sizeof(T)); return b; }
template <class T> static inline T wg_from(uint64_t b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }
__attribute__((noinline)) static uint64_t __ballot(bool x) {
  const void* site = __builtin_return_address(0);
  wg::bring(x ? 1 : 0, site);
  const wg::Wave& w = wg::my_wave();
  uint64_t b = 0;
  for (uint32_t l = 0; l < 64; l++)
    if (((w.mask >> l) & 1ull) && w.slot[l]) b |= 1ull << l;
  wg::leave(site);
  return b;
}
template <class T> __attribute__((noinline)) static T __shfl(T v, int src, int = 64) {
  const void* site = __builtin_return_address(0);
  wg::bring(wg_bits(v), site);
  const T r = wg_from<T>(wg::my_wave().slot[src & 63]);
  wg::leave(site);
  return r;
}
template <class T> __attribute__((noinline)) static T __shfl_up(T v, int off, int = 64) {
  const uint32_t lane = threadIdx.x & 63u;
  const void* site = __builtin_return_address(0);
  wg::bring(wg_bits(v), site);
  const T r = lane >= (uint32_t)off ? wg_from<T>(wg::my_wave().slot[lane - off]) : v;
  wg::leave(site);
  return r;
}
template <class T> __attribute__((noinline)) static T __shfl_down(T v, int off, int = 64) {
  const uint32_t lane = threadIdx.x & 63u;
  const void* site = __builtin_return_address(0);
  wg::bring(wg_bits(v), site);
  const T r = lane + (uint32_t)off < 64u ? wg_from<T>(wg::my_wave().slot[lane + off]) : v;
  wg::leave(site);
  return r;
}
template <class T> __attribute__((noinline)) static T __shfl_xor(T v, int off, int = 64) {
  const uint32_t lane = threadIdx.x & 63u;
  const void* site = __builtin_return_address(0);
  wg::bring(wg_bits(v), site);
  const T r = wg_from<T>(wg::my_wave().slot[(lane ^ (uint32_t)off) & 63u]);
  wg::leave(site);
  return r;
}
static inline uint32_t __builtin_amdgcn_readlane(uint32_t v, int l) { return __shfl(v, l); }
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_fetch_or(p, v, order, scope) atomicOr((p), (v))
#define __hip_atomic_fetch_and(p, v, order, scope) atomicAnd((p), (v))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (decltype(*(p) + 0))(v))
using std::max;
using std::min;
'''

WG_HARNESS = r'''
#define JG_BLOCK 256
#include "jg_route.h"   // (jg_device.h, jg_sparse.h, jg_votes.h)
#include "host.h"

// self-test of the stand-in: a workgroup scan (shuffles + LDS + barriers), ballots under divergence, a tally
static __global__ void k_selftest(const uint32_t* in, uint32_t n, uint32_t* excl, uint32_t* total, uint64_t* ballots, uint32_t* odd_sum) {
  const uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x;
  uint32_t tot = 0;
  const uint32_t e = jg_block_exclusive_scan(i < n ? in[i] : 0u, &tot);
  if (i < n) excl[i] = e;
  if (threadIdx.x == 0) total[blockIdx.x] = tot;
  uint32_t v = (i < n && (in[i] & 1u)) ? in[i] : 0u;
  for (int off = 32; off; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63u) == 0) atomicAdd(odd_sum, v);
  if (i >= n) return;  // (the lanes that are left take part in what follows, the others do not: their bit is 0)
  const uint64_t b = __ballot(in[i] & 1u);
  if ((threadIdx.x & 63u) == 0) ballots[i >> 6] = b;
}
VIS void hw_selftest(const uint32_t* in, uint32_t n, uint32_t* excl, uint32_t* total, uint64_t* ballots, uint32_t* odd_sum) {
  wg::launch(dim3((n + JG_BLOCK - 1) / JG_BLOCK), JG_BLOCK, [&] { k_selftest(in, n, excl, total, ballots, odd_sum); });
}

// ---- the routed round's transport under JG_ROUTE_VOTE_WORDS, kernel by kernel as round_routed_impl launches it ------
struct WgSender {           // what one sender emitted this round
  const JgXqRec* xq;        // its exceptional queue
  uint32_t xq_n, seq_base;
  uint32_t rec_n, rec_per_row, rec_step;  // one sparse step's output region (rec_n = 0: none); rec_step: its PHASE of the round
  uint32_t phases;                        // the sender's steps of the round as phases (jg_route_phase): for its exceptional queue's rows
  const uint32_t* msg_cnt;
  const jg_msg_row* msg;
  const uint32_t* fsm_cnt;
};
struct WgRoute {
  uint32_t R, G, group_bits, ord_bits, cap, n_seg;
  const uint32_t* member_id;
  uint64_t *key, *key_alt;
  uint32_t *idx, *idx_alt;
  jg_msg_row* row;
  uint32_t* count;   // [R][R + 4] | [JG_ROUTE_SEGS] cursors | [R] keep | [R] kinds
  uint32_t* bk;      // bucket scratch
  uint32_t bk_words;
  JgRouteCols cols;  // the sorted command columns (cap entries each)
  uint32_t words;    // bit 0: JG_ROUTE_VOTE_WORDS; bit 1: a repeated pass
};
VIS int hw_route(const WgRoute* rt, const WgSender* snd, const JgVoteMail* vm, uint32_t* total_out) {
  const uint32_t R = rt->R, ROUTE_WORDS = R + 4;
  uint32_t* d_cursor = rt->count + (size_t)R * ROUTE_WORDS;
  uint32_t* d_keep_n = d_cursor + JG_ROUTE_SEGS;
  const size_t words = (size_t)R * ROUTE_WORDS + JG_ROUTE_SEGS + 2 * R;
  auto table = [&](uint32_t s) {
    JgRouteTable t{};
    t.R = R, t.src = s;
    for (uint32_t n = 0; n < R; n++) t.member_id[n] = rt->member_id[n];
    t.src_id = t.member_id[s];
    t.group_bits = rt->group_bits, t.ord_bits = rt->ord_bits, t.cap = rt->cap;
    t.seg_cap = rt->cap / rt->n_seg, t.seg_mask = rt->n_seg - 1;
    t.key = rt->key, t.idx = rt->idx, t.row = rt->row;
    t.cursor = d_cursor;
    t.count = rt->count + (size_t)s * ROUTE_WORDS;
    t.kinds = d_keep_n + R;
    return t;
  };
  std::vector<JgRouteRecJob> rjobs;
  std::vector<JgRouteXqJob> xjobs;
  uint32_t widest_r = 0;
  for (uint32_t s = 0; s < R; s++) {
    const JgRouteTable t = table(s);
    if (snd[s].rec_n) {
      JgRouteRecJob j{};
      j.t = t, j.n = snd[s].rec_n, j.per_row = snd[s].rec_per_row, j.step = snd[s].rec_step;
      j.msg_cnt = snd[s].msg_cnt, j.msg = snd[s].msg, j.fsm_cnt = snd[s].fsm_cnt;
      rjobs.push_back(j);
      widest_r = std::max(widest_r, j.n);
    }
    JgRouteXqJob j{};
    j.t = t, j.xq = snd[s].xq, j.xq_n = &snd[s].xq_n, j.xq_cap = snd[s].xq_n + 1, j.seq_base = snd[s].seq_base, j.phases = snd[s].phases;
    xjobs.push_back(j);
  }
  JgRouteBuckets bk{};
  const uint32_t tile_bits = std::min<uint32_t>(JG_ROUTE_TILE_BITS, rt->group_bits);
  bk.n_buckets = R << (rt->group_bits - tile_bits);
  const uint32_t n_tiles = (bk.n_buckets + JG_ROUTE_SCAN_TILE - 1) / JG_ROUTE_SCAN_TILE;
  if ((size_t)n_tiles * JG_ROUTE_SCAN_TILE + bk.n_buckets + 1 + n_tiles + 1 > rt->bk_words) return -1;
  bk.hist = rt->bk, bk.cur = bk.hist + (size_t)n_tiles * JG_ROUTE_SCAN_TILE, bk.done = bk.cur + bk.n_buckets, bk.tile = bk.done + 1;
  const uint32_t bk_clear = n_tiles * JG_ROUTE_SCAN_TILE + bk.n_buckets + 1;
  const JgVoteMail m = *vm;
  const uint32_t xgrid = 3;  // (the kernels stride: a few workgroups per queue are as good as 256 here)
  if ((rt->words & 1u) && !(rt->words & 2u)) {  // (the census once: a repeated delivering pass finds it done)
    wg::launch(dim3(std::max<uint32_t>((widest_r + JG_BLOCK - 1) / JG_BLOCK, xgrid), (uint32_t)(rjobs.size() + xjobs.size())), JG_BLOCK,
               [&] { k_votes_census_multi(rjobs.data(), (uint32_t)rjobs.size(), xjobs.data(), m); });
    wg::launch(dim3(2), JG_BLOCK, [&] { k_votes_validate(m, R - 1u); });
  }
  wg::launch(dim3(2), JG_BLOCK, [&] { k_route_clear(rt->count, (uint32_t)words, bk.hist, bk_clear); });
  // the delivering pass, ONE launch as round_routed_impl issues it: the sparse steps' slots, the queues and (words) the expansion by blockIdx.y
  const uint32_t rec_x = (widest_r + JG_BLOCK * JG_ROUTE_ITEMS - 1) / (JG_BLOCK * JG_ROUTE_ITEMS);
  const bool w = rt->words & 1u;
  const dim3 dgrid(std::max<uint32_t>(rec_x, xgrid), (uint32_t)(rjobs.size() + xjobs.size() * (w ? 2 : 1)));
  if (w) wg::launch(dgrid, JG_BLOCK, [&] { k_route_deliver_multi_words(rjobs.data(), (uint32_t)rjobs.size(), xjobs.data(), (uint32_t)xjobs.size(), m); });
  else wg::launch(dgrid, JG_BLOCK, [&] { k_route_deliver_multi(rjobs.data(), (uint32_t)rjobs.size(), xjobs.data(), (uint32_t)xjobs.size()); });
  uint32_t total = 0, fullest = 0;
  for (uint32_t k = 0; k < rt->n_seg; k++) total += d_cursor[k], fullest = std::max(fullest, d_cursor[k]);
  if (fullest > rt->cap / rt->n_seg) return -2;  // (the host would grow the staging and repeat)
  for (uint32_t s = 0; s < R; s++)
    if (rt->count[(size_t)s * ROUTE_WORDS + R + JG_ROUTE_OVERFLOW]) return -3;  // (... or repeat with the wide index field)
  *total_out = total;
  if (!total) return 0;
  // the ordering pass
  bk.shift = rt->ord_bits + 3 + JG_ROUTE_STEP_BITS + tile_bits;
  const uint32_t seg_cap = rt->cap / rt->n_seg;
  const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>((std::min(fullest, seg_cap) + JG_BLOCK - 1) / JG_BLOCK, 4096 / rt->n_seg));
  wg::launch(dim3(grid, rt->n_seg), JG_BLOCK, [&] { k_route_hist(d_cursor, seg_cap, rt->key, bk); });
  wg::launch(dim3(n_tiles), JG_BLOCK, [&] { k_route_scan_all(bk, JgRouteXqDone{}); });  // (its last workgroup scans the tiles' totals; the queues are the harness's)
  wg::launch(dim3(grid, rt->n_seg), JG_BLOCK, [&] { k_route_scatter(d_cursor, seg_cap, rt->key, rt->idx, bk, rt->key_alt, rt->idx_alt); });
  wg::launch(dim3((bk.n_buckets + 2) / 3), JG_BLOCK, [&] { k_route_sort_build(bk, rt->key_alt, rt->idx_alt, rt->row, rt->cols, 3); });
  return 0;
}

// ---- the receiving half as the kernel runs it: every node in one launch (each node's next step number: seq_out) ---------
VIS int hw_vote_half_multi(Host** nodes, uint32_t R, uint32_t* seq_out, uint32_t step, uint64_t now, const JgVoteMail* in, const JgVoteMail* out,
                           uint32_t grid_x) {
  JgVoteHalfJobs jobs{};
  for (uint32_t n = 0; n < R; n++) {
    Host* h = nodes[n];
    h->seq++;
    seq_out[n] = h->seq;
    h->d.xq = h->xq.data(), h->d.xq_cap = (uint32_t)h->xq.size();
    JgVoteHalfJob j{};
    j.d = h->d, j.self = n, j.seq = h->seq, j.step = step, j.need = R - 1u, j.now = now;
    jobs.j[n] = j;
  }
  const JgVoteMail a = *in, b = *out;
  wg::launch(dim3(grid_x, R), JG_BLOCK, [&] { k_vote_half_multi(jobs, a, b); });
  int rc = 0;
  for (uint32_t n = 0; n < R; n++) {
    Host* h = nodes[n];
    for (uint64_t& v : h->blk_dec) h->decisions += v, v = 0;
    const uint32_t nf = *h->d.fault_q_n;
    std::vector<JgFaultRec> f(h->d.fault_q, h->d.fault_q + nf);
    std::stable_sort(f.begin(), f.end(), [](const JgFaultRec& x, const JgFaultRec& y) { return x.seq != y.seq ? x.seq < y.seq : x.group < y.group; });
    h->faults.insert(h->faults.end(), f.begin(), f.end());
    *h->d.fault_q_n = 0;
    rc = rc ? rc : (int)h->status[0];
  }
  return rc;
}
// a node's exceptional queue as the kernels left it (unordered), and emptied
VIS size_t hw_take_xq(Host* h, JgXqRec* out, size_t cap) {
  const uint32_t n = *h->d.xq_n;
  for (size_t i = 0; i < n && i < cap; i++) out[i] = h->xq[i];
  *h->d.xq_n = 0;
  h->d.xq = nullptr, h->d.xq_cap = 0;
  return n;
}
VIS void hw_votes_clear(const JgVoteMail* m) {
  const JgVoteMail a = *m;
  wg::launch(dim3(3), JG_BLOCK, [&] { k_votes_clear(a, nullptr, 0, nullptr, 0); });
}
'''

XQ_DTYPE = np.dtype([("row", capi.MSG_DTYPE), ("seq", "<u4"), ("k", "<u4")])
_lib = None


def build():
    """g++ -> a shared library in a temporary directory (once per process)"""
    global _lib
    if _lib is not None:
        return _lib
    tmp = tempfile.mkdtemp(prefix="jg_host_workgroups_")
    os.makedirs(os.path.join(tmp, "shim", "hip"))
    open(os.path.join(tmp, "shim", "hip", "hip_runtime.h"), "w").write(WG_SHIM)
    open(os.path.join(tmp, "host.h"), "w").write(HOST_H)
    cpp, so = os.path.join(tmp, "wg.cpp"), os.path.join(tmp, "libhost_workgroups.so")
    open(cpp, "w").write(WG_HARNESS)
    cc = ["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-Wno-unused-function", "-Wno-unused-variable", "-Wl,-Bsymbolic", "-fvisibility=hidden",
          f"-I{os.path.join(tmp, 'shim')}", f"-I{CSRC}", f"-I{os.path.join(ROOT, 'include')}", f"-I{tmp}"]
    if os.environ.get("JG_HOST_SANITIZE"):  # (as tests/host_compiled.py; the sanitizer follows swapcontext, not a hand-written switch)
        cc += ["-g", "-fno-omit-frame-pointer", "-DWG_UCONTEXT", f"-fsanitize={os.environ['JG_HOST_SANITIZE']}", "-fno-sanitize-recover=all"]
    subprocess.run(cc + ["-o", so, cpp], check=True)
    lib = C.CDLL(so)
    lib.hw_selftest.restype = None
    lib.hw_selftest.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 4
    lib.hw_route.argtypes = [C.c_void_p] * 4
    lib.hw_vote_half_multi.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.hw_take_xq.restype = C.c_size_t
    lib.hw_take_xq.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.hw_votes_clear.restype = None
    lib.hw_votes_clear.argtypes = [C.c_void_p]
    _lib = lib
    return lib


class _WgSender(C.Structure):
    _fields_ = [("xq", C.c_void_p), ("xq_n", C.c_uint32), ("seq_base", C.c_uint32), ("rec_n", C.c_uint32), ("rec_per_row", C.c_uint32),
                ("rec_step", C.c_uint32), ("phases", C.c_uint32), ("msg_cnt", C.c_void_p), ("msg", C.c_void_p), ("fsm_cnt", C.c_void_p)]


PHASES_IDENTITY = sum(i << (3 * i) for i in range(1, 8))  # step i of the round IS phase i


class _JgRouteCols(C.Structure):
    _fields_ = [("kind", C.c_void_p), ("flag", C.c_void_p), ("group", C.c_void_p), ("from_", C.c_void_p), ("term", C.c_void_p), ("id", C.c_void_p),
                ("aux", C.c_void_p)]


class _WgRoute(C.Structure):
    _fields_ = [("R", C.c_uint32), ("G", C.c_uint32), ("group_bits", C.c_uint32), ("ord_bits", C.c_uint32), ("cap", C.c_uint32), ("n_seg", C.c_uint32),
                ("member_id", C.c_void_p), ("key", C.c_void_p), ("key_alt", C.c_void_p), ("idx", C.c_void_p), ("idx_alt", C.c_void_p), ("row", C.c_void_p),
                ("count", C.c_void_p), ("bk", C.c_void_p), ("bk_words", C.c_uint32), ("cols", _JgRouteCols), ("words", C.c_uint32)]


ROUTE_SEGS = 8


class Transport:
    """the routed round's transport (jg_route.h) in the device's kernels on the host: emitted rows in, every addressee's
    next command batch out, in the staging's order"""

    def __init__(self, R, G, member_ids, words, cap=None, ord_bits=12):
        self.lib = build()
        self.R, self.G, self.words = R, G, words
        self.member_ids = np.ascontiguousarray(member_ids, np.uint32)
        self.group_bits = 1
        while self.group_bits < 32 and (G - 1) >> self.group_bits:
            self.group_bits += 1
        self.count = np.zeros(R * (R + 4) + ROUTE_SEGS + 2 * R, np.uint32)
        self.bk = np.zeros(4 * 1024 + 4 * (R << self.group_bits) + 64, np.uint32)
        self._alloc(cap or ROUTE_SEGS * 256)
        self.ord_bits = ord_bits
        self.repeats = 0

    def _alloc(self, cap):
        cap -= cap % ROUTE_SEGS
        self.cap = cap
        self.key, self.key_alt = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        self.idx, self.idx_alt = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        self.row = np.zeros(cap, capi.MSG_DTYPE)
        self.cols = dict(kind=np.zeros(cap, np.uint8), flag=np.zeros(cap, np.uint8), group=np.zeros(cap, np.uint32), from_=np.zeros(cap, np.uint32),
                         term=np.zeros(cap, np.uint64), id=np.zeros(cap, np.uint64), aux=np.zeros(cap, np.uint64))

    def route(self, senders, mail):
        """senders: per slot dict(xq=XQ_DTYPE array, seq_base, rec=None | (phase, msg_cnt [n], msg [n, per_row] rows)[, phases = the
        map from the queue rows' steps to the round's phases: 3 bits per step, default the identity]); returns
        (per addressee: command columns, rows per (sender -> addressee), kinds per addressee, rows that stay per sender)"""
        R = self.R
        arr = (_WgSender * R)()
        keep = []
        for s, sd in enumerate(senders):
            xq = np.ascontiguousarray(sd["xq"])
            keep.append(xq)
            arr[s].xq, arr[s].xq_n, arr[s].seq_base, arr[s].phases = xq.ctypes.data, len(xq), sd["seq_base"], sd.get("phases", PHASES_IDENTITY)
            if sd.get("rec") is not None:
                step, cnt, msg = sd["rec"]
                cnt, msg = np.ascontiguousarray(cnt, np.uint32), np.ascontiguousarray(msg)
                fsm = np.zeros(len(cnt), np.uint32)
                keep += [cnt, msg, fsm]
                arr[s].rec_n, arr[s].rec_per_row, arr[s].rec_step = len(cnt), msg.shape[1], step
                arr[s].msg_cnt, arr[s].msg, arr[s].fsm_cnt = cnt.ctypes.data, msg.ctypes.data, fsm.ctypes.data
        for attempt in range(8):  # (as the engine: a segment that ran over -> a larger staging, and the pass again; it modifies nothing)
            c = self.cols
            rt = _WgRoute(R, self.G, self.group_bits, self.ord_bits, self.cap, ROUTE_SEGS, self.member_ids.ctypes.data, self.key.ctypes.data, self.key_alt.ctypes.data,
                          self.idx.ctypes.data, self.idx_alt.ctypes.data, self.row.ctypes.data, self.count.ctypes.data, self.bk.ctypes.data, len(self.bk),
                          _JgRouteCols(c["kind"].ctypes.data, c["flag"].ctypes.data, c["group"].ctypes.data, c["from_"].ctypes.data, c["term"].ctypes.data,
                                       c["id"].ctypes.data, c["aux"].ctypes.data), (1 if self.words else 0) | (2 if attempt else 0))
            total = C.c_uint32(0)
            rc = self.lib.hw_route(C.byref(rt), arr, C.addressof(mail.c) if mail is not None else C.addressof(VoteMail(R, self.G).c), C.byref(total))
            if rc != -2:
                break
            self.repeats += 1
            self._alloc(4 * self.cap)
        assert rc == 0, f"host workgroups: the delivering pass does not settle ({rc})"
        per = self.count[:R * (R + 4)].reshape(R, R + 4)
        to = per[:, :R].sum(axis=0)
        assert int(to.sum()) == total.value, (to, total.value)
        out, off = [], 0
        for n in range(R):
            out.append({k: v[off:off + int(to[n])].copy() for k, v in c.items()})
            off += int(to[n])
        kinds = self.count[R * (R + 4) + ROUTE_SEGS + R:][:R].copy()
        stays = per[:, R + 0] + per[:, R + 3]  # JG_ROUTE_KEPT + JG_ROUTE_KEPT_XQ
        return out, per[:, :R].copy(), kinds, stays.copy()
