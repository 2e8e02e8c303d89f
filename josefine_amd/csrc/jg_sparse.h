// jg_sparse.h — the whole Raft state machine over a batch of command rows.
//
// Rows are sorted by group (stable: a group's rows keep their stream order).  The
// lane that sits on the first row of a group's run owns that group for the step:
// it loads the group's state, applies the run in order through RaftHandle::apply
// (mod.rs:471-479 -> jg_apply) and stores the state back.  No CSR arrays are needed,
// so a batch can be produced, sorted and kept entirely on the device.
//
// Output rows go to fixed per-command regions (row i owns msg_per_row message slots
// and fsm_per_row fsm slots; a run of k rows owns k regions back to back), sized by
// the per-command upper bounds derived in josefine_gpu.hip.  The count of rows a run
// produced is written at the run's first row; k_gather_rows compacts at drain time,
// which yields rows group-major and, within a group, in the reference's emission order.
#pragma once
#include "jg_dense.h"

struct JgRowsArgs {
  uint32_t n;  // command rows
  const uint32_t* group;
  const uint8_t* kind;
  const uint32_t* from;
  const uint64_t* term;
  const uint64_t* id;
  const uint64_t* aux;
  const uint8_t* flag;
  const uint64_t* blk_id;   // Vec<Block> side arrays (chain.rs:86-91)
  const uint64_t* blk_next;
  uint32_t msg_per_row, fsm_per_row;
  jg_msg_row* msg_out;  // [n * msg_per_row]
  jg_fsm_row* fsm_out;  // [n * fsm_per_row]
  uint32_t* msg_cnt;    // [n] rows produced by the run starting here (0 elsewhere)
  uint32_t* fsm_cnt;
  uint32_t* err;        // 1: output bound exceeded, 2: rows not sorted by group, 3: group out of range
  uint64_t now;
  uint32_t seq;
};

__global__ __launch_bounds__(JG_BLOCK) void k_apply_rows(JgDev d, JgRowsArgs a) {
  uint32_t dec = 0;
  for (uint32_t i = blockIdx.x * JG_BLOCK + threadIdx.x; i < a.n; i += gridDim.x * JG_BLOCK) {
    const uint32_t g = a.group[i];
    const uint32_t gp = i ? a.group[i - 1] : 0xffffffffu;
    a.msg_cnt[i] = 0;
    a.fsm_cnt[i] = 0;
    if (i && gp == g) continue;  // not the head of a run
    if (i && gp > g) *a.err = 2;
    if (g >= d.G) {
      *a.err = 3;
      continue;
    }
    uint32_t j = i + 1;
    while (j < a.n && a.group[j] == g) j++;
    JgLane L;
    jg_load(d, L, g);
    L.now = a.now;
    L.seq = a.seq;
    jg_msg_row* m0 = a.msg_out + (size_t)i * a.msg_per_row;
    jg_fsm_row* f0 = a.fsm_out + (size_t)i * a.fsm_per_row;
    L.mp = m0;
    L.mend = m0 + (size_t)(j - i) * a.msg_per_row;
    L.fp = f0;
    L.fend = f0 + (size_t)(j - i) * a.fsm_per_row;
    for (uint32_t k = i; k < j; k++) {
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

// drain-time compaction: copy each run's rows from its region to its final offset
template <typename Row>
__global__ void k_gather_rows(uint32_t n, uint32_t per_row, const uint32_t* __restrict__ cnt,
                              const uint64_t* __restrict__ dst_off, const Row* __restrict__ src,
                              Row* __restrict__ dst) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t c = cnt[i];
    if (!c) continue;
    const Row* s = src + (size_t)i * per_row;
    Row* t = dst + dst_off[i];
    for (uint32_t k = 0; k < c; k++) t[k] = s[k];
  }
}
