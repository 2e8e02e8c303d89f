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
  uint64_t n_blocks;        // entries in the side arrays (an AppendEntries row must stay inside them)
  uint32_t msg_per_row, fsm_per_row;
  jg_msg_row* msg_out;  // [n * msg_per_row]
  jg_fsm_row* fsm_out;  // [n * fsm_per_row]
  uint32_t* msg_cnt;    // [n] rows produced by the run starting here (0 elsewhere)
  uint32_t* fsm_cnt;
  uint64_t* bsum_m;     // [ceil(n / JG_BLOCK)] sums of msg_cnt / fsm_cnt over each workgroup-sized tile of rows:
  uint64_t* bsum_f;     //   what the drain's scan starts from (written here: no separate counting launch)
  uint32_t* err;        // 1: output bound exceeded, 2: rows not sorted by group, 3: group out of range,
                        // 5: an AppendEntries row's block range leaves the side arrays (the row is not applied)
  uint64_t now;
  uint32_t seq;
};

// Every lane loads the columns of ITS row (coalesced); the lane that owns a run then reads row i+t of
// its run out of lane+t's registers.  The first version had the owner load the six columns of each of
// its rows itself: one dependent trip to memory per row — a candidate's sixteen VoteResponses took the
// batch's owner lanes sixteen round trips (profiles/README.md, the routed round).  Rows of a run that
// continue in the next wave are still loaded by the owner.
template <uint32_t KINDS = JG_KINDS_ALL>
__device__ __forceinline__ void jg_apply_rows_body(const JgDev& d, const JgRowsArgs& a) {
  uint32_t dec = 0;
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t base = blockIdx.x * JG_BLOCK; base < a.n; base += gridDim.x * JG_BLOCK) {  // (workgroup-uniform trips)
    const uint32_t i = base + threadIdx.x;
    const bool in = i < a.n;
    const uint32_t g = in ? a.group[i] : 0xffffffffu;
    const uint32_t gp = (in && i) ? a.group[i - 1] : 0xffffffffu;
    JgCmd mine{};
    if (in) {
      mine.kind = a.kind[i];
      mine.from = a.from[i];
      mine.flag = a.flag[i];
      mine.term = a.term[i];
      mine.id = a.id[i];
      mine.aux = a.aux[i];
      a.msg_cnt[i] = 0;
      a.fsm_cnt[i] = 0;
      // a device-resident batch is not validated by the host (jg_submit's rows are): a forged
      // AppendEntries row must not read past the block side arrays
      if (mine.kind == JG_CMD_APPEND_ENTRIES && a.blk_id && (mine.aux > a.n_blocks || mine.id > a.n_blocks - mine.aux)) {
        *a.err = 5;
        mine.kind = JG_CMD_NOOP;
      }
    }
    const bool start = in && !(i && gp == g);  // first row of a run
    if (start && i && gp > g) *a.err = 2;
    if (start && g >= d.G) *a.err = 3;
    const bool owner = start && g < d.G;
    // the run ends at the next run start (or the end of the batch); beyond this wave: look it up
    const uint64_t bounds = __ballot(start || !in);
    const uint64_t above = lane == 63 ? 0ull : bounds >> (lane + 1);
    uint32_t run = 0;
    if (owner) {
      if (above) {
        run = (uint32_t)__ffsll((long long)above);
      } else {
        uint32_t j = i + (64 - lane);
        while (j < a.n && a.group[j] == g) j++;
        run = j - i;
      }
    }
    uint32_t longest = run;
#pragma unroll
    for (int off = 32; off; off >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, off, 64));
    JgLane L;
    jg_msg_row* m0 = nullptr;
    jg_fsm_row* f0 = nullptr;
    if (owner) {
      jg_load(d, L, g);
      L.now = a.now;
      L.seq = a.seq;
      m0 = a.msg_out + (size_t)i * a.msg_per_row;
      f0 = a.fsm_out + (size_t)i * a.fsm_per_row;
      L.mp = m0;
      L.mend = m0 + (size_t)run * a.msg_per_row;
      L.fp = f0;
      L.fend = f0 + (size_t)run * a.fsm_per_row;
    }
    for (uint32_t t = 0; t < longest; t++) {  // (wave-uniform)
      const int src = (int)((lane + t) & 63u);
      JgCmd c;
      c.kind = (uint8_t)__shfl((int)mine.kind, src, 64);
      c.from = (uint32_t)__shfl((int)mine.from, src, 64);
      c.flag = (uint8_t)__shfl((int)mine.flag, src, 64);
      c.term = __shfl(mine.term, src, 64);
      c.id = __shfl(mine.id, src, 64);
      c.aux = __shfl(mine.aux, src, 64);
      if (owner && t < run) {
        if (lane + t >= 64) {  // the run continues in the next wave's rows
          const uint32_t k = i + t;
          c.kind = a.kind[k];
          c.from = a.from[k];
          c.flag = a.flag[k];
          c.term = a.term[k];
          c.id = a.id[k];
          c.aux = a.aux[k];
          if (c.kind == JG_CMD_APPEND_ENTRIES && a.blk_id && (c.aux > a.n_blocks || c.id > a.n_blocks - c.aux)) c.kind = JG_CMD_NOOP;
        }
        jg_apply<KINDS>(d, L, c, a.blk_id, a.blk_next);
      }
    }
    uint32_t cm = 0, cf = 0;
    if (owner) {
      cm = (uint32_t)(L.mp - m0), cf = (uint32_t)(L.fp - f0);
      a.msg_cnt[i] = cm;
      a.fsm_cnt[i] = cf;
      if (L.overflow) *a.err = 1;
      dec += L.decisions;
      jg_store<KINDS != JG_KINDS_ELECTION>(d, L);  // (jg_store_dirty measured no gain here and costs 26 VGPRs: the run-per-lane body has it)
    }
    // the tile's sums (tile = the JG_BLOCK rows this workgroup just served)
    __shared__ uint32_t red_m[JG_BLOCK / 64], red_f[JG_BLOCK / 64];
#pragma unroll
    for (int off = 32; off; off >>= 1) {
      cm += __shfl_down(cm, off, 64);
      cf += __shfl_down(cf, off, 64);
    }
    if (lane == 0) red_m[threadIdx.x >> 6] = cm, red_f[threadIdx.x >> 6] = cf;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tm = 0, tf = 0;
#pragma unroll
      for (int w = 0; w < JG_BLOCK / 64; w++) tm += red_m[w], tf += red_f[w];
      a.bsum_m[base / JG_BLOCK] = tm;
      a.bsum_f[base / JG_BLOCK] = tf;
    }
    __syncthreads();
  }
  jg_block_count(d.blk_decisions, dec);
}

__global__ __launch_bounds__(JG_BLOCK) JG_GSM_OCC void k_apply_rows(JgDev d, JgRowsArgs a) { jg_apply_rows_body(d, a); }
// The steps of several engines that share a stream in ONE launch (blockIdx.y = engine): the routed round of a
// cluster applies up to two batches per node and round - seven launches of ~30 us one behind the other, the
// longest (the candidate node's) setting the pace of each.  Jobs live in device memory.
struct JgApplyJob {
  JgDev d;
  JgRowsArgs a;
};
__global__ __launch_bounds__(JG_BLOCK) JG_GSM_OCC void k_apply_rows_multi(const JgApplyJob* __restrict__ jobs) {
  const JgApplyJob& j = jobs[blockIdx.y];  // (through the reference: 64 us per launch; a by-value copy of the job went to scratch: 198 us)
  jg_apply_rows_body(j.d, j.a);
}

// ---- a RUN per lane -----------------------------------------------------------------------------------
// The batches a cluster's transport delivers are runs of 4 to 16 rows per group (a voter's four copies of a
// VoteRequest, candidate.rs:30-37; a candidate's sixteen VoteResponses): with a lane per ROW, one lane of the
// wave walks the run while the run's other lanes wait - 10 lanes of 64 at work, every wave as slow as its
// longest run (104 us for the 1.3 M delivered rows of a configs[4] round).  Here a workgroup stages a tile of
// JG_RUN_TILE rows in LDS (coalesced loads, as before), compacts the run starts of the tile into a list, and
// every lane takes a RUN off that list: ~160 of 256 lanes at work.  A run that crosses the tile's end is
// finished from global memory by the lane that started it.  Outputs, counts and tile sums exactly as
// jg_apply_rows_body leaves them.
// A small batch (a round of a 3-replica cluster's elections: 20 k rows per node in runs of 2-4) takes tiles of
// JG_RUN_TILE_SMALL rows: four times the workgroups (21 tiles of 1024 do not fill a device of 256 CUs), and a tile's
// ~100 runs are one trip of its 256 lanes where the ~400 runs of a 1024-row tile were two, one behind the other.
#define JG_RUN_TILE 1024u
#define JG_RUN_TILE_SMALL 256u
#define JG_RUN_SMALL_BATCH 200000u  // rows: batches up to here take the small tile
template <uint32_t KINDS = JG_KINDS_ALL, uint32_t TILE = JG_RUN_TILE>
__device__ __forceinline__ void jg_apply_runs_body(const JgDev& d, const JgRowsArgs& a) {
  static_assert(TILE % JG_BLOCK == 0 && TILE <= 65536, "tile geometry");
  constexpr uint32_t T = TILE, SUB = TILE / JG_BLOCK;
  __shared__ uint64_t s_term[T], s_id[T], s_aux[T];
  __shared__ uint32_t s_group[T], s_from[T];
  __shared__ uint8_t s_kind[T], s_flag[T];
  __shared__ uint16_t s_start[T];
  __shared__ uint32_t s_n, s_bm[SUB], s_bf[SUB];
  uint32_t dec = 0;
  for (uint32_t tile0 = blockIdx.x * T; tile0 < a.n; tile0 += gridDim.x * T) {  // (workgroup-uniform trips)
    if (threadIdx.x == 0) s_n = 0;
    if (threadIdx.x < SUB) s_bm[threadIdx.x] = 0, s_bf[threadIdx.x] = 0;
#pragma unroll
    for (uint32_t k = 0; k < SUB; k++) {
      const uint32_t j = k * JG_BLOCK + threadIdx.x, i = tile0 + j;
      if (i >= a.n) continue;
      uint8_t kind = a.kind[i];
      const uint64_t id = a.id[i], aux = a.aux[i];
      // (a device-resident batch is not validated by the host: see jg_apply_rows_body)
      if (kind == JG_CMD_APPEND_ENTRIES && a.blk_id && (aux > a.n_blocks || id > a.n_blocks - aux)) {
        *a.err = 5;
        kind = JG_CMD_NOOP;
      }
      s_kind[j] = kind, s_flag[j] = a.flag[i], s_from[j] = a.from[i], s_group[j] = a.group[i];
      s_term[j] = a.term[i], s_id[j] = id, s_aux[j] = aux;
      a.msg_cnt[i] = 0;
      a.fsm_cnt[i] = 0;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < SUB; k++) {
      const uint32_t j = k * JG_BLOCK + threadIdx.x, i = tile0 + j;
      if (i >= a.n) continue;
      const uint32_t g = s_group[j];
      const uint32_t gp = j ? s_group[j - 1] : (i ? a.group[i - 1] : 0xffffffffu);
      const bool start = !(i && gp == g);
      if (start && i && gp > g) *a.err = 2;
      if (start && g >= d.G) *a.err = 3;
      if (start && g < d.G) s_start[atomicAdd(&s_n, 1u)] = (uint16_t)j;  // (the order of the list is immaterial: groups are independent)
    }
    __syncthreads();
    const uint32_t n_runs = s_n;
    for (uint32_t r = threadIdx.x; r < n_runs; r += JG_BLOCK) {
      const uint32_t j = s_start[r], i = tile0 + j, g = s_group[j];
      uint32_t run = 1;
      while (j + run < T && i + run < a.n && s_group[j + run] == g) run++;
      if (j + run == T)
        while (i + run < a.n && a.group[i + run] == g) run++;
      JgLane L;
      jg_load(d, L, g);
      const JgLane O = L;
      L.now = a.now;
      L.seq = a.seq;
      jg_msg_row* const m0 = a.msg_out + (size_t)i * a.msg_per_row;
      jg_fsm_row* const f0 = a.fsm_out + (size_t)i * a.fsm_per_row;
      L.mp = m0;
      L.mend = m0 + (size_t)run * a.msg_per_row;
      L.fp = f0;
      L.fend = f0 + (size_t)run * a.fsm_per_row;
      for (uint32_t t = 0; t < run; t++) {
        JgCmd c;
        if (j + t < T) {
          c.kind = s_kind[j + t], c.from = s_from[j + t], c.flag = s_flag[j + t];
          c.term = s_term[j + t], c.id = s_id[j + t], c.aux = s_aux[j + t];
        } else {  // the run continues in the next tile's rows
          const uint32_t k = i + t;
          c.kind = a.kind[k], c.from = a.from[k], c.flag = a.flag[k];
          c.term = a.term[k], c.id = a.id[k], c.aux = a.aux[k];
          if (c.kind == JG_CMD_APPEND_ENTRIES && a.blk_id && (c.aux > a.n_blocks || c.id > a.n_blocks - c.aux)) c.kind = JG_CMD_NOOP;
        }
        jg_apply<KINDS>(d, L, c, a.blk_id, a.blk_next);
      }
      const uint32_t cm = (uint32_t)(L.mp - m0), cf = (uint32_t)(L.fp - f0);
      a.msg_cnt[i] = cm;
      a.fsm_cnt[i] = cf;
      if (cm) atomicAdd(&s_bm[j / JG_BLOCK], cm);
      if (cf) atomicAdd(&s_bf[j / JG_BLOCK], cf);
      if (L.overflow) *a.err = 1;
      dec += L.decisions;
      jg_store_dirty<KINDS != JG_KINDS_ELECTION>(d, L, O);
    }
    __syncthreads();
    // the sums of the JG_BLOCK-row tiles the drain's scan starts from
    if (threadIdx.x < SUB && tile0 + threadIdx.x * JG_BLOCK < a.n) {
      a.bsum_m[tile0 / JG_BLOCK + threadIdx.x] = s_bm[threadIdx.x];
      a.bsum_f[tile0 / JG_BLOCK + threadIdx.x] = s_bf[threadIdx.x];
    }
    __syncthreads();
  }
  jg_block_count(d.blk_decisions, dec);
}
// (one batch, arguments by value: what jg_step takes with JG_APPLY_RUNS=1 - the fuzz suites then run through this body)
__global__ __launch_bounds__(JG_BLOCK) JG_GSM_OCC void k_apply_runs(JgDev d, JgRowsArgs a) { jg_apply_runs_body(d, a); }
__global__ __launch_bounds__(JG_BLOCK) JG_GSM_OCC void k_apply_runs_small(JgDev d, JgRowsArgs a) { jg_apply_runs_body<JG_KINDS_ALL, JG_RUN_TILE_SMALL>(d, a); }
__global__ __launch_bounds__(JG_BLOCK) JG_GSM_OCC void k_apply_runs_multi(const JgApplyJob* __restrict__ jobs) {
  const JgApplyJob& j = jobs[blockIdx.y];
  jg_apply_runs_body(j.d, j.a);
}
__global__ __launch_bounds__(JG_BLOCK) void k_apply_vote_runs_multi(const JgApplyJob* __restrict__ jobs) {
  const JgApplyJob& j = jobs[blockIdx.y];
  jg_apply_runs_body<JG_KINDS_ELECTION>(j.d, j.a);
}
__global__ __launch_bounds__(JG_BLOCK) JG_GSM_OCC void k_apply_runs_multi_small(const JgApplyJob* __restrict__ jobs) {
  const JgApplyJob& j = jobs[blockIdx.y];
  jg_apply_runs_body<JG_KINDS_ALL, JG_RUN_TILE_SMALL>(j.d, j.a);
}
__global__ __launch_bounds__(JG_BLOCK) void k_apply_vote_runs_multi_small(const JgApplyJob* __restrict__ jobs) {
  const JgApplyJob& j = jobs[blockIdx.y];
  jg_apply_runs_body<JG_KINDS_ELECTION, JG_RUN_TILE_SMALL>(j.d, j.a);
}

// ---- drain-time compaction, entirely on the device -------------------------------------------
// cnt[i] rows sit at src + i*per_row; the drained sequence is their concatenation in row
// order (= group order, emission order within a group).  Exclusive scan of cnt in three
// steps: per-workgroup sums (JG_SCAN_TILE counts each; at step time, by k_apply_rows itself), a single-workgroup scan of
// those sums (one launch for all pending steps at drain time), then each workgroup re-scans
// its counts (wave64 shuffles + an LDS hop across the four waves), adds its base and
// copies its rows to their final place in ONE buffer per queue, which travels to the pinned
// host queue in one copy.
#define JG_SCAN_ITEMS 1  // (a tile = the rows one workgroup of k_apply_rows serves: it writes the tile sums itself)
#define JG_SCAN_TILE (JG_BLOCK * JG_SCAN_ITEMS)

// inclusive scan of one value per lane across the workgroup; returns the exclusive prefix of
// the calling thread and the workgroup total through *total
__device__ __forceinline__ uint32_t jg_block_exclusive_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wave_tot[JG_BLOCK / 64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t up = __shfl_up(inc, off, 64);
    if (lane >= (uint32_t)off) inc += up;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (uint32_t w = 0; w < JG_BLOCK / 64; w++) {
    base += w < wave ? wave_tot[w] : 0u;
    tot += wave_tot[w];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// tile sums of both count arrays again (after the cluster transport took rows out of a step's slots)
__global__ __launch_bounds__(JG_BLOCK) void k_count_block_sums(const uint32_t* __restrict__ cnt_m,
                                                               const uint32_t* __restrict__ cnt_f, uint32_t n,
                                                               uint64_t* __restrict__ bsum_m,
                                                               uint64_t* __restrict__ bsum_f) {
  const uint32_t base = blockIdx.x * JG_SCAN_TILE + threadIdx.x * JG_SCAN_ITEMS;
  uint32_t vm = 0, vf = 0;
#pragma unroll
  for (int k = 0; k < JG_SCAN_ITEMS; k++) {
    vm += (base + k < n) ? cnt_m[base + k] : 0u;
    vf += (base + k < n) ? cnt_f[base + k] : 0u;
  }
  uint32_t tot_m, tot_f;
  (void)jg_block_exclusive_scan(vm, &tot_m);
  (void)jg_block_exclusive_scan(vf, &tot_f);
  if (threadIdx.x == 0) {
    bsum_m[blockIdx.x] = tot_m;
    bsum_f[blockIdx.x] = tot_f;
  }
}

// drain time, one launch for every pending step: workgroup b turns the tile sums of job b
// (one count array of one step) into exclusive prefixes in place and writes the job's grand
// total.  The job table and the totals live in pinned host memory (read / written in place).
struct JgScanJob {
  uint64_t* bsum;
  uint32_t nb, pad;
};
__global__ __launch_bounds__(JG_BLOCK) void k_scan_block_sums(const JgScanJob* __restrict__ jobs,
                                                              uint64_t* __restrict__ totals) {
  __shared__ uint64_t carry_s;
  uint64_t* bsum = jobs[blockIdx.x].bsum;
  const uint32_t nb = jobs[blockIdx.x].nb;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += JG_BLOCK) {
    const uint32_t i = base + threadIdx.x;
    const uint64_t v = i < nb ? bsum[i] : 0;
    // tile sums fit 32 bits (<= 1024 counts of at most a few rows each)
    uint32_t tot;
    const uint32_t ex = jg_block_exclusive_scan((uint32_t)v, &tot);
    const uint64_t carry = carry_s;
    if (i < nb) bsum[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

template <typename Row>
__global__ __launch_bounds__(JG_BLOCK) void k_scan_gather(const uint32_t* __restrict__ cnt, uint32_t n,
                                                          const uint64_t* __restrict__ bsum, uint32_t per_row,
                                                          const Row* __restrict__ src, Row* __restrict__ dst) {
  const uint32_t base = blockIdx.x * JG_SCAN_TILE + threadIdx.x * JG_SCAN_ITEMS;
  uint32_t c[JG_SCAN_ITEMS], v = 0;
#pragma unroll
  for (int k = 0; k < JG_SCAN_ITEMS; k++) {
    c[k] = (base + k < n) ? cnt[base + k] : 0u;
    v += c[k];
  }
  uint32_t tot;
  uint64_t off = bsum[blockIdx.x] + jg_block_exclusive_scan(v, &tot);
#pragma unroll
  for (int k = 0; k < JG_SCAN_ITEMS; k++) {
    const Row* s = src + (size_t)(base + k) * per_row;
    for (uint32_t r = 0; r < c[k]; r++) dst[off + r] = s[r];
    off += c[k];
  }
}
