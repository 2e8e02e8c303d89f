// josefine_gpu.hip — C ABI (include/josefine_gpu.h) of the MI355X batched
// Chained-Raft engine: host-side marshalling around the gfx950 kernels in
// jg_kernels.h.  There is deliberately no CPU implementation in this library:
// every entry point that computes does so on the device or fails with
// JG_EDEVICE.  What the host does do: validate arguments, bucket command rows by
// group (a stable radix sort of row indices — marshalling, not Raft), move bytes,
// and launch.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>  // plain library sort of the drained fault records
#include <rocprim/device/device_select.hpp>      // jg_step_node's rare path: order-preserving compaction of the general-path rows
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "jg_kernels.h"
#include "jg_route.h"
#include "jg_follower.h"
#include "jg_node.h"

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess)                                                                                \
      return fail(JG_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(_e) + " (no CPU fallback)"); \
  } while (0)

namespace {

// One sparse step whose output rows have not been drained yet.
struct StepRec {
  uint32_t n = 0;  // command rows
  uint32_t seq = 0;
  uint32_t msg_per_row = 0, fsm_per_row = 0;
  uint32_t *d_msg_cnt = nullptr, *d_fsm_cnt = nullptr;
  uint64_t *d_bsum_m = nullptr, *d_bsum_f = nullptr;  // tile sums -> exclusive prefixes at drain
  jg_msg_row* d_msg = nullptr;
  jg_fsm_row* d_fsm = nullptr;
};

// Grow-only device arena for the per-step command blobs and output regions:
// bump allocation, reset when every pending step has been drained.  Keeps
// hipMalloc/hipFree (hundreds of microseconds each) off the per-step path.
struct Arena {
  struct Chunk {
    char* p;
    size_t cap, off;
  };
  std::vector<Chunk> chunks;
  hipError_t alloc(size_t bytes, void** out) {
    bytes = (bytes + 255) & ~size_t(255);
    if (chunks.empty() || chunks.back().off + bytes > chunks.back().cap) {
      size_t cap = std::max<size_t>(bytes, chunks.empty() ? (size_t)32 << 20 : chunks.back().cap * 2);
      void* p = nullptr;
      hipError_t e = hipMalloc(&p, cap);
      if (e != hipSuccess) return e;
      chunks.push_back(Chunk{(char*)p, cap, 0});
    }
    Chunk& c = chunks.back();
    *out = c.p + c.off;
    c.off += bytes;
    return hipSuccess;
  }
  // one chunk of at least `bytes` up front (an idle arena only): a hipMalloc of a few hundred MB takes
  // 6-8 ms on some boxes, and a workload whose steps grow slowly (the routed round of configs[4]) otherwise pays
  // one per doubling and node inside its timed region
  hipError_t reserve(size_t bytes) {
    for (const Chunk& c : chunks)
      if (c.off) return hipSuccess;  // in use: leave it alone
    if (!chunks.empty() && chunks.back().cap >= bytes) return hipSuccess;
    destroy();
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return e;
    chunks.push_back(Chunk{(char*)p, bytes, 0});
    return hipSuccess;
  }
  void reset() {  // keep the largest chunk
    while (chunks.size() > 1) {
      (void)hipFree(chunks.front().p);
      chunks.erase(chunks.begin());
    }
    if (!chunks.empty()) chunks.back().off = 0;
  }
  void destroy() {
    for (Chunk& c : chunks) (void)hipFree(c.p);
    chunks.clear();
  }
};

// Host-side output queue in pinned memory: device rows land here with one async copy at PCIe
// speed (a pageable destination costs a staged, blocking copy), and jg_drain_*_view hands the
// rows to the caller without another pass.
template <typename Row>
struct PinnedQueue {
  Row* p = nullptr;
  size_t cap = 0, n = 0;
  // rows [0, viewed) were handed out by a *_view call: they stay where the caller was pointed
  // until the next drain of THIS queue (any other call may only append behind them)
  size_t viewed = 0;
  std::vector<Row*> retired;  // buffers a view may still point into
  hipError_t reserve(size_t want) {
    if (want <= cap) return hipSuccess;
    size_t ncap = std::max<size_t>(want, std::max<size_t>(cap * 2, 4096));
    Row* q = nullptr;
    hipError_t e = hipHostMalloc((void**)&q, ncap * sizeof(Row), hipHostMallocDefault);
    if (e != hipSuccess) return e;
    if (n) std::memcpy(q, p, n * sizeof(Row));
    if (p) {
      if (viewed) retired.push_back(p);
      else (void)hipHostFree(p);
    }
    p = q;
    cap = ncap;
    return hipSuccess;
  }
  void release_view() {
    if (!viewed) return;
    if (n > viewed) std::memmove(p, p + viewed, (n - viewed) * sizeof(Row));
    n -= viewed;
    viewed = 0;
    for (Row* r : retired) (void)hipHostFree(r);
    retired.clear();
  }
  void destroy() {
    for (Row* r : retired) (void)hipHostFree(r);
    retired.clear();
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = n = viewed = 0;
  }
};

// A column of the commands queued by jg_submit, in pinned host memory: jg_step_node uploads it as it is
// (no staging copy), and jg_submit_reserve hands its tail out for the caller to fill in place.
template <typename T>
struct PinnedVec {
  T* p = nullptr;
  size_t n = 0, cap = 0;
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  void clear() { n = 0; }
  T* data() { return p; }
  const T* data() const { return p; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  hipError_t reserve(size_t want) {
    if (want <= cap) return hipSuccess;
    const size_t ncap = std::max<size_t>(want + want / 2, 4096);
    T* q = nullptr;
    hipError_t e = hipHostMalloc((void**)&q, ncap * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) return e;
    if (n) std::memcpy(q, p, n * sizeof(T));
    if (p) (void)hipHostFree(p);
    p = q;
    cap = ncap;
    return hipSuccess;
  }
  // append src[0..k) (or k zeros)
  hipError_t append(const T* src, size_t k) {
    hipError_t e = reserve(n + k);
    if (e != hipSuccess) return e;
    if (src) std::memcpy(p + n, src, k * sizeof(T));
    else std::memset(p + n, 0, k * sizeof(T));
    n += k;
    return hipSuccess;
  }
  // Two buffers: flip() makes the other one current (empty) and leaves this one as it is - an asynchronous copy out of
  // it may still be in flight (jg_step_node returns before its uploads have completed; the buffer comes round again two
  // steps later, behind that step's synchronisation).  Only the current buffer ever grows or is freed.
  T* alt = nullptr;
  size_t alt_cap = 0;
  void flip() {
    std::swap(p, alt);
    std::swap(cap, alt_cap);
    n = 0;
  }
  void destroy() {
    if (p) (void)hipHostFree(p);
    if (alt) (void)hipHostFree(alt);
    p = alt = nullptr;
    n = cap = alt_cap = 0;
  }
};

// A run of queued output rows that came out of one step (multi-device merge: jg_multi.h).
struct JgSeg {
  uint32_t seq;
  size_t n;
};

}  // namespace

struct jg_engine {
  jg_config cfg;
  JgDev dev;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;  // non-null while a jg_dense_cluster has this node on its lead node's stream: the stream to destroy
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_stage = nullptr, ev_order = nullptr;
  std::vector<void*> allocs;
  uint32_t count_slots = 0;  // workgroup slots of dev.blk_decisions
  uint32_t dense_grid = 0;
  JgDev* d_dev = nullptr;  // device copy of `dev` (k_leader_tick_dense's general path): d_dev2[cur_set]
  JgDev* d_dev2[2] = {nullptr, nullptr};  // one per set of the fault / exceptional-row queues
  int uniform_self = 0;  // the own replica slot if it is the same for every group, else -1
  // device status block {err, irregular_seen, deferred_seen, fault_q_n, xq_n, cold_seen, fault_q_n', xq_n'}
  // (the primed words: the second set of the fault / exceptional-row queues): read back with one
  // copy into its pinned mirror at every synchronisation point
  uint32_t* d_status = nullptr;
  uint32_t* h_status = nullptr;
  uint32_t* d_err = nullptr;
  uint64_t* d_acks_staging = nullptr;  // [R][G] for the host-buffer dense entry point
  uint64_t* d_ones = nullptr;  // one all-ones word: the stride-0 stand-in for an absent ack block / HeartbeatResponse column
  // commands queued by jg_submit (host SoA)
  PinnedVec<uint8_t> p_kind, p_flag;
  PinnedVec<uint32_t> p_group, p_from;
  PinnedVec<uint64_t> p_term, p_id, p_aux, p_blk_id, p_blk_next;
  // which optional columns some jg_submit since the last step actually provided (an absent column is
  // all zeros: jg_step_node does not upload it)
  bool p_has_from = false, p_has_term = false, p_has_aux = false, p_has_flag = false;
  bool p_unchecked = false;  // some rows were committed with JG_COL_UNCHECKED: only jg_step_node may take this batch
  // JG_COL_UPLOAD_NOW: the committed batch on its way to the device before the step is called, on a copy stream of
  // its own (the two directions of the bus are independent: the previous step's outputs travel home meanwhile).
  // Two device buffers by turns: a step's rows are read until the step is settled, the next upload must not wait for that
  struct RowLayout {
    size_t n = 0, nb = 0, bytes = 0;
    bool has_from = false, has_term = false, has_aux = false, has_flag = false;
    size_t o_id = 0, o_term = 0, o_aux = 0, o_bid = 0, o_bnext = 0, o_group = 0, o_from = 0, o_kind = 0, o_flag = 0;
    bool same_batch(const RowLayout& o) const {
      return n == o.n && nb == o.nb && has_from == o.has_from && has_term == o.has_term && has_aux == o.has_aux && has_flag == o.has_flag;
    }
  };
  struct EarlyUpload {
    hipStream_t st = nullptr;
    hipEvent_t ev_up = nullptr;               // behind the copies of the batch in flight
    hipEvent_t ev_free[2] = {nullptr, nullptr};  // behind the last step (and its settling) that read buf[k]
    bool read[2] = {false, false};
    char* buf[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0};
    int turn = 0;        // the buffer the next upload takes
    int last_used = -1;  // the buffer the last node step read its rows from (-1: the arena's)
    bool valid = false;  // a batch is on its way / there, laid out as `lay`
    RowLayout lay;
  } up;
  uint32_t p_kinds_seen = 0;  // bit 0: an AppendEntries row is queued, bit 1: a Heartbeat row
  // pinned staging for the upload of one step (reused; guarded by ev_stage)
  char* stage = nullptr;
  size_t stage_cap = 0;
  bool stage_busy = false;
  Arena arenas[2];   // [cur_arena]: steps since the last prefetch point; the other: the batch in transfer
  int cur_arena = 0;
  std::vector<StepRec> recs;
  // jg_drain_prefetch: one batch of steps whose compaction + transfer to the host queues runs on
  // `copy_stream` while the engine keeps stepping (phase 1: scan enqueued, 2: gathers enqueued)
  struct DrainBatch {
    std::vector<StepRec> recs;
    int arena = 0, set = 0, phase = 0;
    uint32_t seq_hi = 0;  // the engine's step number when the batch was formed (every record of it is at or below)
    bool to_landing = false;  // rows go to l_msgs / l_fsm (from offset 0) instead of behind q_msgs / q_fsm
    size_t at_m = 0, at_f = 0, add_m = 0, add_f = 0;
    uint32_t nf = 0, nx = 0;
    uint64_t irr_gen = 0;  // e->irr_gen at the prefetch point
  } inflight;
  bool pipelined = false;  // drains deliver up to the latest prefetch point and never synchronise later steps
  // The engine's own drain thread (created at the first jg_drain_prefetch): it waits for the scan
  // of the batch in transfer, issues phase B the moment the totals are known - whatever the
  // caller's thread is doing - and waits for the batch to land.
  struct DrainThread {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    int state = 0;  // 0 idle, 1 batch posted, 2 batch landed (or failed)
    bool quit = false;
    int rc = 0;
    std::string err;
  }* drain_thread = nullptr;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_steps = nullptr, ev_scan = nullptr, ev_done = nullptr;
  // two sets of the device-side fault / exceptional-row queues: kernels append to [cur_set] while
  // the other one is being copied out
  JgFaultRec* fq[2] = {nullptr, nullptr};
  JgXqRec* xqb[2] = {nullptr, nullptr};
  int cur_set = 0;
  uint32_t* h_cnt = nullptr;  // pinned: {fault_q_n, xq_n} of the batch in transfer, then the 8 status words at its prefetch point
  PinnedQueue<jg_msg_row> q_msgs;
  PinnedQueue<jg_fsm_row> q_fsm;
  // pipelined drains: the batch in transfer lands in queues of its own (nothing is ever moved
  // behind rows a view still covers); each is handed over - a pointer swap when the consumer has
  // taken everything before it - at the next drain call of its kind
  PinnedQueue<jg_msg_row> l_msgs;
  PinnedQueue<jg_fsm_row> l_fsm;
  bool landed_m = false, landed_f = false;
  std::vector<jg_fault_row> q_faults;
  std::vector<jg_compact_row> q_compacted;  // jg_chain_compact_resident -> jg_drain_compacted
  JgCompactRow* d_compact = nullptr;        // device list of one compact pass (lazily allocated)
  uint32_t* d_compact_n = nullptr;
  uint32_t compact_cap = 0;
  PinnedQueue<jg_fault_row> h_faults;  // pinned landing buffers of the device queues (faults: sorted rows + steps)
  PinnedQueue<uint32_t> h_fault_seq;
  uint64_t *fs_k0 = nullptr, *fs_k1 = nullptr;  // device scratch of the fault sort (grow-only)
  uint32_t *fs_v0 = nullptr, *fs_v1 = nullptr, *fs_seq = nullptr;
  jg_fault_row* fs_rows = nullptr;
  void* fs_tmp = nullptr;
  size_t fs_cap = 0, fs_tmp_bytes = 0;
  uint32_t* fs_bk = nullptr;  // the fault sort's bucket counters (grow-only)
  size_t fs_bk_words = 0;
  uint32_t fault_floor[2] = {0, 0};  // per buffer set: a step number below every record the set can hold
  PinnedQueue<JgXqRec> h_xq;
  // the gathers compact into device memory and ONE copy per queue takes the rows to the pinned host
  // queue (a DMA engine's work, not the gather kernels' across PCIe)
  void *d_stage_m = nullptr, *d_stage_f = nullptr;
  size_t stage_m_cap = 0, stage_f_cap = 0;
  std::vector<JgXqRec> xq_tmp;
  JgScanJob* h_jobs = nullptr;  // pinned: drain-time scan jobs and their totals
  uint64_t* h_totals = nullptr;
  size_t scan_cap = 0;
  uint32_t seq = 0;
  // multi-device (jg_multi.h): a parent owns a router and no device state; its shards record which
  // step every queued output row belongs to
  struct JgRouter* router = nullptr;
  jg_engine* parent = nullptr;
  bool track_segs = false;
  std::vector<JgSeg> seg_m, seg_f;
  std::vector<uint32_t> q_fault_seq;
  bool stepped = false;
  // Some group's chain may have left FAST form (then k_dense_slow runs behind the
  // dense kernel).  Set by every sparse step, cleared at the next synchronisation
  // point if the device-side flag is still 0.
  bool maybe_irregular = false;
  bool flag_check_pending = false;
  uint64_t irr_gen = 0;  // bumped by every step that sets flag_check_pending (a pipelined status snapshot settles the flag only if nothing did since)
  bool slow_scheduled_ever = false;  // some dense launch had k_dense_slow behind it
  uint64_t n_cmds = 0, n_dense = 0, n_launch = 0;
  // set while a jg_dense_cluster round is being captured into a hipGraph: the node kernels then
  // take logical time and step number from this device-resident clock instead of their arguments
  JgClock* replay_clock = nullptr;
  uint32_t replay_slot = 0;
  // jg_step_node: the inbox / outbox columns of the node step, their pinned host mirrors, rocPRIM scratch
  struct NodeStep {
    bool ready = false;
    JgNodeCols cols{};
    jg_leader_beat* o_beat = nullptr;  // device outbox
    uint64_t *o_ae = nullptr, *o_answer = nullptr, *o_hbc = nullptr;
    jg_leader_beat* h_beat = nullptr;  // pinned mirrors
    uint64_t *h_ae = nullptr, *h_answer = nullptr, *h_hbc = nullptr;
    uint64_t *h_in_answers = nullptr, *h_in_hbc = nullptr;  // pinned [R][G]: column inbound (jg_node_inbox_columns)
    uint32_t col_mask = 0, col_hbc_mask = 0;                // slots handed out for the next step / with their hb_commit column
    uint32_t* d_nsparse = nullptr;     // {general-path rows}
    uint32_t* h_nsparse = nullptr;     // pinned
    // the general path's rows as (group << 32 | arrival index, arrival index) pairs, appended by k_node_route (grow-only),
    // and the bucket pass that orders them (jg_route.h: hist / scan / scatter + k_bucket_order)
    uint64_t* sp_key = nullptr;
    uint32_t* sp_idx = nullptr;
    size_t sp_cap = 0;
    uint32_t* bk_mem = nullptr;
    uint32_t bk_words = 0, bk_buckets = 0, bk_tile_bits = 0;
    uint32_t group_bits = 1;
    hipEvent_t ev_out = nullptr;
    hipEvent_t ev_cols = nullptr;      // behind the uploads of the handed-out columns: the pinned buffers are free again
    bool cols_in_flight = false;
    // JG_NODE_ASYNC: a step that returned without looking at its general-path row count (settled by node_settle)
    struct Pending {
      bool on = false;
      JgNodeRows rows{};
      size_t n = 0, nb = 0, fsm_rec_seq = 0;
      uint64_t now_ms = 0;
      uint32_t flags = 0, col_mask = 0, seq_general = 0, seq_leader = 0, seq_follower = 0, seq_end = 0;
    } pending;
    jg_node_outbox last{};
    uint32_t last_flags = 0;
    // multi-device parent: the shards' columns concatenated
    std::vector<jg_leader_beat> cat_beat;
    std::vector<uint64_t> cat_ae, cat_answer, cat_hbc;
  } node;
  // jg_kernel_timing: HIP event pairs around the dense tick kernel itself (not the slow kernel
  // behind it), a ring of the most recent launches, read after the fact
  static constexpr int KT_RING = 256;
  std::vector<hipEvent_t> kt_ev;  // 2 * KT_RING once enabled
  bool kt_on = false;
  uint64_t kt_n = 0;
  uint32_t kt_every = 1, kt_seen = 0;  // every kt_every-th dense launch is timed (two event records cost the stream a few microseconds)
};

namespace {

template <typename T>
int dev_alloc(jg_engine* e, T** p, size_t n) {
  void* q = nullptr;
  size_t bytes = std::max<size_t>(n * sizeof(T), 16);
  HIPCHK(hipMalloc(&q, bytes));
  HIPCHK(hipMemsetAsync(q, 0, bytes, e->stream));
  e->allocs.push_back(q);
  *p = (T*)q;
  return JG_OK;
}

inline uint32_t grid_for(size_t n, uint32_t cap) {
  size_t b = (n + JG_BLOCK - 1) / JG_BLOCK;
  if (b < 1) b = 1;
  return (uint32_t)std::min<size_t>(b, cap);
}

// Output-row bounds per command (messages, fsm rows) for R replicas — the maximum
// over all roles and kinds:
//   Tick / Timeout      : DROP + (R-1) VoteRequest + Heartbeat (R = 1) | Heartbeat + (R-1) AppendEntries  -> R+1
//   HeartbeatResponse   : replicate(): R-1 AppendEntries
//   Heartbeat           : FLUSH + HeartbeatResponse; one Apply range
//   VoteResponse        : DROP + Heartbeat on elect()
//   ClientRequest       : forward / queue; Notify + Apply range
inline uint32_t msg_bound(uint32_t R) { return R + 1 < 2 ? 2 : R + 1; }
inline uint32_t fsm_bound() { return 2; }

template <int R>
void launch_dense(jg_engine* e, const uint64_t* acks, uint32_t n_ticks, const JgLeaderNode* nd) {
  const size_t stride = (size_t)e->cfg.n_groups * e->cfg.n_replicas;
  struct Lap {  // (event pair around the one launch below, when jg_kernel_timing is on)
    jg_engine* e;
    bool on;
    explicit Lap(jg_engine* e_) : e(e_), on(e_->kt_on && e_->kt_seen++ % e_->kt_every == 0) {
      if (on) (void)hipEventRecord(e->kt_ev[2 * (e->kt_n % jg_engine::KT_RING)], e->stream);
    }
    ~Lap() {
      if (on) (void)hipEventRecord(e->kt_ev[2 * (e->kt_n++ % jg_engine::KT_RING) + 1], e->stream);
    }
  } lap(e);
  if (nd) {  // node tick: HeartbeatResponses in, the Tick's outbox out
    // (an absent input column is a stride-0 view of one all-ones word for this kernel: no branch around loads)
    if (nd->fsm_delta)  // jg_step_node: the tick leaves its fsm_tx output behind as one word per group
      hipLaunchKernelGGL((k_leader_node_tick<R, true>), dim3(e->dense_grid), dim3(JG_BLOCK), 0, e->stream,
                         jg_dense_hot_of(e->dev), (const JgDev*)e->d_dev, acks ? acks : (const uint64_t*)e->d_ones, e->seq,
                         e->uniform_self, *nd);
    else
      hipLaunchKernelGGL((k_leader_node_tick<R, false>), dim3(e->dense_grid), dim3(JG_BLOCK), 0, e->stream,
                         jg_dense_hot_of(e->dev), (const JgDev*)e->d_dev, acks ? acks : (const uint64_t*)e->d_ones, e->seq,
                         e->uniform_self, *nd);
  }
  else if (n_ticks > 1)  // temporal fusion: state read once, written once per launch
    hipLaunchKernelGGL(k_leader_tick_dense_n<R>, dim3(e->dense_grid), dim3(JG_BLOCK), 0, e->stream, e->dev, acks,
                       n_ticks, stride, e->seq, e->uniform_self);
  else if (e->maybe_irregular)  // k_dense_slow is scheduled behind it: the kernel hands its general path to that one too
    hipLaunchKernelGGL((k_leader_tick_dense<R, true>), dim3(e->dense_grid), dim3(JG_BLOCK), 0, e->stream,
                       jg_dense_hot_of(e->dev), (const JgDev*)e->d_dev, acks, e->seq, e->uniform_self);
  else
    hipLaunchKernelGGL((k_leader_tick_dense<R, false>), dim3(e->dense_grid), dim3(JG_BLOCK), 0, e->stream,
                       jg_dense_hot_of(e->dev), (const JgDev*)e->d_dev, acks, e->seq, e->uniform_self);
}

int node_settle(jg_engine* e);  // (jg_step_node with JG_NODE_ASYNC: the step's general path, if it has one, runs when the step is settled)
int dense_step(jg_engine* e, const uint64_t* acks_dev, uint32_t n_ticks = 1, const JgLeaderNode* nd = nullptr) {
  if (!(nd && nd->sparse_mode == 2u)) {  // (not from inside node_settle's own catch-up pass)
    const int rc = node_settle(e);
    if (rc) return rc;
  }
  e->stepped = true;
  e->seq++;  // tick t of this launch carries sequence number seq + t
  switch (e->cfg.n_replicas) {
    case 1: launch_dense<1>(e, acks_dev, n_ticks, nd); break;
    case 2: launch_dense<2>(e, acks_dev, n_ticks, nd); break;
    case 3: launch_dense<3>(e, acks_dev, n_ticks, nd); break;
    case 4: launch_dense<4>(e, acks_dev, n_ticks, nd); break;
    case 5: launch_dense<5>(e, acks_dev, n_ticks, nd); break;
    case 6: launch_dense<6>(e, acks_dev, n_ticks, nd); break;
    case 7: launch_dense<7>(e, acks_dev, n_ticks, nd); break;
    default: launch_dense<8>(e, acks_dev, n_ticks, nd); break;
  }
  e->n_launch++;
  // the slow kernel behind it: when a sparse step may have left a leader with an irregular
  // chain, and behind every node tick and every T-tick launch (their general path: a
  // HeartbeatResponse without the commit, an escaped lag field, an ack above the head)
  if (e->maybe_irregular || nd || n_ticks > 1) {
    e->slow_scheduled_ever = true;
    JgLeaderNode none{};
    if (nd)
      hipLaunchKernelGGL(k_dense_slow<true>, dim3(JG_SHARDS), dim3(JG_BLOCK), 0, e->stream, e->dev, acks_dev, n_ticks,
                         (size_t)e->cfg.n_groups * e->cfg.n_replicas, e->seq, *nd);
    else
      hipLaunchKernelGGL(k_dense_slow<false>, dim3(JG_SHARDS), dim3(JG_BLOCK), 0, e->stream, e->dev, acks_dev, n_ticks,
                         (size_t)e->cfg.n_groups * e->cfg.n_replicas, e->seq, none);
    e->n_launch++;
  }
  HIPCHK(hipGetLastError());
  e->seq += n_ticks - 1;
  e->n_dense += (uint64_t)e->cfg.n_groups * n_ticks;
  return JG_OK;
}

// The exceptional-message queue of the dense node steps, allocated at their first use:
// (R + 3) rows per group bound what one tick can emit outside the mailbox vocabulary.
// The device-resident copy of `dev` the ack-only dense kernel reads on its general path.
// `dev` as the kernels of buffer set k see it
JgDev dev_for_set(const jg_engine* e, int k) {
  JgDev d = e->dev;
  d.fault_q = e->fq[k];
  d.fault_q_n = e->d_status + (k ? 6 : 3);
  d.xq = e->dev.xq ? e->xqb[k] : nullptr;
  d.xq_n = e->d_status + (k ? 7 : 4);
  return d;
}
int push_dev_copy(jg_engine* e) {
  for (int k = 0; k < 2; k++) {
    const JgDev d = dev_for_set(e, k);
    HIPCHK(hipMemcpyAsync(e->d_dev2[k], &d, sizeof(JgDev), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));  // (`d` is a local)
  }
  e->d_dev = e->d_dev2[e->cur_set];
  return JG_OK;
}

int ensure_xq(jg_engine* e) {
  if (e->dev.xq) return JG_OK;
  const size_t cap = std::max<size_t>((size_t)(e->cfg.n_replicas + 3) * e->cfg.n_groups, 65536);
  if (cap > 0xffffffffull) return fail(JG_EINVAL, "too many groups for the dense node tick");
  for (int k = 0; k < 2; k++) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, cap * sizeof(JgXqRec)));
    e->allocs.push_back(p);
    e->xqb[k] = (JgXqRec*)p;
  }
  e->dev.xq = e->xqb[e->cur_set];
  e->dev.xq_cap = (uint32_t)cap;
  return push_dev_copy(e);
}

// device-side error flags of a status block {err, irregular_seen, deferred_seen, fault_q_n, xq_n, cold_seen, fault_q_n', xq_n'}
int status_check(const jg_engine* e, const uint32_t* st) {
  const uint32_t err = st[0];
  if (err == 1) return fail(JG_EDEVICE, "internal: an output row exceeded its per-command bound");
  if (err == 2) return fail(JG_EINVAL, "device command rows were not sorted by group");
  if (err == 3) return fail(JG_EINVAL, "device command rows name a group out of range");
  if (err == 4) return fail(JG_EDEVICE, "internal: deferred-group list overflow");
  if (err == 5) return fail(JG_EINVAL, "device command rows: an AppendEntries row's block range is outside the side arrays");
  if (err == 6) return fail(JG_EINVAL, "jg_step_node: a row names a sender whose answers arrived as a column (jg_node_inbox_columns) in the same step");
  if (err == 7) return fail(JG_EINVAL, "jg_step_node: a row committed with JG_COL_UNCHECKED names a group or a kind out of range (it was not applied)");
  if (st[4] > e->dev.xq_cap || st[7] > e->dev.xq_cap)
    return fail(JG_ECAPACITY, "exceptional-message queue overflow: drain the messages more often");
  return JG_OK;
}

// Everything that needs the stream idle first calls this: synchronise, surface
// device-side error flags, and settle the lazily-read irregular-chain flag.
int sync_and_check(jg_engine* e) {
  HIPCHK(hipSetDevice(e->device));
  {
    const int rc = node_settle(e);
    if (rc) return rc;
  }
  HIPCHK(hipMemcpyAsync(e->h_status, e->d_status, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->stage_busy = false;
  const uint32_t irregular = e->h_status[1], deferred = e->h_status[2];
  {
    const int rc = status_check(e, e->h_status);
    if (rc) return rc;
  }
  if (e->flag_check_pending) {
    e->maybe_irregular = irregular != 0;  // sticky on the device: once seen, the slow kernel stays scheduled
    e->flag_check_pending = false;
  }
  // the ack-only kernel ran its in-kernel general path (escaped lag fields, acks above the head):
  // from here on k_dense_slow is scheduled behind it and takes those groups with dense lanes
  if (e->h_status[5]) e->maybe_irregular = true;
  // Assertion: irregular chains only come out of sparse steps, and every dense launch after
  // a sparse step has k_dense_slow behind it until the device flag is read back as 0 — so
  // while no slow kernel was ever scheduled the dense kernel cannot have deferred a group.
  if (!e->slow_scheduled_ever && deferred)
    return fail(JG_EDEVICE, "internal: irregular chain reached the fast-only dense path");
  return JG_OK;
}

// Fault records leave the device in atomic-append order; the drained order is (step, group), ties
// in queue order (= emission order: one lane owns a group for a step).  They are sorted on the
// device, on the stream that drains them: a stable LSD radix sort (rocPRIM) of (step << 32 | group)
// keys.  (On the host this was the largest single cost of a configs[4] drain: 0.9 ms per 160 k records.)
// Round 4: no library sort here either.  key = (step - floor) << bits(G) | group with `floor` below every step of the
// batch, value = the record's position in the queue; the bucket pass of jg_route.h (a bucket = the key's top 16 bits or
// fewer) + k_fault_order, which ranks a bucket's pairs by (key, queue position): equal keys keep their queue order.
__global__ void k_fault_split(const JgFaultRec* __restrict__ q, uint32_t n, uint32_t floor, uint32_t group_bits,
                              uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    keys[i] = ((uint64_t)(q[i].seq - floor) << group_bits) | q[i].group;
    vals[i] = i;
  }
}
__global__ __launch_bounds__(JG_BLOCK) void k_fault_order(JgRouteBuckets b, const uint64_t* __restrict__ key, const uint32_t* __restrict__ val,
                                                          uint32_t* __restrict__ val_out) {
  __shared__ uint64_t s_key[JG_ROUTE_SORT_CAP];
  __shared__ uint32_t s_val[JG_ROUTE_SORT_CAP];
  const uint32_t lo = b.off(blockIdx.x), n = b.off(blockIdx.x + 1) - lo;
  if (!n) return;
  const bool lds = n <= JG_ROUTE_SORT_CAP;
  if (lds) {
    for (uint32_t i = threadIdx.x; i < n; i += JG_BLOCK) s_key[i] = key[lo + i], s_val[i] = val[lo + i];
    __syncthreads();
  }
  for (uint32_t i = threadIdx.x; i < n; i += JG_BLOCK) {
    const uint64_t k = lds ? s_key[i] : key[lo + i];
    const uint32_t v = lds ? s_val[i] : val[lo + i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; j++) {
      const uint64_t kj = lds ? s_key[j] : key[lo + j];
      const uint32_t vj = lds ? s_val[j] : val[lo + j];
      rank += kj < k || (kj == k && vj < v);
    }
    val_out[lo + rank] = v;
  }
}
// the same for the rows of jg_chain_compact_resident: key = (group, position in the walk), value = id
__global__ void k_compact_split(const JgCompactRow* __restrict__ q, uint32_t n, uint64_t* __restrict__ keys,
                                uint64_t* __restrict__ vals) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    keys[i] = ((uint64_t)q[i].group << 8) | q[i].pad;
    vals[i] = q[i].id;
  }
}
__global__ void k_compact_join(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ vals, uint32_t n,
                               jg_compact_row* __restrict__ rows) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    rows[i] = jg_compact_row{(uint32_t)(keys[i] >> 8), 0, vals[i]};
}

__global__ void k_fault_join(const JgFaultRec* __restrict__ q, const uint32_t* __restrict__ order, uint32_t n,
                             jg_fault_row* __restrict__ rows, uint32_t* __restrict__ seqs) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const JgFaultRec r = q[order[i]];
    rows[i] = jg_fault_row{r.group, r.code};
    seqs[i] = r.seq;
  }
}

// ---- drains ------------------------------------------------------------------------------------
// Finished steps' output rows and the device-side queues travel to the host queues in two
// phases, both entirely on the device: A. one scan launch over the per-step tile sums (the host
// learns the totals), B. one gather per step straight into the pinned host queue + the copies of
// the fault / exceptional-row queues.  A synchronous drain runs them back to back on the engine's
// stream; jg_drain_prefetch runs them on `copy_stream` behind an event while the engine keeps
// stepping (phase B is issued by whichever API call first notices that the scan has finished).
inline void seg_add(std::vector<JgSeg>& v, uint32_t seq, size_t n) {
  if (!n) return;
  if (!v.empty() && v.back().seq == seq) v.back().n += n;
  else v.push_back(JgSeg{seq, n});
}

// phase A: job table + one scan launch (totals land in pinned host memory)
int drain_scan(jg_engine* e, const std::vector<StepRec>& recs, hipStream_t st) {
  const size_t nrec = recs.size();
  if (!nrec) return JG_OK;
  if (e->scan_cap < 2 * nrec) {
    if (e->h_jobs) HIPCHK(hipHostFree(e->h_jobs));
    if (e->h_totals) HIPCHK(hipHostFree(e->h_totals));
    e->h_jobs = nullptr, e->h_totals = nullptr;
    e->scan_cap = std::max<size_t>(4 * nrec, 64);
    HIPCHK(hipHostMalloc((void**)&e->h_jobs, e->scan_cap * sizeof(JgScanJob), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&e->h_totals, e->scan_cap * sizeof(uint64_t), hipHostMallocDefault));
  }
  for (size_t k = 0; k < nrec; k++) {
    const StepRec& r = recs[k];
    const uint32_t nb = (r.n + JG_SCAN_TILE - 1) / JG_SCAN_TILE;
    e->h_jobs[2 * k] = JgScanJob{r.d_bsum_m, r.d_bsum_m ? nb : 0u, 0};  // (a node step's record has fsm rows only)
    e->h_jobs[2 * k + 1] = JgScanJob{r.d_bsum_f, nb, 0};
  }
  hipLaunchKernelGGL(k_scan_block_sums, dim3(2 * nrec), dim3(JG_BLOCK), 0, st, (const JgScanJob*)e->h_jobs, e->h_totals);
  HIPCHK(hipGetLastError());
  return JG_OK;
}

// phase B: the gathers compact into a device staging buffer, one copy per queue moves the rows into the
// pinned host queues, then the two device queues of buffer set `set`
int drain_gather(jg_engine* e, jg_engine::DrainBatch& b, const std::vector<StepRec>& recs, hipStream_t st) {
  static const bool trace = std::getenv("JG_TRACE_DRAIN") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double g0 = now();
  double g1 = g0, g2 = g0, g3 = g0, g4 = g0;
  const uint64_t* totals = e->h_totals;
  b.add_m = b.add_f = 0;
  for (size_t k = 0; k < recs.size(); k++) b.add_m += totals[2 * k], b.add_f += totals[2 * k + 1];
  PinnedQueue<jg_msg_row>& qm = b.to_landing ? e->l_msgs : e->q_msgs;
  PinnedQueue<jg_fsm_row>& qf = b.to_landing ? e->l_fsm : e->q_fsm;
  b.at_m = qm.n, b.at_f = qf.n;
  // (a quarter of headroom when the queue has to grow: the row count wobbles from batch to batch and
  // re-pinning a 30 MB buffer costs milliseconds)
  if (qm.cap < b.at_m + b.add_m) HIPCHK(qm.reserve(b.at_m + b.add_m + b.add_m / 4));
  if (qf.cap < b.at_f + b.add_f) HIPCHK(qf.reserve(b.at_f + b.add_f + b.add_f / 4));
  // (the gather kernels used to write into the pinned host queue themselves: PCIe-bound for 90 us per
  // 16-tick batch of configs[4], during which the tick kernels beside them ran 2-4 x slower; now they
  // compact in HBM and a copy engine moves the rows.  JG_DRAIN_DIRECT=1: the old way, for an A/B)
  static const bool staged = std::getenv("JG_DRAIN_DIRECT") == nullptr;
  jg_msg_row* dst_m = qm.p + b.at_m;
  jg_fsm_row* dst_f = qf.p + b.at_f;
  if (staged) {
    auto grow = [](void*& p, size_t& cap, size_t bytes) -> hipError_t {
      if (bytes <= cap) return hipSuccess;
      if (p) (void)hipFree(p);
      cap = bytes + bytes / 4;
      return hipMalloc(&p, cap);
    };
    HIPCHK(grow(e->d_stage_m, e->stage_m_cap, b.add_m * sizeof(jg_msg_row)));
    HIPCHK(grow(e->d_stage_f, e->stage_f_cap, b.add_f * sizeof(jg_fsm_row)));
    dst_m = (jg_msg_row*)e->d_stage_m, dst_f = (jg_fsm_row*)e->d_stage_f;
  }
  uint64_t off_m = 0, off_f = 0;
  for (size_t k = 0; k < recs.size(); k++) {
    const StepRec& r = recs[k];
    const uint32_t nb = (r.n + JG_SCAN_TILE - 1) / JG_SCAN_TILE;
    if (totals[2 * k]) {
      hipLaunchKernelGGL(k_scan_gather<jg_msg_row>, dim3(nb), dim3(JG_BLOCK), 0, st, r.d_msg_cnt, r.n, r.d_bsum_m,
                         r.msg_per_row, r.d_msg, dst_m + off_m);
      off_m += totals[2 * k];
    }
    if (totals[2 * k + 1]) {
      hipLaunchKernelGGL(k_scan_gather<jg_fsm_row>, dim3(nb), dim3(JG_BLOCK), 0, st, r.d_fsm_cnt, r.n, r.d_bsum_f,
                         r.fsm_per_row, r.d_fsm, dst_f + off_f);
      off_f += totals[2 * k + 1];
    }
  }
  HIPCHK(hipGetLastError());
  if (staged) {
    if (b.add_m) HIPCHK(hipMemcpyAsync(qm.p + b.at_m, e->d_stage_m, b.add_m * sizeof(jg_msg_row), hipMemcpyDeviceToHost, st));
    if (b.add_f) HIPCHK(hipMemcpyAsync(qf.p + b.at_f, e->d_stage_f, b.add_f * sizeof(jg_fsm_row), hipMemcpyDeviceToHost, st));
  }
  g1 = now();
  uint32_t* d_cnt = e->d_status + (b.set ? 6 : 3);  // {fault_q_n, xq_n} of this buffer set
  if (b.nx) {
    HIPCHK(e->h_xq.reserve(b.nx));
    HIPCHK(hipMemcpyAsync(e->h_xq.p, e->xqb[b.set], (size_t)b.nx * sizeof(JgXqRec), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemsetAsync(d_cnt + 1, 0, sizeof(uint32_t), st));
  }
  if (b.nf) {
    if (b.nf > e->dev.fault_q_cap) return fail(JG_EDEVICE, "fault queue overflow");
    const size_t n = b.nf;
    if (e->fs_cap < n) {  // (grow-only; hipFree synchronises, so this happens a handful of times per engine)
      for (void* p : {(void*)e->fs_k0, (void*)e->fs_k1, (void*)e->fs_v0, (void*)e->fs_v1, (void*)e->fs_seq, (void*)e->fs_rows})
        if (p) HIPCHK(hipFree(p));
      e->fs_cap = std::max<size_t>(2 * n, 4096);
      HIPCHK(hipMalloc((void**)&e->fs_k0, e->fs_cap * 8));
      HIPCHK(hipMalloc((void**)&e->fs_k1, e->fs_cap * 8));
      HIPCHK(hipMalloc((void**)&e->fs_v0, e->fs_cap * 4));
      HIPCHK(hipMalloc((void**)&e->fs_v1, e->fs_cap * 4));
      HIPCHK(hipMalloc((void**)&e->fs_seq, e->fs_cap * 4));
      HIPCHK(hipMalloc((void**)&e->fs_rows, e->fs_cap * sizeof(jg_fault_row)));
    }
    // the key's layout and the buckets: (step - floor) << bits(G) | group, a bucket = its top 16 bits at most
    uint32_t gb = 1;
    while (gb < 32 && (e->cfg.n_groups - 1) >> gb) gb++;
    const uint32_t floor = e->fault_floor[b.set];
    const uint64_t k_max = ((uint64_t)(b.seq_hi - floor) << gb) | (e->cfg.n_groups - 1);
    uint32_t bits = 0;
    while (bits < 64 && (k_max >> bits)) bits++;
    JgRouteBuckets bk{};
    bk.shift = bits > 16 ? bits - 16 : 0;
    bk.n_buckets = (uint32_t)(k_max >> bk.shift) + 1;
    const uint32_t bk_tiles = (bk.n_buckets + JG_ROUTE_SCAN_TILE - 1) / JG_ROUTE_SCAN_TILE;
    const size_t bk_words = (size_t)bk_tiles * JG_ROUTE_SCAN_TILE + bk.n_buckets + bk_tiles + 1;
    if (e->fs_bk_words < bk_words) {
      if (e->fs_bk) HIPCHK(hipFree(e->fs_bk));
      e->fs_bk_words = bk_words;
      HIPCHK(hipMalloc((void**)&e->fs_bk, bk_words * 4));
    }
    bk.hist = e->fs_bk, bk.cur = bk.hist + (size_t)bk_tiles * JG_ROUTE_SCAN_TILE, bk.tile = bk.cur + bk.n_buckets;
    g2 = now();
    const uint32_t grid = grid_for(n, 1024);
    const uint32_t* d_n = e->d_status + (b.set ? 6 : 3);  // (the queue's own count word: the bucket pass reads it on the device)
    hipLaunchKernelGGL(k_fault_split, dim3(grid), dim3(JG_BLOCK), 0, st, (const JgFaultRec*)e->fq[b.set], (uint32_t)n, floor, gb,
                       e->fs_k0, e->fs_v0);
    hipLaunchKernelGGL(k_route_clear, dim3(64), dim3(JG_BLOCK), 0, st, bk.hist, bk_tiles * JG_ROUTE_SCAN_TILE + bk.n_buckets, bk.tile, bk_tiles + 1);
    hipLaunchKernelGGL(k_route_hist, dim3(grid, 1), dim3(JG_BLOCK), 0, st, d_n, (uint32_t)e->fs_cap, (const uint64_t*)e->fs_k0, bk);
    hipLaunchKernelGGL(k_route_scan, dim3(bk_tiles), dim3(JG_BLOCK), 0, st, bk);
    hipLaunchKernelGGL(k_route_scan_tiles, dim3(1), dim3(JG_BLOCK), 0, st, bk);
    hipLaunchKernelGGL(k_route_scatter, dim3(grid, 1), dim3(JG_BLOCK), 0, st, d_n, (uint32_t)e->fs_cap, (const uint64_t*)e->fs_k0,
                       (const uint32_t*)e->fs_v0, bk, e->fs_k1, e->fs_v1);
    hipLaunchKernelGGL(k_fault_order, dim3(bk.n_buckets), dim3(JG_BLOCK), 0, st, bk, (const uint64_t*)e->fs_k1, (const uint32_t*)e->fs_v1, e->fs_v0);
    hipLaunchKernelGGL(k_fault_join, dim3(grid), dim3(JG_BLOCK), 0, st, (const JgFaultRec*)e->fq[b.set], (const uint32_t*)e->fs_v0, (uint32_t)n,
                       e->fs_rows, e->fs_seq);
    HIPCHK(hipGetLastError());
    g3 = now();
    HIPCHK(e->h_faults.reserve(2 * n));  // (headroom: the count wobbles from batch to batch, pinned reallocation is slow)
    HIPCHK(e->h_fault_seq.reserve(2 * n));
    HIPCHK(hipMemcpyAsync(e->h_faults.p, e->fs_rows, n * sizeof(jg_fault_row), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(e->h_fault_seq.p, e->fs_seq, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(uint32_t), st));
    g4 = now();
  }
  if (trace)
    std::fprintf(stderr, "[jg drain phase B] reserve + %zu gather launches %.3f ms, sort scratch %.3f ms, sort launches %.3f ms, copies %.3f ms\n",
                 recs.size(), g1 - g0, g2 - g1, g3 - g2, g4 - g3);
  return JG_OK;
}

// host tail, once everything of the batch has landed: the rows count as queued, exceptional rows
// are merged in by step sequence number, fault records are put in (step, group) order
int drain_finish(jg_engine* e, jg_engine::DrainBatch& b, std::vector<StepRec>& recs, Arena& arena) {
  const size_t nrec = recs.size();
  const uint64_t* totals = e->h_totals;
  PinnedQueue<jg_msg_row>& qm = b.to_landing ? e->l_msgs : e->q_msgs;
  PinnedQueue<jg_fsm_row>& qf = b.to_landing ? e->l_fsm : e->q_fsm;
  qm.n = b.at_m + b.add_m;
  qf.n = b.at_f + b.add_f;
  if (b.to_landing) e->landed_m = e->landed_f = true;
  if (e->track_segs)
    for (size_t k = 0; k < nrec; k++) {
      if (!b.nx) seg_add(e->seg_m, recs[k].seq, totals[2 * k]);  // (else: in the merge below)
      seg_add(e->seg_f, recs[k].seq, totals[2 * k + 1]);
    }
  if (b.nx) {
    // Merge by step sequence number: the rows of sparse step k (already in the queue, step
    // order) carry rec.seq; exceptional rows carry the seq of their dense step.  Rare path.
    e->xq_tmp.assign(e->h_xq.p, e->h_xq.p + b.nx);
    std::vector<JgXqRec>& xr = e->xq_tmp;
    std::sort(xr.begin(), xr.end(), [](const JgXqRec& x, const JgXqRec& y) {
      if (x.seq != y.seq) return x.seq < y.seq;
      if (x.row.group != y.row.group) return x.row.group < y.row.group;
      return x.k < y.k;
    });
    const size_t old_n = b.at_m, nx = b.nx;  // rows queued before this batch
    std::vector<jg_msg_row> merged;
    merged.reserve(b.add_m + nx);
    size_t xi = 0, off = old_n;
    for (size_t k = 0; k < nrec; k++) {
      while (xi < nx && xr[xi].seq < recs[k].seq) {
        if (e->track_segs) seg_add(e->seg_m, xr[xi].seq, 1);
        merged.push_back(xr[xi++].row);
      }
      const size_t cnt = totals[2 * k];
      merged.insert(merged.end(), qm.p + off, qm.p + off + cnt);
      if (e->track_segs) seg_add(e->seg_m, recs[k].seq, cnt);
      off += cnt;
    }
    while (xi < nx) {
      if (e->track_segs) seg_add(e->seg_m, xr[xi].seq, 1);
      merged.push_back(xr[xi++].row);
    }
    HIPCHK(qm.reserve(old_n + merged.size()));
    if (!merged.empty()) std::memcpy(qm.p + old_n, merged.data(), merged.size() * sizeof(jg_msg_row));
    qm.n = old_n + merged.size();
  }
  if (nrec) {
    recs.clear();
    arena.reset();
  }
  if (b.nf) {  // (sorted on the device)
    e->q_faults.insert(e->q_faults.end(), e->h_faults.p, e->h_faults.p + b.nf);
    if (e->track_segs) e->q_fault_seq.insert(e->q_fault_seq.end(), e->h_fault_seq.p, e->h_fault_seq.p + b.nf);
  }
  return JG_OK;
}

// ---- the batch in transfer (jg_drain_prefetch) ---------------------------------------------------
int inflight_phase_b(jg_engine* e) {
  static const bool trace = std::getenv("JG_TRACE_DRAIN") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  jg_engine::DrainBatch& b = e->inflight;
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipEventSynchronize(e->ev_scan));
  const double t1 = now();
  b.nf = e->h_cnt[0], b.nx = e->h_cnt[1];
  if (b.nx > e->dev.xq_cap) return fail(JG_ECAPACITY, "exceptional-message queue overflow: drain the messages more often");
  int rc = drain_gather(e, b, b.recs, e->copy_stream);
  if (rc) return rc;
  HIPCHK(hipEventRecord(e->ev_done, e->copy_stream));
  const double t2 = now();
  HIPCHK(hipEventSynchronize(e->ev_done));
  if (trace)
    std::fprintf(stderr, "[jg drain thread] %zu steps: waited %.3f ms for the scan, issued phase B in %.3f ms, landed after %.3f ms\n",
                 b.recs.size(), t1 - t0, t2 - t1, now() - t2);
  return JG_OK;
}
void drain_thread_main(jg_engine* e) {
  jg_engine::DrainThread& t = *e->drain_thread;
  std::unique_lock<std::mutex> lk(t.m);
  for (;;) {
    t.cv.wait(lk, [&] { return t.state == 1 || t.quit; });
    if (t.quit) return;
    lk.unlock();
    g_err.clear();
    const int rc = inflight_phase_b(e);
    lk.lock();
    t.rc = rc;
    t.err = rc ? g_err : std::string();
    t.state = 2;
    t.cv.notify_all();
  }
}
// a landed batch joins the queue the consumer drains: a pointer swap if the consumer has taken
// everything before it, an append behind what it has not taken yet otherwise
template <typename Row>
int handover(PinnedQueue<Row>& q, PinnedQueue<Row>& l, bool& flag) {
  if (!flag) return JG_OK;
  flag = false;
  if (q.n == 0 && !q.viewed) {
    std::swap(q.p, l.p);
    std::swap(q.cap, l.cap);
    q.n = l.n;
    l.n = 0;
    return JG_OK;
  }
  HIPCHK(q.reserve(q.n + l.n));
  if (l.n) std::memcpy(q.p + q.n, l.p, l.n * sizeof(Row));
  q.n += l.n;
  l.n = 0;
  return JG_OK;
}

bool inflight_landed(jg_engine* e) {
  if (!e->inflight.phase) return true;
  std::lock_guard<std::mutex> lk(e->drain_thread->m);
  return e->drain_thread->state == 2;
}
// wait for the batch in transfer (never for the engine's own stream) and queue its rows
int inflight_finish(jg_engine* e) {
  jg_engine::DrainBatch& b = e->inflight;
  if (!b.phase) return JG_OK;
  jg_engine::DrainThread& t = *e->drain_thread;
  {
    std::unique_lock<std::mutex> lk(t.m);
    t.cv.wait(lk, [&] { return t.state == 2; });
    t.state = 0;
  }
  b.phase = 0;
  if (t.rc) return fail(t.rc, "drain thread: " + t.err);
  {  // what sync_and_check does with the status block, on the snapshot taken at the prefetch point
    const uint32_t* st = e->h_cnt + 2;
    const int rc = status_check(e, st);
    if (rc) return rc;
    if (e->flag_check_pending && b.irr_gen == e->irr_gen) {  // no step since could have left an irregular chain
      e->maybe_irregular = st[1] != 0;
      e->flag_check_pending = false;
    }
    if (st[5]) e->maybe_irregular = true;
  }
  return drain_finish(e, b, b.recs, e->arenas[b.arena]);
}

// `wait`: jg_drain_flush (block until the previous batch has landed); jg_drain_prefetch never
// blocks: while a batch is still in transfer it starts nothing (the next call takes more steps)
int drain_prefetch(jg_engine* e, bool wait) {
  HIPCHK(hipSetDevice(e->device));
  {
    const int rc = node_settle(e);
    if (rc) return rc;
  }
  e->pipelined = true;
  if (!wait && !inflight_landed(e)) return JG_OK;
  int rc = inflight_finish(e);
  if (rc) return rc;
  if ((rc = handover(e->q_msgs, e->l_msgs, e->landed_m))) return rc;  // the landing queues must be free
  if ((rc = handover(e->q_fsm, e->l_fsm, e->landed_f))) return rc;
  if (!e->stepped) return JG_OK;
  if (!e->drain_thread) {  // first use: the second stream (a second hardware queue: not before it is needed), its events, the thread
    HIPCHK(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&e->ev_steps, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->ev_scan, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming));
    e->drain_thread = new jg_engine::DrainThread();
    e->drain_thread->th = std::thread(drain_thread_main, e);
  }
  jg_engine::DrainBatch& b = e->inflight;
  b.to_landing = true;
  b.recs.swap(e->recs);
  b.arena = e->cur_arena;
  e->cur_arena ^= 1;
  // kernels launched from here on append to the other fault / exceptional-row queues
  b.set = e->cur_set;
  b.seq_hi = e->seq;
  e->cur_set ^= 1;
  e->fault_floor[e->cur_set] = e->seq;  // (what the other set collects from here on is later than this point)
  e->dev = dev_for_set(e, e->cur_set);
  e->d_dev = e->d_dev2[e->cur_set];  // (both device copies were written up front: nothing to upload here)
  HIPCHK(hipEventRecord(e->ev_steps, e->stream));
  HIPCHK(hipStreamWaitEvent(e->copy_stream, e->ev_steps, 0));
  rc = drain_scan(e, b.recs, e->copy_stream);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(e->h_cnt, e->d_status + (b.set ? 6 : 3), 2 * sizeof(uint32_t), hipMemcpyDeviceToHost,
                        e->copy_stream));
  // the status block as of the prefetch point: a pipelined engine never reaches sync_and_check through its
  // drains, so this copy is where device-side error flags surface and the irregular-chain flag settles
  HIPCHK(hipMemcpyAsync(e->h_cnt + 2, e->d_status, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, e->copy_stream));
  b.irr_gen = e->irr_gen;
  HIPCHK(hipEventRecord(e->ev_scan, e->copy_stream));
  b.phase = 1;
  {
    std::lock_guard<std::mutex> lk(e->drain_thread->m);
    e->drain_thread->state = 1;
  }
  e->drain_thread->cv.notify_all();
  return JG_OK;
}

// Synchronous drain: everything stepped so far (unless the engine is pipelined: then exactly the
// batches up to the latest prefetch point, without synchronising later steps).  `release_mask`:
// bit 0 / bit 1 = the caller is a drain of the message / fsm queue, which ends the life of that
// queue's outstanding view.
int collect(jg_engine* e, int release_mask) {
  static const bool trace = std::getenv("JG_TRACE_DRAIN") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  // (first: the batch in transfer lands BEHIND the rows a view may still cover; only then may
  // the queue be compacted)
  if (e->pipelined && !inflight_landed(e)) return JG_OK;  // nothing new yet; the queues are the drain thread's
  int rc = inflight_finish(e);
  if (rc) return rc;
  if (release_mask & 1) {
    e->q_msgs.release_view();  // the caller is done with that queue's last view
    if ((rc = handover(e->q_msgs, e->l_msgs, e->landed_m))) return rc;
  }
  if (release_mask & 2) {
    e->q_fsm.release_view();
    if ((rc = handover(e->q_fsm, e->l_fsm, e->landed_f))) return rc;
  }
  if (e->pipelined) {
    if (trace) std::fprintf(stderr, "[jg drain] pipelined: %.3f ms on the host (%zu msg rows queued)\n", now() - t0, e->q_msgs.n);
    return JG_OK;
  }
  rc = sync_and_check(e);
  if (rc) return rc;
  const double t1 = now();
  jg_engine::DrainBatch b;
  b.set = e->cur_set;
  b.seq_hi = e->seq;
  b.nf = e->h_status[b.set ? 6 : 3], b.nx = e->h_status[b.set ? 7 : 4];
  const size_t nrec = e->recs.size();
  rc = drain_scan(e, e->recs, e->stream);
  if (rc) return rc;
  if (nrec) HIPCHK(hipStreamSynchronize(e->stream));
  const double t2 = now();
  rc = drain_gather(e, b, e->recs, e->stream);
  if (rc) return rc;
  if (nrec || b.nf || b.nx) HIPCHK(hipStreamSynchronize(e->stream));
  const double t3 = now();
  rc = drain_finish(e, b, e->recs, e->arenas[e->cur_arena]);
  e->fault_floor[b.set] = b.seq_hi;  // (the set is empty again: whatever it collects next is later than this batch)
  if (trace && nrec)
    std::fprintf(stderr, "[jg drain] %zu steps: sync %.3f ms, scan %.3f ms, gather+copy %.3f ms, host tail %.3f ms (%zu msg rows, %u faults)\n",
                 nrec, t1 - t0, t2 - t1, t3 - t2, now() - t3, e->q_msgs.n, b.nf);
  return rc;
}

template <typename Row>
int drain(jg_engine* e, PinnedQueue<Row>& q, int mask, Row* out, size_t cap, size_t* n) {
  if (!e || !n) return fail(JG_EINVAL, "null argument");
  if (e->pipelined && !inflight_landed(e)) {  // a batch is in transfer: nothing new to deliver yet
    *n = 0;
    return JG_OK;
  }
  int rc = collect(e, mask);
  if (rc) return rc;
  *n = q.n;
  if (!out) return JG_OK;
  if (cap < q.n) return fail(JG_ECAPACITY, "output buffer too small");
  if (q.n) std::memcpy(out, q.p, q.n * sizeof(Row));
  q.n = 0;
  return JG_OK;
}
template <typename Row>
int drain_view(jg_engine* e, PinnedQueue<Row>& q, int mask, const Row** rows, size_t* n) {
  if (!e || !rows || !n) return fail(JG_EINVAL, "null argument");
  if (e->pipelined && !inflight_landed(e)) {  // a batch is in transfer: nothing new (an earlier view stays valid)
    *rows = q.p;
    *n = 0;
    return JG_OK;
  }
  int rc = collect(e, mask);
  if (rc) return rc;
  *rows = q.p;
  *n = q.n;
  q.viewed = q.n;  // consumed: the rows stay where they are until the next drain of this queue
  return JG_OK;
}

// The optional columns (from, term, aux, flag) of rows [at, at + n) of the pending batch: copied where the
// caller provided one, zero-filled LAZILY otherwise - a column nobody provides between two steps is never
// written (jg_step_node then does not upload it either); the first submit that does provide it zero-fills
// the rows queued before it, and from then on absent columns are zero-filled as they come.
template <typename T>
hipError_t pending_col(PinnedVec<T>& v, bool& has, size_t at, size_t n, const T* src) {
  hipError_t e = v.reserve(at + n);
  if (e != hipSuccess) return e;
  if (src) {
    if (!has && at) std::memset(v.p, 0, at * sizeof(T));
    has = true;
    std::memcpy(v.p + at, src, n * sizeof(T));
  } else if (has) {
    std::memset(v.p + at, 0, n * sizeof(T));
  }
  v.n = at + n;
  return hipSuccess;
}
int pending_optional(jg_engine* e, size_t at, size_t n, const uint32_t* from, const uint64_t* term, const uint64_t* aux,
                     const uint8_t* flag) {
  HIPCHK(pending_col(e->p_from, e->p_has_from, at, n, from));
  HIPCHK(pending_col(e->p_term, e->p_has_term, at, n, term));
  HIPCHK(pending_col(e->p_aux, e->p_has_aux, at, n, aux));
  HIPCHK(pending_col(e->p_flag, e->p_has_flag, at, n, flag));
  return JG_OK;
}
// every optional column materialised (the general step gathers all seven)
void pending_materialise(jg_engine* e) {
  const size_t n = e->p_kind.size();
  if (!e->p_has_from && n) std::memset(e->p_from.p, 0, n * 4);
  if (!e->p_has_term && n) std::memset(e->p_term.p, 0, n * 8);
  if (!e->p_has_aux && n) std::memset(e->p_aux.p, 0, n * 8);
  if (!e->p_has_flag && n) std::memset(e->p_flag.p, 0, n);
  e->p_has_from = e->p_has_term = e->p_has_aux = e->p_has_flag = true;
}

// jg_submit's argument checks (shared with the multi-device router)
int validate_batch(uint32_t n_groups, const jg_cmd_batch* b, uint32_t* kinds_seen = nullptr) {
  if (b->n && (!b->kind || !b->group)) return fail(JG_EINVAL, "kind/group columns are required");
  if (b->n_blocks && (!b->blk_id || !b->blk_next)) return fail(JG_EINVAL, "block side arrays are required");
  // (two branch-free passes the compiler vectorises - a batch is millions of rows per tick through
  // jg_step_node - and the per-row checks only where an AppendEntries row is present)
  uint32_t bad_group = 0, bad_kind = 0, has_ae = 0, has_hb = 0;
  for (size_t i = 0; i < b->n; i++) bad_group |= b->group[i] >= n_groups;
  for (size_t i = 0; i < b->n; i++) {
    bad_kind |= b->kind[i] >= JG_CMD__COUNT;
    has_ae |= b->kind[i] == JG_CMD_APPEND_ENTRIES;
    has_hb |= b->kind[i] == JG_CMD_HEARTBEAT;
  }
  if (bad_group) return fail(JG_EINVAL, "group out of range");
  if (bad_kind) return fail(JG_EINVAL, "unknown command kind");
  if (kinds_seen) *kinds_seen = (has_ae ? 1u : 0u) | (has_hb ? 2u : 0u);
  if (has_ae) {
    if (!b->id || !b->aux) return fail(JG_EINVAL, "AppendEntries needs id/aux columns");
    for (size_t i = 0; i < b->n; i++)
      if (b->kind[i] == JG_CMD_APPEND_ENTRIES && (b->aux[i] > b->n_blocks || b->id[i] > b->n_blocks - b->aux[i]))  // (overflow-safe)
        return fail(JG_EINVAL, "block side-array range out of bounds");
  }
  return JG_OK;
}

// element width of a jg_read_state column
size_t field_width(int field) {
  switch (field) {
    case JG_FIELD_TERM: case JG_FIELD_COMMIT: case JG_FIELD_HEAD: case JG_FIELD_ID_GEN: case JG_FIELD_MATCH:
    case JG_FIELD_ELECTION_TIME: case JG_FIELD_HEARTBEAT_TIME: return 8;
    case JG_FIELD_VOTED_FOR: case JG_FIELD_LEADER_ID: case JG_FIELD_ELECTION_TIMEOUT: case JG_FIELD_QUEUED_REQS: return 4;
    default: return 1;
  }
}

// Stable LSD radix sort of row indices by group id: per-group stream order = row order.
void sort_rows_by_group(const uint32_t* group, size_t n, uint32_t n_groups, std::vector<uint32_t>& order) {
  order.resize(n);
  std::iota(order.begin(), order.end(), 0u);
  bool sorted = true;
  for (size_t i = 1; i < n && sorted; i++) sorted = group[i - 1] <= group[i];
  if (sorted) return;
  std::vector<uint32_t> tmp(n);
  uint32_t bits = 1;
  while (bits < 32 && (n_groups - 1) >> bits) bits++;
  const uint32_t RADIX = 11, BUCKETS = 1u << RADIX;
  std::vector<uint32_t> count(BUCKETS);
  for (uint32_t shift = 0; shift < bits; shift += RADIX) {
    std::fill(count.begin(), count.end(), 0u);
    for (size_t i = 0; i < n; i++) count[(group[order[i]] >> shift) & (BUCKETS - 1)]++;
    uint32_t sum = 0;
    for (uint32_t b = 0; b < BUCKETS; b++) {
      uint32_t c = count[b];
      count[b] = sum;
      sum += c;
    }
    for (size_t i = 0; i < n; i++) tmp[count[(group[order[i]] >> shift) & (BUCKETS - 1)]++] = order[i];
    order.swap(tmp);
  }
}

// Launch k_apply_rows over device-resident, group-sorted command columns.
// everything of a k_apply_rows step but the launch: output regions, the step record, the host-side bookkeeping
int prepare_rows(jg_engine* e, uint32_t n, const uint32_t* group, const uint8_t* kind, const uint32_t* from,
                 const uint64_t* term, const uint64_t* id, const uint64_t* aux, const uint8_t* flag,
                 const uint64_t* blk_id, const uint64_t* blk_next, uint64_t n_blocks, uint64_t now_ms, JgRowsArgs* out,
                 uint32_t msg_per_row = 0) {
  // msg_per_row != 0: the caller knows the kinds of its rows and with them a tighter bound on the message rows one
  // command can emit (the slots of a command lie msg_per_row rows apart: what reads them back reads that much less)
  StepRec rec;
  rec.n = n;
  rec.msg_per_row = msg_per_row ? msg_per_row : msg_bound(e->cfg.n_replicas);
  rec.fsm_per_row = fsm_bound();
  if ((uint64_t)n * rec.msg_per_row > 0xffffffffull) return fail(JG_EINVAL, "batch too large: split it");
  HIPCHK(e->arenas[e->cur_arena].alloc((size_t)n * 4, (void**)&rec.d_msg_cnt));
  HIPCHK(e->arenas[e->cur_arena].alloc((size_t)n * 4, (void**)&rec.d_fsm_cnt));
  HIPCHK(e->arenas[e->cur_arena].alloc((size_t)n * rec.msg_per_row * sizeof(jg_msg_row), (void**)&rec.d_msg));
  HIPCHK(e->arenas[e->cur_arena].alloc((size_t)n * rec.fsm_per_row * sizeof(jg_fsm_row), (void**)&rec.d_fsm));
  const uint32_t n_tiles = (n + JG_SCAN_TILE - 1) / JG_SCAN_TILE;
  HIPCHK(e->arenas[e->cur_arena].alloc((size_t)n_tiles * 8, (void**)&rec.d_bsum_m));
  HIPCHK(e->arenas[e->cur_arena].alloc((size_t)n_tiles * 8, (void**)&rec.d_bsum_f));
  JgRowsArgs a;
  a.n = n;
  a.group = group;
  a.kind = kind;
  a.from = from;
  a.term = term;
  a.id = id;
  a.aux = aux;
  a.flag = flag;
  a.blk_id = blk_id;
  a.blk_next = blk_next;
  a.n_blocks = n_blocks;
  a.msg_per_row = rec.msg_per_row;
  a.fsm_per_row = rec.fsm_per_row;
  a.msg_out = rec.d_msg;
  a.fsm_out = rec.d_fsm;
  a.msg_cnt = rec.d_msg_cnt;
  a.fsm_cnt = rec.d_fsm_cnt;
  a.bsum_m = rec.d_bsum_m;
  a.bsum_f = rec.d_bsum_f;
  a.err = e->d_err;
  a.now = now_ms;
  a.seq = e->seq;
  rec.seq = e->seq;
  e->n_launch += 1;
  e->recs.push_back(rec);
  e->n_cmds += n;
  e->maybe_irregular = true;  // until the device flag says otherwise (sync_and_check)
  e->flag_check_pending = true;
  e->irr_gen++;
  *out = a;
  return JG_OK;
}
int launch_rows(jg_engine* e, uint32_t n, const uint32_t* group, const uint8_t* kind, const uint32_t* from,
                const uint64_t* term, const uint64_t* id, const uint64_t* aux, const uint8_t* flag,
                const uint64_t* blk_id, const uint64_t* blk_next, uint64_t n_blocks, uint64_t now_ms) {
  JgRowsArgs a;
  const int rc = prepare_rows(e, n, group, kind, from, term, id, aux, flag, blk_id, blk_next, n_blocks, now_ms, &a);
  if (rc) return rc;
  // JG_APPLY_RUNS=1 (test hook): the run-per-lane body the cluster transport's batches take (jg_apply_runs_body) for
  // every batch - the fuzz and parity suites then hold it to the oracle with runs of every length across its tiles
  // (JG_APPLY_RUNS=small: with the 256-row tiles small batches take)
  static const char* runs_env = std::getenv("JG_APPLY_RUNS");
  static const bool runs = runs_env != nullptr, runs_small = runs_env && std::string(runs_env) == "small";
  if (runs_small)
    hipLaunchKernelGGL(k_apply_runs_small, dim3(std::min<uint32_t>((n + JG_RUN_TILE_SMALL - 1) / JG_RUN_TILE_SMALL, e->count_slots)), dim3(JG_BLOCK), 0,
                       e->stream, e->dev, a);
  else if (runs)
    hipLaunchKernelGGL(k_apply_runs, dim3(std::min<uint32_t>((n + JG_RUN_TILE - 1) / JG_RUN_TILE, e->count_slots)), dim3(JG_BLOCK), 0, e->stream,
                       e->dev, a);
  else
    hipLaunchKernelGGL(k_apply_rows, dim3(grid_for(n, e->count_slots)), dim3(JG_BLOCK), 0, e->stream, e->dev, a);
  HIPCHK(hipGetLastError());
  return JG_OK;
}

template <int R>
void launch_calib(jg_engine* e, const uint64_t* rot, const uint64_t* a8, uint64_t* b8, const uint32_t* c4) {
  hipLaunchKernelGGL(k_stream_calib<R>, dim3(e->dense_grid), dim3(JG_BLOCK), 0, e->stream, rot, a8, b8, c4,
                     e->cfg.n_groups);
}

}  // namespace

#include "jg_multi.h"

extern "C" {

const char* jg_last_error(void) { return g_err.c_str(); }
uint32_t jg_abi_version(void) { return JG_ABI_VERSION; }

int jg_engine_create(const jg_config* cfg, jg_engine** out) {
  if (!cfg || !out) return fail(JG_EINVAL, "null argument");
  if (cfg->abi_version != JG_ABI_VERSION) return fail(JG_EINVAL, "abi version mismatch");
  if (cfg->n_replicas < 1 || cfg->n_replicas > JG_MAX_REPLICAS) return fail(JG_EINVAL, "n_replicas out of range");
  for (uint32_t r = 0; r < cfg->n_replicas; r++) {
    if (cfg->node_ids[r] == 0) return fail(JG_EINVAL, "id cannot be 0");  // config.rs:64-66
    for (uint32_t q = 0; q < r; q++)
      if (cfg->node_ids[q] == cfg->node_ids[r]) return fail(JG_EINVAL, "duplicate node id");
  }
  if (cfg->heartbeat_timeout_ms < 5) return fail(JG_EINVAL, "heartbeat timeout is too low");  // config.rs:70-72
  // thread_rng().gen_range(min..max) panics on an empty range (follower.rs:105)
  if (cfg->election_timeout_max_ms <= cfg->election_timeout_min_ms) return fail(JG_EINVAL, "election timeout range is empty");
  if (cfg->n_groups == 0) return fail(JG_EINVAL, "n_groups cannot be 0");
  if (cfg->n_devices > JG_MAX_DEVICES) return fail(JG_EINVAL, "n_devices out of range");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  for (uint32_t d = 0; d < cfg->n_devices; d++)
    if (cfg->device_ids[d] < 0 || cfg->device_ids[d] >= ndev) return fail(JG_EDEVICE, "no such HIP device (no CPU fallback)");
  if (cfg->n_devices > 1) return router_create(cfg, out);  // one shard per listed device, one handle
  const int device_id = cfg->n_devices == 1 ? cfg->device_ids[0] : cfg->device_id;
  if (device_id < 0 || device_id >= ndev) return fail(JG_EDEVICE, "no such HIP device (no CPU fallback)");
  HIPCHK(hipSetDevice(device_id));

  jg_engine* e = new jg_engine();
  e->cfg = *cfg;
  e->device = device_id;
  int rc = JG_OK;
  auto bail = [&](int code) {
    jg_engine_destroy(e);
    return code;
  };
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(JG_EDEVICE, "hipStreamCreate failed"));
  if (hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_stage, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_order, hipEventDisableTiming) != hipSuccess)
    return bail(fail(JG_EDEVICE, "hipEventCreate failed"));

  const size_t G = cfg->n_groups, R = cfg->n_replicas;
  JgDev& d = e->dev;
  std::memset(&d, 0, sizeof d);
  d.G = (uint32_t)G;
  d.R = (uint32_t)R;
  for (uint32_t r = 0; r < JG_MAX_REPLICAS; r++) d.node_ids[r] = r < R ? cfg->node_ids[r] : 0;
  d.hb_timeout = cfg->heartbeat_timeout_ms;
  d.el_min = cfg->election_timeout_min_ms;
  d.el_max = cfg->election_timeout_max_ms;
  d.cfg_flags = cfg->flags;
  d.seed = cfg->seed;
  d.group_base = cfg->group_base;
  const char* env_grid = std::getenv("JG_DENSE_GRID");
  uint32_t cap = env_grid ? (uint32_t)std::atoi(env_grid) : 8192u;  // measured best (profiles/README.md)
  if (cap < 1) cap = 1;
  e->dense_grid = grid_for(G, cap);
  e->count_slots = std::max<uint32_t>(e->dense_grid, 4096);
#define A(ptr, n) \
  if ((rc = dev_alloc(e, &ptr, (n))) != JG_OK) return bail(rc)
  A(d.term, G);
  A(d.commit, G);
  A(d.head, G);
  A(d.id_gen, G);
  A(d.run_hi, G);
  A(d.mlag, G);
  A(d.match_wide, G * R);
  A(d.heartbeat_time, G);
  A(d.win_lo, G * JG_CHAIN_WINDOW);
  A(d.win_hi, G * JG_CHAIN_WINDOW);
  A(d.win_next, G * JG_CHAIN_WINDOW);
  A(d.flags, G);
  A(d.cold.t, G);
  A(d.cold.v, G);
  A(d.fvote_id, G * JG_FOREIGN_VOTERS);
  A(d.blk_decisions, e->count_slots);
  d.fault_q_cap = (uint32_t)std::max<size_t>(2 * G, 1024);
  A(e->fq[0], d.fault_q_cap);
  A(e->fq[1], d.fault_q_cap);
  d.fault_q = e->fq[0];
  A(e->d_status, 8);
  e->d_err = e->d_status;
  d.err = e->d_status;
  d.irregular_seen = e->d_status + 1;
  d.deferred_seen = e->d_status + 2;
  d.fault_q_n = e->d_status + 3;
  d.xq_n = e->d_status + 4;
  d.cold_seen = e->d_status + 5;
  if (hipHostMalloc((void**)&e->h_status, 8 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&e->h_cnt, 10 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess)
    return bail(fail(JG_EDEVICE, "hipHostMalloc failed"));
  {  // deferred lists: shard = workgroup & (JG_SHARDS-1); generous per-shard capacity, bounds-checked
    const size_t n_wg = (G + JG_BLOCK - 1) / JG_BLOCK;
    d.slow_cap = (uint32_t)((3 * ((n_wg + JG_SHARDS - 1) / JG_SHARDS) + 2) * JG_BLOCK);
    // (>= the groups of one shard of the deferral bitmap: ceil(ceil(G/64)/JG_SHARDS) * 64)
    const size_t shard_groups = ((((G + 63) / 64) + JG_SHARDS - 1) / JG_SHARDS) * 64;
    if (d.slow_cap < shard_groups) d.slow_cap = (uint32_t)shard_groups;
  }
  A(d.slow_list, (size_t)JG_SHARDS * d.slow_cap);
  A(d.slow_cnt, JG_SHARDS);
  A(d.defer_bits, (G + 63) / 64);
  A(d.fdefer_bits, 2 * ((G + 63) / 64));
  A(e->d_ones, 2);
  A(e->d_dev2[0], 1);
  A(e->d_dev2[1], 1);
#undef A
  if (hipMemsetAsync(e->d_ones, 0xff, 16, e->stream) != hipSuccess) return bail(fail(JG_EDEVICE, "hipMemsetAsync failed"));
  if ((rc = push_dev_copy(e)) != JG_OK) return bail(rc);
  hipLaunchKernelGGL(k_init_groups, dim3(grid_for(G, 2048)), dim3(JG_BLOCK), 0, e->stream, e->dev,
                     (const uint8_t*)nullptr);
  {  // one launch of the general-path kernel over its (empty) lists: it is the only kernel with
     // scratch memory, which the runtime sets up at a kernel's first launch (~150 us) — here, not
     // inside somebody's first node tick
    JgLeaderNode none{};
    hipLaunchKernelGGL(k_dense_slow<true>, dim3(JG_SHARDS), dim3(JG_BLOCK), 0, e->stream, e->dev, (const uint64_t*)nullptr,
                       0u, (size_t)0, 0u, none);
  }
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
    return bail(fail(JG_EDEVICE, "k_init_groups failed: is this a gfx950 device? (no CPU fallback)"));
  *out = e;
  return JG_OK;
}

void jg_engine_destroy(jg_engine* e) {
  if (!e) return;
  if (e->parent) return;  // a shard handle: owned by its parent
  if (e->router) {
    router_destroy(e);
    delete e;
    return;
  }
  (void)hipSetDevice(e->device);
  if (e->drain_thread) {
    jg_engine::DrainThread& t = *e->drain_thread;
    {
      std::unique_lock<std::mutex> lk(t.m);
      t.cv.wait(lk, [&] { return t.state != 1; });  // a batch in transfer lands first
      t.quit = true;
    }
    t.cv.notify_all();
    if (t.th.joinable()) t.th.join();
    delete e->drain_thread;
    e->drain_thread = nullptr;
  }
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
  e->arenas[0].destroy();
  e->arenas[1].destroy();
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->d_acks_staging) (void)hipFree(e->d_acks_staging);
  if (e->stage) (void)hipHostFree(e->stage);
  if (e->h_status) (void)hipHostFree(e->h_status);
  if (e->h_cnt) (void)hipHostFree(e->h_cnt);
  e->h_faults.destroy();
  e->h_fault_seq.destroy();
  e->h_xq.destroy();
  for (void* p : {(void*)e->fs_k0, (void*)e->fs_k1, (void*)e->fs_v0, (void*)e->fs_v1, (void*)e->fs_seq, (void*)e->fs_rows, e->fs_tmp, e->d_stage_m, e->d_stage_f})
    if (p) (void)hipFree(p);
  for (hipEvent_t ev : e->kt_ev) (void)hipEventDestroy(ev);
  if (e->ev_steps) (void)hipEventDestroy(e->ev_steps);
  if (e->ev_scan) (void)hipEventDestroy(e->ev_scan);
  if (e->ev_done) (void)hipEventDestroy(e->ev_done);
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  if (e->h_jobs) (void)hipHostFree(e->h_jobs);
  if (e->h_totals) (void)hipHostFree(e->h_totals);
  e->p_kind.destroy(), e->p_flag.destroy(), e->p_group.destroy(), e->p_from.destroy(), e->p_term.destroy();
  e->p_id.destroy(), e->p_aux.destroy(), e->p_blk_id.destroy(), e->p_blk_next.destroy();
  for (void* p : {(void*)e->node.h_beat, (void*)e->node.h_ae, (void*)e->node.h_answer, (void*)e->node.h_hbc, (void*)e->node.h_nsparse,
                  (void*)e->node.h_in_answers, (void*)e->node.h_in_hbc})
    if (p) (void)hipHostFree(p);
  if (e->up.st) {
    (void)hipStreamSynchronize(e->up.st);
    (void)hipStreamDestroy(e->up.st);
    (void)hipEventDestroy(e->up.ev_up);
    for (hipEvent_t ev : e->up.ev_free) (void)hipEventDestroy(ev);
  }
  for (char* p : e->up.buf)
    if (p) (void)hipFree(p);
  if (e->fs_bk) (void)hipFree(e->fs_bk);
  if (e->node.sp_key) (void)hipFree(e->node.sp_key);
  if (e->node.sp_idx) (void)hipFree(e->node.sp_idx);
  if (e->node.ev_out) (void)hipEventDestroy(e->node.ev_out);
  if (e->node.ev_cols) (void)hipEventDestroy(e->node.ev_cols);
  e->q_msgs.destroy();
  e->q_fsm.destroy();
  e->l_msgs.destroy();
  e->l_fsm.destroy();
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->ev_stage) (void)hipEventDestroy(e->ev_stage);
  if (e->ev_order) (void)hipEventDestroy(e->ev_order);
  // (an engine destroyed while still in a jg_dense_cluster - against the documented order - must not
  // destroy the lead node's stream it was lent: its own one is the one to release)
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  else if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

uint32_t jg_shard_count(const jg_engine* e) { return !e ? 0u : e->router ? (uint32_t)e->router->D() : 1u; }

int jg_get_shard(jg_engine* e, uint32_t shard, jg_shard_info* out) {
  if (!e || !out) return fail(JG_EINVAL, "null argument");
  if (shard >= jg_shard_count(e)) return fail(JG_EINVAL, "shard out of range");
  jg_engine* s = e->router ? e->router->sh[shard] : e;
  out->engine = s;
  out->device_id = s->device;
  out->group_lo = e->router ? e->router->lo[shard] : 0;
  out->n_groups = s->cfg.n_groups;
  out->reserved = 0;
  return JG_OK;
}

int jg_set_self_slots(jg_engine* e, const uint8_t* slots) {
  if (!e || !slots) return fail(JG_EINVAL, "null argument");
  if (e->router) return router_set_self_slots(e, slots);
  if (e->stepped) return fail(JG_EINVAL, "self slots are fixed after the first step");
  for (uint32_t g = 0; g < e->cfg.n_groups; g++)
    if (slots[g] >= e->cfg.n_replicas) return fail(JG_EINVAL, "self slot out of range");
  e->uniform_self = slots[0];
  for (uint32_t g = 1; g < e->cfg.n_groups; g++)
    if (slots[g] != slots[0]) e->uniform_self = -1;
  HIPCHK(hipSetDevice(e->device));
  uint8_t* d_slots = nullptr;
  HIPCHK(hipMalloc((void**)&d_slots, std::max<size_t>(e->cfg.n_groups, 16)));
  HIPCHK(hipMemcpyAsync(d_slots, slots, e->cfg.n_groups, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_init_groups, dim3(grid_for(e->cfg.n_groups, 2048)), dim3(JG_BLOCK), 0, e->stream, e->dev,
                     (const uint8_t*)d_slots);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipFree(d_slots));
  return JG_OK;
}

int jg_submit(jg_engine* e, const jg_cmd_batch* b) {
  if (!e || !b) return fail(JG_EINVAL, "null argument");
  if (e->router) return router_submit(e, b);
  {
    uint32_t seen = 0;
    const int rc = validate_batch(e->cfg.n_groups, b, &seen);
    if (rc) return rc;
    e->p_kinds_seen |= seen;
  }
  const size_t at = e->p_kind.size(), n = b->n;
  if (e->up.valid) {  // rows behind an early upload (JG_COL_UPLOAD_NOW): the step uploads the whole batch itself
    HIPCHK(hipEventSynchronize(e->up.ev_up));  // (the columns may move when they grow)
    e->up.valid = false;
  }
  const uint64_t blk_shift = e->p_blk_id.size();
  HIPCHK(e->p_kind.append(b->kind, n));
  HIPCHK(e->p_group.append(b->group, n));
  HIPCHK(e->p_id.append(b->id, n));
  {
    const int rc = pending_optional(e, at, n, b->from, b->term, b->aux, b->flag);
    if (rc) return rc;
  }
  if (blk_shift && b->n_blocks)  // side arrays of successive submits are concatenated
    for (size_t i = 0; i < n; i++)
      if (b->kind[i] == JG_CMD_APPEND_ENTRIES) e->p_id[at + i] += blk_shift;
  if (b->n_blocks) {
    HIPCHK(e->p_blk_id.append(b->blk_id, b->n_blocks));
    HIPCHK(e->p_blk_next.append(b->blk_next, b->n_blocks));
  }
  return JG_OK;
}

namespace {
// the device image of a node step's rows: one section per column that is present, 16-byte aligned
void node_row_layout(const jg_engine* e, size_t n, size_t nb, jg_engine::RowLayout& l) {
  l = jg_engine::RowLayout{};
  l.n = n, l.nb = nb;
  l.has_from = e->p_has_from, l.has_term = e->p_has_term, l.has_aux = e->p_has_aux, l.has_flag = e->p_has_flag;
  size_t off = 0;
  auto sect = [&](size_t bytes) {
    size_t at = off;
    off = (off + bytes + 15) & ~size_t(15);
    return at;
  };
  l.o_id = sect(n * 8), l.o_term = sect(l.has_term ? n * 8 : 0), l.o_aux = sect(l.has_aux ? n * 8 : 0), l.o_bid = sect(nb * 8);
  l.o_bnext = sect(nb * 8), l.o_group = sect(n * 4), l.o_from = sect(l.has_from ? n * 4 : 0), l.o_kind = sect(n);
  l.o_flag = sect(l.has_flag ? n : 0);
  l.bytes = off;
}
// the pinned columns -> the device image at B, on stream st
int upload_node_rows(jg_engine* e, const jg_engine::RowLayout& l, char* B, hipStream_t st, uint64_t* bytes_up) {
  const size_t n = l.n, nb = l.nb;
  auto up = [&](size_t at, const void* src, size_t nbytes) -> hipError_t {
    if (bytes_up) *bytes_up += nbytes;
    return hipMemcpyAsync(B + at, src, nbytes, hipMemcpyHostToDevice, st);
  };
  HIPCHK(up(l.o_id, e->p_id.data(), n * 8));
  if (l.has_term) HIPCHK(up(l.o_term, e->p_term.data(), n * 8));
  if (l.has_aux) HIPCHK(up(l.o_aux, e->p_aux.data(), n * 8));
  HIPCHK(up(l.o_group, e->p_group.data(), n * 4));
  if (l.has_from) HIPCHK(up(l.o_from, e->p_from.data(), n * 4));
  HIPCHK(up(l.o_kind, e->p_kind.data(), n));
  if (l.has_flag) HIPCHK(up(l.o_flag, e->p_flag.data(), n));
  if (nb) {
    HIPCHK(up(l.o_bid, e->p_blk_id.data(), nb * 8));
    HIPCHK(up(l.o_bnext, e->p_blk_next.data(), nb * 8));
  }
  return JG_OK;
}
// JG_COL_UPLOAD_NOW: everything committed so far leaves for the device
int upload_rows_now(jg_engine* e) {
  jg_engine::EarlyUpload& u = e->up;
  u.valid = false;
  const size_t n = e->p_kind.size(), nb = e->p_blk_id.size();
  if (!n || n > 0x7fffffffull) return JG_OK;  // (the step says what is wrong with such a batch)
  HIPCHK(hipSetDevice(e->device));
  if (!u.st) {
    HIPCHK(hipStreamCreateWithFlags(&u.st, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&u.ev_up, hipEventDisableTiming));
    for (hipEvent_t& ev : u.ev_free) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  jg_engine::RowLayout l;
  node_row_layout(e, n, nb, l);
  const int k = u.turn;
  if (u.cap[k] < l.bytes) {  // (grow-only; hipFree waits for whoever still reads the old one)
    if (u.buf[k]) HIPCHK(hipFree(u.buf[k]));
    u.buf[k] = nullptr, u.read[k] = false;
    u.cap[k] = l.bytes + l.bytes / 2;
    HIPCHK(hipMalloc((void**)&u.buf[k], u.cap[k]));
  }
  if (u.read[k]) HIPCHK(hipStreamWaitEvent(u.st, u.ev_free[k], 0));  // (the step before last read its rows here)
  int rc = upload_node_rows(e, l, u.buf[k], u.st, nullptr);
  if (rc) return rc;
  HIPCHK(hipEventRecord(u.ev_up, u.st));
  u.lay = l, u.valid = true;
  return JG_OK;
}
}  // namespace

int jg_submit_reserve(jg_engine* e, size_t n, size_t n_blocks, jg_cmd_cols* cols) {
  if (!e || !cols) return fail(JG_EINVAL, "null argument");
  if (e->router) return fail(JG_EINVAL, "jg_submit_reserve: the columns are per shard: call this on a shard handle (jg_get_shard)");
  const size_t at = e->p_kind.size(), bat = e->p_blk_id.size();
  if (e->up.valid) HIPCHK(hipEventSynchronize(e->up.ev_up));  // (rows behind an early upload: the columns may move when they grow)
  HIPCHK(e->p_kind.reserve(at + n));
  HIPCHK(e->p_group.reserve(at + n));
  HIPCHK(e->p_from.reserve(at + n));
  HIPCHK(e->p_term.reserve(at + n));
  HIPCHK(e->p_id.reserve(at + n));
  HIPCHK(e->p_aux.reserve(at + n));
  HIPCHK(e->p_flag.reserve(at + n));
  HIPCHK(e->p_blk_id.reserve(bat + n_blocks));
  HIPCHK(e->p_blk_next.reserve(bat + n_blocks));
  cols->kind = e->p_kind.p + at, cols->group = e->p_group.p + at, cols->from = e->p_from.p + at, cols->term = e->p_term.p + at;
  cols->id = e->p_id.p + at, cols->aux = e->p_aux.p + at, cols->flag = e->p_flag.p + at;
  cols->blk_id = e->p_blk_id.p + bat, cols->blk_next = e->p_blk_next.p + bat;
  return JG_OK;
}

int jg_submit_commit(jg_engine* e, size_t n, size_t n_blocks, uint32_t optional_columns) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) return fail(JG_EINVAL, "jg_submit_commit: the columns are per shard: call this on a shard handle (jg_get_shard)");
  if (optional_columns & ~63u) return fail(JG_EINVAL, "unknown column bit");
  const size_t at = e->p_kind.size(), bat = e->p_blk_id.size();
  if (at + n > e->p_kind.cap || at + n > e->p_group.cap || at + n > e->p_id.cap || bat + n_blocks > e->p_blk_id.cap)
    return fail(JG_EINVAL, "jg_submit_commit: more rows than jg_submit_reserve made room for");
  jg_cmd_batch b{};  // what was written in place, as a batch: the same checks as jg_submit
  b.n = n, b.kind = e->p_kind.p + at, b.group = e->p_group.p + at, b.id = e->p_id.p + at, b.aux = e->p_aux.p + at;
  b.n_blocks = n_blocks, b.blk_id = e->p_blk_id.p + bat, b.blk_next = e->p_blk_next.p + bat;
  uint32_t seen = 0;
  if (optional_columns & JG_COL_UNCHECKED) {
    // no pass over the rows on the host (2.5 ms per 9 M rows): jg_step_node's classification checks group and kind on
    // the device; what the rows may hold is assumed (a Heartbeat; an AppendEntries if the aux column is there)
    seen = 2u | ((optional_columns & JG_COL_AUX) ? 1u : 0u);
    e->p_unchecked = true;
  } else {
    int rc = validate_batch(e->cfg.n_groups, &b, &seen);
    if (rc) return rc;
  }
  if ((seen & 1u) && !(optional_columns & JG_COL_AUX)) return fail(JG_EINVAL, "AppendEntries needs id/aux columns");
  e->p_kinds_seen |= seen;
  e->p_kind.n = e->p_group.n = e->p_id.n = at + n;
  // an optional column the caller filled is adopted where it lies (src == its own place: no copy)
  auto adopt = [&](auto& v, bool& has, bool given) {
    using T = typename std::remove_reference<decltype(*v.p)>::type;
    if (given) {
      if (!has && at) std::memset(v.p, 0, at * sizeof(T));
      has = true;
    } else if (has) {
      std::memset(v.p + at, 0, n * sizeof(T));
    }
    v.n = at + n;
  };
  adopt(e->p_from, e->p_has_from, (optional_columns & JG_COL_FROM) != 0);
  adopt(e->p_term, e->p_has_term, (optional_columns & JG_COL_TERM) != 0);
  adopt(e->p_aux, e->p_has_aux, (optional_columns & JG_COL_AUX) != 0);
  adopt(e->p_flag, e->p_has_flag, (optional_columns & JG_COL_FLAG) != 0);
  if (bat && n_blocks)
    for (size_t i = 0; i < n; i++)
      if (b.kind[i] == JG_CMD_APPEND_ENTRIES) e->p_id[at + i] += bat;
  e->p_blk_id.n = e->p_blk_next.n = bat + n_blocks;
  if (optional_columns & JG_COL_UPLOAD_NOW) return upload_rows_now(e);
  e->up.valid = false;  // (rows behind an early upload: the step uploads the whole batch itself)
  return JG_OK;
}

int jg_step(jg_engine* e, uint64_t now_ms) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) return router_step(e, now_ms);
  {
    const int rc = node_settle(e);
    if (rc) return rc;
  }
  e->stepped = true;
  const size_t n = e->p_kind.size();
  if (!n) return JG_OK;
  if (n > 0x7fffffffull) return fail(JG_EINVAL, "batch too large: split it");
  if (e->p_unchecked) return fail(JG_EINVAL, "rows committed with JG_COL_UNCHECKED are validated by jg_step_node's classification only: call jg_step_node");
  HIPCHK(hipSetDevice(e->device));
  e->seq++;
  pending_materialise(e);
  std::vector<uint32_t> order;
  sort_rows_by_group(e->p_group.data(), n, e->cfg.n_groups, order);
  const size_t nb = e->p_blk_id.size();

  // one blob: 8-byte columns first, then 4-byte, then 1-byte (16-byte aligned sections)
  size_t off = 0;
  auto sect = [&](size_t bytes) {
    size_t at = off;
    off = (off + bytes + 15) & ~size_t(15);
    return at;
  };
  const size_t o_term = sect(n * 8), o_id = sect(n * 8), o_aux = sect(n * 8), o_bid = sect(nb * 8),
               o_bnext = sect(nb * 8), o_group = sect(n * 4), o_from = sect(n * 4), o_kind = sect(n),
               o_flag = sect(n);
  const size_t bytes = off;
  if (e->stage_busy) {  // the previous step's upload may still be reading the pinned buffer
    HIPCHK(hipEventSynchronize(e->ev_stage));
    e->stage_busy = false;
  }
  if (e->stage_cap < bytes) {
    if (e->stage) HIPCHK(hipHostFree(e->stage));
    e->stage = nullptr;
    e->stage_cap = std::max(bytes * 2, (size_t)1 << 20);
    HIPCHK(hipHostMalloc((void**)&e->stage, e->stage_cap, hipHostMallocDefault));
  }
  char* S = e->stage;
  uint64_t *s_term = (uint64_t*)(S + o_term), *s_id = (uint64_t*)(S + o_id), *s_aux = (uint64_t*)(S + o_aux);
  uint32_t *s_group = (uint32_t*)(S + o_group), *s_from = (uint32_t*)(S + o_from);
  uint8_t *s_kind = (uint8_t*)(S + o_kind), *s_flag = (uint8_t*)(S + o_flag);
  for (size_t k = 0; k < n; k++) {
    const uint32_t i = order[k];
    s_term[k] = e->p_term[i];
    s_id[k] = e->p_id[i];
    s_aux[k] = e->p_aux[i];
    s_group[k] = e->p_group[i];
    s_from[k] = e->p_from[i];
    s_kind[k] = e->p_kind[i];
    s_flag[k] = e->p_flag[i];
  }
  if (nb) {
    std::memcpy(S + o_bid, e->p_blk_id.data(), nb * 8);
    std::memcpy(S + o_bnext, e->p_blk_next.data(), nb * 8);
  }
  char* B = nullptr;
  HIPCHK(e->arenas[e->cur_arena].alloc(bytes, (void**)&B));
  HIPCHK(hipMemcpyAsync(B, S, bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipEventRecord(e->ev_stage, e->stream));
  e->stage_busy = true;
  int rc = launch_rows(e, (uint32_t)n, (const uint32_t*)(B + o_group), (const uint8_t*)(B + o_kind),
                       (const uint32_t*)(B + o_from), (const uint64_t*)(B + o_term), (const uint64_t*)(B + o_id),
                       (const uint64_t*)(B + o_aux), (const uint8_t*)(B + o_flag), (const uint64_t*)(B + o_bid),
                       (const uint64_t*)(B + o_bnext), nb, now_ms);
  if (rc) return rc;
  e->up.valid = false;
  e->p_kind.clear();
  e->p_flag.clear();
  e->p_group.clear();
  e->p_from.clear();
  e->p_term.clear();
  e->p_id.clear();
  e->p_aux.clear();
  e->p_blk_id.clear();
  e->p_blk_next.clear();
  e->p_has_from = e->p_has_term = e->p_has_aux = e->p_has_flag = false;
  e->p_kinds_seen = 0;
  return JG_OK;
}

int jg_step_device_rows(jg_engine* e, const jg_cmd_batch* b, uint64_t now_ms) {
  if (!e || !b) return fail(JG_EINVAL, "null argument");
  if (e->router) return fail(JG_EINVAL, "device pointers are per shard: call this on a shard handle (jg_get_shard)");
  if (!e->p_kind.empty()) return fail(JG_EINVAL, "commands are queued: call jg_step first");
  {
    const int rc = node_settle(e);
    if (rc) return rc;
  }
  e->stepped = true;
  if (!b->n) return JG_OK;
  if (b->n > 0x7fffffffull) return fail(JG_EINVAL, "batch too large: split it");
  if (!b->kind || !b->group || !b->from || !b->term || !b->id || !b->aux || !b->flag)
    return fail(JG_EINVAL, "all seven device columns are required");
  HIPCHK(hipSetDevice(e->device));
  e->seq++;
  if (b->n_blocks && (!b->blk_id || !b->blk_next)) return fail(JG_EINVAL, "block side arrays are required");
  // every AppendEntries row's block range is checked against n_blocks on the device (error word 5 -> JG_EINVAL at the
  // next synchronising call; the row is not applied): a batch without side arrays can only carry empty AppendEntries
  const uint64_t* none = (const uint64_t*)e->d_ones;
  return launch_rows(e, (uint32_t)b->n, b->group, b->kind, b->from, b->term, b->id, b->aux, b->flag,
                     b->n_blocks ? b->blk_id : none, b->n_blocks ? b->blk_next : none, b->n_blocks, now_ms);
}

int jg_step_dense_acks_device(jg_engine* e, const uint64_t* acks_dev) {
  if (!e || !acks_dev) return fail(JG_EINVAL, "null argument");
  if (e->router) return fail(JG_EINVAL, "one block per shard: jg_step_dense_acks_shards");
  if (!e->p_kind.empty()) return fail(JG_EINVAL, "commands are queued: call jg_step first");
  HIPCHK(hipSetDevice(e->device));
  return dense_step(e, acks_dev);
}

int jg_step_dense_acks_shards(jg_engine* e, const uint64_t* const* acks_dev, uint32_t n_ticks) {
  if (!e || !acks_dev) return fail(JG_EINVAL, "null argument");
  if (!n_ticks) return JG_OK;
  if (e->router) return router_step_dense_acks_shards(e, acks_dev, n_ticks);
  return jg_step_dense_acks_device_n(e, acks_dev[0], n_ticks);
}

int jg_step_dense_acks_device_n(jg_engine* e, const uint64_t* acks_dev, uint32_t n_ticks) {
  if (!e || !acks_dev) return fail(JG_EINVAL, "null argument");
  if (!n_ticks) return JG_OK;
  if (e->router) return fail(JG_EINVAL, "one block per shard: jg_step_dense_acks_shards");
  if (!e->p_kind.empty()) return fail(JG_EINVAL, "commands are queued: call jg_step first");
  HIPCHK(hipSetDevice(e->device));
  return dense_step(e, acks_dev, n_ticks);
}

int jg_step_dense_acks(jg_engine* e, const uint64_t* acks_host) {
  if (!e || !acks_host) return fail(JG_EINVAL, "null argument");
  if (e->router) return router_step_dense_acks(e, acks_host);
  if (!e->p_kind.empty()) return fail(JG_EINVAL, "commands are queued: call jg_step first");
  HIPCHK(hipSetDevice(e->device));
  const size_t bytes = (size_t)e->cfg.n_groups * e->cfg.n_replicas * 8;
  if (!e->d_acks_staging) HIPCHK(hipMalloc((void**)&e->d_acks_staging, std::max<size_t>(bytes, 16)));
  HIPCHK(hipMemcpyAsync(e->d_acks_staging, acks_host, bytes, hipMemcpyHostToDevice, e->stream));
  int rc = dense_step(e, e->d_acks_staging);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));  // the host buffer is only borrowed for the call
  return JG_OK;
}

int jg_step_dense_leader(jg_engine* e, uint64_t now_ms, const jg_leader_inbox* in, const jg_leader_outbox* out) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) return fail(JG_EINVAL, "device pointers are per shard: call this on a shard handle (jg_get_shard)");
  if (!e->p_kind.empty()) return fail(JG_EINVAL, "commands are queued: call jg_step first");
  if (out && (!out->beat || !out->ae)) return fail(JG_EINVAL, "every outbox column is required");
  if (in && in->answers && !in->hbr_commit) return fail(JG_EINVAL, "answers need hbr_commit");
  HIPCHK(hipSetDevice(e->device));
  int rc = ensure_xq(e);
  if (rc) return rc;
  JgLeaderNode nd{};
  nd.clock = e->replay_clock, nd.clock_slot = e->replay_slot;
  nd.hbr_commit = in ? in->hbr_commit : nullptr;
  nd.packed = 1;
  if (out) {
    nd.o_beat = out->beat;
    nd.o_ae = out->ae;
  }
  nd.now = now_ms;
  const uint64_t* acks = in ? in->answers : nullptr;
  if (!acks && !out) return JG_OK;  // nothing to apply
  nd.ack_stride = acks ? 1 : 0;
  return dense_step(e, acks, 1, &nd);
}

namespace {
// the two launches of a follower half; `fsm_*`: jg_step_node's fsm delta columns (or null)
int follower_half(jg_engine* e, uint64_t now_ms, const jg_follower_inbox* in, const jg_follower_outbox* out, int tick,
                  uint32_t* fsm_delta, uint64_t* fsm_prev, const uint64_t* sparse_bits = nullptr, uint32_t sparse_mode = 0) {
  if (sparse_mode != 2u) {  // (not from inside node_settle's own catch-up pass)
    const int rc = node_settle(e);
    if (rc) return rc;
  }
  e->stepped = true;
  e->seq++;
  JgFollowerArgs a{};
  a.clock = e->replay_clock, a.clock_slot = e->replay_slot;
  a.leader = in->leader;
  a.leader_id = in->leader_id;
  a.beat = in->beat;
  a.ae = in->ae;
  a.o_answer = out->answer;
  a.o_hbc = out->hb_commit;
  a.now = now_ms;
  a.seq = e->seq;
  a.tick = tick ? 1 : 0;
  a.fsm_delta = fsm_delta;
  a.fsm_prev = fsm_prev;
  a.sparse_bits = sparse_bits, a.sparse_mode = sparse_mode;
  hipLaunchKernelGGL(k_follower_tick_dense, dim3(e->dense_grid), dim3(JG_BLOCK), 0, e->stream, e->dev, a);
  // always scheduled: which groups need the general state machine is only known on the device
  // (empty lists cost a few microseconds)
  e->slow_scheduled_ever = true;
  hipLaunchKernelGGL(k_follower_slow, dim3(JG_SHARDS), dim3(JG_BLOCK), 0, e->stream, e->dev, a);
  HIPCHK(hipGetLastError());
  e->n_launch += 2;
  e->n_dense += e->cfg.n_groups;
  // a deferred follower may have become a candidate / changed its chain: like a sparse step
  e->maybe_irregular = true;
  e->flag_check_pending = true;
  e->irr_gen++;
  return JG_OK;
}
}  // namespace

int jg_step_dense_follower(jg_engine* e, uint64_t now_ms, const jg_follower_inbox* in, const jg_follower_outbox* out,
                           int tick) {
  if (!e || !in || !out) return fail(JG_EINVAL, "null argument");
  if (e->router) return fail(JG_EINVAL, "device pointers are per shard: call this on a shard handle (jg_get_shard)");
  if (!e->p_kind.empty()) return fail(JG_EINVAL, "commands are queued: call jg_step first");
  if (!in->beat || !in->ae) return fail(JG_EINVAL, "every inbox column is required");
  if (!out->answer || !out->hb_commit) return fail(JG_EINVAL, "every outbox column is required");
  if (!in->leader && !in->leader_id) return fail(JG_EINVAL, "id cannot be 0");  // config.rs:64-66
  HIPCHK(hipSetDevice(e->device));
  int rc = ensure_xq(e);
  if (rc) return rc;
  return follower_half(e, now_ms, in, out, tick, nullptr, nullptr);
}


// ---- jg_step_node: a node's whole tick from host rows (jg_node.h) -----------------------------------
namespace {
int node_ensure(jg_engine* e) {
  jg_engine::NodeStep& n = e->node;
  if (n.ready) return JG_OK;
  const size_t G = e->cfg.n_groups, R = e->cfg.n_replicas;
  int rc = JG_OK;
#define A(ptr, cnt) \
  if ((rc = dev_alloc(e, &ptr, (cnt))) != JG_OK) return rc
  A(n.cols.answers, R * G);
  A(n.cols.hbr_commit, R * G);
  A(n.cols.token, G);
  A(n.cols.f_beat, G);
  A(n.cols.f_ae, G);
  A(n.cols.f_leader, G);
  A(n.cols.cls, G);
  A(n.cols.lt_max, G);
  A(n.cols.lt_min, G);
  A(n.cols.lf_max, G);
  A(n.cols.lf_min, G);
  A(n.cols.fsm_delta, G);
  A(n.cols.fsm_prev, G);
  A(n.cols.fsm_mid, G);
  A(n.cols.arr, 2 * R * G);
  A(n.cols.fo, 2 * G);
  A(n.cols.sparse_bits, (G + 63) / 64);
  A(n.o_beat, G);
  A(n.o_ae, R * G);
  HIPCHK(hipMemsetAsync(n.o_ae, 0xff, std::max<size_t>(R * G * 8, 16), e->stream));  // (the own slot's row is never written: JG_NO_ACK once)
  A(n.o_answer, G);
  A(n.o_hbc, G);
  A(n.d_nsparse, 4);
#define A2(ptr, cnt) \
  if ((rc = dev_alloc(e, &ptr, (cnt))) != JG_OK) return rc
#undef A
  HIPCHK(hipHostMalloc((void**)&n.h_beat, std::max<size_t>(G * sizeof(jg_leader_beat), 16), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&n.h_ae, std::max<size_t>(R * G * 8, 16), hipHostMallocDefault));
  std::memset(n.h_ae, 0xff, std::max<size_t>(R * G * 8, 16));  // (the own slot's row is not downloaded while it is the same for every group)
  HIPCHK(hipHostMalloc((void**)&n.h_answer, std::max<size_t>(G * 8, 16), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&n.h_hbc, std::max<size_t>(G * 8, 16), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&n.h_nsparse, 16, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&n.h_in_answers, std::max<size_t>(R * G * 8, 16), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&n.h_in_hbc, std::max<size_t>(R * G * 8, 16), hipHostMallocDefault));
  HIPCHK(hipEventCreateWithFlags(&n.ev_out, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&n.ev_cols, hipEventDisableTiming));
  while (n.group_bits < 32 && (G - 1) >> n.group_bits) n.group_bits++;
  n.bk_tile_bits = std::min<uint32_t>(JG_ROUTE_TILE_BITS, n.group_bits);
  n.bk_buckets = ((uint32_t)G + (1u << n.bk_tile_bits) - 1u) >> n.bk_tile_bits;
  const uint32_t bk_tiles = (n.bk_buckets + JG_ROUTE_SCAN_TILE - 1) / JG_ROUTE_SCAN_TILE;
  n.bk_words = bk_tiles * JG_ROUTE_SCAN_TILE + n.bk_buckets + bk_tiles + 1;  // hist (whole tiles) | cur | tile
  A2(n.bk_mem, n.bk_words);
  n.ready = true;
  return JG_OK;
}

int node_dense_halves(jg_engine* e, uint64_t now_ms, uint32_t flags, uint32_t col_mask, uint32_t sparse_mode, uint64_t* bytes_down);
int node_general(jg_engine* e, const JgNodeRows& rows, size_t n, size_t nb, uint32_t n_sparse, uint64_t now_ms);

int node_step(jg_engine* e, uint64_t now_ms, uint32_t flags) {
  jg_engine::NodeStep& nd = e->node;
  const uint32_t halves = flags & (JG_NODE_LEADER_HALF | JG_NODE_FOLLOWER_HALF);
  // JG_NODE_ASYNC: no synchronisation inside the step - the general-path row count is looked at when the step is settled
  const bool async = (flags & JG_NODE_ASYNC) != 0;
  HIPCHK(hipSetDevice(e->device));
  int rc = node_ensure(e);
  if (rc) return rc;
  if ((rc = node_settle(e))) return rc;  // (an earlier asynchronous step)
  if ((rc = ensure_xq(e))) return rc;
  e->stepped = true;
  jg_engine::NodeStep::Pending& pend = nd.pending;
  pend = jg_engine::NodeStep::Pending{};
  const uint32_t seq0 = e->seq;
  const size_t n = e->p_kind.size(), nb = e->p_blk_id.size();
  if (n > 0x7fffffffull) return fail(JG_EINVAL, "batch too large: split it");
  const uint32_t G = e->cfg.n_groups, R = e->cfg.n_replicas;
  static const bool trace = std::getenv("JG_TRACE_NODE") != nullptr;
  auto clk = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double T0 = clk();
  double T1 = T0, T2 = T0;
  const uint32_t both_beats = (e->p_kinds_seen & 3u) == 3u;  // (a batch with Heartbeat AND AppendEntries rows: their consistency columns are needed)
  const uint32_t ggrid = grid_for(G, 4096);
  uint64_t bytes_up = 0;
  // column inbound: the handed-out slots' columns go up as they are (8 bytes per partition and peer instead of two rows)
  if (nd.col_mask && !(halves & JG_NODE_LEADER_HALF))  // (never dropped silently: the leader half is what applies them)
    return fail(JG_EINVAL, "jg_step_node: a column was handed out (jg_node_inbox_columns) but the leader half does not run");
  const uint32_t col_mask = nd.col_mask;
  for (uint32_t r = 0; r < R;) {  // (neighbouring slots travel in one copy: a copy costs ~10 us before its first byte)
    if (!((col_mask >> r) & 1u)) {
      r++;
      continue;
    }
    uint32_t r1 = r + 1;
    while (r1 < R && ((col_mask >> r1) & 1u) && (((nd.col_hbc_mask >> r1) & 1u) == ((nd.col_hbc_mask >> r) & 1u))) r1++;
    const size_t at = (size_t)r * G, len = (size_t)(r1 - r) * G * 8;
    HIPCHK(hipMemcpyAsync(nd.cols.answers + at, nd.h_in_answers + at, len, hipMemcpyHostToDevice, e->stream));
    if ((nd.col_hbc_mask >> r) & 1u)
      HIPCHK(hipMemcpyAsync(nd.cols.hbr_commit + at, nd.h_in_hbc + at, len, hipMemcpyHostToDevice, e->stream));
    else
      HIPCHK(hipMemsetAsync(nd.cols.hbr_commit + at, 0, len, e->stream));
    bytes_up += len * (((nd.col_hbc_mask >> r) & 1u) ? 2 : 1);
    r = r1;
  }
  if (col_mask) {  // (jg_node_inbox_columns waits for this before it hands the same pinned buffers out again)
    HIPCHK(hipEventRecord(nd.ev_cols, e->stream));
    nd.cols_in_flight = true;
  }
  nd.col_mask = nd.col_hbc_mask = 0;  // (a hand-out covers one step)
  hipLaunchKernelGGL(k_node_prefill, dim3(ggrid), dim3(JG_BLOCK), 0, e->stream, e->dev, nd.cols, e->uniform_self,
                     halves & JG_NODE_LEADER_HALF, halves & JG_NODE_FOLLOWER_HALF, both_beats, col_mask);
  uint32_t n_sparse = 0;
  // the general path's sequence number is taken whether or not it runs: the shards of a multi-device engine must leave
  // one node step with the same numbers (the router merges their rows by step number first: jg_multi.h)
  e->seq = seq0 + 1;
  if (n) {
    // the rows in stream order, straight out of the pinned columns jg_submit (or the caller, in place:
    // jg_submit_reserve) filled: one copy per column that is present - an optional column nobody
    // provided is all zeros and is not uploaded at all (an AppendResponse row is 18 bytes then, not 34)
    jg_engine::RowLayout lay;
    node_row_layout(e, n, nb, lay);
    const bool has_from = lay.has_from, has_term = lay.has_term, has_aux = lay.has_aux, has_flag = lay.has_flag;
    const size_t o_id = lay.o_id, o_term = lay.o_term, o_aux = lay.o_aux, o_bid = lay.o_bid, o_bnext = lay.o_bnext, o_group = lay.o_group,
                 o_from = lay.o_from, o_kind = lay.o_kind, o_flag = lay.o_flag;
    if (nd.sp_cap < n) {  // (room for every row on the general path; grow-only, like the pinned columns)
      if (nd.sp_key) HIPCHK(hipFree(nd.sp_key));
      if (nd.sp_idx) HIPCHK(hipFree(nd.sp_idx));
      nd.sp_cap = n + n / 2;
      HIPCHK(hipMalloc((void**)&nd.sp_key, nd.sp_cap * 8));
      HIPCHK(hipMalloc((void**)&nd.sp_idx, nd.sp_cap * 4));
    }
    char* B = nullptr;
    jg_engine::EarlyUpload& u = e->up;
    if (u.last_used >= 0) {  // whoever read the last step's rows (its settling included) is in the stream by now
      HIPCHK(hipEventRecord(u.ev_free[u.last_used], e->stream));
      u.read[u.last_used] = true;
      u.last_used = -1;
    }
    if (u.valid && u.lay.same_batch(lay)) {
      // JG_COL_UPLOAD_NOW: the batch left when it was committed - the kernels wait for its copies, nothing else does
      B = u.buf[u.turn];
      HIPCHK(hipStreamWaitEvent(e->stream, u.ev_up, 0));
      bytes_up += n * (8u + 4u + 1u + (has_term ? 8u : 0u) + (has_aux ? 8u : 0u) + (has_from ? 4u : 0u) + (has_flag ? 1u : 0u)) + nb * 16u;
      u.last_used = u.turn;
      u.turn ^= 1;
    } else {
      Arena& ar = e->arenas[e->cur_arena];
      HIPCHK(ar.alloc(lay.bytes, (void**)&B));
      if ((rc = upload_node_rows(e, lay, B, e->stream, &bytes_up))) return rc;
    }
    u.valid = false;
    // (the pinned columns are free again after the synchronisation below)
    JgNodeRows rows{};
    rows.n = (uint32_t)n;
    rows.group = (const uint32_t*)(B + o_group), rows.kind = (const uint8_t*)(B + o_kind);
    rows.from = has_from ? (const uint32_t*)(B + o_from) : nullptr, rows.term = has_term ? (const uint64_t*)(B + o_term) : nullptr;
    rows.id = (const uint64_t*)(B + o_id), rows.aux = has_aux ? (const uint64_t*)(B + o_aux) : nullptr;
    rows.flag = has_flag ? (const uint8_t*)(B + o_flag) : nullptr;
    rows.blk_id = (const uint64_t*)(B + o_bid), rows.blk_next = (const uint64_t*)(B + o_bnext), rows.n_blocks = nb;
    const uint32_t rgrid = grid_for(n, 4096);
    HIPCHK(hipMemsetAsync(nd.d_nsparse, 0, 8, e->stream));
    hipLaunchKernelGGL(k_node_classify, dim3(rgrid), dim3(JG_BLOCK), 0, e->stream, e->dev, nd.cols, rows, e->uniform_self,
                       halves, both_beats, col_mask);
    hipLaunchKernelGGL(k_node_route, dim3(rgrid), dim3(JG_BLOCK), 0, e->stream, e->dev, nd.cols, rows, e->uniform_self,
                       both_beats, nd.sp_key, nd.sp_idx, nd.d_nsparse);
    HIPCHK(hipGetLastError());
    e->n_launch += 3;
    HIPCHK(hipMemcpyAsync(nd.h_nsparse, nd.d_nsparse, 4, hipMemcpyDeviceToHost, e->stream));
    if (!async) {
      // the one synchronisation of a synchronous step: how many rows take the general path sizes that launch
      T1 = clk();
      HIPCHK(hipStreamSynchronize(e->stream));
      T2 = clk();
      n_sparse = nd.h_nsparse[0];
      if (n_sparse && (rc = node_general(e, rows, n, nb, n_sparse, now_ms))) return rc;
    }
    pend.rows = rows, pend.n = n, pend.nb = nb;
    // the pinned columns: the OTHER set from here on (an asynchronous step's uploads may still be reading this one)
    e->p_kind.flip(), e->p_flag.flip(), e->p_group.flip(), e->p_from.flip(), e->p_term.flip(), e->p_id.flip();
    e->p_aux.flip(), e->p_blk_id.flip(), e->p_blk_next.flip();
    e->p_has_from = e->p_has_term = e->p_has_aux = e->p_has_flag = false;
    e->p_kinds_seen = 0;
    e->p_unchecked = false;
  }
  // the dense halves: every partition, the ones whose rows went the general way included (they are ticked here) -
  // except in an asynchronous step, whose halves leave those partitions to the catch-up pass (node_settle)
  pend.now_ms = now_ms, pend.flags = flags, pend.col_mask = col_mask;
  e->seq = seq0 + 1;  // (the general path's number: taken above whether or not it runs)
  uint64_t bytes_down = 0;
  if ((rc = node_dense_halves(e, now_ms, flags, col_mask, async && n ? 1u : 0u, &bytes_down))) return rc;
  {  // fsm_tx rows of the dense halves -> a step record of its own (per-group regions; compacted by the drains)
    StepRec rec;
    rec.n = G;
    rec.seq = e->seq;
    rec.msg_per_row = 0;
    rec.fsm_per_row = JGN_FSM_ROWS;
    Arena& ar = e->arenas[e->cur_arena];
    const uint32_t n_tiles = (G + JG_SCAN_TILE - 1) / JG_SCAN_TILE;
    HIPCHK(ar.alloc((size_t)G * 4, (void**)&rec.d_fsm_cnt));
    HIPCHK(ar.alloc((size_t)G * JGN_FSM_ROWS * sizeof(jg_fsm_row), (void**)&rec.d_fsm));
    HIPCHK(ar.alloc((size_t)n_tiles * 8, (void**)&rec.d_bsum_f));
    hipLaunchKernelGGL(k_node_fsm_build, dim3(n_tiles), dim3(JG_BLOCK), 0, e->stream, e->dev, nd.cols, rec.d_fsm, rec.d_fsm_cnt,
                       rec.d_bsum_f);
    HIPCHK(hipGetLastError());
    e->n_launch++;
    e->recs.push_back(rec);
    pend.fsm_rec_seq = rec.seq;
  }
  pend.seq_general = seq0 + 1, pend.seq_end = e->seq;
  HIPCHK(hipEventRecord(nd.ev_out, e->stream));
  if (trace)
    std::fprintf(stderr, "[jg node] %zu rows: uploads + classify + route issued in %.0f us, waited %.0f us (H2D %.1f MB), halves + fsm build + outbox copies issued in %.0f us\n",
                 n, T1 - T0, T2 - T1, bytes_up / 1e6, clk() - T2);
  nd.last = jg_node_outbox{};
  nd.last.rows = n, nd.last.rows_general = n_sparse, nd.last.bytes_h2d = bytes_up, nd.last.bytes_d2h = bytes_down;
  nd.last_flags = flags;
  pend.on = async && n != 0;  // (nothing is pending when there were no rows: no general path to come back for)
  return JG_OK;
}

// The dense halves of a node step + the downloads of their outbox columns.  sparse_mode: 0 every partition; 1 all but
// the partitions whose rows take the general path (an asynchronous step, first pass); 2 only those (its catch-up pass).
int node_dense_halves(jg_engine* e, uint64_t now_ms, uint32_t flags, uint32_t col_mask, uint32_t sparse_mode, uint64_t* bytes_down) {
  jg_engine::NodeStep& nd = e->node;
  const uint32_t halves = flags & (JG_NODE_LEADER_HALF | JG_NODE_FOLLOWER_HALF);
  const bool tick = (flags & JG_NODE_TICK) != 0;
  const uint32_t G = e->cfg.n_groups, R = e->cfg.n_replicas;
  int rc = JG_OK;
  if (halves & JG_NODE_LEADER_HALF) {
    JgLeaderNode ln{};
    ln.hbr_commit = nd.cols.hbr_commit;
    ln.packed = 1;
    ln.ack_stride = 1;
    if (tick) ln.o_beat = nd.o_beat, ln.o_ae = nd.o_ae;
    ln.now = now_ms;
    ln.fsm_delta = nd.cols.fsm_delta, ln.fsm_prev = nd.cols.fsm_prev, ln.fsm_mid = nd.cols.fsm_mid;
    ln.arr = nd.cols.arr, ln.col_mask = col_mask;  // (the slow kernel replays its groups in arrival order)
    if (sparse_mode) ln.sparse_bits = nd.cols.sparse_bits, ln.sparse_mode = sparse_mode;
    if ((rc = dense_step(e, nd.cols.answers, 1, &ln))) return rc;
    if (tick) {
      HIPCHK(hipMemcpyAsync(nd.h_beat, nd.o_beat, (size_t)G * sizeof(jg_leader_beat), hipMemcpyDeviceToHost, e->stream));
      // (the own slot's row is JG_NO_ACK on both sides and stays there: not written, not downloaded)
      const uint32_t own = e->uniform_self >= 0 ? (uint32_t)e->uniform_self : R;
      if (own > 0) HIPCHK(hipMemcpyAsync(nd.h_ae, nd.o_ae, (size_t)std::min(own, R) * G * 8, hipMemcpyDeviceToHost, e->stream));
      if (own + 1 < R)
        HIPCHK(hipMemcpyAsync(nd.h_ae + (size_t)(own + 1) * G, nd.o_ae + (size_t)(own + 1) * G, (size_t)(R - own - 1) * G * 8, hipMemcpyDeviceToHost,
                              e->stream));
      *bytes_down += (size_t)G * (sizeof(jg_leader_beat) + (size_t)(own < R ? R - 1 : R) * 8);
    }
  }
  if (halves & JG_NODE_FOLLOWER_HALF) {
    jg_follower_inbox fi{};
    fi.leader = nd.cols.f_leader, fi.beat = nd.cols.f_beat, fi.ae = nd.cols.f_ae;
    const jg_follower_outbox fo{nd.o_answer, nd.o_hbc};
    if ((rc = follower_half(e, now_ms, &fi, &fo, tick ? 1 : 0, nd.cols.fsm_delta, nd.cols.fsm_prev,
                            sparse_mode ? nd.cols.sparse_bits : nullptr, sparse_mode)))
      return rc;
    HIPCHK(hipMemcpyAsync(nd.h_answer, nd.o_answer, (size_t)G * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(nd.h_hbc, nd.o_hbc, (size_t)G * 8, hipMemcpyDeviceToHost, e->stream));
    *bytes_down += (size_t)G * 16;
  }
  return JG_OK;
}

// The general path of a node step: the rows k_node_route listed (in no particular order) are put into group-major
// order, a group's rows in the order they arrived, by the bucket pass of jg_route.h - key = group << 32 | arrival index,
// a bucket = 256 groups, one workgroup ranks a bucket - and become the batch k_apply_rows takes: exactly jg_submit +
// jg_step for those partitions, in stream order.  (Round 3: rocprim::select + radix_sort_pairs.)
int node_general(jg_engine* e, const JgNodeRows& rows, size_t n, size_t nb, uint32_t n_sparse, uint64_t now_ms) {
  jg_engine::NodeStep& nd = e->node;
  Arena& ar = e->arenas[e->cur_arena];
  int rc = JG_OK;
  (void)n;
  uint64_t* key_alt = nullptr;
  uint32_t *idx_alt = nullptr, *order = nullptr;
  HIPCHK(ar.alloc((size_t)n_sparse * 8, (void**)&key_alt));
  HIPCHK(ar.alloc((size_t)n_sparse * 4, (void**)&idx_alt));
  HIPCHK(ar.alloc((size_t)n_sparse * 4, (void**)&order));
  JgRouteBuckets bk{};
  bk.n_buckets = nd.bk_buckets, bk.shift = 32 + nd.bk_tile_bits;
  const uint32_t bk_tiles = (bk.n_buckets + JG_ROUTE_SCAN_TILE - 1) / JG_ROUTE_SCAN_TILE;
  bk.hist = nd.bk_mem, bk.cur = bk.hist + (size_t)bk_tiles * JG_ROUTE_SCAN_TILE, bk.tile = bk.cur + bk.n_buckets;
  hipStream_t st = e->stream;
  hipLaunchKernelGGL(k_route_clear, dim3(64), dim3(JG_BLOCK), 0, st, bk.hist, bk_tiles * JG_ROUTE_SCAN_TILE + bk.n_buckets, bk.tile, bk_tiles + 1);
  const uint32_t grid = std::min<uint32_t>((n_sparse + JG_BLOCK - 1) / JG_BLOCK, 4096);
  hipLaunchKernelGGL(k_route_hist, dim3(grid, 1), dim3(JG_BLOCK), 0, st, (const uint32_t*)nd.d_nsparse, (uint32_t)nd.sp_cap, (const uint64_t*)nd.sp_key, bk);
  hipLaunchKernelGGL(k_route_scan, dim3(bk_tiles), dim3(JG_BLOCK), 0, st, bk);
  hipLaunchKernelGGL(k_route_scan_tiles, dim3(1), dim3(JG_BLOCK), 0, st, bk);
  hipLaunchKernelGGL(k_route_scatter, dim3(grid, 1), dim3(JG_BLOCK), 0, st, (const uint32_t*)nd.d_nsparse, (uint32_t)nd.sp_cap, (const uint64_t*)nd.sp_key,
                     (const uint32_t*)nd.sp_idx, bk, key_alt, idx_alt);
  hipLaunchKernelGGL(k_bucket_order, dim3(bk.n_buckets), dim3(JG_BLOCK), 0, st, bk, (const uint64_t*)key_alt, (const uint32_t*)idx_alt, order);
  const uint32_t sgrid = (n_sparse + JG_BLOCK - 1) / JG_BLOCK;
  JgNodeSorted so{};
  char* M = nullptr;
  const size_t ns = n_sparse;
  HIPCHK(ar.alloc(ns * 34 + 64, (void**)&M));  // 3 x 8 + 2 x 4 + 2 x 1 bytes per row, widest columns first
  so.term = (uint64_t*)M, M += ns * 8;
  so.id = (uint64_t*)M, M += ns * 8;
  so.aux = (uint64_t*)M, M += ns * 8;
  so.group = (uint32_t*)M, M += ns * 4;
  so.from = (uint32_t*)M, M += ns * 4;
  so.kind = (uint8_t*)M, M += ns;
  so.flag = (uint8_t*)M;
  hipLaunchKernelGGL(k_node_gather_rows, dim3(sgrid), dim3(JG_BLOCK), 0, st, n_sparse, (const uint32_t*)order, rows, so);
  HIPCHK(hipGetLastError());
  e->n_launch += 7;
  if ((rc = launch_rows(e, n_sparse, so.group, so.kind, so.from, so.term, so.id, so.aux, so.flag,
                        nb ? rows.blk_id : (const uint64_t*)e->d_ones, nb ? rows.blk_next : (const uint64_t*)e->d_ones, nb, now_ms)))
    return rc;
  return JG_OK;
}

// An asynchronous node step is settled the first time anything looks at the engine again: the general-path row count
// has landed by then; if it is not zero, those rows are applied now (their sequence number was reserved) and the dense
// halves come back for exactly the partitions they left alone - same results, same record order, one pass later.
int node_settle(jg_engine* e) {
  jg_engine::NodeStep& nd = e->node;
  jg_engine::NodeStep::Pending& pd = nd.pending;
  if (!pd.on) return JG_OK;
  pd.on = false;
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  const uint32_t n_sparse = nd.h_nsparse[0];
  nd.last.rows_general = n_sparse;
  if (!n_sparse) return JG_OK;
  int rc = JG_OK;
  const uint32_t seq_end = e->seq;
  e->seq = pd.seq_general;
  const size_t recs_before = e->recs.size();
  if ((rc = node_general(e, pd.rows, pd.n, pd.nb, n_sparse, pd.now_ms))) return rc;
  // the general path's record belongs BEFORE the dense halves' fsm record (steps in order)
  if (e->recs.size() == recs_before + 1) {
    size_t at = recs_before;
    while (at > 0 && e->recs[at - 1].seq > pd.seq_general) at--;
    std::rotate(e->recs.begin() + at, e->recs.begin() + recs_before, e->recs.end());
  }
  uint64_t bytes_down = 0;
  e->seq = pd.seq_general;  // (the halves number themselves from here exactly as in the first pass)
  if ((rc = node_dense_halves(e, pd.now_ms, pd.flags, pd.col_mask, 2u, &bytes_down))) return rc;
  for (StepRec& rec : e->recs)
    if (rec.seq == pd.fsm_rec_seq && rec.fsm_per_row == JGN_FSM_ROWS && rec.msg_per_row == 0) {
      const uint32_t n_tiles = (e->cfg.n_groups + JG_SCAN_TILE - 1) / JG_SCAN_TILE;
      hipLaunchKernelGGL(k_node_fsm_build, dim3(n_tiles), dim3(JG_BLOCK), 0, e->stream, e->dev, nd.cols, rec.d_fsm, rec.d_fsm_cnt, rec.d_bsum_f);
      e->n_launch++;
    }
  HIPCHK(hipGetLastError());
  e->seq = seq_end;
  HIPCHK(hipStreamSynchronize(e->stream));
  return JG_OK;
}
}  // namespace

int jg_step_node(jg_engine* e, uint64_t now_ms, uint32_t flags) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (!(flags & (JG_NODE_LEADER_HALF | JG_NODE_FOLLOWER_HALF)) || (flags & ~15u))
    return fail(JG_EINVAL, "jg_step_node: flags = JG_NODE_LEADER_HALF and / or JG_NODE_FOLLOWER_HALF [| JG_NODE_TICK] [| JG_NODE_ASYNC]");
  if (e->router) return router_step_node(e, now_ms, flags);
  if (e->inflight.phase) return fail(JG_EINVAL, "a drain is in transfer: jg_drain_wait first");
  return node_step(e, now_ms, flags);
}

int jg_node_inbox_columns(jg_engine* e, uint32_t slot, uint64_t** answer, uint64_t** hb_commit) {
  if (!e || !answer) return fail(JG_EINVAL, "null argument");
  if (e->router) return fail(JG_EINVAL, "jg_node_inbox_columns: the columns are per shard: call this on a shard handle (jg_get_shard)");
  if (slot >= e->cfg.n_replicas) return fail(JG_EINVAL, "slot out of range");
  if (e->uniform_self >= 0 && (uint32_t)e->uniform_self == slot)
    return fail(JG_EINVAL, "jg_node_inbox_columns: the own slot's word carries the append count");
  HIPCHK(hipSetDevice(e->device));
  int rc = node_ensure(e);
  if (rc) return rc;
  jg_engine::NodeStep& nd = e->node;
  const size_t G = e->cfg.n_groups;
  if (nd.cols_in_flight) {  // the previous step's uploads out of these buffers (a step without rows never synchronises)
    HIPCHK(hipEventSynchronize(nd.ev_cols));
    nd.cols_in_flight = false;
  }
  *answer = nd.h_in_answers + (size_t)slot * G;
  nd.col_mask |= 1u << slot;
  if (hb_commit) {
    *hb_commit = nd.h_in_hbc + (size_t)slot * G;
    nd.col_hbc_mask |= 1u << slot;
  } else {
    nd.col_hbc_mask &= ~(1u << slot);
  }
  return JG_OK;
}

int jg_node_outbox_view(jg_engine* e, jg_node_outbox* out) {
  if (!e || !out) return fail(JG_EINVAL, "null argument");
  if (e->router) return router_node_outbox(e, out);
  jg_engine::NodeStep& nd = e->node;
  if (!nd.ready || !nd.last_flags) return fail(JG_EINVAL, "no jg_step_node yet");
  int rc = sync_and_check(e);  // (the columns have landed; device-side error flags surface here)
  if (rc) return rc;
  *out = nd.last;
  if ((nd.last_flags & JG_NODE_LEADER_HALF) && (nd.last_flags & JG_NODE_TICK)) out->beat = nd.h_beat, out->ae = nd.h_ae;
  if (nd.last_flags & JG_NODE_FOLLOWER_HALF) out->answer = nd.h_answer, out->hb_commit = nd.h_hbc;
  return JG_OK;
}

struct jg_dense_cluster {
  std::vector<jg_engine*> nodes;
  uint32_t G = 0, R = 0, lead = 0;
  uint32_t lead_id = 0;
  uint64_t *acks = nullptr, *hbr_commit = nullptr, *o_ae = nullptr;  // acks: the lead node's inbox answer words
  jg_leader_beat* o_beat = nullptr;
  std::vector<void*> bufs;
  // one protocol round captured as a hipGraph (ten launches and nine cross-stream dependencies per
  // round cost more host time than the round's kernels take on the device)
  JgClock* clock = nullptr;
  JgFollowerJob* d_jobs = nullptr;  // the follower halves of a replayed round as ONE launch (k_follower_tick_dense_multi)
  bool failed = false;  // a routed round failed after it had consumed the delivered rows: the in-flight votes are gone
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  // ... and `many_rounds` consecutive rounds as ONE graph (the clock advances itself: JgClock): a graph launch costs
  // a few microseconds of its own beside its nodes', shared by the rounds it holds
  hipGraph_t graph_many = nullptr;
  hipGraphExec_t exec_many = nullptr;
  uint32_t many_rounds = 0;
  uint64_t sig = 0, graph_dt = 0;
  uint64_t* offered = nullptr;  // [G] the ClientRequests per round as set by jg_dense_cluster_set_appends
  // per-partition leadership (lead == JG_CLUSTER_ANY_LEADER at creation): every node runs both halves over the
  // cluster's mailboxes; owner[g] = whose Tick the columns carry this round (k_cluster_claim)
  bool any = false;
  uint8_t* owner = nullptr;
  char *any_h_jobs = nullptr, *any_d_jobs = nullptr;  // the round's job tables (leader halves | follower halves): pinned / device
  hipEvent_t any_ev = nullptr;  // behind the upload of an eager round's tables: the pinned copy may be rewritten
  bool any_ev_pending = false;
  static constexpr size_t ANY_SLICE = 8192;
  // While clustered, nodes that share the lead node's device run on ITS stream: the halves of a round
  // are bandwidth-bound, so running them side by side buys nothing (each then takes 50-60 us instead
  // of 20), five streams do not fit four hardware queues (two follower halves ended up behind each
  // other anyway), and every cross-stream dependency is a host call.  The nodes' own streams are
  // restored when the cluster is destroyed.
  std::vector<hipStream_t> own_stream;
  // jg_dense_cluster_round_routed: per destination node, the staging the senders' rows are scattered
  // into, its sort scratch, and the command columns of the node's next round (all grow-only)
  struct Route {
    uint32_t* d_count = nullptr;  // [R][R+4] per sender: rows per destination + JG_ROUTE_*; then the JG_ROUTE_SEGS staging cursors; then [R] kept exceptional rows; then [R] the kinds delivered per destination
    std::vector<uint32_t> kinds_in;  // per node: the census of command kinds of the rows waiting for its next round (bit k: JG_CMD_* k)
    uint32_t* h_count = nullptr;  // pinned mirror
    // staging shared by all destinations, its sort scratch, the sorted command columns (node n's rows
    // are the slice [in_off[n], in_off[n] + n_in[n]) of every column); all grow-only
    uint64_t *key = nullptr, *key_alt = nullptr;
    uint32_t *idx = nullptr, *idx_alt = nullptr;
    jg_msg_row* row = nullptr;
    uint32_t cap = 0;
    JgRouteCols cols{};
    char* cols_mem = nullptr;
    std::vector<uint32_t> n_in, in_off;
    std::vector<JgXqRec*> xq_keep;  // per node, lazily: where the exceptional rows that stay are compacted
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    uint32_t* bk_hist = nullptr;  // bucket counts / offsets, scatter cursors, scan-tile bases (k_route_hist ... _sort_build)
    uint32_t bk_cap = 0;
    // job tables of the round's multi launches (one launch for all nodes / senders / steps): a pinned staging
    // the host fills and its device copy, in slices of JOB_SLICE bytes
    static constexpr size_t JOB_SLICE = 16384;
    char *h_jobs = nullptr, *d_jobs = nullptr;
    uint32_t group_bits = 1;
    bool ready = false;
    // JG_CLUSTER_OPT_VOTE_WORDS: the election vocabulary as mailbox words (jg_votes.h) - two rounds' mail, used in turn
    bool vote_words = false;
    JgVoteMail vm[2]{};
    void* vm_mem = nullptr;
    uint32_t vm_turn = 0;
    hipEvent_t ev_counts = nullptr;               // behind the delivering pass's counts on their way to the host
    uint32_t last_total = 0, last_fullest_seg = 0;  // the previous round's rows: what the ordering pass is sized for before the counts are in
  } rt;
};

int jg_dense_cluster_create(jg_engine* const* nodes, uint32_t n_nodes, uint32_t lead, jg_dense_cluster** out) {
  const bool any = lead == JG_CLUSTER_ANY_LEADER;
  if (any) lead = 0;  // (the node whose stream and device the cluster's work is issued on)
  if (!nodes || !out || !n_nodes || lead >= n_nodes) return fail(JG_EINVAL, "bad argument");
  for (uint32_t r = 0; r < n_nodes; r++) {
    if (!nodes[r] || nodes[r]->router) return fail(JG_EINVAL, "a dense cluster takes single-device engines (or shard handles)");
    if (nodes[r]->cfg.n_groups != nodes[0]->cfg.n_groups || nodes[r]->cfg.n_replicas != n_nodes)
      return fail(JG_EINVAL, "every node hosts the same groups, one replica slot each");
    if (any && (nodes[r]->device != nodes[0]->device || nodes[r]->uniform_self != (int)r))
      return fail(JG_EINVAL, "per-partition leadership: the nodes share a device and nodes[r] hosts replica slot r of every group");
  }
  if (any && n_nodes > JG_LEADER_MULTI) return fail(JG_EINVAL, "per-partition leadership: at most 6 nodes");
  jg_dense_cluster* c = new jg_dense_cluster();
  c->any = any;
  c->nodes.assign(nodes, nodes + n_nodes);
  c->G = nodes[0]->cfg.n_groups, c->R = n_nodes, c->lead = lead;
  c->lead_id = nodes[lead]->cfg.node_ids[lead];
  jg_engine* L = nodes[lead];
  const size_t G = c->G, R = c->R;
  auto alloc = [&](size_t bytes, void** p) {
    int rc = jg_device_alloc(L, bytes, p);
    if (!rc) c->bufs.push_back(*p);
    return rc;
  };
  int rc = JG_OK;
  if ((rc = alloc(8 * R * G, (void**)&c->acks)) || (rc = alloc(8 * R * G, (void**)&c->hbr_commit)) ||
      (rc = alloc(16 * G, (void**)&c->o_beat)) || (rc = alloc(8 * R * G, (void**)&c->o_ae)) ||
      (rc = alloc(8 * G, (void**)&c->offered))) {
    jg_dense_cluster_destroy(c);
    return rc;
  }
  std::vector<uint64_t> a(R * G, JG_NO_ACK);  // nothing from anybody ...
  // ... and the lead node's own slot carries the number of appends: zero, with no HeartbeatResponse
  // (JG_NO_ACK there is outside the own slot's domain: JG_FAULT_ENGINE_DENSE_APPENDS)
  for (size_t g = 0; g < G; g++) a[(size_t)lead * G + g] = JG_ANSWER(0, JG_HB_NONE);
  if (any) {  // (whoever owns a group reads its own slot's word from `offered`: every row of the inbox is a peer's)
    const size_t ob = (G + 15) & ~size_t(15);
    if ((rc = alloc(ob, (void**)&c->owner)) || hipMemsetAsync(c->owner, 0xff, ob, L->stream) != hipSuccess ||
        hipHostMalloc((void**)&c->any_h_jobs, 2 * jg_dense_cluster::ANY_SLICE, hipHostMallocDefault) != hipSuccess ||
        (rc = alloc(2 * jg_dense_cluster::ANY_SLICE, (void**)&c->any_d_jobs))) {
      jg_dense_cluster_destroy(c);
      return rc ? rc : fail(JG_EDEVICE, "per-partition leadership: allocation failed");
    }
  }
  if ((rc = jg_device_upload(L, c->offered, a.data() + (size_t)lead * G, G * 8))) {
    jg_dense_cluster_destroy(c);
    return rc;
  }
  if (any)
    for (size_t g = 0; g < G; g++) a[(size_t)lead * G + g] = JG_NO_ACK;
  if ((rc = jg_device_upload(L, c->acks, a.data(), a.size() * 8)) ||
      // (the lead node's own row of the AppendEntries block is never written by its kernel: JG_NO_ACK once)
      hipMemsetAsync(c->o_ae, 0xff, 8 * R * G, L->stream) != hipSuccess) {
    jg_dense_cluster_destroy(c);
    return rc;
  }
  static const bool own_streams = std::getenv("JG_CLUSTER_OWN_STREAMS") != nullptr;
  c->own_stream.assign(n_nodes, nullptr);
  for (uint32_t r = 0; r < n_nodes && !own_streams; r++) {
    jg_engine* e = nodes[r];
    if (e == L || e->device != L->device) continue;
    if ((rc = sync_and_check(e))) {  // nothing of its own is in flight when the stream changes hands
      jg_dense_cluster_destroy(c);
      return rc;
    }
    c->own_stream[r] = e->stream;
    e->own_stream = e->stream;
    e->stream = L->stream;
  }
  *out = c;
  return JG_OK;
}

void jg_dense_cluster_destroy(jg_dense_cluster* c) {
  if (!c) return;
  for (size_t r = 0; r < c->own_stream.size(); r++)
    if (c->own_stream[r]) {
      (void)hipStreamSynchronize(c->nodes[r]->stream);
      c->nodes[r]->stream = c->own_stream[r];
      c->nodes[r]->own_stream = nullptr;
    }
  if (c->exec) (void)hipGraphExecDestroy(c->exec);
  if (c->graph) (void)hipGraphDestroy(c->graph);
  if (c->exec_many) (void)hipGraphExecDestroy(c->exec_many);
  if (c->graph_many) (void)hipGraphDestroy(c->graph_many);
  if (c->any_h_jobs) (void)hipHostFree(c->any_h_jobs);
  if (c->any_ev) (void)hipEventDestroy(c->any_ev);
  if (c->rt.ev_counts) (void)hipEventDestroy(c->rt.ev_counts);
  for (void* p : c->bufs) (void)jg_device_free(c->nodes[c->lead], p);
  for (void* p : {(void*)c->rt.key, (void*)c->rt.key_alt, (void*)c->rt.idx, (void*)c->rt.idx_alt, (void*)c->rt.row, (void*)c->rt.cols_mem})
    if (p) (void)hipFree(p);
  for (JgXqRec* p : c->rt.xq_keep)
    if (p) (void)hipFree(p);
  if (c->rt.sort_tmp) (void)hipFree(c->rt.sort_tmp);
  if (c->rt.vm_mem) (void)hipFree(c->rt.vm_mem);
  if (c->rt.h_jobs) (void)hipHostFree(c->rt.h_jobs);
  if (c->rt.d_jobs) (void)hipFree(c->rt.d_jobs);
  if (c->rt.bk_hist) (void)hipFree(c->rt.bk_hist);
  if (c->rt.d_count) (void)hipFree(c->rt.d_count);
  if (c->rt.h_count) (void)hipHostFree(c->rt.h_count);
  delete c;
}

int jg_dense_cluster_set_option(jg_dense_cluster* c, uint32_t option, uint64_t value) {
  if (!c) return fail(JG_EINVAL, "null argument");
  switch (option) {
    case JG_CLUSTER_OPT_VOTE_WORDS:
      if (c->rt.ready) return fail(JG_EINVAL, "JG_CLUSTER_OPT_VOTE_WORDS is fixed before the cluster's first routed round");
      if (value > 1) return fail(JG_EINVAL, "JG_CLUSTER_OPT_VOTE_WORDS takes 0 or 1");
      c->rt.vote_words = value != 0;
      return JG_OK;
    default:
      return fail(JG_EINVAL, "unknown cluster option");
  }
}

int jg_dense_cluster_set_appends(jg_dense_cluster* c, uint64_t uniform, const uint64_t* per_group) {
  if (!c) return fail(JG_EINVAL, "null argument");
  std::vector<uint64_t> v(c->G);  // the own slot's answer words: JG_ANSWER(#appends, no HeartbeatResponse)
  for (uint32_t g = 0; g < c->G; g++) {
    const uint64_t n = per_group ? per_group[g] : uniform;
    v[g] = n < JG_MAILBOX_NONE ? JG_ANSWER(n, JG_HB_NONE) : JG_NO_ACK;  // (out of range stays out of range: JG_FAULT_ENGINE_DENSE_APPENDS)
  }
  const uint64_t* src = v.data();
  int rc = jg_device_upload(c->nodes[c->lead], c->offered, src, (size_t)c->G * 8);
  if (rc || c->any) return rc;  // (per-partition leadership: the kernels read `offered` itself)
  return jg_device_upload(c->nodes[c->lead], c->acks + (size_t)c->lead * c->G, src, (size_t)c->G * 8);
}

int jg_dense_cluster_offer_appends(jg_dense_cluster* c, const uint32_t* groups_dev, uint32_t n, uint64_t per_round) {
  if (!c || (n && !groups_dev)) return fail(JG_EINVAL, "null argument");
  if (per_round >= JG_MAILBOX_NONE) return fail(JG_EINVAL, "appends per round: out of the own slot's domain");
  if (!n) return JG_OK;
  jg_engine* L = c->nodes[c->lead];
  HIPCHK(hipSetDevice(L->device));
  hipLaunchKernelGGL(k_offer_appends, dim3((n + 255) / 256), dim3(256), 0, L->stream, n, groups_dev, c->G, per_round, c->offered,
                     c->any ? (uint64_t*)nullptr : c->acks + (size_t)c->lead * c->G);
  HIPCHK(hipGetLastError());
  return JG_OK;
}
int jg_dense_cluster_withdraw_appends(jg_dense_cluster* c, const uint32_t* groups_dev, uint32_t n) {
  return jg_dense_cluster_offer_appends(c, groups_dev, n, 0);
}

int jg_dense_cluster_mailboxes(jg_dense_cluster* c, jg_leader_inbox* in, jg_leader_outbox* out) {
  if (!c) return fail(JG_EINVAL, "null argument");
  if (in) *in = jg_leader_inbox{c->acks, c->hbr_commit};
  if (out) *out = jg_leader_outbox{c->o_beat, c->o_ae};
  return JG_OK;
}

namespace {
// the body of one round; `leading_waits`: the leader's stream first waits for the followers' last answers
// the follower job of node r as the replayed round's kernels see it
JgFollowerJob cluster_job(const jg_dense_cluster* c, uint32_t r) {
  const jg_engine* e = c->nodes[r];
  JgFollowerJob j{};
  j.d = e->dev;
  j.a.clock = c->clock, j.a.clock_slot = r;
  j.a.leader = nullptr, j.a.leader_id = c->lead_id;
  j.a.beat = c->o_beat, j.a.ae = c->o_ae + (size_t)r * c->G;
  j.a.o_answer = c->acks + (size_t)r * c->G, j.a.o_hbc = c->hbr_commit + (size_t)r * c->G;
  j.a.tick = 1;
  return j;
}

// the follower halves of an eager round as jobs (with this round's time and step numbers) + the host-side bookkeeping
// of the two launches that serve them
int cluster_follower_jobs(jg_dense_cluster* c, uint64_t now_ms, std::vector<JgFollowerJob>& jobs) {
  int rc = JG_OK;
  for (uint32_t r = 0; r < c->R; r++) {
    if (r == c->lead) continue;
    jg_engine* e = c->nodes[r];
    if ((rc = ensure_xq(e))) return rc;
    e->stepped = true;
    e->seq++;
    JgFollowerJob j = cluster_job(c, r);
    j.a.clock = nullptr, j.a.now = now_ms, j.a.seq = e->seq;
    jobs.push_back(j);
    e->slow_scheduled_ever = true;
    e->n_launch += 2;
    e->n_dense += e->cfg.n_groups;
    e->maybe_irregular = true, e->flag_check_pending = true, e->irr_gen++;
  }
  return JG_OK;
}
// `prepared`: the caller has the jobs already (cluster_follower_jobs) and their device copy at d_slice is on its way
int cluster_round_body(jg_dense_cluster* c, uint64_t now_ms, bool leading_waits, bool multi = false, char* h_slice = nullptr,
                       char* d_slice = nullptr, const std::vector<JgFollowerJob>* prepared = nullptr) {
  jg_engine* L = c->nodes[c->lead];
  const size_t G = c->G;
  const jg_leader_inbox in{c->acks, c->hbr_commit};
  const jg_leader_outbox out{c->o_beat, c->o_ae};
  int rc = JG_OK;
  if (leading_waits)
    for (uint32_t r = 0; r < c->R; r++)
      if (r != c->lead && (rc = jg_stream_wait(L, c->nodes[r]))) return rc;
  if ((rc = jg_step_dense_leader(L, now_ms, &in, &out))) return rc;
  if (multi) {  // (a captured round whose nodes share the lead node's stream) every follower half in ONE launch
    hipLaunchKernelGGL(k_follower_tick_dense_multi, dim3(L->dense_grid, c->R - 1), dim3(JG_BLOCK), 0, L->stream, (const JgFollowerJob*)c->d_jobs);
    {
      JgFollowerJobs kj{};
      uint32_t k = 0;
      for (uint32_t r = 0; r < c->R; r++)
        if (r != c->lead) kj.j[k++] = cluster_job(c, r);
      hipLaunchKernelGGL(k_follower_slow_multi, dim3(JG_SHARDS, c->R - 1), dim3(JG_BLOCK), 0, L->stream, kj);
    }
    HIPCHK(hipGetLastError());
    return JG_OK;  // (the host-side bookkeeping of a replayed round is done per graph launch)
  }
  if (h_slice) {  // an eager round whose nodes share the lead node's stream: the same two launches, jobs with this round's time
    std::vector<JgFollowerJob> own;
    if (!prepared && (rc = cluster_follower_jobs(c, now_ms, own))) return rc;
    const std::vector<JgFollowerJob>& jobs = prepared ? *prepared : own;
    if (!jobs.empty()) {
      if (!prepared) {
        std::memcpy(h_slice, jobs.data(), jobs.size() * sizeof(JgFollowerJob));
        HIPCHK(hipMemcpyAsync(d_slice, h_slice, jobs.size() * sizeof(JgFollowerJob), hipMemcpyHostToDevice, L->stream));
      }
      hipLaunchKernelGGL(k_follower_tick_dense_multi, dim3(L->dense_grid, (uint32_t)jobs.size()), dim3(JG_BLOCK), 0, L->stream, (const JgFollowerJob*)d_slice);
      JgFollowerJobs kj{};
      for (size_t k = 0; k < jobs.size(); k++) kj.j[k] = jobs[k];
      hipLaunchKernelGGL(k_follower_slow_multi, dim3(JG_SHARDS, (uint32_t)jobs.size()), dim3(JG_BLOCK), 0, L->stream, kj);
      HIPCHK(hipGetLastError());
    }
    return JG_OK;
  }
  for (uint32_t r = 0; r < c->R; r++) {
    if (r == c->lead) continue;
    if ((rc = jg_stream_wait(c->nodes[r], L))) return rc;
    jg_follower_inbox fi{};
    fi.leader = nullptr, fi.leader_id = c->lead_id;
    fi.beat = c->o_beat, fi.ae = c->o_ae + (size_t)r * G;
    const jg_follower_outbox fo{c->acks + (size_t)r * G, c->hbr_commit + (size_t)r * G};
    if ((rc = jg_step_dense_follower(c->nodes[r], now_ms, &fi, &fo, 1))) return rc;
  }
  return JG_OK;
}

// ---- a round with per-partition leadership (JG_CLUSTER_ANY_LEADER) --------------------------------
// Five launches on the cluster's stream: k_cluster_claim (who owns each group's columns this round), the leader
// halves of all nodes (k_leader_node_tick_any, blockIdx.y = node), their slow kernels (k_dense_slow_multi), the
// follower halves of all nodes (k_follower_tick_dense_any), their slow kernels (k_follower_slow_multi).  Every node
// takes TWO steps per round (leader half, follower half).  `replay`: the round is being captured - time and step
// numbers come from the device-resident clock, the host-side bookkeeping is done per graph launch.
int cluster_tables_any(jg_dense_cluster* c, uint64_t now_ms, bool replay) {
  jg_engine* L = c->nodes[c->lead];
  const uint32_t R = c->R;
  const size_t G = c->G;
  int rc = JG_OK;
  JgLeaderJob* lj = (JgLeaderJob*)c->any_h_jobs;
  JgFollowerJob* fj = (JgFollowerJob*)(c->any_h_jobs + jg_dense_cluster::ANY_SLICE);
  static_assert(JG_LEADER_MULTI * sizeof(JgLeaderJob) <= jg_dense_cluster::ANY_SLICE, "job slice too small");
  static_assert(JG_LEADER_MULTI * sizeof(JgFollowerJob) <= jg_dense_cluster::ANY_SLICE, "job slice too small");
  if (!c->any_ev) HIPCHK(hipEventCreateWithFlags(&c->any_ev, hipEventDisableTiming));
  if (c->any_ev_pending) {  // (the previous eager round's upload out of the same pinned tables)
    HIPCHK(hipEventSynchronize(c->any_ev));
    c->any_ev_pending = false;
  }
  for (uint32_t r = 0; r < R; r++) {
    jg_engine* e = c->nodes[r];
    if (!replay) {
      if ((rc = ensure_xq(e))) return rc;
      e->stepped = true;
      e->seq += 2;  // leader half: seq - 1, follower half: seq
      e->slow_scheduled_ever = true;
      e->n_launch += 4;
      e->n_dense += 2 * G;
      e->maybe_irregular = true, e->flag_check_pending = true, e->irr_gen++;
    }
    JgLeaderNode nd{};
    nd.clock = replay ? c->clock : nullptr, nd.clock_slot = r;
    nd.ack_stride = 1, nd.packed = 1;
    nd.hbr_commit = c->hbr_commit;
    nd.o_beat = c->o_beat, nd.o_ae = c->o_ae;
    nd.now = now_ms;
    nd.owner = c->owner, nd.offered = c->offered;
    JgLeaderJob& j = lj[r];
    j.h = jg_dense_hot_of(e->dev), j.dp = e->d_dev, j.acks = c->acks, j.seq = e->seq - 1, j.us = (int)r, j.nd = nd;
    JgFollowerJob& f = fj[r];
    f = JgFollowerJob{};
    f.d = e->dev;
    f.a.clock = replay ? c->clock : nullptr, f.a.clock_slot = r, f.a.seq_off = 1;
    f.a.leader = nullptr, f.a.leader_id = 0;
    f.a.beat = c->o_beat, f.a.ae = c->o_ae + (size_t)r * G;
    f.a.o_answer = c->acks + (size_t)r * G, f.a.o_hbc = c->hbr_commit + (size_t)r * G;
    f.a.now = now_ms, f.a.seq = e->seq, f.a.tick = 1;
    f.a.owner = c->owner, f.a.self_slot = r;
  }
  // (replay: the tables are written once, outside the capture; an eager round's carry its time and step numbers)
  if (replay) HIPCHK(hipMemcpy(c->any_d_jobs, c->any_h_jobs, 2 * jg_dense_cluster::ANY_SLICE, hipMemcpyHostToDevice));
  else {
    HIPCHK(hipMemcpyAsync(c->any_d_jobs, c->any_h_jobs, 2 * jg_dense_cluster::ANY_SLICE, hipMemcpyHostToDevice, L->stream));
    HIPCHK(hipEventRecord(c->any_ev, L->stream));
    c->any_ev_pending = true;
  }
  return JG_OK;
}
// (the launches, separately: a capture writes its tables before hipStreamBeginCapture)
int cluster_launch_any(jg_dense_cluster* c) {
  jg_engine* L = c->nodes[c->lead];
  const uint32_t R = c->R;
  hipStream_t st = L->stream;
  JgClaimArgs ca{};
  ca.R = R, ca.G = c->G, ca.owner = c->owner;
  for (uint32_t r = 0; r < R; r++) ca.flags[r] = c->nodes[r]->dev.flags;
  hipLaunchKernelGGL(k_cluster_claim, dim3(grid_for((c->G + 3) / 4, 2048)), dim3(JG_BLOCK), 0, st, ca);
  const JgLeaderJob* lj = (const JgLeaderJob*)c->any_d_jobs;
#define JG_LAUNCH_ANY(RR) hipLaunchKernelGGL((k_leader_node_tick_any<RR>), dim3(L->dense_grid, R), dim3(JG_BLOCK), 0, st, lj)
  switch (R) {
    case 1: JG_LAUNCH_ANY(1); break;
    case 2: JG_LAUNCH_ANY(2); break;
    case 3: JG_LAUNCH_ANY(3); break;
    case 4: JG_LAUNCH_ANY(4); break;
    case 5: JG_LAUNCH_ANY(5); break;
    default: JG_LAUNCH_ANY(6); break;
  }
#undef JG_LAUNCH_ANY
  // the slow kernels' jobs as kernel arguments: rebuilt from the tables the fast kernels read
  JgLeaderSlowJobs sj{};
  JgFollowerJobs fsj{};
  const JgLeaderJob* hl = (const JgLeaderJob*)c->any_h_jobs;
  const JgFollowerJob* hf = (const JgFollowerJob*)(c->any_h_jobs + jg_dense_cluster::ANY_SLICE);
  for (uint32_t r = 0; r < R; r++) {
    sj.j[r].d = c->nodes[r]->dev, sj.j[r].acks = hl[r].acks, sj.j[r].seq0 = hl[r].seq, sj.j[r].nd = hl[r].nd;
    fsj.j[r] = hf[r];
  }
  hipLaunchKernelGGL(k_dense_slow_multi, dim3(JG_SHARDS, R), dim3(JG_BLOCK), 0, st, sj);
  hipLaunchKernelGGL(k_follower_tick_dense_any, dim3(L->dense_grid, R), dim3(JG_BLOCK), 0, st,
                     (const JgFollowerJob*)(c->any_d_jobs + jg_dense_cluster::ANY_SLICE));
  hipLaunchKernelGGL(k_follower_slow_multi, dim3(JG_SHARDS, R), dim3(JG_BLOCK), 0, st, fsj);
  HIPCHK(hipGetLastError());
  return JG_OK;
}

// what a captured round depends on besides the mailboxes: recapture when any of it changes
uint64_t cluster_signature(const jg_dense_cluster* c, uint64_t dt) {
  uint64_t h = 0x9e3779b97f4a7c15ull ^ dt;
  for (const jg_engine* e : c->nodes) {
    h = h * 0x100000001b3ull ^ (uint64_t)e->cur_set;
    h = h * 0x100000001b3ull ^ (uint64_t)(uintptr_t)e->dev.xq;
    h = h * 0x100000001b3ull ^ (uint64_t)e->kt_on;
  }
  return h;
}

int cluster_capture(jg_dense_cluster* c, uint64_t dt_ms, uint32_t rounds = 1) {
  jg_engine* L = c->nodes[c->lead];
  hipGraph_t& graph = rounds > 1 ? c->graph_many : c->graph;
  hipGraphExec_t& exec = rounds > 1 ? c->exec_many : c->exec;
  if (exec) (void)hipGraphExecDestroy(exec), exec = nullptr;
  if (graph) (void)hipGraphDestroy(graph), graph = nullptr;
  struct Saved {
    uint32_t seq;
    uint64_t n_dense, n_launch;
  };
  std::vector<Saved> saved;
  for (uint32_t r = 0; r < c->R; r++) {
    jg_engine* e = c->nodes[r];
    saved.push_back(Saved{e->seq, e->n_dense, e->n_launch});
    e->replay_clock = c->clock, e->replay_slot = r;
  }
  int rc = JG_OK;
  // all nodes on the lead node's stream (the default while clustered): the R - 1 follower halves are one launch,
  // their slow kernels another; the jobs are written here, outside the capture
  bool multi = c->R > 1 && !c->any;
  for (jg_engine* e : c->nodes) multi = multi && e->stream == L->stream;
  static const bool no_multi = std::getenv("JG_CLUSTER_SEPARATE_HALVES") != nullptr;  // (A/B: one launch per follower half, as in round 2)
  if (no_multi) multi = false;
  if (multi) {
    if (!c->d_jobs) {
      HIPCHK(hipMalloc((void**)&c->d_jobs, (size_t)JG_MAX_REPLICAS * sizeof(JgFollowerJob)));
      c->bufs.push_back(c->d_jobs);
    }
    std::vector<JgFollowerJob> jobs;
    for (uint32_t r = 0; r < c->R; r++)
      if (r != c->lead) jobs.push_back(cluster_job(c, r));
    HIPCHK(hipMemcpy(c->d_jobs, jobs.data(), jobs.size() * sizeof(JgFollowerJob), hipMemcpyHostToDevice));
  }
  if (c->any && (rc = cluster_tables_any(c, 0, true))) return rc;  // (the tables, before the capture begins)
  hipError_t he = hipStreamBeginCapture(L->stream, hipStreamCaptureModeRelaxed);
  if (he == hipSuccess) {
    for (uint32_t k = 0; k < rounds && !rc; k++) {
      rc = c->any ? cluster_launch_any(c) : cluster_round_body(c, 0, false, multi);
      for (uint32_t r = 0; r < c->R && !rc; r++)  // every forked stream joins the leader's again
        if (r != c->lead) rc = jg_stream_wait(L, c->nodes[r]);
    }
    he = hipStreamEndCapture(L->stream, &graph);
  }
  for (uint32_t r = 0; r < c->R; r++) {  // nothing has run: the host-side bookkeeping of the captured calls is undone
    jg_engine* e = c->nodes[r];
    e->replay_clock = nullptr;
    e->seq = saved[r].seq, e->n_dense = saved[r].n_dense, e->n_launch = saved[r].n_launch;
  }
  if (rc) return rc;
  if (he != hipSuccess) return fail(JG_EDEVICE, std::string("hipGraph capture: ") + hipGetErrorString(he));
  HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  if (rounds > 1) c->many_rounds = rounds;
  else c->sig = cluster_signature(c, dt_ms);
  return JG_OK;
}
}  // namespace

int jg_dense_cluster_rounds(jg_dense_cluster* c, uint64_t now_ms, uint64_t dt_ms, uint32_t n_rounds) {
  if (!c) return fail(JG_EINVAL, "null argument");
  if (!n_rounds) return JG_OK;
  jg_engine* L = c->nodes[c->lead];
  int rc = JG_OK;
  static const bool no_graph = std::getenv("JG_NO_GRAPH") != nullptr;
  bool same_device = true;
  for (jg_engine* e : c->nodes) same_device = same_device && e->device == L->device;
  if (c->any && (no_graph || n_rounds < 2)) {  // per-partition leadership, eager (the nodes share a device and a stream)
    HIPCHK(hipSetDevice(L->device));
    for (jg_engine* e : c->nodes)
      if (!e->p_kind.empty()) return fail(JG_EINVAL, "commands are queued: call jg_step first");
    for (uint32_t k = 0; k < n_rounds; k++, now_ms += dt_ms)
      if ((rc = cluster_tables_any(c, now_ms, false)) || (rc = cluster_launch_any(c))) return rc;
    return JG_OK;
  }
  if (no_graph || !same_device || n_rounds < 2) {  // eager: one round at a time
    for (uint32_t k = 0; k < n_rounds; k++, now_ms += dt_ms)
      if ((rc = cluster_round_body(c, now_ms, true))) return rc;
    for (uint32_t r = 0; r < c->R; r++)  // the leader's stream ends behind the last answers
      if (r != c->lead && (rc = jg_stream_wait(L, c->nodes[r]))) return rc;
    return JG_OK;
  }
  HIPCHK(hipSetDevice(L->device));
  for (jg_engine* e : c->nodes) {  // nothing may allocate or synchronise inside a capture
    if (!e->p_kind.empty()) return fail(JG_EINVAL, "commands are queued: call jg_step first");
    if ((rc = ensure_xq(e))) return rc;
  }
  if (!c->clock) {
    HIPCHK(hipMalloc((void**)&c->clock, sizeof(JgClock)));
    c->bufs.push_back(c->clock);
  }
  // (JG_CLUSTER_ROUNDS_PER_GRAPH: how many rounds the long graph holds - 1: none, the A/B; kernel timing brackets single launches)
  static const uint32_t per_graph = std::getenv("JG_CLUSTER_ROUNDS_PER_GRAPH") ? (uint32_t)std::atoi(std::getenv("JG_CLUSTER_ROUNDS_PER_GRAPH")) : 8u;
  if (!c->exec || c->sig != cluster_signature(c, dt_ms)) {
    if (c->exec_many) (void)hipGraphExecDestroy(c->exec_many), c->exec_many = nullptr;  // (captured against the same state: both again)
    if ((rc = cluster_capture(c, dt_ms))) return rc;
  }
  bool timing = false;
  for (jg_engine* e : c->nodes) timing = timing || e->kt_on;
  const uint32_t many = (per_graph > 1 && !timing && n_rounds >= per_graph) ? per_graph : 0u;
  if (many && (!c->exec_many || c->many_rounds != many))
    if ((rc = cluster_capture(c, dt_ms, many))) return rc;
  for (uint32_t r = 0; r < c->R; r++)  // the followers' earlier work first
    if (r != c->lead && (rc = jg_stream_wait(L, c->nodes[r]))) return rc;
  JgClock init{};  // the first replayed round's time and step numbers; the rounds advance it themselves (JgClock)
  init.dt = dt_ms, init.n_nodes = c->R, init.seq_step = c->any ? 2 : 1;
  init.v[0].now = now_ms;
  for (uint32_t r = 0; r < c->R; r++) init.v[0].seq[r] = c->nodes[r]->seq + 1;
  hipLaunchKernelGGL(k_clock_set, dim3(1), dim3(1), 0, L->stream, c->clock, init);
  for (uint32_t k = 0; k < n_rounds;) {
    const uint32_t held = (many && n_rounds - k >= many) ? many : 1u;
    HIPCHK(hipGraphLaunch(held > 1 ? c->exec_many : c->exec, L->stream));
    k += held;
    for (uint32_t r = 0; r < c->R; r++) {  // what the eager calls would have recorded on the host
      jg_engine* e = c->nodes[r];
      e->seq += (c->any ? 2 : 1) * held;
      e->stepped = true;
      e->n_dense += (c->any ? 2 * (uint64_t)c->G : c->G) * held;
      e->n_launch += (c->any ? 4 : 2) * held;
      e->slow_scheduled_ever = true;
      if (c->any || r != c->lead) e->maybe_irregular = true, e->flag_check_pending = true, e->irr_gen += held;
    }
  }
  HIPCHK(hipGetLastError());
  for (uint32_t r = 0; r < c->R; r++)  // later work on the followers' own streams comes behind the replayed rounds
    if (r != c->lead && (rc = jg_stream_wait(c->nodes[r], L))) return rc;
  return JG_OK;
}

namespace {
constexpr uint32_t ROUTE_WORDS = JG_MAX_REPLICAS + 4;  // per sender: rows per destination, kept, fsm rows, overflow, kept exceptional rows

int route_grow(jg_dense_cluster::Route& d, size_t need) {
  if (need <= d.cap) return JG_OK;
  for (void* p : {(void*)d.key, (void*)d.key_alt, (void*)d.idx, (void*)d.idx_alt, (void*)d.row, (void*)d.cols_mem})
    if (p) HIPCHK(hipFree(p));
  const size_t cap = (std::max<size_t>(need + need / 2, 65536) + 63) & ~size_t(63);  // (whole segments: jg_route_reserve)
  if (cap > 0x7fffffffull) return fail(JG_ECAPACITY, "routed round: too many rows");
  HIPCHK(hipMalloc((void**)&d.key, cap * 8));
  HIPCHK(hipMalloc((void**)&d.key_alt, cap * 8));
  HIPCHK(hipMalloc((void**)&d.idx, cap * 4));
  HIPCHK(hipMalloc((void**)&d.idx_alt, cap * 4));
  HIPCHK(hipMalloc((void**)&d.row, cap * sizeof(jg_msg_row)));
  HIPCHK(hipMalloc((void**)&d.cols_mem, cap * 34));  // 3 x 8 + 2 x 4 + 2 x 1 bytes per row, widest columns first
  char* m = d.cols_mem;
  d.cols.term = (uint64_t*)m, m += cap * 8;
  d.cols.id = (uint64_t*)m, m += cap * 8;
  d.cols.aux = (uint64_t*)m, m += cap * 8;
  d.cols.group = (uint32_t*)m, m += cap * 4;
  d.cols.from = (uint32_t*)m, m += cap * 4;
  d.cols.kind = (uint8_t*)m, m += cap;
  d.cols.flag = (uint8_t*)m;
  d.cap = (uint32_t)cap;
  return JG_OK;
}
}  // namespace

static int round_routed_impl(jg_dense_cluster* c, uint64_t now_ms, const jg_cmd_batch* inject, jg_route_stats* stats, bool* started);
int jg_dense_cluster_round_routed(jg_dense_cluster* c, uint64_t now_ms, const jg_cmd_batch* inject, jg_route_stats* stats) {
  if (!c) return fail(JG_EINVAL, "null argument");
  // Everything that can be checked is checked before the first launch; an error AFTER the round has begun to
  // consume the rows the transport delivered (a HIP failure, an internal inconsistency) leaves in-flight messages
  // lost or half-routed: the cluster says so from then on instead of carrying on quietly.
  if (c->failed) return fail(JG_EDEVICE, "jg_dense_cluster_round_routed: an earlier routed round failed half-way: destroy the cluster");
  bool started = false;
  const int rc = round_routed_impl(c, now_ms, inject, stats, &started);
  if (rc && started) c->failed = true;
  return rc;
}
static int round_routed_impl(jg_dense_cluster* c, uint64_t now_ms, const jg_cmd_batch* inject, jg_route_stats* stats, bool* started) {
  jg_engine* L = c->nodes[c->lead];
  const uint32_t R = c->R;
  int rc = JG_OK;
  for (jg_engine* e : c->nodes) {
    if (e->device != L->device) return fail(JG_EINVAL, "routed rounds take nodes that share a device");
    if (!e->p_kind.empty()) return fail(JG_EINVAL, "commands are queued: call jg_step first");
    if (e->inflight.phase) return fail(JG_EINVAL, "a drain is in transfer: jg_drain_flush first");
    if ((rc = ensure_xq(e))) return rc;
  }
  for (uint32_t n = 0; inject && n < R; n++) {  // (checked before anything is launched)
    if (!inject[n].n) continue;
    if (inject[n].n_blocks) return fail(JG_EINVAL, "injected rows cannot carry blocks");
    if (inject[n].n > 0x7fffffffull) return fail(JG_EINVAL, "batch too large: split it");
    if (!inject[n].kind || !inject[n].group || !inject[n].from || !inject[n].term || !inject[n].id || !inject[n].aux || !inject[n].flag)
      return fail(JG_EINVAL, "all seven device columns are required");
  }
  HIPCHK(hipSetDevice(L->device));
  static const bool trace = std::getenv("JG_TRACE_ROUTE") != nullptr;
  auto clk = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double T0 = clk();
  double T1 = T0, T2 = T0, T3 = T0, T4 = T0;
  jg_dense_cluster::Route& rt = c->rt;
  const size_t words = (size_t)R * ROUTE_WORDS + JG_ROUTE_SEGS + 2 * R;
  if (!rt.ready) {
    HIPCHK(hipMalloc((void**)&rt.d_count, words * 4));
    HIPCHK(hipHostMalloc((void**)&rt.h_count, words * 4, hipHostMallocDefault));
    rt.n_in.assign(R, 0), rt.in_off.assign(R, 0), rt.kinds_in.assign(R, 0);
    rt.xq_keep.assign(R, nullptr);
    while (rt.group_bits < 32 && (c->G - 1) >> rt.group_bits) rt.group_bits++;
    if (rt.group_bits > 29) return fail(JG_EINVAL, "routed rounds: too many groups for the transport's ordering key");
    // room for a round's worth of rows everywhere, allocated once: 2 rows per partition and node with their output
    // regions (this is a 288 GB device: 640 B per partition and node is 3.2 GB at 5 x 1 M)
    if ((rc = route_grow(rt, (size_t)c->G * 2))) return rc;
    for (jg_engine* e : c->nodes)
      if (e->recs.empty()) HIPCHK(e->arenas[e->cur_arena].reserve((size_t)c->G * 640));
    rt.ready = true;
  }
  // -- 1. what the transport delivered last round, then this round's injected rows (per group: in that order).
  // Nodes that share the lead node's stream take each of the two in ONE launch (k_apply_rows_multi).
  bool one_stream = true;
  for (jg_engine* e : c->nodes) one_stream = one_stream && e->stream == L->stream;
  static const bool no_multi = std::getenv("JG_ROUTE_SEPARATE_LAUNCHES") != nullptr;  // (A/B: round 2's launch per node / sender / step)
  const bool multi = one_stream && !no_multi;
  // JG_CLUSTER_OPT_VOTE_WORDS (jg_dense_cluster_set_option; jg_votes.h): an election's traffic travels as mailbox words - the campaigns' broadcasts are
  // counted into request words by a census of the emitted rows and are not staged, the answers are written as words by the
  // receiving half (k_vote_half_multi) and never become rows - wherever EVERYTHING a node receives for a partition in a
  // round is such words; every other partition's mail travels as rows, as without the switch.  Fixed for a cluster's life.
  const bool vwords = rt.vote_words && multi && R >= 2 && std::getenv("JG_ROUTE_LIBRARY_SORT") == nullptr;
  if (vwords && !rt.vm_mem) {
    const size_t RG = (size_t)R * c->G, wd = ((size_t)c->G + 63) / 64;
    const size_t per = RG * (8 + 8 + 8 + 4 + 4) + 2 * R * wd * 8;
    HIPCHK(hipMalloc(&rt.vm_mem, 2 * per));
    HIPCHK(hipMemsetAsync(rt.vm_mem, 0, 2 * per, L->stream));  // (the first round reads the mail of a round that never was: none)
    char* p = (char*)rt.vm_mem;
    for (int k = 0; k < 2; k++) {
      JgVoteMail& m = rt.vm[k];
      m.R = R, m.G = c->G, m.words = (uint32_t)wd;
      m.q_term = (uint64_t*)p, p += RG * 8;
      m.q_head = (uint64_t*)p, p += RG * 8;
      m.a_term = (uint64_t*)p, p += RG * 8;
      m.q_ctl = (uint32_t*)p, p += RG * 4;
      m.a_ctl = (uint32_t*)p, p += RG * 4;
      m.rowmail = (uint64_t*)p, p += R * wd * 8;
      m.wordmail = (uint64_t*)p, p += R * wd * 8;
    }
  }
  const JgVoteMail vprev = rt.vm[rt.vm_turn ^ 1u], vcur = rt.vm[rt.vm_turn];  // (last round's mail is read, this round's filled)
  if (vwords) {
    hipLaunchKernelGGL(k_votes_clear, dim3((vcur.words + JG_VOTE_CHUNK - 1) / JG_VOTE_CHUNK), dim3(JG_BLOCK), 0, L->stream, vcur);  // (a workgroup per chunk of the bitmaps)
    HIPCHK(hipGetLastError());
  }
  if (!rt.h_jobs) {
    HIPCHK(hipHostMalloc((void**)&rt.h_jobs, 6 * jg_dense_cluster::Route::JOB_SLICE, hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&rt.d_jobs, 6 * jg_dense_cluster::Route::JOB_SLICE));
  }
  static_assert(JG_MAX_REPLICAS * sizeof(JgApplyJob) <= jg_dense_cluster::Route::JOB_SLICE, "job slice too small");
  static_assert(JG_MAX_REPLICAS * sizeof(JgFollowerJob) <= jg_dense_cluster::Route::JOB_SLICE, "job slice too small");
  static_assert(sizeof(JgVoteHalfJobs) + 2 * sizeof(JgVoteMail) <= 4096, "kernel arguments");
  auto slice_h = [&](int k) { return rt.h_jobs + (size_t)k * jg_dense_cluster::Route::JOB_SLICE; };
  auto slice_d = [&](int k) { return rt.d_jobs + (size_t)k * jg_dense_cluster::Route::JOB_SLICE; };
  // (multi: the job tables of the whole round - both sparse steps, the follower halves, the delivering pass - are
  // written first and travel in ONE copy: every table is host bookkeeping only, and a copy costs ~10 us of stream time)
  static const bool row_per_lane = std::getenv("JG_ROUTE_ROW_PER_LANE") != nullptr;  // (A/B: the row-per-lane kernels for the delivered rows)
  static const bool big_tiles_only = std::getenv("JG_ROUTE_BIG_TILES") != nullptr;  // (A/B: 1024-row tiles whatever the batch)
  auto small_tiles = [&](uint32_t widest) { return widest <= JG_RUN_SMALL_BATCH && !big_tiles_only; };
  auto run_grid = [&](uint32_t widest) {
    const uint32_t tile = small_tiles(widest) ? JG_RUN_TILE_SMALL : JG_RUN_TILE;
    return std::min<uint32_t>(std::max<uint32_t>((widest + tile - 1) / tile, 1u), L->count_slots);
  };
  auto apply_all = [&](int slice, std::vector<JgApplyJob>& jobs, uint32_t widest) -> int {
    if (jobs.empty()) return JG_OK;
    if (!multi) {
      for (size_t k = 0; k < jobs.size(); k++)
        hipLaunchKernelGGL(k_apply_rows, dim3(grid_for(jobs[k].a.n, L->count_slots)), dim3(JG_BLOCK), 0, L->stream, jobs[k].d, jobs[k].a);
    } else if (slice == 0 && !row_per_lane) {  // delivered rows: runs of 4-16 rows per group, a RUN per lane (jg_apply_runs_body)
      hipLaunchKernelGGL(small_tiles(widest) ? k_apply_runs_multi_small : k_apply_runs_multi, dim3(run_grid(widest), (uint32_t)jobs.size()),
                         dim3(JG_BLOCK), 0, L->stream, (const JgApplyJob*)slice_d(slice));
    } else {
      hipLaunchKernelGGL(k_apply_rows_multi, dim3(grid_for(widest, L->count_slots), (uint32_t)jobs.size()), dim3(JG_BLOCK), 0, L->stream,
                         (const JgApplyJob*)slice_d(slice));
    }
    HIPCHK(hipGetLastError());
    return JG_OK;
  };
  std::vector<uint32_t> seq_base(R);
  for (uint32_t n = 0; n < R; n++) seq_base[n] = c->nodes[n]->seq;
  *started = true;
  // (jobs_v: delivered batches that hold an election's traffic only - the transport's census says so - take the
  // kernel without the chain code: k_apply_votes_multi.  JG_ROUTE_NO_VOTES_KERNEL=1: the general one for all, an A/B)
  static const bool no_votes_kernel = std::getenv("JG_ROUTE_NO_VOTES_KERNEL") != nullptr;
  std::vector<JgApplyJob> jobs_a, jobs_v, jobs_b;
  uint32_t widest_a = 0, widest_v = 0, widest_b = 0;
  {
    for (uint32_t n = 0; n < R; n++) {
      const bool votes = multi && !no_votes_kernel && jg_kinds_within(rt.kinds_in[n], JG_KINDS_ELECTION);
      std::vector<JgApplyJob>& jobs = votes ? jobs_v : jobs_a;
      uint32_t& widest = votes ? widest_v : widest_a;
      jg_engine* e = c->nodes[n];
      if (!rt.n_in[n]) {
        if (vwords) e->stepped = true, e->seq++;  // (the receiving half of the vote mail is this step too: it has a number on every node)
        continue;
      }
      const size_t o = rt.in_off[n];
      e->stepped = true;
      e->seq++;
      JgApplyJob j{};
      j.d = e->dev;
      // VoteRequest -> one VoteResponse; VoteResponse -> DROP + Heartbeat on elect() (candidate.rs:108-113): two slots per row
      const bool two = votes && jg_kinds_within(rt.kinds_in[n], (1u << JG_CMD_VOTE_REQUEST) | (1u << JG_CMD_VOTE_RESPONSE));
      if ((rc = prepare_rows(e, rt.n_in[n], rt.cols.group + o, rt.cols.kind + o, rt.cols.from + o, rt.cols.term + o, rt.cols.id + o,
                             rt.cols.aux + o, rt.cols.flag + o, nullptr, nullptr, 0, now_ms, &j.a, two ? 2u : 0u)))
        return rc;
      if (e->stream != L->stream) {  // (its own stream: its own launch)
        hipLaunchKernelGGL(k_apply_rows, dim3(grid_for(j.a.n, e->count_slots)), dim3(JG_BLOCK), 0, e->stream, j.d, j.a);
      } else {
        widest = std::max(widest, j.a.n);
        jobs.push_back(j);
      }
      rt.n_in[n] = 0;
    }
  }
  {
    std::vector<JgApplyJob>& jobs = jobs_b;
    uint32_t& widest = widest_b;
    for (uint32_t n = 0; inject && n < R; n++) {
      jg_engine* e = c->nodes[n];
      const jg_cmd_batch& b = inject[n];
      if (!b.n) continue;
      e->stepped = true;
      e->seq++;
      JgApplyJob j{};
      j.d = e->dev;
      const uint64_t* none = (const uint64_t*)e->d_ones;  // (injected rows carry no blocks: checked above)
      if ((rc = prepare_rows(e, (uint32_t)b.n, b.group, b.kind, b.from, b.term, b.id, b.aux, b.flag, none, none, 0, now_ms, &j.a))) return rc;
      if (e->stream != L->stream) {
        hipLaunchKernelGGL(k_apply_rows, dim3(grid_for(j.a.n, e->count_slots)), dim3(JG_BLOCK), 0, e->stream, j.d, j.a);
      } else {
        widest = std::max(widest, j.a.n);
        jobs.push_back(j);
      }
    }
  }
  std::vector<JgFollowerJob> fjobs;
  if (c->any) {
    if (!one_stream) return fail(JG_EINVAL, "per-partition leadership: the nodes share the cluster's stream");
    if ((rc = cluster_tables_any(c, now_ms, false))) return rc;  // (its own copy, behind the sparse steps' tables)
  } else if (multi && (rc = cluster_follower_jobs(c, now_ms, fjobs))) return rc;
  hipStream_t st = L->stream;
  uint32_t* d_cursor = rt.d_count + (size_t)R * ROUTE_WORDS;
  uint32_t* d_keep_n = d_cursor + JG_ROUTE_SEGS;
  // the staging in segments, one cursor each (jg_route_reserve); the library sort of the A/B wants it in one piece
  static const bool library_sort = std::getenv("JG_ROUTE_LIBRARY_SORT") != nullptr;
  static const bool one_cursor = library_sort || std::getenv("JG_ROUTE_ONE_CURSOR") != nullptr;
  const uint32_t n_seg = one_cursor ? 1u : JG_ROUTE_SEGS;
  // (JG_ROUTE_NARROW_BITS: test hook - a field too narrow for the trace exercises the repeat with the wide one)
  static const uint32_t narrow = std::getenv("JG_ROUTE_NARROW_BITS") ? (uint32_t)std::atoi(std::getenv("JG_ROUTE_NARROW_BITS")) : JG_ROUTE_ORD_BITS_FAST;
  uint32_t ord_bits = std::min<uint32_t>(std::max<uint32_t>(narrow, 1u), JG_ROUTE_ORD_BITS);
  auto table = [&](uint32_t s) {
    JgRouteTable t{};
    t.R = R, t.src = s;
    for (uint32_t n = 0; n < R; n++) t.member_id[n] = c->nodes[n]->cfg.node_ids[n];
    t.src_id = t.member_id[s];
    t.group_bits = rt.group_bits, t.ord_bits = ord_bits, t.cap = rt.cap;
    t.seg_cap = rt.cap / n_seg, t.seg_mask = n_seg - 1;
    t.key = rt.key, t.idx = rt.idx, t.row = rt.row;
    t.cursor = d_cursor;
    t.count = rt.d_count + (size_t)s * ROUTE_WORDS;
    t.kinds = d_keep_n + R;
    return t;
  };
  for (uint32_t s = 0; s < R; s++)
    for (const StepRec& r : c->nodes[s]->recs)
      if (r.seq > seq_base[s] && r.seq - seq_base[s] > 7)
        return fail(JG_ECAPACITY, "routed round: more steps than the transport's ordering key numbers");
  const uint32_t* h_cursor = rt.h_count + (size_t)R * ROUTE_WORDS;
  // the bucket pass's counters (their size does not depend on the round's rows): cleared with the tallies, in one launch
  JgRouteBuckets bk{};
  const uint32_t tile_bits = std::min<uint32_t>(JG_ROUTE_TILE_BITS, rt.group_bits);
  bk.n_buckets = R << (rt.group_bits - tile_bits);
  const uint32_t n_tiles = (bk.n_buckets + JG_ROUTE_SCAN_TILE - 1) / JG_ROUTE_SCAN_TILE;
  {
    const size_t bk_words = (size_t)n_tiles * JG_ROUTE_SCAN_TILE + bk.n_buckets + n_tiles + 1;  // hist (whole tiles) | cur | tile
    if (rt.bk_cap < bk_words) {
      if (rt.bk_hist) HIPCHK(hipFree(rt.bk_hist));
      rt.bk_cap = (uint32_t)bk_words;
      HIPCHK(hipMalloc((void**)&rt.bk_hist, bk_words * 4));
    }
    bk.hist = rt.bk_hist, bk.cur = bk.hist + (size_t)n_tiles * JG_ROUTE_SCAN_TILE, bk.tile = bk.cur + bk.n_buckets;
  }
  const uint32_t bk_clear = n_tiles * JG_ROUTE_SCAN_TILE + bk.n_buckets;  // counts and cursors
  // the delivering pass's jobs: every (sender, step) in one launch, every sender's exceptional-row queue in another
  std::vector<JgRouteRecJob> rjobs;
  std::vector<JgRouteXqJob> xjobs;
  uint32_t widest_r = 0;
  size_t rb = 0, xb = 0;
  auto route_jobs = [&]() -> int {  // (again on a repeated attempt: the table carries the staging's size and the key layout)
    rjobs.clear(), xjobs.clear(), widest_r = 0;
    for (uint32_t s = 0; s < R; s++) {
      jg_engine* e = c->nodes[s];
      const JgRouteTable t = table(s);
      for (const StepRec& r : e->recs)
        if (r.seq > seq_base[s] && r.d_msg) {
          JgRouteRecJob j{};
          j.t = t, j.n = r.n, j.per_row = r.msg_per_row, j.step = r.seq - seq_base[s];
          j.msg_cnt = r.d_msg_cnt, j.msg = r.d_msg, j.fsm_cnt = r.d_fsm_cnt;
          rjobs.push_back(j);
          widest_r = std::max(widest_r, r.n);
        }
      JgRouteXqJob j{};
      j.t = t, j.xq = e->dev.xq, j.xq_n = e->dev.xq_n, j.xq_cap = e->dev.xq_cap, j.seq_base = seq_base[s];
      xjobs.push_back(j);
    }
    rb = rjobs.size() * sizeof(JgRouteRecJob), xb = xjobs.size() * sizeof(JgRouteXqJob);
    if (rb + xb > jg_dense_cluster::Route::JOB_SLICE) return fail(JG_ECAPACITY, "routed round: too many undrained steps for the transport's job table");
    std::memcpy(slice_h(3), rjobs.data(), rb);
    std::memcpy(slice_h(3) + rb, xjobs.data(), xb);
    return JG_OK;
  };
  if ((rc = route_jobs())) return rc;
  JgVoteHalfJobs vjobs{};  // the vote mail's receiving half on every node (the delivered step's number): kernel arguments
  if (vwords) {
    for (uint32_t n = 0; n < R; n++) {
      jg_engine* e = c->nodes[n];
      JgVoteHalfJob& j = vjobs.j[n];
      j.d = e->dev, j.self = n, j.seq = seq_base[n] + 1u, j.step = 1u, j.need = R - 1u, j.now = now_ms;
      if (e->seq < j.seq) return fail(JG_EDEVICE, "internal: routed round: the delivered step has no number");
    }
  }
  if (multi) {  // slices 0-3 in one copy
    if (!jobs_a.empty()) std::memcpy(slice_h(0), jobs_a.data(), jobs_a.size() * sizeof(JgApplyJob));
    if (!jobs_v.empty()) std::memcpy(slice_h(0) + jobs_a.size() * sizeof(JgApplyJob), jobs_v.data(), jobs_v.size() * sizeof(JgApplyJob));
    if (!jobs_b.empty()) std::memcpy(slice_h(1), jobs_b.data(), jobs_b.size() * sizeof(JgApplyJob));
    if (!fjobs.empty()) std::memcpy(slice_h(2), fjobs.data(), fjobs.size() * sizeof(JgFollowerJob));
    HIPCHK(hipMemcpyAsync(slice_d(0), slice_h(0), 4 * jg_dense_cluster::Route::JOB_SLICE, hipMemcpyHostToDevice, st));
  }
  // -- 1. (launches) what the transport delivered last round, then this round's injected rows
  if ((rc = apply_all(0, jobs_a, widest_a))) return rc;
  if (!jobs_v.empty()) {  // (different nodes than jobs_a's: the two launches are independent of each other)
    if (row_per_lane)
      hipLaunchKernelGGL(k_apply_votes_multi, dim3(grid_for(widest_v, L->count_slots), (uint32_t)jobs_v.size()), dim3(JG_BLOCK), 0, L->stream,
                         (const JgApplyJob*)slice_d(0) + jobs_a.size());
    else
      hipLaunchKernelGGL(small_tiles(widest_v) ? k_apply_vote_runs_multi_small : k_apply_vote_runs_multi,
                         dim3(run_grid(widest_v), (uint32_t)jobs_v.size()), dim3(JG_BLOCK), 0, L->stream,
                         (const JgApplyJob*)slice_d(0) + jobs_a.size());
    HIPCHK(hipGetLastError());
  }
  if (vwords) {  // (partitions other than the rows': a partition's mail of a round is words or rows, never both)
    uint32_t slots = L->count_slots;
    for (jg_engine* e : c->nodes) slots = std::min(slots, e->count_slots);
    const uint32_t n_chunks = (vprev.words + JG_VOTE_CHUNK - 1) / JG_VOTE_CHUNK;  // (a workgroup per chunk of the bitmap; its counter slot is blockIdx.x)
    hipLaunchKernelGGL(k_vote_half_multi, dim3(std::max(1u, std::min(n_chunks, slots)), R), dim3(JG_BLOCK), 0, L->stream, vjobs, vprev, vcur);
    HIPCHK(hipGetLastError());
  }
  if ((rc = apply_all(1, jobs_b, widest_b))) return rc;
  // -- 2. the dense round; ClientRequests only where the lead node (still) leads
  if (c->any) {  // (whoever owns a group reads `offered`: nothing to mask)
    if ((rc = cluster_launch_any(c))) return rc;
  } else {
    hipLaunchKernelGGL(k_route_mask_appends, dim3((c->G + 255) / 256), dim3(256), 0, L->stream, c->G, (const uint32_t*)L->dev.flags,
                       (const uint64_t*)c->offered, c->acks + (size_t)c->lead * c->G);
    if ((rc = cluster_round_body(c, now_ms, true, false, multi ? slice_h(2) : nullptr, multi ? slice_d(2) : nullptr, multi ? &fjobs : nullptr))) return rc;
  }
  T1 = clk();
  // -- 3. the transport, on the lead node's stream behind everybody's round
  for (uint32_t r = 0; r < R; r++)
    if (r != c->lead && (rc = jg_stream_wait(L, c->nodes[r]))) return rc;
  // The ordering pass (bucket by (destination, group tile), sort every bucket in LDS: jg_route.h) takes everything it
  // needs to know about the round's rows from the device - the segments' cursors - so it is launched BEHIND the
  // delivering pass before the host has seen the counts: the host then waits for the counts' copy only (an event),
  // with the ordering still queued, and does its bookkeeping and the next round's preparation while the device
  // works.  (Waiting first left the device idle for the wake-up and the five launches: JG_ROUTE_SYNC_FIRST=1, the A/B.)
  // A pass that has to be repeated (staging too small, emission index too wide) repeats the ordering with it.
  static const bool sync_first = std::getenv("JG_ROUTE_SYNC_FIRST") != nullptr;
  const bool optimistic = multi && !library_sort && !sync_first && rt.last_total != 0;
  if (!rt.ev_counts) HIPCHK(hipEventCreateWithFlags(&rt.ev_counts, hipEventDisableTiming));
  auto launch_order = [&](uint32_t fullest_seg) {
    bk.shift = ord_bits + 3 + JG_ROUTE_STEP_BITS + tile_bits;  // (its counters were cleared with the tallies, before the delivering pass)
    const uint32_t seg_cap = rt.cap / n_seg;
    const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>((std::min(fullest_seg, seg_cap) + JG_BLOCK - 1) / JG_BLOCK, 4096 / n_seg));
    hipLaunchKernelGGL(k_route_hist, dim3(grid, n_seg), dim3(JG_BLOCK), 0, st, (const uint32_t*)d_cursor, seg_cap, (const uint64_t*)rt.key, bk);
    hipLaunchKernelGGL(k_route_scan, dim3(n_tiles), dim3(JG_BLOCK), 0, st, bk);
    hipLaunchKernelGGL(k_route_scan_tiles, dim3(1), dim3(JG_BLOCK), 0, st, bk);
    hipLaunchKernelGGL(k_route_scatter, dim3(grid, n_seg), dim3(JG_BLOCK), 0, st, (const uint32_t*)d_cursor, seg_cap, (const uint64_t*)rt.key,
                       (const uint32_t*)rt.idx, bk, rt.key_alt, rt.idx_alt);
    hipLaunchKernelGGL(k_route_sort_build, dim3(bk.n_buckets), dim3(JG_BLOCK), 0, st, bk, rt.key_alt, rt.idx_alt, (const jg_msg_row*)rt.row,
                       rt.cols);
  };
  bool ordered = false;
  if (vwords) {  // the census of everything the round emitted (once: a repeated delivering pass finds it done)
    if (!rjobs.empty())
      hipLaunchKernelGGL(k_votes_census_rec_multi, dim3((widest_r + JG_BLOCK - 1) / JG_BLOCK, (uint32_t)rjobs.size()), dim3(JG_BLOCK), 0, st,
                         (const JgRouteRecJob*)slice_d(3), vcur);
    hipLaunchKernelGGL(k_votes_census_xq_multi, dim3(256, (uint32_t)xjobs.size()), dim3(JG_BLOCK), 0, st, (const JgRouteXqJob*)(slice_d(3) + rb), vcur);
    hipLaunchKernelGGL(k_votes_validate, dim3((vcur.words + JG_VOTE_CHUNK - 1) / JG_VOTE_CHUNK), dim3(JG_BLOCK), 0, st, vcur, R - 1u);
    HIPCHK(hipGetLastError());
  }
  for (int attempt = 0;; attempt++) {  // (repeated once when the staging turns out too small: the pass modifies nothing)
    hipLaunchKernelGGL(k_route_clear, dim3(64), dim3(JG_BLOCK), 0, st, rt.d_count, (uint32_t)words, bk.hist, bk_clear);
    if (attempt) {  // (every attempt ends with a synchronisation - the counts - so the staging is free again)
      if ((rc = route_jobs())) return rc;
      if (multi) HIPCHK(hipMemcpyAsync(slice_d(3), slice_h(3), rb + xb, hipMemcpyHostToDevice, st));
    }
    if (vwords) {  // (implies multi) the delivering pass leaves the words' copies where they are; the answer words that must be rows after all
      if (!rjobs.empty())
        hipLaunchKernelGGL(k_route_rec_multi_words, dim3((widest_r + JG_BLOCK * JG_ROUTE_ITEMS - 1) / (JG_BLOCK * JG_ROUTE_ITEMS), (uint32_t)rjobs.size()),
                           dim3(JG_BLOCK), 0, st, (const JgRouteRecJob*)slice_d(3), vcur);
      hipLaunchKernelGGL(k_route_xq_multi_words, dim3(256, (uint32_t)xjobs.size()), dim3(JG_BLOCK), 0, st, (const JgRouteXqJob*)(slice_d(3) + rb), vcur);
      hipLaunchKernelGGL(k_votes_expand_multi, dim3(std::min<uint32_t>((vcur.words + JG_VOTE_CHUNK - 1) / JG_VOTE_CHUNK, 512u), (uint32_t)xjobs.size()), dim3(JG_BLOCK), 0, st,
                         (const JgRouteXqJob*)(slice_d(3) + rb), vcur);
    } else if (multi) {
      if (!rjobs.empty())
        hipLaunchKernelGGL(k_route_rec_multi, dim3((widest_r + JG_BLOCK * JG_ROUTE_ITEMS - 1) / (JG_BLOCK * JG_ROUTE_ITEMS), (uint32_t)rjobs.size()), dim3(JG_BLOCK), 0, st,
                           (const JgRouteRecJob*)slice_d(3));
      // (256 workgroups per queue: a round's queue holds a few ten thousand rows, and every workgroup - busy or not - pays the tally)
      hipLaunchKernelGGL(k_route_xq_multi, dim3(256, (uint32_t)xjobs.size()), dim3(JG_BLOCK), 0, st, (const JgRouteXqJob*)(slice_d(3) + rb));
    } else {
      for (const JgRouteRecJob& j : rjobs)
        hipLaunchKernelGGL(k_route_rec, dim3((j.n + JG_BLOCK * JG_ROUTE_ITEMS - 1) / (JG_BLOCK * JG_ROUTE_ITEMS)), dim3(JG_BLOCK), 0, st, j.t, j.n, j.per_row,
                           j.step, j.msg_cnt, j.msg, j.fsm_cnt);
      for (const JgRouteXqJob& j : xjobs)
        hipLaunchKernelGGL(k_route_xq<false>, dim3(1024), dim3(JG_BLOCK), 0, st, j.t, j.xq, j.xq_n, j.xq_cap, j.seq_base, (JgXqRec*)nullptr,
                           (uint32_t*)nullptr);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(rt.h_count, rt.d_count, words * 4, hipMemcpyDeviceToHost, st));
    ordered = false;
    if (optimistic) {
      HIPCHK(hipEventRecord(rt.ev_counts, st));
      launch_order(2 * rt.last_fullest_seg + JG_BLOCK);  // (a round's rows come in the numbers the last round's did; the kernels stride)
      HIPCHK(hipGetLastError());
      ordered = true;
    }
    T2 = clk();
    if (optimistic) HIPCHK(hipEventSynchronize(rt.ev_counts));
    else HIPCHK(hipStreamSynchronize(st));
    T3 = clk();
    bool wide = false;  // some group emitted more rows in one step than the narrow index field numbers
    for (uint32_t s = 0; s < R; s++) wide = wide || rt.h_count[(size_t)s * ROUTE_WORDS + R + JG_ROUTE_OVERFLOW];
    uint64_t fullest = 0;  // (a segment that ran over: every segment gets that much room, and the pass is repeated)
    for (uint32_t k = 0; k < n_seg; k++) fullest = std::max<uint64_t>(fullest, h_cursor[k]);
    const bool fits = fullest <= rt.cap / n_seg;
    if (fits && !wide) break;
    if (attempt >= 2) return fail(JG_EDEVICE, "internal: routed round: the delivering pass does not settle");
    if (wide) {
      if (ord_bits == JG_ROUTE_ORD_BITS) return fail(JG_ECAPACITY, "routed round: a group emitted too many rows in one step");
      ord_bits = JG_ROUTE_ORD_BITS;
    }
    if (!fits && (rc = route_grow(rt, fullest * n_seg))) return rc;
  }
  uint32_t total = 0, fullest_seg = 0;
  for (uint32_t k = 0; k < n_seg; k++) total += h_cursor[k], fullest_seg = std::max(fullest_seg, h_cursor[k]);
  std::vector<uint64_t> to(R, 0), from(R, 0);
  uint64_t kept = 0, fsm = 0;
  for (uint32_t s = 0; s < R; s++) {
    const uint32_t* h = rt.h_count + (size_t)s * ROUTE_WORDS;
    for (uint32_t n = 0; n < R; n++) to[n] += h[n], from[s] += h[n];
    kept += h[R + JG_ROUTE_KEPT] + h[R + JG_ROUTE_KEPT_XQ];
    fsm += h[R + JG_ROUTE_FSM];
  }
  // senders that keep rows for the host: the delivered ones leave their slots / the exceptional queue
  JgWordList emptied{};  // exceptional-row queues that were delivered whole: their counts go to zero in one launch
  for (uint32_t s = 0; s < R; s++) {
    jg_engine* e = c->nodes[s];
    const uint32_t* h = rt.h_count + (size_t)s * ROUTE_WORDS;
    if (!from[s] && !vwords) continue;  // (a sender whose only mail was words' copies: they leave its queue too)
    const JgRouteTable t = table(s);
    if (h[R + JG_ROUTE_KEPT] || h[R + JG_ROUTE_FSM])  // (its steps stay queued for a drain: without the delivered rows)
      for (const StepRec& r : e->recs) {
        if (r.seq <= seq_base[s]) continue;
        hipLaunchKernelGGL(k_route_rec_compact, dim3((r.n + JG_BLOCK - 1) / JG_BLOCK), dim3(JG_BLOCK), 0, st, t, r.n, r.msg_per_row,
                           r.d_msg_cnt, r.d_msg);
        hipLaunchKernelGGL(k_count_block_sums, dim3((r.n + JG_SCAN_TILE - 1) / JG_SCAN_TILE), dim3(JG_BLOCK), 0, st, r.d_msg_cnt,
                           r.d_fsm_cnt, r.n, r.d_bsum_m, r.d_bsum_f);
      }
    const uint32_t kx = h[R + JG_ROUTE_KEPT_XQ];
    if (kx) {
      if (!rt.xq_keep[s]) HIPCHK(hipMalloc((void**)&rt.xq_keep[s], (size_t)e->dev.xq_cap * sizeof(JgXqRec)));
      hipLaunchKernelGGL(k_route_xq<true>, dim3(256), dim3(JG_BLOCK), 0, st, t, (const JgXqRec*)e->dev.xq, (const uint32_t*)e->dev.xq_n,
                         e->dev.xq_cap, seq_base[s], rt.xq_keep[s], d_keep_n + s);
      HIPCHK(hipMemcpyAsync(e->dev.xq, rt.xq_keep[s], (size_t)kx * sizeof(JgXqRec), hipMemcpyDeviceToDevice, st));
      HIPCHK(hipMemcpyAsync(e->dev.xq_n, d_keep_n + s, 4, hipMemcpyDeviceToDevice, st));
    } else {
      emptied.p[emptied.n++] = e->dev.xq_n;
    }
  }
  if (emptied.n) hipLaunchKernelGGL(k_route_clear_words, dim3(1), dim3(64), 0, st, emptied);
  // the staged rows in (destination, group, sender, step, emission) order -> the command columns of every
  // node's next round: bucket by (destination, group tile), sort every bucket in LDS (jg_route.h); the
  // library sort stays behind JG_ROUTE_LIBRARY_SORT=1 for an A/B
  rt.last_total = total, rt.last_fullest_seg = fullest_seg;
  if (total && !library_sort) {
    if (!ordered) launch_order(fullest_seg);
  } else if (total) {
    const uint32_t end_bit = ord_bits + 3 + JG_ROUTE_STEP_BITS + rt.group_bits + 3;
    size_t need = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, need, rt.key, rt.key_alt, rt.idx, rt.idx_alt, (size_t)total, 0, end_bit, st));
    if (rt.sort_tmp_bytes < need) {
      if (rt.sort_tmp) HIPCHK(hipFree(rt.sort_tmp));
      rt.sort_tmp_bytes = 2 * need;
      HIPCHK(hipMalloc(&rt.sort_tmp, rt.sort_tmp_bytes));
    }
    size_t bytes = rt.sort_tmp_bytes;
    HIPCHK(rocprim::radix_sort_pairs(rt.sort_tmp, bytes, rt.key, rt.key_alt, rt.idx, rt.idx_alt, (size_t)total, 0, end_bit, st));
    hipLaunchKernelGGL(k_route_build, dim3((total + JG_BLOCK - 1) / JG_BLOCK), dim3(JG_BLOCK), 0, st, total,
                       (const uint32_t*)rt.idx_alt, (const jg_msg_row*)rt.row, rt.cols);
  }
  uint32_t off = 0;
  for (uint32_t n = 0; n < R; n++) {
    rt.kinds_in[n] = rt.h_count[(size_t)R * ROUTE_WORDS + JG_ROUTE_SEGS + R + n];
    rt.in_off[n] = off, rt.n_in[n] = (uint32_t)to[n];
    off += (uint32_t)to[n];
  }
  if (off != total) return fail(JG_EDEVICE, "internal: routed round: row counts disagree");
  HIPCHK(hipGetLastError());
  for (uint32_t r = 0; r < R; r++)  // the nodes' next steps come behind the transport
    if (r != c->lead && (rc = jg_stream_wait(c->nodes[r], L))) return rc;
  // a round whose sparse steps left nothing for the host needs no drain: its output regions are released here
  for (uint32_t s = 0; s < R; s++) {
    jg_engine* e = c->nodes[s];
    const uint32_t* h = rt.h_count + (size_t)s * ROUTE_WORDS;
    if (h[R + JG_ROUTE_KEPT] || h[R + JG_ROUTE_FSM]) continue;
    while (!e->recs.empty() && e->recs.back().seq > seq_base[s]) e->recs.pop_back();
    if (e->recs.empty()) {
      // (nothing of this round reads those regions any more: the delivering pass has completed, and
      // no compaction pass was launched for this sender)
      e->arenas[e->cur_arena].reset();
    }
  }
  if (vwords) rt.vm_turn ^= 1u;
  T4 = clk();
  if (trace)
    std::fprintf(stderr, "[jg route] steps+round issued %.0f us, delivering pass issued %.0f us, wait %.0f us, sort+build issued %.0f us (%u rows)\n",
                 T1 - T0, T2 - T1, T3 - T2, T4 - T3, total);
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    for (uint32_t n = 0; n < R; n++) stats->delivered[n] = to[n];
    stats->kept = kept;
    stats->fsm_rows = fsm;
  }
  return JG_OK;
}

int jg_chain_compact(jg_engine* e, size_t n_trees, const uint64_t* off, const uint64_t* ids, const uint64_t* nexts,
                     const uint64_t* commits, uint8_t* removed) {
  if (!e || !off || !commits) return fail(JG_EINVAL, "null argument");
  if (!n_trees) return JG_OK;
  if (e->router) e = e->router->sh[0];  // a pure function: any shard's device will do
  HIPCHK(hipSetDevice(e->device));
  const size_t n = off[n_trees];
  if (n && (!ids || !nexts || !removed)) return fail(JG_EINVAL, "null argument");
  uint64_t *d_off = nullptr, *d_ids = nullptr, *d_next = nullptr, *d_commit = nullptr;
  uint8_t* d_rem = nullptr;
  HIPCHK(hipMalloc((void**)&d_off, (n_trees + 1) * 8));
  HIPCHK(hipMalloc((void**)&d_ids, std::max<size_t>(n * 8, 16)));
  HIPCHK(hipMalloc((void**)&d_next, std::max<size_t>(n * 8, 16)));
  HIPCHK(hipMalloc((void**)&d_commit, n_trees * 8));
  HIPCHK(hipMalloc((void**)&d_rem, std::max<size_t>(n, 16)));
  HIPCHK(hipMemcpyAsync(d_off, off, (n_trees + 1) * 8, hipMemcpyHostToDevice, e->stream));
  if (n) {
    HIPCHK(hipMemcpyAsync(d_ids, ids, n * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d_next, nexts, n * 8, hipMemcpyHostToDevice, e->stream));
  }
  HIPCHK(hipMemcpyAsync(d_commit, commits, n_trees * 8, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemsetAsync(d_rem, 0, std::max<size_t>(n, 16), e->stream));
  hipLaunchKernelGGL(k_chain_compact, dim3(grid_for(n_trees, 2048)), dim3(JG_BLOCK), 0, e->stream, n_trees, d_off,
                     d_ids, d_next, d_commit, d_rem);
  HIPCHK(hipGetLastError());
  e->n_launch++;
  if (n) HIPCHK(hipMemcpyAsync(removed, d_rem, n, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipFree(d_off));
  HIPCHK(hipFree(d_ids));
  HIPCHK(hipFree(d_next));
  HIPCHK(hipFree(d_commit));
  HIPCHK(hipFree(d_rem));
  return JG_OK;
}

int jg_chain_compact_resident(jg_engine* e, size_t* n_removed) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (n_removed) *n_removed = 0;
  if (e->router) {  // shard by shard; rows rebased to the parent's group numbers
    for (size_t d = 0; d < e->router->D(); d++) {
      jg_engine* s = e->router->sh[d];
      size_t n = 0;
      const int rc = jg_chain_compact_resident(s, &n);
      if (rc) return rc;
      for (jg_compact_row r : s->q_compacted) {
        r.group += e->router->lo[d];
        e->q_compacted.push_back(r);
      }
      s->q_compacted.clear();
      if (n_removed) *n_removed += n;
    }
    return JG_OK;
  }
  if (!e->p_kind.empty()) return fail(JG_EINVAL, "commands are queued: call jg_step first");
  HIPCHK(hipSetDevice(e->device));
  if (!e->d_compact) {
    // one pass removes at most one block per segment: (JG_CHAIN_WINDOW + 1) rows per group
    e->compact_cap = (uint32_t)std::min<size_t>((size_t)(JG_CHAIN_WINDOW + 1) * e->cfg.n_groups, 0x7fffffffu);
    HIPCHK(hipMalloc((void**)&e->d_compact, (size_t)e->compact_cap * sizeof(JgCompactRow)));
    HIPCHK(hipMalloc((void**)&e->d_compact_n, 16));
    e->allocs.push_back(e->d_compact);
    e->allocs.push_back(e->d_compact_n);
  }
  e->stepped = true;
  e->seq++;
  HIPCHK(hipMemsetAsync(e->d_compact_n, 0, sizeof(uint32_t), e->stream));
  hipLaunchKernelGGL(k_compact_resident, dim3(grid_for(e->cfg.n_groups, 4096)), dim3(JG_BLOCK), 0, e->stream, e->dev,
                     e->d_compact, e->d_compact_n, e->compact_cap, e->seq);
  HIPCHK(hipGetLastError());
  e->n_launch++;
  e->maybe_irregular = true;  // a leader's run may have lost its top: like a sparse step
  e->flag_check_pending = true;
  e->irr_gen++;
  uint32_t n = 0;
  HIPCHK(hipMemcpyAsync(&n, e->d_compact_n, sizeof n, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (n > e->compact_cap) return fail(JG_ECAPACITY, "more blocks removed than the compaction list holds (the chains ARE compacted)");
  if (n) {  // order the rows on the device: group ascending, then the order of the walk (ids descending)
    uint64_t *k0 = nullptr, *k1 = nullptr, *v0 = nullptr, *v1 = nullptr;
    jg_compact_row *d_rows = nullptr, *h_rows = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    HIPCHK(hipMalloc((void**)&k0, (size_t)n * 8));
    HIPCHK(hipMalloc((void**)&k1, (size_t)n * 8));
    HIPCHK(hipMalloc((void**)&v0, (size_t)n * 8));
    HIPCHK(hipMalloc((void**)&v1, (size_t)n * 8));
    HIPCHK(hipMalloc((void**)&d_rows, (size_t)n * sizeof(jg_compact_row)));
    HIPCHK(hipHostMalloc((void**)&h_rows, (size_t)n * sizeof(jg_compact_row), hipHostMallocDefault));
    HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0, 40, e->stream));
    HIPCHK(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)));
    const uint32_t grid = grid_for(n, 4096);
    hipLaunchKernelGGL(k_compact_split, dim3(grid), dim3(JG_BLOCK), 0, e->stream, (const JgCompactRow*)e->d_compact, n, k0, v0);
    HIPCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0, 40, e->stream));
    hipLaunchKernelGGL(k_compact_join, dim3(grid), dim3(JG_BLOCK), 0, e->stream, (const uint64_t*)k1, (const uint64_t*)v1, n, d_rows);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_rows, d_rows, (size_t)n * sizeof(jg_compact_row), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->q_compacted.insert(e->q_compacted.end(), h_rows, h_rows + n);
    for (void* p : {(void*)k0, (void*)k1, (void*)v0, (void*)v1, (void*)d_rows, tmp}) HIPCHK(hipFree(p));
    HIPCHK(hipHostFree(h_rows));
  }
  if (n_removed) *n_removed = n;
  return JG_OK;
}

int jg_drain_compacted(jg_engine* e, jg_compact_row* out, size_t cap, size_t* n) {
  if (!e || !n) return fail(JG_EINVAL, "null argument");
  *n = e->q_compacted.size();
  if (!out) return JG_OK;
  if (cap < *n) return fail(JG_ECAPACITY, "output buffer too small");
  if (*n) std::memcpy(out, e->q_compacted.data(), *n * sizeof(jg_compact_row));
  e->q_compacted.clear();
  return JG_OK;
}

int jg_sync(jg_engine* e) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) return e->router->run([&](size_t d) { return sync_and_check(e->router->sh[d]); });
  return sync_and_check(e);
}

int jg_stream_wait(jg_engine* waiter, jg_engine* signal) {
  if (!waiter || !signal) return fail(JG_EINVAL, "null argument");
  if (waiter == signal) return JG_OK;
  if (waiter->router || signal->router) {  // shard by shard (same ownership on both sides)
    if (!waiter->router || !signal->router || waiter->router->lo != signal->router->lo)
      return fail(JG_EINVAL, "jg_stream_wait: the two engines are sharded differently");
    for (size_t d = 0; d < waiter->router->D(); d++) {
      const int rc = jg_stream_wait(waiter->router->sh[d], signal->router->sh[d]);
      if (rc) return rc;
    }
    return JG_OK;
  }
  if (waiter->stream == signal->stream) return JG_OK;  // (nodes of a jg_dense_cluster share a stream: already in order)
  HIPCHK(hipSetDevice(signal->device));
  HIPCHK(hipEventRecord(signal->ev_order, signal->stream));
  HIPCHK(hipSetDevice(waiter->device));
  HIPCHK(hipStreamWaitEvent(waiter->stream, signal->ev_order, 0));
  return JG_OK;
}

int jg_drain_messages(jg_engine* e, jg_msg_row* out, size_t cap, size_t* n) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) return router_drain(e, e->router->msgs, 1, out, cap, n);
  return drain(e, e->q_msgs, 1, out, cap, n);
}
int jg_drain_applies(jg_engine* e, jg_fsm_row* out, size_t cap, size_t* n) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) return router_drain(e, e->router->fsm, 2, out, cap, n);
  return drain(e, e->q_fsm, 2, out, cap, n);
}
int jg_drain_messages_view(jg_engine* e, const jg_msg_row** rows, size_t* n) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) return router_drain_view(e, e->router->msgs, e->router->msgs_view, 1, rows, n);
  return drain_view(e, e->q_msgs, 1, rows, n);
}
int jg_drain_applies_view(jg_engine* e, const jg_fsm_row** rows, size_t* n) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) return router_drain_view(e, e->router->fsm, e->router->fsm_view, 2, rows, n);
  return drain_view(e, e->q_fsm, 2, rows, n);
}
int jg_drain_prefetch(jg_engine* e) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) {  // all shards or none: their batches must cover the same steps
    e->pipelined = true;
    for (jg_engine* s : e->router->sh) s->pipelined = true;
    if (!router_all_landed(e)) return JG_OK;
    return e->router->run([&](size_t d) { return drain_prefetch(e->router->sh[d], true); });
  }
  return drain_prefetch(e, false);
}

int jg_drain_wait(jg_engine* e) {
  if (!e) return fail(JG_EINVAL, "null argument");
  auto wait = [](jg_engine* s) {
    if (!s->inflight.phase) return (int)JG_OK;
    jg_engine::DrainThread& t = *s->drain_thread;
    std::unique_lock<std::mutex> lk(t.m);
    t.cv.wait(lk, [&] { return t.state == 2; });
    return (int)JG_OK;
  };
  if (e->router) {
    for (jg_engine* s : e->router->sh) wait(s);
    return JG_OK;
  }
  return wait(e);
}

int jg_drain_flush(jg_engine* e) {
  if (!e) return fail(JG_EINVAL, "null argument");
  auto flush = [](jg_engine* s) {
    int rc = drain_prefetch(s, true);   // the batch in transfer lands; whatever was stepped since starts
    if (rc) return rc;
    return inflight_finish(s);          // ... and lands too
  };
  if (e->router) {
    e->pipelined = true;
    for (jg_engine* s : e->router->sh) s->pipelined = true;
    return e->router->run([&](size_t d) { return flush(e->router->sh[d]); });
  }
  return flush(e);
}

int jg_drain_faults(jg_engine* e, jg_fault_row* out, size_t cap, size_t* n) {
  if (!e || !n) return fail(JG_EINVAL, "null argument");
  if (!e->router && e->pipelined && !inflight_landed(e)) {
    *n = 0;
    return JG_OK;
  }
  int rc = e->router ? router_collect(e, 0) : collect(e, 0);
  if (rc) return rc;
  std::vector<jg_fault_row>& q = e->router ? e->router->faults : e->q_faults;
  *n = q.size();
  if (!out) return JG_OK;
  if (cap < q.size()) return fail(JG_ECAPACITY, "output buffer too small");
  if (!q.empty()) std::memcpy(out, q.data(), q.size() * sizeof(jg_fault_row));
  q.clear();
  e->q_fault_seq.clear();
  return JG_OK;
}

int jg_read_state(jg_engine* e, int field, uint32_t replica, void* out, uint32_t g0, uint32_t n) {
  if (!e || (!out && n)) return fail(JG_EINVAL, "null argument");
  if (e->router) return router_read_state(e, field, replica, out, g0, n);
  if ((uint64_t)g0 + n > e->cfg.n_groups) return fail(JG_EINVAL, "group range out of bounds");
  if (field < 0 || field >= JG_FIELD__COUNT) return fail(JG_EINVAL, "unknown field");
  if (field == JG_FIELD_MATCH && replica >= e->cfg.n_replicas) return fail(JG_EINVAL, "replica out of range");
  if (!n) return JG_OK;
  int rc = sync_and_check(e);
  if (rc) return rc;
  const JgDev& d = e->dev;
  std::vector<uint32_t> fl(n);
  HIPCHK(hipMemcpy(fl.data(), d.flags + g0, (size_t)n * 4, hipMemcpyDeviceToHost));
  auto role = [&](uint32_t i) { return fl[i] & JGF_ROLE_MASK; };
  std::vector<uint64_t> t64;
  std::vector<uint32_t> t32;
  auto get64 = [&](const uint64_t* col) -> int {
    t64.resize(n);
    HIPCHK(hipMemcpy(t64.data(), col + g0, (size_t)n * 8, hipMemcpyDeviceToHost));
    return JG_OK;
  };
  // a field of one of the two 16-byte cold records (JgColdCols), for groups [g0, g0 + n): a strided copy
  auto cold_field = [&](const uint4* col, size_t offset, size_t width, void* dst) -> int {
    HIPCHK(hipMemcpy2D(dst, width, (const char*)(col + g0) + offset, sizeof(uint4), width, n, hipMemcpyDeviceToHost));
    return JG_OK;
  };
  auto cold32 = [&](size_t offset) -> int {
    t32.resize(n);
    return cold_field(d.cold.v, offset, 4, t32.data());
  };
  auto lag_base = [&](std::vector<uint64_t>& base) -> int {  // what a leader's lags are relative to (jg_lag_base_is_run_hi)
    std::vector<uint64_t> top(n);
    HIPCHK(hipMemcpy(base.data(), d.head + g0, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(top.data(), d.run_hi + g0, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++)
      if (jg_lag_base_is_run_hi(fl[i])) base[i] = top[i];
    return JG_OK;
  };
  auto copy64 = [&](const uint64_t* col) -> int {  // straight column -> caller's buffer
    HIPCHK(hipMemcpy(out, col + g0, (size_t)n * 8, hipMemcpyDeviceToHost));
    return JG_OK;
  };
  uint64_t* o64 = (uint64_t*)out;
  uint32_t* o32 = (uint32_t*)out;
  uint8_t* o8 = (uint8_t*)out;
  switch (field) {
    case JG_FIELD_TERM: return copy64(d.term);
    case JG_FIELD_COMMIT: {  // leaders: packed as a lag below the head (field R of mlag), escape -> column
      std::vector<uint64_t> head(n), col(n);
      if ((rc = lag_base(head))) return rc;
      HIPCHK(hipMemcpy(col.data(), d.commit + g0, (size_t)n * 8, hipMemcpyDeviceToHost));
      if ((rc = get64(d.mlag))) return rc;
      for (uint32_t i = 0; i < n; i++) {
        const uint64_t f = jg_lag_field(t64[i], d.R, d.R);
        o64[i] = (role(i) != JG_ROLE_LEADER || jg_lag_wide(f, d.R)) ? col[i] : head[i] - f;
      }
      return JG_OK;
    }
    case JG_FIELD_HEAD: return copy64(d.head);
    case JG_FIELD_ELECTION_TIME: return cold_field(d.cold.t, JG_COLD_T_ELECTION_TIME, 8, out);
    case JG_FIELD_ELECTION_TIMEOUT: return cold_field(d.cold.t, JG_COLD_T_ELECTION_TIMEOUT, 4, out);
    case JG_FIELD_QUEUED_REQS: return cold_field(d.cold.v, JG_COLD_V_QUEUED, 4, out);
    case JG_FIELD_ID_GEN: {  // implicit (head + 1) while the chain is in FAST form
      std::vector<uint64_t> head(n);
      HIPCHK(hipMemcpy(head.data(), d.head + g0, (size_t)n * 8, hipMemcpyDeviceToHost));
      if ((rc = get64(d.id_gen))) return rc;
      for (uint32_t i = 0; i < n; i++) o64[i] = (fl[i] & JGF_FAST) ? head[i] + 1 : t64[i];
      return JG_OK;
    }
    case JG_FIELD_MATCH: {  // delta-packed: head - lag, or the wide column where the lag field is the escape
      std::vector<uint64_t> head(n), wide(n);
      if ((rc = lag_base(head))) return rc;
      HIPCHK(hipMemcpy(wide.data(), d.match_wide + (size_t)replica * d.G + g0, (size_t)n * 8, hipMemcpyDeviceToHost));
      if ((rc = get64(d.mlag))) return rc;
      for (uint32_t i = 0; i < n; i++) {
        const uint64_t f = jg_lag_field(t64[i], replica, d.R);
        o64[i] = role(i) != JG_ROLE_LEADER ? 0 : jg_lag_wide(f, d.R) ? wide[i] : head[i] - f;
      }
      return JG_OK;
    }
    case JG_FIELD_HEARTBEAT_TIME:
      if ((rc = get64(d.heartbeat_time))) return rc;
      for (uint32_t i = 0; i < n; i++) o64[i] = role(i) == JG_ROLE_LEADER ? t64[i] : 0;
      return JG_OK;
    case JG_FIELD_VOTED_FOR:
      if ((rc = cold32(JG_COLD_V_VOTED_FOR))) return rc;
      for (uint32_t i = 0; i < n; i++) o32[i] = (fl[i] & JGF_VOTED) ? t32[i] : 0;
      return JG_OK;
    case JG_FIELD_LEADER_ID:
      if ((rc = cold32(JG_COLD_V_LEADER_ID))) return rc;
      for (uint32_t i = 0; i < n; i++)
        o32[i] = (role(i) == JG_ROLE_FOLLOWER && (fl[i] & JGF_HAS_LEADER)) ? t32[i] : 0;
      return JG_OK;
    case JG_FIELD_VOTE_SEEN:
    case JG_FIELD_VOTE_GRANTED:
      if ((rc = cold32(JG_COLD_V_VOTES))) return rc;
      for (uint32_t i = 0; i < n; i++) {
        uint32_t v = field == JG_FIELD_VOTE_SEEN ? (t32[i] & 0xff) : ((t32[i] >> 8) & 0xff);
        o8[i] = role(i) == JG_ROLE_CANDIDATE ? (uint8_t)v : 0;
      }
      return JG_OK;
    case JG_FIELD_HAS_VOTED:
      for (uint32_t i = 0; i < n; i++) o8[i] = (fl[i] & JGF_VOTED) ? 1 : 0;
      return JG_OK;
    case JG_FIELD_ROLE:
      for (uint32_t i = 0; i < n; i++) o8[i] = (uint8_t)role(i);
      return JG_OK;
    case JG_FIELD_REPL_STATE:
      for (uint32_t i = 0; i < n; i++)
        o8[i] = role(i) == JG_ROLE_LEADER ? (uint8_t)((fl[i] & JGF_REPL_MASK) >> JGF_REPL_SHIFT) : 0;
      return JG_OK;
    case JG_FIELD_FAULT:
      for (uint32_t i = 0; i < n; i++) o8[i] = (uint8_t)((fl[i] & JGF_FAULT_MASK) >> JGF_FAULT_SHIFT);
      return JG_OK;
    case JG_FIELD_HAS_LEADER:
      for (uint32_t i = 0; i < n; i++) o8[i] = (role(i) == JG_ROLE_FOLLOWER && (fl[i] & JGF_HAS_LEADER)) ? 1 : 0;
      return JG_OK;
    case JG_FIELD_SELF_SLOT:
      for (uint32_t i = 0; i < n; i++) o8[i] = (uint8_t)((fl[i] & JGF_SELF_MASK) >> JGF_SELF_SHIFT);
      return JG_OK;
    default: return fail(JG_EINVAL, "unknown field");
  }
}

int jg_get_counters(jg_engine* e, uint64_t out[4]) {
  if (!e || !out) return fail(JG_EINVAL, "null argument");
  if (e->router) return router_get_counters(e, out);
  int rc = sync_and_check(e);
  if (rc) return rc;
  std::vector<uint64_t> slots(e->count_slots);
  HIPCHK(hipMemcpy(slots.data(), e->dev.blk_decisions, slots.size() * 8, hipMemcpyDeviceToHost));
  uint64_t dec = 0;
  for (uint64_t v : slots) dec += v;
  out[0] = e->n_cmds;
  out[1] = dec;
  out[2] = e->n_dense;
  out[3] = e->n_launch;
  return JG_OK;
}

int jg_device_alloc(jg_engine* e, size_t bytes, void** dev_ptr) {
  if (!e || !dev_ptr) return fail(JG_EINVAL, "null argument");
  if (e->router) return fail(JG_EINVAL, "device pointers are per shard: call this on a shard handle (jg_get_shard)");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMalloc(dev_ptr, std::max<size_t>(bytes, 16)));
  HIPCHK(hipMemsetAsync(*dev_ptr, 0, std::max<size_t>(bytes, 16), e->stream));
  return JG_OK;
}
int jg_device_free(jg_engine* e, void* dev_ptr) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) return fail(JG_EINVAL, "device pointers are per shard: call this on a shard handle (jg_get_shard)");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipFree(dev_ptr));
  return JG_OK;
}
int jg_device_upload(jg_engine* e, void* dev_dst, const void* host_src, size_t bytes) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) return fail(JG_EINVAL, "device pointers are per shard: call this on a shard handle (jg_get_shard)");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return JG_OK;
}
int jg_device_download(jg_engine* e, void* host_dst, const void* dev_src, size_t bytes) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) return fail(JG_EINVAL, "device pointers are per shard: call this on a shard handle (jg_get_shard)");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return JG_OK;
}
int jg_timer_start(jg_engine* e) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) {  // every shard's stream
    for (jg_engine* s : e->router->sh) {
      const int rc = jg_timer_start(s);
      if (rc) return rc;
    }
    return JG_OK;
  }
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipEventRecord(e->ev0, e->stream));
  return JG_OK;
}
int jg_timer_stop(jg_engine* e, float* ms) {
  if (!e || !ms) return fail(JG_EINVAL, "null argument");
  if (e->router) {  // the slowest shard
    *ms = 0;
    for (jg_engine* s : e->router->sh) {
      float v = 0;
      const int rc = jg_timer_stop(s, &v);
      if (rc) return rc;
      *ms = std::max(*ms, v);
    }
    return JG_OK;
  }
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipEventRecord(e->ev1, e->stream));
  {  // poll for a while before sleeping on the event: an interrupt-driven wake-up costs tens of
     // microseconds, which is a visible fraction of a 20-launch timed region
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t q;
    while ((q = hipEventQuery(e->ev1)) == hipErrorNotReady &&
           std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(20)) {
    }
    if (q != hipSuccess && q != hipErrorNotReady) return fail(JG_EDEVICE, std::string("hipEventQuery: ") + hipGetErrorString(q));
  }
  HIPCHK(hipEventSynchronize(e->ev1));
  HIPCHK(hipEventElapsedTime(ms, e->ev0, e->ev1));
  return JG_OK;
}

int jg_kernel_timing(jg_engine* e, int enable) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) {
    for (jg_engine* s : e->router->sh) {
      const int rc = jg_kernel_timing(s, enable);
      if (rc) return rc;
    }
    return JG_OK;
  }
  HIPCHK(hipSetDevice(e->device));
  if (enable && e->kt_ev.empty()) {
    e->kt_ev.resize(2 * jg_engine::KT_RING);
    for (hipEvent_t& ev : e->kt_ev) HIPCHK(hipEventCreate(&ev));
  }
  e->kt_on = enable != 0;
  e->kt_every = enable > 1 ? (uint32_t)enable : 1u;
  e->kt_n = 0, e->kt_seen = 0;
  return JG_OK;
}

int jg_kernel_timing_read(jg_engine* e, float* avg_us, uint32_t* n_launches) {
  if (!e || !avg_us || !n_launches) return fail(JG_EINVAL, "null argument");
  if (e->router) {  // the slowest shard's average
    *avg_us = 0, *n_launches = 0;
    for (jg_engine* s : e->router->sh) {
      float v = 0;
      uint32_t k = 0;
      const int rc = jg_kernel_timing_read(s, &v, &k);
      if (rc) return rc;
      if (v > *avg_us) *avg_us = v, *n_launches = k;
    }
    return JG_OK;
  }
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  const uint64_t n = std::min<uint64_t>(e->kt_n, jg_engine::KT_RING);
  double sum = 0;
  for (uint64_t k = 0; k < n; k++) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e->kt_ev[2 * k], e->kt_ev[2 * k + 1]));
    sum += ms;
  }
  *avg_us = n ? (float)(sum * 1e3 / (double)n) : 0.0f;
  *n_launches = (uint32_t)n;
  return JG_OK;
}

int jg_calibrate_stream(jg_engine* e, uint32_t iters, float* avg_us) {
  if (!e || !avg_us || !iters) return fail(JG_EINVAL, "null argument");
  if (e->router) e = e->router->sh[0];
  HIPCHK(hipSetDevice(e->device));
  const size_t G = e->cfg.n_groups, R = e->cfg.n_replicas;
  const size_t blk = std::max<size_t>(G * R * 8, 16);
  // enough ack-sized blocks to overflow the 256 MiB Infinity Cache, like the real ack stream
  const size_t nbuf = std::min<size_t>(std::max<size_t>(((size_t)640 << 20) / blk + 1, 2), 64);
  char* rot = nullptr;
  uint64_t *a8 = nullptr, *b8 = nullptr;
  uint32_t* c4 = nullptr;
  HIPCHK(hipMalloc((void**)&rot, blk * nbuf));
  HIPCHK(hipMalloc((void**)&a8, std::max<size_t>(G * 8, 16)));
  HIPCHK(hipMalloc((void**)&b8, std::max<size_t>(G * 8, 16)));
  HIPCHK(hipMalloc((void**)&c4, std::max<size_t>(G * 4, 16)));
  HIPCHK(hipMemsetAsync(rot, 0, blk * nbuf, e->stream));
  HIPCHK(hipMemsetAsync(a8, 0, std::max<size_t>(G * 8, 16), e->stream));
  HIPCHK(hipMemsetAsync(b8, 0, std::max<size_t>(G * 8, 16), e->stream));
  HIPCHK(hipMemsetAsync(c4, 0, std::max<size_t>(G * 4, 16), e->stream));
  const uint32_t warm = 5;
  for (uint32_t i = 0; i < warm + iters; i++) {
    if (i == warm) HIPCHK(hipEventRecord(e->ev0, e->stream));
    const uint64_t* r = (const uint64_t*)(rot + (i % nbuf) * blk);
    switch (R) {
      case 1: launch_calib<1>(e, r, a8, b8, c4); break;
      case 2: launch_calib<2>(e, r, a8, b8, c4); break;
      case 3: launch_calib<3>(e, r, a8, b8, c4); break;
      case 4: launch_calib<4>(e, r, a8, b8, c4); break;
      case 5: launch_calib<5>(e, r, a8, b8, c4); break;
      case 6: launch_calib<6>(e, r, a8, b8, c4); break;
      case 7: launch_calib<7>(e, r, a8, b8, c4); break;
      default: launch_calib<8>(e, r, a8, b8, c4); break;
    }
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e->ev1, e->stream));
  HIPCHK(hipEventSynchronize(e->ev1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  *avg_us = ms * 1000.0f / (float)iters;
  HIPCHK(hipFree(rot));
  HIPCHK(hipFree(a8));
  HIPCHK(hipFree(b8));
  HIPCHK(hipFree(c4));
  return JG_OK;
}

int jg_synth_fill_acks_device(jg_engine* e, uint32_t mode, uint64_t tick, uint64_t* sim_dev, uint64_t* acks_dev) {
  if (!e || !sim_dev || !acks_dev) return fail(JG_EINVAL, "null argument");
  if (e->router) return fail(JG_EINVAL, "device pointers are per shard: call this on a shard handle (jg_get_shard)");
  if (mode > 1) return fail(JG_EINVAL, "unknown synth mode");
  HIPCHK(hipSetDevice(e->device));
  hipLaunchKernelGGL(k_synth_acks, dim3(grid_for(e->cfg.n_groups, 4096)), dim3(JG_BLOCK), 0, e->stream, e->dev, mode,
                     tick, sim_dev, acks_dev);
  HIPCHK(hipGetLastError());
  return JG_OK;
}

}  // extern "C"
