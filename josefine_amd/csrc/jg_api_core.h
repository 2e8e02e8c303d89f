// jg_api_core.h - the engine behind the C ABI: error reporting, the step records and output arenas, the pinned host queues,
// struct jg_engine, and what every entry point shares (status checks, the deferred lists, the fault queue's ordering).
// One of the pieces of josefine_gpu.hip's ONE translation unit (included there, in order; not a header for anybody else).
#pragma once
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
  bool p_id32 = false;       // JG_COL_ID32: the pending id column holds 32-bit values (the step's one commit)
  bool p_packed = false;     // JG_COL_PACKED_KIND: the pending kind column holds kind | sender slot << 4 | flag << 7
  bool p_unchecked = false;  // some rows were committed with JG_COL_UNCHECKED: only jg_step_node may take this batch
  // JG_COL_UPLOAD_NOW: the committed batch on its way to the device before the step is called, on a copy stream of
  // its own (the two directions of the bus are independent: the previous step's outputs travel home meanwhile).
  // Two device buffers by turns: a step's rows are read until the step is settled, the next upload must not wait for that
  struct RowLayout {
    size_t n = 0, nb = 0, bytes = 0;
    bool has_from = false, has_term = false, has_aux = false, has_flag = false, packed = false, id32 = false;
    size_t o_id = 0, o_term = 0, o_aux = 0, o_bid = 0, o_bnext = 0, o_group = 0, o_from = 0, o_kind = 0, o_flag = 0;
    bool same_batch(const RowLayout& o) const {
      return n == o.n && nb == o.nb && has_from == o.has_from && has_term == o.has_term && has_aux == o.has_aux && has_flag == o.has_flag && packed == o.packed && id32 == o.id32;
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
  bool cluster_mask_offers = false;  // set by a routed round around its single-lead leader half (JgLeaderNode::mask_offers)
  uint64_t* cluster_aec = nullptr;  // set by a jg_dense_cluster around ITS dense halves: the cluster's common AppendEntries column (JgLeaderNode::o_aec)
  uint32_t replay_slot = 0;
  // jg_step_node: the inbox / outbox columns of the node step, their pinned host mirrors, rocPRIM scratch
  // JG_NODE_ASYNC: a step that returned without looking at its general-path row count (settled by node_settle)
  struct NodePending {
    bool on = false;
    JgNodeRows rows{};
    size_t n = 0, nb = 0, fsm_rec_seq = 0;
    uint64_t now_ms = 0;
    uint32_t flags = 0, col_mask = 0, seq_general = 0, seq_leader = 0, seq_follower = 0, seq_end = 0;
  };
  // What ONE node step leaves for the host.  A step owns a set until the next step begins - or, taken with JG_NODE_KEEP,
  // until its outbox has been viewed: two such steps may be outstanding (the event loop's two ticks in flight), the newer
  // one in NodeStep's own fields, the older one in NodeStep::spare (the sets change places when a step begins).
  struct NodeOut {
    jg_leader_beat* h_beat = nullptr;  // pinned mirrors of the outbox columns
    uint64_t *h_ae = nullptr, *h_answer = nullptr, *h_hbc = nullptr, *h_aec = nullptr;
    // device outbox: a set has its own, so that a kept step's columns travel home (on a stream of their own) while the next
    // step's kernels already write theirs
    jg_leader_beat* o_beat = nullptr;
    uint64_t *o_answer = nullptr, *o_hbc = nullptr;
    uint64_t* o_aec = nullptr;         // JG_NODE_COMMON_AE: the common AppendEntries word (JgLeaderNode::o_aec), allocated at first use
    uint64_t* o_ae = nullptr;          // the AppendEntries words by addressee (JG_NODE_COMMON_AE fetches them when a partition needs them)
    bool ae_rows_landed = false;       // JG_NODE_COMMON_AE: h_ae holds the step's rows (fetched when a partition needs them)
    uint32_t* h_nsparse = nullptr;     // pinned: the step's copy of d_nsparse
    hipEvent_t ev_out = nullptr;
    NodePending pending;
    jg_node_outbox last{};
    uint32_t last_flags = 0;
    // -- JG_NODE_KEEP ----------------------------------------------------------------------------------------------------
    bool keep = false;                 // the step was taken with JG_NODE_KEEP
    bool out = false;                  // ... and its outbox has not been viewed yet
    int set = 0, arena = 0;            // the fault / exceptional-row queues and the arena its kernels used
    hipEvent_t ev_early = nullptr;     // behind the copy of the general-path row count (the row passes are done)
    hipEvent_t ev_kernels = nullptr;   // behind the kernels whose output the down stream copies next
    uint32_t* h_status = nullptr;      // pinned [8]: the device status block behind the step's last kernel
    uint64_t* h_total = nullptr;       // pinned: the fsm_tx rows of the step's dense halves ...
    JgScanJob* h_job = nullptr;        // pinned: ... and the scan job that counts them
    uint32_t* d_fsm_cnt = nullptr;     // the step's fsm regions (its arena's): counts, rows, tile sums, the compacted rows
    jg_fsm_row *d_fsm = nullptr, *d_stage = nullptr;
    uint64_t* d_bsum = nullptr;
    PinnedQueue<jg_fsm_row> l_fsm;     // where the compacted rows land: the first `fsm_copied` came with the step's own copy
    size_t fsm_copied = 0;
    bool fsm_landed = false;           // l_fsm is to be handed to q_fsm by the next drain of that queue
    uint64_t irr_gen = 0;
    uint32_t seq_lo = 0, seq_hi = 0;   // the engine's step numbers before and after the step
  };
  struct NodeStep : NodeOut {
    bool ready = false;
    JgNodeCols cols{};
    uint64_t *h_in_answers = nullptr, *h_in_hbc = nullptr;  // pinned [R][G]: column inbound (jg_node_inbox_columns)
    uint32_t col_mask = 0, col_hbc_mask = 0;                // slots handed out for the next step / with their hb_commit column
    uint32_t* d_nsparse = nullptr;     // {general-path rows, -, partitions whose AppendEntries words differ by addressee (JG_NODE_COMMON_AE), -}
    // the general path's rows as (group << 32 | arrival index, arrival index) pairs, appended by k_node_route (grow-only),
    // and the bucket pass that orders them (jg_route.h: hist / scan / scatter + k_bucket_order)
    uint64_t* sp_key = nullptr;
    uint32_t* sp_idx = nullptr;
    size_t sp_cap = 0;
    uint32_t* bk_mem = nullptr;
    uint32_t bk_words = 0, bk_buckets = 0, bk_tile_bits = 0;
    // the tiled row pass (jg_node.h, k_node_bin_* / k_node_tile): the chunks' counts per tile, the tiles' first rows, and the
    // binned copies of a step's rows (grow-only; 38 bytes per row of room)
    uint32_t *bin_cnt = nullptr, *bin_off = nullptr;
    char* bin_mem = nullptr;
    size_t bin_cap = 0;
    uint32_t n_tiles = 0;
    uint32_t group_bits = 1;
    hipEvent_t ev_cols = nullptr;      // behind the uploads of the handed-out columns: the pinned buffers are free again
    bool cols_in_flight = false;
    using Pending = NodePending;
    // JG_NODE_KEEP: the other set (the OLDER outstanding step's while two are, a free one otherwise)
    NodeOut spare;
    // ... and the stream a kept step's outputs travel home on (the outbox columns, the fsm rows, the status block): the step's
    // kernels do not queue up behind the previous step's copies - a tick's kernels are 0.3 ms, its trip home 1 ms
    hipStream_t down = nullptr;
    uint32_t kept_n = 0;               // kept steps whose outbox has not been viewed (0 .. 2)
    bool in_step = false;              // node_step is running (its halves' own node_settle calls are not another caller's)
    bool viewed_spare = false;         // the outbox viewed last is the spare set's
    size_t fsm_guess = 0;              // fsm rows of the kept step finished last: what the next one's own copy takes along
    bool fsm_guess_known = false;
    NodeOut* fsm_visible = nullptr;    // the set whose l_fsm the next drain of the fsm queue hands over
    NodeOut& own() { return *this; }
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
// JG_NODE_KEEP: while kept node steps are outstanding nothing else may step the engine - what it produced would have to be
// delivered between two steps whose outputs are not due yet (the node step's own halves and its settling pass go through)
inline int kept_refuse(const jg_engine* e) {
  if (e->node.kept_n && !e->node.in_step) return fail(JG_EINVAL, "kept node steps are outstanding (JG_NODE_KEEP): jg_node_outbox_view first");
  return JG_OK;
}
int dense_step(jg_engine* e, const uint64_t* acks_dev, uint32_t n_ticks = 1, const JgLeaderNode* nd = nullptr) {
  {
    const int rc = kept_refuse(e);
    if (rc) return rc;
  }
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
__global__ void k_fault_join(const JgFaultRec* __restrict__ q, const uint32_t* __restrict__ order, uint32_t n,
                             jg_fault_row* __restrict__ rows, uint32_t* __restrict__ seqs) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const JgFaultRec r = q[order[i]];
    rows[i] = jg_fault_row{r.group, r.code};
    seqs[i] = r.seq;
  }
}

}  // namespace
