// josefine_gpu.hip — C ABI (include/josefine_gpu.h) of the MI355X batched
// Chained-Raft engine: host-side marshalling around the gfx950 kernels in
// jg_kernels.h.  There is deliberately no CPU implementation in this library:
// every entry point that computes does so on the device or fails with
// JG_EDEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "jg_kernels.h"

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      return fail(JG_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(_e) + " (no CPU fallback)"); \
  } while (0)

namespace {

struct StepRec {
  uint32_t n_active = 0;
  void* blob = nullptr;  // one device allocation per step
  uint32_t *d_msg_base = nullptr, *d_fsm_base = nullptr, *d_msg_cnt = nullptr, *d_fsm_cnt = nullptr;
  jg_msg_row* d_msg = nullptr;
  jg_fsm_row* d_fsm = nullptr;
};

}  // namespace

struct jg_engine {
  jg_config cfg;
  JgDev dev;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<void*> allocs;
  uint32_t count_slots = 0;  // workgroup slots of dev.blk_decisions
  uint32_t dense_grid = 0;
  int dense_variant = 1;
  uint32_t* d_err = nullptr;
  uint64_t* d_acks_staging = nullptr;  // [R][G] for the host-buffer dense entry point
  // commands queued by jg_submit (host SoA)
  std::vector<uint8_t> p_kind, p_flag;
  std::vector<uint32_t> p_group, p_from;
  std::vector<uint64_t> p_term, p_id, p_aux, p_blk_id, p_blk_next;
  std::vector<StepRec> recs;
  std::vector<jg_msg_row> q_msgs;
  std::vector<jg_fsm_row> q_fsm;
  std::vector<jg_fault_row> q_faults;
  uint32_t seq = 0;
  bool stepped = false;
  bool maybe_irregular = false;  // some leader's chain may have left FAST form
  uint64_t n_cmds = 0, n_dense = 0, n_launch = 0;
};

namespace {

template <typename T>
int dev_alloc(jg_engine* e, T** p, size_t n, bool zero = true) {
  void* q = nullptr;
  size_t bytes = std::max<size_t>(n * sizeof(T), 16);
  HIPCHK(hipMalloc(&q, bytes));
  if (zero) HIPCHK(hipMemsetAsync(q, 0, bytes, e->stream));
  e->allocs.push_back(q);
  *p = (T*)q;
  return JG_OK;
}

inline uint32_t grid_for(size_t n, uint32_t cap) {
  size_t b = (n + JG_BLOCK - 1) / JG_BLOCK;
  if (b < 1) b = 1;
  return (uint32_t)std::min<size_t>(b, cap);
}

// output-row bounds per command kind (messages, fsm rows) for R replicas
inline void row_bounds(uint8_t kind, uint32_t R, uint32_t* m, uint32_t* f) {
  switch (kind) {
    case JG_CMD_TICK:
    case JG_CMD_TIMEOUT: *m = R + 1; *f = 0; break;       // DROP + (R-1) VoteRequest + Heartbeat | Heartbeat + (R-1) AppendEntries
    case JG_CMD_HEARTBEAT_RESPONSE: *m = R; *f = 0; break;  // replicate()
    case JG_CMD_HEARTBEAT: *m = 2; *f = 1; break;          // FLUSH + HeartbeatResponse; Apply
    case JG_CMD_VOTE_RESPONSE: *m = 2; *f = 0; break;      // DROP + Heartbeat on elect()
    case JG_CMD_CLIENT_REQUEST: *m = 1; *f = 2; break;     // proxy/queue; Notify + Apply
    case JG_CMD_APPEND_RESPONSE: *m = 0; *f = 1; break;
    case JG_CMD_VOTE_REQUEST:
    case JG_CMD_APPEND_ENTRIES:
    case JG_CMD_CLIENT_RESPONSE: *m = 1; *f = 0; break;
    default: *m = 0; *f = 0; break;
  }
}

template <int R>
void launch_dense(jg_engine* e, const uint64_t* acks, uint32_t n_ticks) {
  const size_t stride = (size_t)e->cfg.n_groups * e->cfg.n_replicas;
  if (n_ticks > 1)  // temporal fusion: state read once, written once per launch
    hipLaunchKernelGGL(k_leader_tick_dense_n<R>, dim3(e->dense_grid), dim3(JG_BLOCK), 0, e->stream, e->dev, acks,
                       n_ticks, stride, e->seq);
  else if (e->dense_variant == 2)  // two groups per lane, 16-B accesses (needs an even G)
    hipLaunchKernelGGL(k_leader_tick_dense_x2<R>, dim3(e->dense_grid), dim3(JG_BLOCK), 0, e->stream, e->dev, acks,
                       e->seq);
  else
    hipLaunchKernelGGL(k_leader_tick_dense<R>, dim3(e->dense_grid), dim3(JG_BLOCK), 0, e->stream, e->dev, acks,
                       e->seq);
}

int dense_step(jg_engine* e, const uint64_t* acks_dev, uint32_t n_ticks = 1) {
  e->stepped = true;
  e->seq++;  // tick t of this launch carries sequence number seq + t
  if (e->maybe_irregular) HIPCHK(hipMemsetAsync(e->dev.slow_n, 0, sizeof(uint32_t), e->stream));
  switch (e->cfg.n_replicas) {
    case 1: launch_dense<1>(e, acks_dev, n_ticks); break;
    case 2: launch_dense<2>(e, acks_dev, n_ticks); break;
    case 3: launch_dense<3>(e, acks_dev, n_ticks); break;
    case 4: launch_dense<4>(e, acks_dev, n_ticks); break;
    case 5: launch_dense<5>(e, acks_dev, n_ticks); break;
    case 6: launch_dense<6>(e, acks_dev, n_ticks); break;
    case 7: launch_dense<7>(e, acks_dev, n_ticks); break;
    default: launch_dense<8>(e, acks_dev, n_ticks); break;
  }
  e->n_launch++;
  if (e->maybe_irregular) {
    hipLaunchKernelGGL(k_dense_slow, dim3(std::min<uint32_t>(e->count_slots, 64)), dim3(JG_BLOCK), 0, e->stream,
                       e->dev, acks_dev, n_ticks, (size_t)e->cfg.n_groups * e->cfg.n_replicas, e->seq);
    e->n_launch++;
  }
  HIPCHK(hipGetLastError());
  e->seq += n_ticks - 1;
  e->n_dense += (uint64_t)e->cfg.n_groups * n_ticks;
  return JG_OK;
}

// Pull finished steps' output rows and the fault queue to the host queues.
int collect(jg_engine* e) {
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  uint32_t err = 0, slow_n = 0;
  HIPCHK(hipMemcpy(&err, e->d_err, sizeof err, hipMemcpyDeviceToHost));
  if (err) return fail(JG_EDEVICE, "internal: an output row exceeded its per-command bound");
  if (!e->maybe_irregular) {
    HIPCHK(hipMemcpy(&slow_n, e->dev.slow_n, sizeof slow_n, hipMemcpyDeviceToHost));
    if (slow_n) return fail(JG_EDEVICE, "internal: irregular chain reached the fast-only dense path");
  }
  for (StepRec& r : e->recs) {
    const uint32_t n = r.n_active;
    std::vector<uint32_t> base(n + 1), cnt(n), off(n);
    for (int pass = 0; pass < 2; pass++) {
      const size_t row = pass == 0 ? sizeof(jg_msg_row) : sizeof(jg_fsm_row);
      HIPCHK(hipMemcpy(cnt.data(), pass == 0 ? r.d_msg_cnt : r.d_fsm_cnt, n * 4, hipMemcpyDeviceToHost));
      uint64_t total = 0;
      for (uint32_t i = 0; i < n; i++) {
        off[i] = (uint32_t)total;
        total += cnt[i];
      }
      if (!total) continue;
      uint32_t* d_off = nullptr;
      void* d_dst = nullptr;
      HIPCHK(hipMalloc((void**)&d_off, n * 4));
      HIPCHK(hipMalloc(&d_dst, total * row));
      HIPCHK(hipMemcpyAsync(d_off, off.data(), n * 4, hipMemcpyHostToDevice, e->stream));
      uint32_t grid = grid_for(n, 1024);
      if (pass == 0) {
        hipLaunchKernelGGL(k_gather_rows<jg_msg_row>, dim3(grid), dim3(JG_BLOCK), 0, e->stream, n, r.d_msg_base,
                           r.d_msg_cnt, d_off, r.d_msg, (jg_msg_row*)d_dst);
        size_t at = e->q_msgs.size();
        e->q_msgs.resize(at + total);
        HIPCHK(hipMemcpyAsync(e->q_msgs.data() + at, d_dst, total * row, hipMemcpyDeviceToHost, e->stream));
      } else {
        hipLaunchKernelGGL(k_gather_rows<jg_fsm_row>, dim3(grid), dim3(JG_BLOCK), 0, e->stream, n, r.d_fsm_base,
                           r.d_fsm_cnt, d_off, r.d_fsm, (jg_fsm_row*)d_dst);
        size_t at = e->q_fsm.size();
        e->q_fsm.resize(at + total);
        HIPCHK(hipMemcpyAsync(e->q_fsm.data() + at, d_dst, total * row, hipMemcpyDeviceToHost, e->stream));
      }
      HIPCHK(hipStreamSynchronize(e->stream));
      HIPCHK(hipFree(d_off));
      HIPCHK(hipFree(d_dst));
    }
    HIPCHK(hipFree(r.blob));
  }
  e->recs.clear();
  // faults
  uint32_t nf = 0;
  HIPCHK(hipMemcpy(&nf, e->dev.fault_q_n, sizeof nf, hipMemcpyDeviceToHost));
  if (nf) {
    if (nf > e->dev.fault_q_cap) return fail(JG_EDEVICE, "fault queue overflow");
    std::vector<JgFaultRec> fr(nf);
    HIPCHK(hipMemcpy(fr.data(), e->dev.fault_q, nf * sizeof(JgFaultRec), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(e->dev.fault_q_n, 0, sizeof(uint32_t)));
    std::stable_sort(fr.begin(), fr.end(), [](const JgFaultRec& a, const JgFaultRec& b) {
      return a.seq != b.seq ? a.seq < b.seq : a.group < b.group;
    });
    for (const JgFaultRec& f : fr) e->q_faults.push_back(jg_fault_row{f.group, f.code});
  }
  return JG_OK;
}

template <typename Row>
int drain(jg_engine* e, std::vector<Row>& q, Row* out, size_t cap, size_t* n) {
  if (!e || !n) return fail(JG_EINVAL, "null argument");
  int rc = collect(e);
  if (rc) return rc;
  *n = q.size();
  if (!out) return JG_OK;
  if (cap < q.size()) return fail(JG_ECAPACITY, "output buffer too small");
  if (!q.empty()) std::memcpy(out, q.data(), q.size() * sizeof(Row));
  q.clear();
  return JG_OK;
}

}  // namespace

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
  if (cfg->election_timeout_max_ms < cfg->election_timeout_min_ms) return fail(JG_EINVAL, "election timeout range");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(JG_EDEVICE, "no such HIP device (no CPU fallback)");
  HIPCHK(hipSetDevice(cfg->device_id));

  jg_engine* e = new jg_engine();
  e->cfg = *cfg;
  e->device = cfg->device_id;
  int rc = JG_OK;
  auto bail = [&](int code) {
    jg_engine_destroy(e);
    return code;
  };
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(JG_EDEVICE, "hipStreamCreate failed"));
  if (hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess)
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
  const char* env_var = std::getenv("JG_DENSE_VARIANT");
  e->dense_variant = env_var ? std::atoi(env_var) : 1;
  if (e->dense_variant != 2 || (G & 1)) e->dense_variant = 1;
  e->dense_grid = grid_for(e->dense_variant == 2 ? G / 2 : G, cap);
  e->count_slots = std::max<uint32_t>(e->dense_grid, 4096);
#define A(ptr, n)                                   \
  if ((rc = dev_alloc(e, &ptr, (n))) != JG_OK) return bail(rc)
  A(d.term, G);
  A(d.commit, G);
  A(d.head, G);
  A(d.id_gen, G);
  A(d.run_hi, G);
  A(d.match, G * R);
  A(d.election_time, G);
  A(d.heartbeat_time, G);
  A(d.win_lo, G * JG_CHAIN_WINDOW);
  A(d.win_hi, G * JG_CHAIN_WINDOW);
  A(d.win_next, G * JG_CHAIN_WINDOW);
  A(d.flags, G);
  A(d.voted_for, G);
  A(d.leader_id, G);
  A(d.election_timeout, G);
  A(d.rng_draws, G);
  A(d.queued, G);
  A(d.votes, G);
  A(d.blk_decisions, e->count_slots);
  d.fault_q_cap = (uint32_t)std::max<size_t>(2 * G, 1024);
  A(d.fault_q, d.fault_q_cap);
  A(d.fault_q_n, 1);
  A(d.slow_list, G);
  A(d.slow_n, 1);
  A(e->d_err, 1);
#undef A
  hipLaunchKernelGGL(k_init_groups, dim3(grid_for(G, 2048)), dim3(JG_BLOCK), 0, e->stream, e->dev,
                     (const uint8_t*)nullptr);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
    return bail(fail(JG_EDEVICE, "k_init_groups failed: is this a gfx950 device? (no CPU fallback)"));
  *out = e;
  return JG_OK;
}

void jg_engine_destroy(jg_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (StepRec& r : e->recs) (void)hipFree(r.blob);
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->d_acks_staging) (void)hipFree(e->d_acks_staging);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int jg_set_self_slots(jg_engine* e, const uint8_t* slots) {
  if (!e || !slots) return fail(JG_EINVAL, "null argument");
  if (e->stepped) return fail(JG_EINVAL, "self slots are fixed after the first step");
  for (uint32_t g = 0; g < e->cfg.n_groups; g++)
    if (slots[g] >= e->cfg.n_replicas) return fail(JG_EINVAL, "self slot out of range");
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
  if (b->n && (!b->kind || !b->group)) return fail(JG_EINVAL, "kind/group columns are required");
  for (size_t i = 0; i < b->n; i++) {
    if (b->group[i] >= e->cfg.n_groups) return fail(JG_EINVAL, "group out of range");
    if (b->kind[i] >= JG_CMD__COUNT) return fail(JG_EINVAL, "unknown command kind");
    if (b->kind[i] == JG_CMD_APPEND_ENTRIES) {
      if (!b->id || !b->aux) return fail(JG_EINVAL, "AppendEntries needs id/aux columns");
      if (b->id[i] + b->aux[i] > b->n_blocks) return fail(JG_EINVAL, "block side-array range out of bounds");
    }
  }
  const uint64_t blk_shift = e->p_blk_id.size();
  for (size_t i = 0; i < b->n; i++) {
    e->p_kind.push_back(b->kind[i]);
    e->p_group.push_back(b->group[i]);
    e->p_from.push_back(b->from ? b->from[i] : 0);
    e->p_term.push_back(b->term ? b->term[i] : 0);
    uint64_t id = b->id ? b->id[i] : 0;
    if (b->kind[i] == JG_CMD_APPEND_ENTRIES) id += blk_shift;  // side arrays are concatenated
    e->p_id.push_back(id);
    e->p_aux.push_back(b->aux ? b->aux[i] : 0);
    e->p_flag.push_back(b->flag ? b->flag[i] : 0);
  }
  for (size_t i = 0; i < b->n_blocks; i++) {
    e->p_blk_id.push_back(b->blk_id[i]);
    e->p_blk_next.push_back(b->blk_next[i]);
  }
  return JG_OK;
}

int jg_step(jg_engine* e, uint64_t now_ms) {
  if (!e) return fail(JG_EINVAL, "null argument");
  e->stepped = true;
  const size_t n = e->p_kind.size();
  if (!n) return JG_OK;
  HIPCHK(hipSetDevice(e->device));
  e->seq++;
  const uint32_t R = e->cfg.n_replicas;
  // stable bucket by group: per-group stream order is row order
  std::vector<uint32_t> order(n);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return e->p_group[a] < e->p_group[b]; });
  std::vector<uint32_t> seg_group, seg_off, msg_base, fsm_base;
  uint64_t mb = 0, fb = 0;
  for (size_t k = 0; k < n; k++) {
    uint32_t g = e->p_group[order[k]];
    if (k == 0 || g != seg_group.back()) {
      seg_group.push_back(g);
      seg_off.push_back((uint32_t)k);
      msg_base.push_back((uint32_t)mb);
      fsm_base.push_back((uint32_t)fb);
    }
    uint32_t m, f;
    uint8_t kind = e->p_kind[order[k]];
    row_bounds(kind, R, &m, &f);
    mb += m;
    fb += f;
    if (kind == JG_CMD_APPEND_ENTRIES || kind == JG_CMD_RESTART) e->maybe_irregular = true;
  }
  if (mb > 0xffffffffull || fb > 0xffffffffull) return fail(JG_EINVAL, "batch too large: split it");
  const uint32_t na = (uint32_t)seg_group.size();
  seg_off.push_back((uint32_t)n);
  msg_base.push_back((uint32_t)mb);
  fsm_base.push_back((uint32_t)fb);
  const size_t nb = e->p_blk_id.size();

  // blob layout (16-byte aligned sections); the first `up_bytes` are uploaded
  size_t off = 0;
  auto sect = [&](size_t bytes) {
    size_t at = off;
    off = (off + bytes + 15) & ~size_t(15);
    return at;
  };
  const size_t o_term = sect(n * 8), o_id = sect(n * 8), o_aux = sect(n * 8), o_bid = sect(nb * 8),
               o_bnext = sect(nb * 8), o_from = sect(n * 4), o_sg = sect(na * 4), o_so = sect((na + 1) * 4),
               o_mb = sect((na + 1) * 4), o_fb = sect((na + 1) * 4), o_kind = sect(n), o_flag = sect(n);
  const size_t up_bytes = off;
  const size_t o_mc = sect(na * 4), o_fc = sect(na * 4), o_msg = sect(mb * sizeof(jg_msg_row)),
               o_fsm = sect(fb * sizeof(jg_fsm_row));
  std::vector<uint8_t> stage(up_bytes);
  auto put = [&](size_t at, auto&& get, size_t count, size_t width) {
    for (size_t k = 0; k < count; k++) {
      auto v = get(k);
      std::memcpy(stage.data() + at + k * width, &v, width);
    }
  };
  put(o_term, [&](size_t k) { return e->p_term[order[k]]; }, n, 8);
  put(o_id, [&](size_t k) { return e->p_id[order[k]]; }, n, 8);
  put(o_aux, [&](size_t k) { return e->p_aux[order[k]]; }, n, 8);
  put(o_from, [&](size_t k) { return e->p_from[order[k]]; }, n, 4);
  put(o_kind, [&](size_t k) { return e->p_kind[order[k]]; }, n, 1);
  put(o_flag, [&](size_t k) { return e->p_flag[order[k]]; }, n, 1);
  if (nb) {
    std::memcpy(stage.data() + o_bid, e->p_blk_id.data(), nb * 8);
    std::memcpy(stage.data() + o_bnext, e->p_blk_next.data(), nb * 8);
  }
  std::memcpy(stage.data() + o_sg, seg_group.data(), na * 4);
  std::memcpy(stage.data() + o_so, seg_off.data(), (na + 1) * 4);
  std::memcpy(stage.data() + o_mb, msg_base.data(), (na + 1) * 4);
  std::memcpy(stage.data() + o_fb, fsm_base.data(), (na + 1) * 4);

  StepRec rec;
  rec.n_active = na;
  HIPCHK(hipMalloc(&rec.blob, off));
  uint8_t* B = (uint8_t*)rec.blob;
  HIPCHK(hipMemcpyAsync(B, stage.data(), up_bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));  // `stage` is pageable and about to go out of scope
  rec.d_msg_base = (uint32_t*)(B + o_mb);
  rec.d_fsm_base = (uint32_t*)(B + o_fb);
  rec.d_msg_cnt = (uint32_t*)(B + o_mc);
  rec.d_fsm_cnt = (uint32_t*)(B + o_fc);
  rec.d_msg = (jg_msg_row*)(B + o_msg);
  rec.d_fsm = (jg_fsm_row*)(B + o_fsm);

  JgStepArgs a;
  a.n_active = na;
  a.seg_group = (const uint32_t*)(B + o_sg);
  a.seg_off = (const uint32_t*)(B + o_so);
  a.kind = B + o_kind;
  a.from = (const uint32_t*)(B + o_from);
  a.term = (const uint64_t*)(B + o_term);
  a.id = (const uint64_t*)(B + o_id);
  a.aux = (const uint64_t*)(B + o_aux);
  a.flag = B + o_flag;
  a.blk_id = (const uint64_t*)(B + o_bid);
  a.blk_next = (const uint64_t*)(B + o_bnext);
  a.msg_base = rec.d_msg_base;
  a.fsm_base = rec.d_fsm_base;
  a.msg_out = rec.d_msg;
  a.fsm_out = rec.d_fsm;
  a.msg_cnt = rec.d_msg_cnt;
  a.fsm_cnt = rec.d_fsm_cnt;
  a.err = e->d_err;
  a.now = now_ms;
  a.seq = e->seq;
  hipLaunchKernelGGL(k_apply_cmds, dim3(grid_for(na, e->count_slots)), dim3(JG_BLOCK), 0, e->stream, e->dev, a);
  HIPCHK(hipGetLastError());
  e->n_launch++;
  e->recs.push_back(rec);
  e->n_cmds += n;
  e->p_kind.clear();
  e->p_flag.clear();
  e->p_group.clear();
  e->p_from.clear();
  e->p_term.clear();
  e->p_id.clear();
  e->p_aux.clear();
  e->p_blk_id.clear();
  e->p_blk_next.clear();
  return JG_OK;
}

int jg_step_dense_acks_device(jg_engine* e, const uint64_t* acks_dev) {
  if (!e || !acks_dev) return fail(JG_EINVAL, "null argument");
  if (!e->p_kind.empty()) return fail(JG_EINVAL, "commands are queued: call jg_step first");
  HIPCHK(hipSetDevice(e->device));
  return dense_step(e, acks_dev);
}

int jg_step_dense_acks_device_n(jg_engine* e, const uint64_t* acks_dev, uint32_t n_ticks) {
  if (!e || !acks_dev) return fail(JG_EINVAL, "null argument");
  if (!n_ticks) return JG_OK;
  if (!e->p_kind.empty()) return fail(JG_EINVAL, "commands are queued: call jg_step first");
  HIPCHK(hipSetDevice(e->device));
  return dense_step(e, acks_dev, n_ticks);
}

int jg_step_dense_acks(jg_engine* e, const uint64_t* acks_host) {
  if (!e || !acks_host) return fail(JG_EINVAL, "null argument");
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

int jg_chain_compact(jg_engine* e, size_t n_trees, const uint64_t* off, const uint64_t* ids, const uint64_t* nexts,
                     const uint64_t* commits, uint8_t* removed) {
  if (!e || !off || !commits) return fail(JG_EINVAL, "null argument");
  if (!n_trees) return JG_OK;
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

int jg_sync(jg_engine* e) {
  if (!e) return fail(JG_EINVAL, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  return JG_OK;
}

int jg_drain_messages(jg_engine* e, jg_msg_row* out, size_t cap, size_t* n) { return drain(e, e->q_msgs, out, cap, n); }
int jg_drain_applies(jg_engine* e, jg_fsm_row* out, size_t cap, size_t* n) { return drain(e, e->q_fsm, out, cap, n); }
int jg_drain_faults(jg_engine* e, jg_fault_row* out, size_t cap, size_t* n) { return drain(e, e->q_faults, out, cap, n); }

int jg_read_state(jg_engine* e, int field, uint32_t replica, void* out, uint32_t g0, uint32_t n) {
  if (!e || (!out && n)) return fail(JG_EINVAL, "null argument");
  if ((uint64_t)g0 + n > e->cfg.n_groups) return fail(JG_EINVAL, "group range out of bounds");
  if (field < 0 || field >= JG_FIELD__COUNT) return fail(JG_EINVAL, "unknown field");
  if (field == JG_FIELD_MATCH && replica >= e->cfg.n_replicas) return fail(JG_EINVAL, "replica out of range");
  if (!n) return JG_OK;
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  const JgDev& d = e->dev;
  std::vector<uint32_t> fl(n);
  HIPCHK(hipMemcpy(fl.data(), d.flags + g0, n * 4, hipMemcpyDeviceToHost));
  auto role = [&](uint32_t i) { return fl[i] & JGF_ROLE_MASK; };
  std::vector<uint64_t> t64;
  std::vector<uint32_t> t32;
  auto get64 = [&](const uint64_t* col) -> int {
    t64.resize(n);
    HIPCHK(hipMemcpy(t64.data(), col + g0, n * 8, hipMemcpyDeviceToHost));
    return JG_OK;
  };
  auto get32 = [&](const uint32_t* col) -> int {
    t32.resize(n);
    HIPCHK(hipMemcpy(t32.data(), col + g0, n * 4, hipMemcpyDeviceToHost));
    return JG_OK;
  };
  int rc = JG_OK;
  uint64_t* o64 = (uint64_t*)out;
  uint32_t* o32 = (uint32_t*)out;
  uint8_t* o8 = (uint8_t*)out;
  switch (field) {
    case JG_FIELD_TERM: return get64(d.term) ? JG_EDEVICE : (std::memcpy(out, t64.data(), n * 8), JG_OK);
    case JG_FIELD_COMMIT: return get64(d.commit) ? JG_EDEVICE : (std::memcpy(out, t64.data(), n * 8), JG_OK);
    case JG_FIELD_HEAD: return get64(d.head) ? JG_EDEVICE : (std::memcpy(out, t64.data(), n * 8), JG_OK);
    case JG_FIELD_ELECTION_TIME:
      return get64(d.election_time) ? JG_EDEVICE : (std::memcpy(out, t64.data(), n * 8), JG_OK);
    case JG_FIELD_ID_GEN: {
      std::vector<uint64_t> head(n);
      HIPCHK(hipMemcpy(head.data(), d.head + g0, n * 8, hipMemcpyDeviceToHost));
      if ((rc = get64(d.id_gen))) return rc;
      for (uint32_t i = 0; i < n; i++) o64[i] = (fl[i] & JGF_FAST) ? head[i] + 1 : t64[i];
      return JG_OK;
    }
    case JG_FIELD_MATCH:
      if ((rc = get64(d.match + (size_t)replica * d.G))) return rc;
      for (uint32_t i = 0; i < n; i++) o64[i] = role(i) == JG_ROLE_LEADER ? t64[i] : 0;
      return JG_OK;
    case JG_FIELD_HEARTBEAT_TIME:
      if ((rc = get64(d.heartbeat_time))) return rc;
      for (uint32_t i = 0; i < n; i++) o64[i] = role(i) == JG_ROLE_LEADER ? t64[i] : 0;
      return JG_OK;
    case JG_FIELD_VOTED_FOR:
      if ((rc = get32(d.voted_for))) return rc;
      for (uint32_t i = 0; i < n; i++) o32[i] = (fl[i] & JGF_VOTED) ? t32[i] : 0;
      return JG_OK;
    case JG_FIELD_LEADER_ID:
      if ((rc = get32(d.leader_id))) return rc;
      for (uint32_t i = 0; i < n; i++)
        o32[i] = (role(i) == JG_ROLE_FOLLOWER && (fl[i] & JGF_HAS_LEADER)) ? t32[i] : 0;
      return JG_OK;
    case JG_FIELD_ELECTION_TIMEOUT:
      return get32(d.election_timeout) ? JG_EDEVICE : (std::memcpy(out, t32.data(), n * 4), JG_OK);
    case JG_FIELD_QUEUED_REQS: return get32(d.queued) ? JG_EDEVICE : (std::memcpy(out, t32.data(), n * 4), JG_OK);
    case JG_FIELD_VOTE_SEEN:
    case JG_FIELD_VOTE_GRANTED:
      if ((rc = get32(d.votes))) return rc;
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
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
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
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMalloc(dev_ptr, std::max<size_t>(bytes, 16)));
  HIPCHK(hipMemsetAsync(*dev_ptr, 0, std::max<size_t>(bytes, 16), e->stream));
  return JG_OK;
}
int jg_device_free(jg_engine* e, void* dev_ptr) {
  if (!e) return fail(JG_EINVAL, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipFree(dev_ptr));
  return JG_OK;
}
int jg_device_upload(jg_engine* e, void* dev_dst, const void* host_src, size_t bytes) {
  if (!e) return fail(JG_EINVAL, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return JG_OK;
}
int jg_device_download(jg_engine* e, void* host_dst, const void* dev_src, size_t bytes) {
  if (!e) return fail(JG_EINVAL, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return JG_OK;
}
int jg_timer_start(jg_engine* e) {
  if (!e) return fail(JG_EINVAL, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipEventRecord(e->ev0, e->stream));
  return JG_OK;
}
int jg_timer_stop(jg_engine* e, float* ms) {
  if (!e || !ms) return fail(JG_EINVAL, "null argument");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipEventRecord(e->ev1, e->stream));
  HIPCHK(hipEventSynchronize(e->ev1));
  HIPCHK(hipEventElapsedTime(ms, e->ev0, e->ev1));
  return JG_OK;
}

int jg_synth_fill_acks_device(jg_engine* e, uint32_t mode, uint64_t tick, uint64_t* sim_dev, uint64_t* acks_dev) {
  if (!e || !sim_dev || !acks_dev) return fail(JG_EINVAL, "null argument");
  if (mode > 1) return fail(JG_EINVAL, "unknown synth mode");
  HIPCHK(hipSetDevice(e->device));
  hipLaunchKernelGGL(k_synth_acks, dim3(grid_for(e->cfg.n_groups, 4096)), dim3(JG_BLOCK), 0, e->stream, e->dev, mode,
                     tick, sim_dev, acks_dev);
  HIPCHK(hipGetLastError());
  return JG_OK;
}

}  // extern "C"
