// jg_api_engine.h - jg_engine_create / destroy, jg_submit, jg_step, the dense ack ticks and the dense node halves
// (jg_step_dense_leader / _follower).  Part of josefine_gpu.hip's one translation unit.
#pragma once
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
  // (one workgroup per 256 groups, no grid-stride trip, up to 16.7 M groups: round 6's sweep - profiles/r06/headline_grid_sweep.txt -
  // at 16 M x 5 172.1 us with the cap of 8192 the rounds before used, 159.9 us = 0.85 of peak in one pass; at 1 M every cap from 4096 up
  // is the same launch, and a persistent grid of 256 CUs x 8 waves loses: 11.05 against 10.67 us)
  uint32_t cap = env_grid ? (uint32_t)std::atoi(env_grid) : 65536u;
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
  for (void* p : {(void*)e->node.h_beat, (void*)e->node.h_ae, (void*)e->node.h_answer, (void*)e->node.h_hbc, (void*)e->node.h_nsparse, (void*)e->node.h_aec,
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
  if (e->node.bin_mem) (void)hipFree(e->node.bin_mem);
  if (e->node.sp_idx) (void)hipFree(e->node.sp_idx);
  if (e->node.ev_out) (void)hipEventDestroy(e->node.ev_out);
  if (e->node.down) {
    (void)hipStreamSynchronize(e->node.down);
    (void)hipStreamDestroy(e->node.down);
  }
  for (jg_engine::NodeOut* o : {&e->node.spare, &e->node.own()}) {  // (JG_NODE_KEEP)
    if (o == &e->node.spare) {
      for (void* p : {(void*)o->h_beat, (void*)o->h_ae, (void*)o->h_answer, (void*)o->h_hbc, (void*)o->h_nsparse, (void*)o->h_aec})
        if (p) (void)hipHostFree(p);
      if (o->ev_out) (void)hipEventDestroy(o->ev_out);
    }
    if (o->h_status) (void)hipHostFree(o->h_status);
    if (o->ev_early) (void)hipEventDestroy(o->ev_early);
    if (o->ev_kernels) (void)hipEventDestroy(o->ev_kernels);
    o->l_fsm.destroy();
  }
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
  if (at && (e->p_packed || e->p_id32)) return fail(JG_EINVAL, "jg_submit: the step's rows so far were committed with JG_COL_PACKED_KIND / JG_COL_ID32");
  if (!at) e->p_packed = e->p_id32 = false;
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
  l.has_from = e->p_has_from, l.has_term = e->p_has_term, l.has_aux = e->p_has_aux, l.has_flag = e->p_has_flag, l.packed = e->p_packed, l.id32 = e->p_id32;
  size_t off = 0;
  auto sect = [&](size_t bytes) {
    size_t at = off;
    off = (off + bytes + 15) & ~size_t(15);
    return at;
  };
  l.o_id = sect(n * (l.id32 ? 4 : 8)), l.o_term = sect(l.has_term ? n * 8 : 0), l.o_aux = sect(l.has_aux ? n * 8 : 0), l.o_bid = sect(nb * 8);
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
  HIPCHK(up(l.o_id, e->p_id.data(), n * (l.id32 ? 4 : 8)));
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
  if (optional_columns & ~255u) return fail(JG_EINVAL, "unknown column bit");
  const size_t at = e->p_kind.size(), bat = e->p_blk_id.size();
  const bool packed = (optional_columns & JG_COL_PACKED_KIND) != 0;
  if (packed && (!(optional_columns & JG_COL_UNCHECKED) || (optional_columns & (JG_COL_FROM | JG_COL_FLAG))))
    return fail(JG_EINVAL, "JG_COL_PACKED_KIND: the byte carries sender and flag (no JG_COL_FROM / JG_COL_FLAG) and is checked on the device (JG_COL_UNCHECKED)");
  if (at && packed != e->p_packed) return fail(JG_EINVAL, "JG_COL_PACKED_KIND: every commit of a step must agree on the kind column's format");
  const bool id32 = (optional_columns & JG_COL_ID32) != 0;
  if (id32 && !(optional_columns & JG_COL_UNCHECKED)) return fail(JG_EINVAL, "JG_COL_ID32 needs JG_COL_UNCHECKED");
  if (at && (id32 || e->p_id32)) return fail(JG_EINVAL, "JG_COL_ID32: only as the step's one commit (rows are pending / were committed with it)");
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
    if (n) e->p_unchecked = true;  // (an empty commit leaves nothing behind: a later jg_submit + jg_step finds no unchecked row)
  } else {
    int rc = validate_batch(e->cfg.n_groups, &b, &seen);
    if (rc) return rc;
  }
  if ((seen & 1u) && !(optional_columns & JG_COL_AUX)) return fail(JG_EINVAL, "AppendEntries needs id/aux columns");
  e->p_kinds_seen |= seen;
  if (!at) e->p_packed = packed, e->p_id32 = id32;  // (the step's first commit says what the kind / id columns hold; later ones agree)
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
      if ((packed ? b.kind[i] & 15u : b.kind[i]) == JG_CMD_APPEND_ENTRIES) {
        if (id32) ((uint32_t*)e->p_id.p)[at + i] += (uint32_t)bat;
        else e->p_id[at + i] += bat;
      }
  e->p_blk_id.n = e->p_blk_next.n = bat + n_blocks;
  if (optional_columns & JG_COL_UPLOAD_NOW) return upload_rows_now(e);
  e->up.valid = false;  // (rows behind an early upload: the step uploads the whole batch itself)
  return JG_OK;
}

int jg_step(jg_engine* e, uint64_t now_ms) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (e->router) return router_step(e, now_ms);
  {
    int rc = kept_refuse(e);
    if (rc || (rc = node_settle(e))) return rc;
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
    int rc = kept_refuse(e);
    if (rc || (rc = node_settle(e))) return rc;
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
  nd.mask_offers = e->cluster_mask_offers ? 1u : 0u;
  if (out) {
    nd.o_beat = out->beat;
    nd.o_ae = out->ae;
    nd.o_aec = e->cluster_aec;
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
  {
    const int rc = kept_refuse(e);
    if (rc) return rc;
  }
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
  a.aec = e->cluster_aec;
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
