// jg_api_drains.h - drains: the finished steps' rows and the device-side queues on their way to the host queues (scan + gather on
// the device, optionally pipelined: jg_drain_prefetch), prepare_rows / the launch of a sparse step.  Part of josefine_gpu.hip's one translation unit.
#pragma once
namespace {
int node_keep_handover(jg_engine* e);  // (jg_api_node.h)
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
  // compact in HBM and a copy engine moves the rows)
  auto grow = [](void*& p, size_t& cap, size_t bytes) -> hipError_t {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    cap = bytes + bytes / 4;
    return hipMalloc(&p, cap);
  };
  HIPCHK(grow(e->d_stage_m, e->stage_m_cap, b.add_m * sizeof(jg_msg_row)));
  HIPCHK(grow(e->d_stage_f, e->stage_f_cap, b.add_f * sizeof(jg_fsm_row)));
  jg_msg_row* dst_m = (jg_msg_row*)e->d_stage_m;
  jg_fsm_row* dst_f = (jg_fsm_row*)e->d_stage_f;
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
  if (b.add_m) HIPCHK(hipMemcpyAsync(qm.p + b.at_m, e->d_stage_m, b.add_m * sizeof(jg_msg_row), hipMemcpyDeviceToHost, st));
  if (b.add_f) HIPCHK(hipMemcpyAsync(qf.p + b.at_f, e->d_stage_f, b.add_f * sizeof(jg_fsm_row), hipMemcpyDeviceToHost, st));
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
  if (e->node.kept_n || e->node.keep) return fail(JG_EINVAL, "jg_drain_prefetch: the engine's node steps keep their outputs (JG_NODE_KEEP) - an engine overlaps its drains one way");
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
  // JG_NODE_KEEP: what a viewed step left is in the queues already (its general-path rows, exceptional rows and faults) or
  // in its landing buffer (the fsm rows of its dense halves: handed over here, a pointer swap when the consumer has taken
  // everything before them); the device is not touched while kept steps are outstanding - their rows are not due yet
  if (e->node.kept_n || e->node.fsm_landed || e->node.spare.fsm_landed) {
    if (release_mask & 1) e->q_msgs.release_view();
    if (release_mask & 2) {
      e->q_fsm.release_view();
      const int rc = node_keep_handover(e);
      if (rc) return rc;
    }
    if (e->node.kept_n) return JG_OK;
  }
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
