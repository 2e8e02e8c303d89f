// jg_api_node.h - jg_step_node: the dense kernels behind the Apply surface.  Part of josefine_gpu.hip's one translation unit.
#pragma once
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
  A(n.bk_mem, n.bk_words);
#if JG_BLOCK == 256
  n.n_tiles = ((uint32_t)G + JGN_TILE - 1u) >> JGN_TILE_BITS;
  A(n.bin_cnt, (size_t)JGN_BIN_WGS * (n.n_tiles + 1u));
  A(n.bin_off, (size_t)n.n_tiles + 3u);
#endif
#undef A
  n.ready = true;
  return JG_OK;
}

// JG_NODE_KEEP: what a set needs beyond node_ensure's - the spare set's own mirrors (`mirrors`), and for either set the
// pinned words its status snapshot, fsm row count and scan job live in
int node_keep_ensure(jg_engine* e, jg_engine::NodeOut& o, bool mirrors) {
  const size_t G = e->cfg.n_groups, R = e->cfg.n_replicas;
  if (mirrors && !o.h_beat) {
    HIPCHK(hipHostMalloc((void**)&o.h_beat, std::max<size_t>(G * sizeof(jg_leader_beat), 16), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&o.h_ae, std::max<size_t>(R * G * 8, 16), hipHostMallocDefault));
    std::memset(o.h_ae, 0xff, std::max<size_t>(R * G * 8, 16));
    HIPCHK(hipHostMalloc((void**)&o.h_answer, std::max<size_t>(G * 8, 16), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&o.h_hbc, std::max<size_t>(G * 8, 16), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&o.h_nsparse, 16, hipHostMallocDefault));
    std::memset(o.h_nsparse, 0, 16);
    HIPCHK(hipEventCreateWithFlags(&o.ev_out, hipEventDisableTiming));
    int rc = dev_alloc(e, &o.o_ae, R * G);
    if (rc || (rc = dev_alloc(e, &o.o_beat, G)) || (rc = dev_alloc(e, &o.o_answer, G)) || (rc = dev_alloc(e, &o.o_hbc, G))) return rc;
    HIPCHK(hipMemsetAsync(o.o_ae, 0xff, std::max<size_t>(R * G * 8, 16), e->stream));
  }
  if (!o.h_status) {
    char* m = nullptr;
    HIPCHK(hipHostMalloc((void**)&m, 64, hipHostMallocDefault));
    std::memset(m, 0, 64);
    o.h_status = (uint32_t*)m, o.h_total = (uint64_t*)(m + 32), o.h_job = (JgScanJob*)(m + 48);
    HIPCHK(hipEventCreateWithFlags(&o.ev_early, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&o.ev_kernels, hipEventDisableTiming));
  }
  if (!e->node.down) HIPCHK(hipStreamCreateWithFlags(&e->node.down, hipStreamNonBlocking));
  // room up front for what a kept tick brings home and builds: pinning a 24 MB landing buffer or growing an arena by a
  // 150 MB chunk takes milliseconds each, and the two sets and the queue they trade buffers with would pay them one after
  // the other over the engine's first ticks (one fsm row per partition and step is the steady state's yield)
  const size_t rows = G + G / 32 + 4096;
  if (o.l_fsm.cap < rows) HIPCHK(o.l_fsm.reserve(rows));
  if (e->q_fsm.cap < rows && !e->q_fsm.viewed) HIPCHK(e->q_fsm.reserve(rows));
  const size_t build = (size_t)G * 4 + 2 * (size_t)G * JGN_FSM_ROWS * sizeof(jg_fsm_row) + ((G + JG_SCAN_TILE - 1) / JG_SCAN_TILE) * 8 + 4096;
  for (Arena& ar : e->arenas) HIPCHK(ar.reserve(build));  // (an arena in use is left alone)
  return JG_OK;
}
// hands a viewed kept step's fsm rows to the queue jg_drain_applies reads (a pointer swap when the consumer has taken everything before them)
int node_keep_handover(jg_engine* e) {
  jg_engine::NodeStep& nd = e->node;
  for (jg_engine::NodeOut* o : {&nd.spare, &nd.own()}) {
    if (!o->fsm_landed) continue;
    const int rc = handover(e->q_fsm, o->l_fsm, o->fsm_landed);
    if (rc) return rc;
  }
  return JG_OK;
}
int node_keep_tail(jg_engine* e, uint32_t flags);
int node_keep_finish(jg_engine* e, jg_engine::NodeOut& o);

int node_dense_halves(jg_engine* e, uint64_t now_ms, uint32_t flags, uint32_t col_mask, uint32_t sparse_mode, uint64_t* bytes_down);
int node_general(jg_engine* e, const JgNodeRows& rows, size_t n, size_t nb, uint32_t n_sparse, uint64_t now_ms);

int node_step(jg_engine* e, uint64_t now_ms, uint32_t flags) {
  jg_engine::NodeStep& nd = e->node;
  const uint32_t halves = flags & (JG_NODE_LEADER_HALF | JG_NODE_FOLLOWER_HALF);
  // JG_NODE_ASYNC: no synchronisation inside the step - the general-path row count is looked at when the step is settled
  const bool async = (flags & JG_NODE_ASYNC) != 0;
  const bool keep = (flags & JG_NODE_KEEP) != 0;
  HIPCHK(hipSetDevice(e->device));
  int rc = node_ensure(e);
  if (rc) return rc;
  if (keep && !async) return fail(JG_EINVAL, "jg_step_node: JG_NODE_KEEP goes with JG_NODE_ASYNC");
  if (keep && (e->pipelined || e->inflight.phase)) return fail(JG_EINVAL, "jg_step_node: JG_NODE_KEEP or jg_drain_prefetch - an engine overlaps its drains one way");
  if (nd.kept_n >= (keep ? 2u : 1u)) return fail(JG_EINVAL, "jg_step_node: kept steps are outstanding (JG_NODE_KEEP): jg_node_outbox_view first");
  if ((rc = node_settle(e))) return rc;  // (an earlier asynchronous step; a kept one: only its row passes are waited for)
  if ((rc = ensure_xq(e))) return rc;
  if (keep) {
    // the sets change places: this step takes the one that is free (the spare: viewed, or never used), the step before -
    // outstanding or not - keeps its own until the step after this one; with one outstanding the device-side queues and
    // the arena change too (what that step's kernels appended and allocated is collected when its outbox is viewed)
    if ((rc = node_keep_ensure(e, nd.spare, true)) || (rc = node_keep_ensure(e, nd.own(), false))) return rc;
    if ((rc = node_keep_handover(e))) return rc;  // (rows nobody drained since their outbox was viewed: the set's landing buffer is about to be reused)
    std::swap(nd.own(), nd.spare);
    nd.viewed_spare = !nd.viewed_spare;
    if (nd.kept_n) {
      e->cur_set ^= 1, e->cur_arena ^= 1;
      e->fault_floor[e->cur_set] = e->seq;
      e->dev = dev_for_set(e, e->cur_set);
      e->d_dev = e->d_dev2[e->cur_set];
    }
    nd.set = e->cur_set, nd.arena = e->cur_arena;
    nd.d_fsm_cnt = nullptr, nd.d_fsm = nd.d_stage = nullptr, nd.d_bsum = nullptr;
    nd.fsm_copied = 0, nd.l_fsm.n = 0;
    nd.irr_gen = e->irr_gen, nd.seq_lo = e->seq;
  }
  nd.keep = keep;
  if (!keep) nd.viewed_spare = false;  // (this step's set is the one a view shows)
  e->stepped = true;
  struct InStep {
    bool& f;
    explicit InStep(bool& b) : f(b) { f = true; }
    ~InStep() { f = false; }
  } in_step(nd.in_step);
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
  // The row passes, TILED (jg_node.h): the rows binned by tile of 256 partitions, one workgroup per tile with the tile's
  // columns in LDS - prefill, classification and scatter in one kernel, whole lines to and from HBM.  JG_NODE_FLAT=1 (an
  // A/B and the tests' statement of it), a step without rows, or more tiles than the binning's LDS table holds: the flat
  // passes (k_node_prefill + k_node_classify + k_node_route: three random accesses per row and pass).
  static const bool flat_env = std::getenv("JG_NODE_FLAT") != nullptr;
#if JG_BLOCK == 256
  const bool tiled = n && !flat_env && !(e->cfg.flags & JG_CFG_FLAT_ROW_PASSES) && nd.n_tiles + 1u <= 16384u;
#else
  const bool tiled = false;
#endif
  if (!tiled)
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
      bytes_up += n * ((lay.id32 ? 4u : 8u) + 4u + 1u + (has_term ? 8u : 0u) + (has_aux ? 8u : 0u) + (has_from ? 4u : 0u) + (has_flag ? 1u : 0u)) + nb * 16u;
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
    rows.packed = lay.packed ? 1u : 0u, rows.id32 = lay.id32 ? 1u : 0u;
    for (uint32_t r = 0; r < JG_MAX_REPLICAS; r++) rows.ids[r] = r < R ? e->cfg.node_ids[r] : 0u;
    rows.blk_id = (const uint64_t*)(B + o_bid), rows.blk_next = (const uint64_t*)(B + o_bnext), rows.n_blocks = nb;
    const uint32_t rgrid = grid_for(n, 4096);
    HIPCHK(hipMemsetAsync(nd.d_nsparse, 0, 8, e->stream));  // (word 0: the general path's rows; word 1: the binning scan's ticket)
#if JG_BLOCK == 256
    if (tiled) {
      if (nd.bin_cap < n) {
        if (nd.bin_mem) HIPCHK(hipFree(nd.bin_mem));
        nd.bin_cap = n + n / 2;
        HIPCHK(hipMalloc((void**)&nd.bin_mem, nd.bin_cap * 41 + 64));
      }
      JgNodeBin bin{};
      bin.n = (uint32_t)n, bin.n_tiles = nd.n_tiles;
      bin.chunk = (uint32_t)(((n + JGN_BIN_WGS - 1) / JGN_BIN_WGS + JG_BLOCK - 1) / JG_BLOCK * JG_BLOCK);
      bin.n_wg = (uint32_t)((n + bin.chunk - 1) / bin.chunk);
      bin.cnt = nd.bin_cnt, bin.tile_off = nd.bin_off, bin.done = nd.d_nsparse + 1;
      char* m = nd.bin_mem;  // widest first: the 16-byte records, then the optional columns the step has
      const size_t cap = nd.bin_cap;
      bin.rows = rows;
      bin.rows.group = nullptr, bin.rows.kind = nullptr, bin.rows.id = nullptr;  // (in the records)
      bin.rec = (JgNodeBinRec*)m, m += cap * 16;
      bin.rows.term = has_term ? (const uint64_t*)m : nullptr, m += cap * 8;
      bin.rows.aux = has_aux ? (const uint64_t*)m : nullptr, m += cap * 8;
      bin.id_hi = lay.id32 ? nullptr : (uint32_t*)m, m += cap * 4;
      bin.rows.from = has_from ? (const uint32_t*)m : nullptr, m += cap * 4;
      bin.rows.flag = has_flag ? (const uint8_t*)m : nullptr;
      const uint32_t nt1 = bin.n_tiles + 1u;
      if (nt1 <= 4096u) hipLaunchKernelGGL((k_node_bin_count<4096>), dim3(bin.n_wg), dim3(JG_BLOCK), 0, e->stream, rows, bin, G);
      else hipLaunchKernelGGL((k_node_bin_count<16384>), dim3(bin.n_wg), dim3(JG_BLOCK), 0, e->stream, rows, bin, G);
      hipLaunchKernelGGL(k_node_bin_scan, dim3((nt1 + JG_BLOCK / JGN_BIN_SEGS - 1) / (JG_BLOCK / JGN_BIN_SEGS)), dim3(JG_BLOCK), 0, e->stream, bin);
      if (nt1 <= 4096u) hipLaunchKernelGGL((k_node_bin_scatter<4096>), dim3(bin.n_wg), dim3(JG_BLOCK), 0, e->stream, e->dev, rows, bin);
      else hipLaunchKernelGGL((k_node_bin_scatter<16384>), dim3(bin.n_wg), dim3(JG_BLOCK), 0, e->stream, e->dev, rows, bin);
#define JG_LAUNCH_TILE(RR)                                                                                                                      \
  if (halves & JG_NODE_FOLLOWER_HALF)                                                                                                            \
    hipLaunchKernelGGL((k_node_tile<RR, true>), dim3(bin.n_tiles), dim3(JG_BLOCK), 0, e->stream, e->dev, nd.cols, bin, e->uniform_self, halves, \
                       both_beats, col_mask, nd.sp_key, nd.sp_idx, nd.d_nsparse);                                                                \
  else                                                                                                                                           \
    hipLaunchKernelGGL((k_node_tile<RR, false>), dim3(bin.n_tiles), dim3(JG_BLOCK), 0, e->stream, e->dev, nd.cols, bin, e->uniform_self, halves, \
                       both_beats, col_mask, nd.sp_key, nd.sp_idx, nd.d_nsparse)
      switch (R) {
        case 1: JG_LAUNCH_TILE(1); break;
        case 2: JG_LAUNCH_TILE(2); break;
        case 3: JG_LAUNCH_TILE(3); break;
        case 4: JG_LAUNCH_TILE(4); break;
        case 5: JG_LAUNCH_TILE(5); break;
        case 6: JG_LAUNCH_TILE(6); break;
        case 7: JG_LAUNCH_TILE(7); break;
        default: JG_LAUNCH_TILE(8); break;
      }
#undef JG_LAUNCH_TILE
      e->n_launch += 4;
    } else
#endif
    {
      hipLaunchKernelGGL(k_node_classify, dim3(rgrid), dim3(JG_BLOCK), 0, e->stream, e->dev, nd.cols, rows, e->uniform_self,
                         halves, both_beats, col_mask);
      hipLaunchKernelGGL(k_node_route, dim3(rgrid), dim3(JG_BLOCK), 0, e->stream, e->dev, nd.cols, rows, e->uniform_self,
                         both_beats, nd.sp_key, nd.sp_idx, nd.d_nsparse);
      e->n_launch += 3;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(nd.h_nsparse, nd.d_nsparse, 4, hipMemcpyDeviceToHost, e->stream));
    if (keep) HIPCHK(hipEventRecord(nd.ev_early, e->stream));  // (what the NEXT kept step waits for: the row passes, not the halves or the trip home)
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
    e->p_unchecked = e->p_packed = e->p_id32 = false;
  } else {
    e->p_unchecked = e->p_packed = e->p_id32 = false;  // (a step without rows: no format of a batch that is not there outlives it)
  }
  // the dense halves: every partition, the ones whose rows went the general way included (they are ticked here) -
  // except in an asynchronous step, whose halves leave those partitions to the catch-up pass (node_settle)
  pend.now_ms = now_ms, pend.flags = flags, pend.col_mask = col_mask;
  e->seq = seq0 + 1;  // (the general path's number: taken above whether or not it runs)
  uint64_t bytes_down = 0;
  if ((rc = node_dense_halves(e, now_ms, flags, col_mask, async && n ? 1u : 0u, &bytes_down))) return rc;
  if (keep) {
    if ((rc = node_keep_tail(e, flags))) return rc;
  } else {  // fsm_tx rows of the dense halves -> a step record of its own (per-group regions; compacted by the drains)
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
                       rec.d_bsum_f, (flags & JG_NODE_FSM_FUSED) ? 1u : 0u);
    HIPCHK(hipGetLastError());
    e->n_launch++;
    e->recs.push_back(rec);
    pend.fsm_rec_seq = rec.seq;
  }
  pend.seq_general = seq0 + 1, pend.seq_end = e->seq;
  HIPCHK(hipEventRecord(nd.ev_out, keep ? nd.down : e->stream));  // (a kept step: everything of it that travels home is on the down stream, behind its kernels)
  if (keep) nd.seq_hi = e->seq, nd.out = true, nd.kept_n++;
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
  // where the outbox columns travel home: a kept step's on the down stream, behind the kernels that wrote them
  hipStream_t ds = nd.keep ? nd.down : e->stream;
  auto behind_the_kernels = [&]() -> hipError_t {
    if (!nd.keep) return hipSuccess;
    hipError_t err = hipEventRecord(nd.ev_kernels, e->stream);
    return err != hipSuccess ? err : hipStreamWaitEvent(nd.down, nd.ev_kernels, 0);
  };
  if (halves & JG_NODE_LEADER_HALF) {
    JgLeaderNode ln{};
    ln.hbr_commit = nd.cols.hbr_commit;
    ln.packed = 1;
    ln.ack_stride = 1;
    const bool common = tick && (flags & JG_NODE_COMMON_AE);
    if (common && !nd.o_aec && (rc = dev_alloc(e, &nd.o_aec, (size_t)G))) return rc;
    if (common && !nd.h_aec) HIPCHK(hipHostMalloc((void**)&nd.h_aec, std::max<size_t>((size_t)G * 8, 16), hipHostMallocDefault));  // (per set: JG_NODE_KEEP)
    if (tick) ln.o_beat = nd.o_beat, ln.o_ae = nd.o_ae;
    if (common) ln.o_aec = nd.o_aec;
    ln.now = now_ms;
    ln.fsm_delta = nd.cols.fsm_delta, ln.fsm_prev = nd.cols.fsm_prev, ln.fsm_mid = nd.cols.fsm_mid;
    ln.arr = nd.cols.arr, ln.col_mask = col_mask;  // (the slow kernel replays its groups in arrival order)
    if (sparse_mode) ln.sparse_bits = nd.cols.sparse_bits, ln.sparse_mode = sparse_mode;
    if ((rc = dense_step(e, nd.cols.answers, 1, &ln))) return rc;
    if (common) {
      // one word per partition; the rows only if some partition's words differ by addressee (fetched by
      // jg_node_outbox_view, which sees the count: none in the steady state)
      HIPCHK(hipMemsetAsync(nd.d_nsparse + 2, 0, 4, e->stream));
      hipLaunchKernelGGL(k_node_count_individual, dim3(grid_for(G, 1024)), dim3(JG_BLOCK), 0, e->stream, (const uint64_t*)nd.o_aec, G, nd.d_nsparse + 2);
      HIPCHK(hipGetLastError());
      e->n_launch++;
      HIPCHK(hipMemcpyAsync(nd.h_nsparse + 2, nd.d_nsparse + 2, 4, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(behind_the_kernels());
      HIPCHK(hipMemcpyAsync(nd.h_beat, nd.o_beat, (size_t)G * sizeof(jg_leader_beat), hipMemcpyDeviceToHost, ds));
      HIPCHK(hipMemcpyAsync(nd.h_aec, nd.o_aec, (size_t)G * 8, hipMemcpyDeviceToHost, ds));
      nd.ae_rows_landed = false;
      *bytes_down += (size_t)G * (sizeof(jg_leader_beat) + 8) + 4;
    } else if (tick) {
      nd.ae_rows_landed = true;
      HIPCHK(behind_the_kernels());
      HIPCHK(hipMemcpyAsync(nd.h_beat, nd.o_beat, (size_t)G * sizeof(jg_leader_beat), hipMemcpyDeviceToHost, ds));
      // (the own slot's row is JG_NO_ACK on both sides and stays there: not written, not downloaded)
      const uint32_t own = e->uniform_self >= 0 ? (uint32_t)e->uniform_self : R;
      if (own > 0) HIPCHK(hipMemcpyAsync(nd.h_ae, nd.o_ae, (size_t)std::min(own, R) * G * 8, hipMemcpyDeviceToHost, ds));
      if (own + 1 < R)
        HIPCHK(hipMemcpyAsync(nd.h_ae + (size_t)(own + 1) * G, nd.o_ae + (size_t)(own + 1) * G, (size_t)(R - own - 1) * G * 8, hipMemcpyDeviceToHost,
                              ds));
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
    HIPCHK(behind_the_kernels());
    HIPCHK(hipMemcpyAsync(nd.h_answer, nd.o_answer, (size_t)G * 8, hipMemcpyDeviceToHost, ds));
    HIPCHK(hipMemcpyAsync(nd.h_hbc, nd.o_hbc, (size_t)G * 8, hipMemcpyDeviceToHost, ds));
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
  if (nd.in_step) return JG_OK;  // (the halves of the step that is running)
  jg_engine::NodeStep::Pending& pd = nd.pending;
  if (!pd.on) return JG_OK;
  pd.on = false;
  HIPCHK(hipSetDevice(e->device));
  // (a kept step: its row passes are what the count depends on - the halves, the fsm rows and the trip home go on)
  if (nd.keep) HIPCHK(hipEventSynchronize(nd.ev_early));
  else HIPCHK(hipStreamSynchronize(e->stream));
  const uint32_t n_sparse = nd.h_nsparse[0];
  nd.last.rows_general = n_sparse;
  if (!n_sparse) return JG_OK;
  if (nd.keep) HIPCHK(hipStreamSynchronize(e->stream));
  struct InStep {
    bool& f;
    explicit InStep(bool& b) : f(b) { f = true; }
    ~InStep() { f = false; }
  } in_step(nd.in_step);
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
  if (nd.keep) {  // the step's fsm rows once more, from the deltas the catch-up pass completed
    if ((rc = node_keep_tail(e, pd.flags))) return rc;
    HIPCHK(hipEventRecord(nd.ev_out, nd.down));
  }
  for (StepRec& rec : e->recs)
    if (!nd.keep && rec.seq == pd.fsm_rec_seq && rec.fsm_per_row == JGN_FSM_ROWS && rec.msg_per_row == 0) {
      const uint32_t n_tiles = (e->cfg.n_groups + JG_SCAN_TILE - 1) / JG_SCAN_TILE;
      hipLaunchKernelGGL(k_node_fsm_build, dim3(n_tiles), dim3(JG_BLOCK), 0, e->stream, e->dev, nd.cols, rec.d_fsm, rec.d_fsm_cnt, rec.d_bsum_f,
                         (pd.flags & JG_NODE_FSM_FUSED) ? 1u : 0u);
      e->n_launch++;
    }
  HIPCHK(hipGetLastError());
  e->seq = seq_end;
  HIPCHK(hipStreamSynchronize(e->stream));
  return JG_OK;
}

// JG_NODE_KEEP: the fsm_tx rows of the step's dense halves, compacted and on their way home behind the step's own kernels -
// what jg_drain_applies does for a step record (scan of the tile sums, gather, one copy), enqueued HERE, the copy sized
// from the kept step finished last (the row count of a tick wobbles by a few percent at most; whatever is missing is
// fetched when the outbox is viewed) - then the status block as the step leaves it.  No drain is issued by the host.
int node_keep_tail(jg_engine* e, uint32_t flags) {
  jg_engine::NodeStep& nd = e->node;
  const uint32_t G = e->cfg.n_groups;
  const uint32_t n_tiles = (G + JG_SCAN_TILE - 1) / JG_SCAN_TILE;
  const size_t cap = (size_t)G * JGN_FSM_ROWS;
  if (!nd.d_fsm) {
    Arena& ar = e->arenas[nd.arena];
    HIPCHK(ar.alloc((size_t)G * 4, (void**)&nd.d_fsm_cnt));
    HIPCHK(ar.alloc(cap * sizeof(jg_fsm_row), (void**)&nd.d_fsm));
    HIPCHK(ar.alloc((size_t)n_tiles * 8, (void**)&nd.d_bsum));
    HIPCHK(ar.alloc(cap * sizeof(jg_fsm_row), (void**)&nd.d_stage));
  }
  hipLaunchKernelGGL(k_node_fsm_build, dim3(n_tiles), dim3(JG_BLOCK), 0, e->stream, e->dev, nd.cols, nd.d_fsm, nd.d_fsm_cnt, nd.d_bsum,
                     (flags & JG_NODE_FSM_FUSED) ? 1u : 0u);
  nd.h_job[0] = JgScanJob{nd.d_bsum, n_tiles, 0};
  hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(JG_BLOCK), 0, e->stream, (const JgScanJob*)nd.h_job, nd.h_total);
  hipLaunchKernelGGL(k_scan_gather<jg_fsm_row>, dim3(n_tiles), dim3(JG_BLOCK), 0, e->stream, nd.d_fsm_cnt, G, nd.d_bsum, (uint32_t)JGN_FSM_ROWS, nd.d_fsm,
                     nd.d_stage);
  HIPCHK(hipGetLastError());
  e->n_launch += 3;
  const size_t last = nd.fsm_guess_known ? nd.fsm_guess : (size_t)G;  // (no kept step finished yet: a row per partition)
  const size_t guess = std::min(cap, last + last / 32 + 4096);  // (3 % of room: every byte crosses the bus)
  nd.l_fsm.n = 0;
  // (the status block is the step's own snapshot: taken on the step's stream, before a newer step's kernels touch it)
  HIPCHK(hipMemcpyAsync(nd.h_status, e->d_status, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipEventRecord(nd.ev_kernels, e->stream));
  HIPCHK(hipStreamWaitEvent(nd.down, nd.ev_kernels, 0));
  if (guess) {
    HIPCHK(nd.l_fsm.reserve(guess));
    HIPCHK(hipMemcpyAsync(nd.l_fsm.p, nd.d_stage, guess * sizeof(jg_fsm_row), hipMemcpyDeviceToHost, nd.down));
  }
  nd.fsm_copied = guess;
  return JG_OK;
}

// JG_NODE_KEEP: the outbox of kept step `o` is being viewed - wait for ITS outputs (not for a newer step's), surface what
// its status block says, collect what only a step outside the steady state leaves behind (general-path records,
// exceptional rows, faults: the synchronous drain, over this step's queues and arena), and make its fsm rows the ones
// the next drain of that queue delivers.
int node_keep_finish(jg_engine* e, jg_engine::NodeOut& o) {
  jg_engine::NodeStep& nd = e->node;
  HIPCHK(hipSetDevice(e->device));
  int rc = JG_OK;
  if (&o == &nd.own() && (rc = node_settle(e))) return rc;  // (the newest step: nobody has looked at its row count yet)
  HIPCHK(hipEventSynchronize(o.ev_out));
  const uint32_t* st = o.h_status;
  if ((rc = status_check(e, st))) return rc;
  if (e->flag_check_pending && o.irr_gen == e->irr_gen) {  // (as a pipelined drain's snapshot: no step since could have left an irregular chain)
    e->maybe_irregular = st[1] != 0;
    e->flag_check_pending = false;
  }
  if (st[5]) e->maybe_irregular = true;
  if (!e->slow_scheduled_ever && st[2]) return fail(JG_EDEVICE, "internal: irregular chain reached the fast-only dense path");
  const uint32_t nf = st[o.set ? 6 : 3], nx = st[o.set ? 7 : 4];
  if ((rc = node_keep_handover(e))) return rc;  // (the step before's rows, if nobody drained them: they come first)
  // (records of THIS step and of what came before it: a newer kept step that somebody settled meanwhile - a read of the
  // engine - keeps its own for its own view)
  size_t nrec = 0;
  while (nrec < e->recs.size() && e->recs[nrec].seq <= o.seq_hi) nrec++;
  if (nrec || nf || nx) {
    // outside the steady state: everything in flight first (the newer step's kernels too), then the synchronous drain
    HIPCHK(hipStreamSynchronize(e->stream));
    jg_engine::DrainBatch b;
    b.set = o.set;
    b.seq_hi = o.seq_hi;
    b.nf = nf, b.nx = nx;
    std::vector<StepRec> mine(e->recs.begin(), e->recs.begin() + nrec);
    e->recs.erase(e->recs.begin(), e->recs.begin() + nrec);
    if ((rc = drain_scan(e, mine, e->stream))) return rc;
    if (nrec) HIPCHK(hipStreamSynchronize(e->stream));
    if ((rc = drain_gather(e, b, mine, e->stream))) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    if ((rc = drain_finish(e, b, mine, e->arenas[o.arena]))) return rc;
    e->fault_floor[b.set] = b.seq_hi;
  }
  const size_t total = (size_t)*o.h_total;
  if (total > o.fsm_copied) {  // (the copy behind the step was sized from the step before: the rest now)
    o.l_fsm.n = o.fsm_copied;
    HIPCHK(o.l_fsm.reserve(total));
    HIPCHK(hipMemcpyAsync(o.l_fsm.p + o.fsm_copied, o.d_stage + o.fsm_copied, (total - o.fsm_copied) * sizeof(jg_fsm_row), hipMemcpyDeviceToHost,
                          e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  o.l_fsm.n = total, o.fsm_landed = total != 0;
  if (e->track_segs) seg_add(e->seg_f, o.seq_hi, total);
  nd.fsm_guess = total, nd.fsm_guess_known = true;
  // everything the step allocated has been read: its arena starts over (records, if it had any, were drained above)
  e->arenas[o.arena].reset();
  o.d_fsm_cnt = nullptr, o.d_fsm = o.d_stage = nullptr, o.d_bsum = nullptr;
  o.out = false;
  nd.kept_n--;
  return JG_OK;
}
}  // namespace

int jg_step_node(jg_engine* e, uint64_t now_ms, uint32_t flags) {
  if (!e) return fail(JG_EINVAL, "null argument");
  if (!(flags & (JG_NODE_LEADER_HALF | JG_NODE_FOLLOWER_HALF)) || (flags & ~127u))
    return fail(JG_EINVAL, "jg_step_node: flags = JG_NODE_LEADER_HALF and / or JG_NODE_FOLLOWER_HALF [| JG_NODE_TICK] [| JG_NODE_ASYNC] [| JG_NODE_COMMON_AE] [| JG_NODE_FSM_FUSED] [| JG_NODE_KEEP]");
  if (e->router && (flags & (JG_NODE_COMMON_AE | JG_NODE_KEEP))) return fail(JG_EINVAL, "jg_step_node: JG_NODE_COMMON_AE and JG_NODE_KEEP are per shard (jg_get_shard)");
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
  if (!nd.ready || !(nd.last_flags | nd.spare.last_flags)) return fail(JG_EINVAL, "no jg_step_node yet");
  // JG_NODE_KEEP: the OLDEST step whose outbox has not been viewed (two outstanding: the spare set's); none outstanding:
  // the one viewed last, again
  jg_engine::NodeOut& o = nd.kept_n == 2 || (nd.kept_n == 0 && nd.viewed_spare) ? nd.spare : nd.own();
  int rc = JG_OK;
  if (o.out) {
    if ((rc = node_keep_finish(e, o))) return rc;  // (waits for THIS step's outputs only)
    nd.viewed_spare = &o == &nd.spare;
  } else if (!nd.kept_n) {
    if ((rc = sync_and_check(e))) return rc;  // (the columns have landed; device-side error flags surface here)
  }
  *out = o.last;
  if ((o.last_flags & JG_NODE_LEADER_HALF) && (o.last_flags & JG_NODE_TICK)) {
    out->beat = o.h_beat, out->ae = o.h_ae;
    if (o.last_flags & JG_NODE_COMMON_AE) {
      out->aec = o.h_aec, out->ae = nullptr;
      if (o.h_nsparse[2]) {  // some partition's words differ by addressee: the rows are wanted after all
        if (!o.ae_rows_landed) {
          const uint32_t G = e->cfg.n_groups, R = e->cfg.n_replicas;
          const uint32_t own = e->uniform_self >= 0 ? (uint32_t)e->uniform_self : R;
          HIPCHK(hipSetDevice(e->device));
          // (a kept step: its own copy of the rows - a newer step has written the other set's since)
          if (own > 0) HIPCHK(hipMemcpyAsync(o.h_ae, o.o_ae, (size_t)std::min(own, R) * G * 8, hipMemcpyDeviceToHost, e->stream));
          if (own + 1 < R)
            HIPCHK(hipMemcpyAsync(o.h_ae + (size_t)(own + 1) * G, o.o_ae + (size_t)(own + 1) * G, (size_t)(R - own - 1) * G * 8, hipMemcpyDeviceToHost,
                                  e->stream));
          HIPCHK(hipStreamSynchronize(e->stream));
          o.last.bytes_d2h += (size_t)G * (size_t)(own < R ? R - 1 : R) * 8;
          out->bytes_d2h = o.last.bytes_d2h;
          o.ae_rows_landed = true;
        }
        out->ae = o.h_ae;
      }
    }
  }
  if (o.last_flags & JG_NODE_FOLLOWER_HALF) out->answer = o.h_answer, out->hb_commit = o.h_hbc;
  return JG_OK;
}
