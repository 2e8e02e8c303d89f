// jg_api_routed.h - jg_dense_cluster_round_routed: a protocol round with the device-side transport (rows, and the election
// vocabulary as mailbox words).  Part of josefine_gpu.hip's one translation unit.
#pragma once
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
  // (nodes that share a device issue their work on the lead node's stream while clustered: jg_dense_cluster_create)
  for (jg_engine* e : c->nodes)
    if (e->stream != L->stream) return fail(JG_EINVAL, "routed rounds take nodes on the cluster's stream");
  // JG_CLUSTER_OPT_VOTE_WORDS (jg_dense_cluster_set_option; jg_votes.h): an election's traffic travels as mailbox words - the campaigns' broadcasts are
  // counted into request words by a census of the emitted rows and are not staged, the answers are written as words by the
  // receiving half (k_vote_half_multi) and never become rows - wherever EVERYTHING a node receives for a partition in a
  // round is such words; every other partition's mail travels as rows, as without the switch.  Fixed for a cluster's life.
  const bool vwords = rt.vote_words && R >= 2;
  if (vwords && !rt.vm_mem) {
    const size_t RG = (size_t)R * c->G, wd = ((size_t)c->G + 63) / 64;
    const size_t per = RG * sizeof(JgVoteRec) + 2 * R * wd * 8;
    HIPCHK(hipMalloc(&rt.vm_mem, 2 * per));
    HIPCHK(hipMemsetAsync(rt.vm_mem, 0, 2 * per, L->stream));  // (the first round reads the mail of a round that never was: none)
    char* p = (char*)rt.vm_mem;
    for (int k = 0; k < 2; k++) {
      JgVoteMail& m = rt.vm[k];
      m.R = R, m.G = c->G, m.words = (uint32_t)wd;
      m.rec = (JgVoteRec*)p, p += RG * sizeof(JgVoteRec);
      m.rowmail = (uint64_t*)p, p += R * wd * 8;
      m.wordmail = (uint64_t*)p, p += R * wd * 8;
    }
  }
  const JgVoteMail vprev = rt.vm[rt.vm_turn ^ 1u], vcur = rt.vm[rt.vm_turn];  // (last round's mail is read, this round's filled)
  if (!rt.h_jobs) {
    HIPCHK(hipHostMalloc((void**)&rt.h_jobs, 6 * jg_dense_cluster::Route::JOB_SLICE, hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&rt.d_jobs, 6 * jg_dense_cluster::Route::JOB_SLICE));
  }
  static_assert(JG_MAX_REPLICAS * sizeof(JgApplyJob) <= jg_dense_cluster::Route::JOB_SLICE, "job slice too small");
  static_assert(JG_MAX_REPLICAS * sizeof(JgFollowerJob) <= jg_dense_cluster::Route::JOB_SLICE, "job slice too small");
  static_assert(sizeof(JgVoteHalfJobs) + 2 * sizeof(JgVoteMail) <= 4096, "kernel arguments");
  auto slice_h = [&](int k) { return rt.h_jobs + (size_t)k * jg_dense_cluster::Route::JOB_SLICE; };
  auto slice_d = [&](int k) { return rt.d_jobs + (size_t)k * jg_dense_cluster::Route::JOB_SLICE; };
  // (the job tables of the whole round - both sparse steps, the follower halves, the delivering pass - are written
  // first and travel in ONE copy: every table is host bookkeeping only, and a copy costs ~10 us of stream time)
  auto small_tiles = [&](uint32_t widest) { return widest <= JG_RUN_SMALL_BATCH; };  // (256-row tiles for small batches: four times the workgroups)
  auto run_grid = [&](uint32_t widest) {
    const uint32_t tile = small_tiles(widest) ? JG_RUN_TILE_SMALL : JG_RUN_TILE;
    return std::min<uint32_t>(std::max<uint32_t>((widest + tile - 1) / tile, 1u), L->count_slots);
  };
  auto apply_all = [&](int slice, std::vector<JgApplyJob>& jobs, uint32_t widest, hipStream_t on) -> int {
    if (jobs.empty()) return JG_OK;
    if (slice == 0) {  // delivered rows: runs of 4-16 rows per group, a RUN per lane (jg_apply_runs_body)
      hipLaunchKernelGGL(small_tiles(widest) ? k_apply_runs_multi_small : k_apply_runs_multi, dim3(run_grid(widest), (uint32_t)jobs.size()),
                         dim3(JG_BLOCK), 0, on, (const JgApplyJob*)slice_d(slice));
    } else {
      hipLaunchKernelGGL(k_apply_rows_multi, dim3(grid_for(widest, L->count_slots), (uint32_t)jobs.size()), dim3(JG_BLOCK), 0, on,
                         (const JgApplyJob*)slice_d(slice));
    }
    HIPCHK(hipGetLastError());
    return JG_OK;
  };
  std::vector<uint32_t> seq_base(R);
  for (uint32_t n = 0; n < R; n++) seq_base[n] = c->nodes[n]->seq;
  // which of a node's steps of this round is which PHASE (jg_route.h: the transport orders a partition's mail by phase,
  // emission index, sender): noted as the steps are numbered below
  std::vector<uint32_t> phases(R, 0);
  auto note_phase = [&](uint32_t n, uint32_t phase) {
    const uint32_t step = c->nodes[n]->seq - seq_base[n];  // (the step that has just been numbered)
    if (step < 8) phases[n] |= phase << (3u * step);
  };
  *started = true;
  // (jobs_v: delivered batches that hold an election's traffic only - the transport's census says so - take the
  // kernel without the chain code: k_apply_vote_runs_multi)
  std::vector<JgApplyJob> jobs_a, jobs_v, jobs_b;
  uint32_t widest_a = 0, widest_v = 0, widest_b = 0;
  {
    for (uint32_t n = 0; n < R; n++) {
      const bool votes = jg_kinds_within(rt.kinds_in[n], JG_KINDS_ELECTION);
      std::vector<JgApplyJob>& jobs = votes ? jobs_v : jobs_a;
      uint32_t& widest = votes ? widest_v : widest_a;
      jg_engine* e = c->nodes[n];
      if (!rt.n_in[n]) {
        if (vwords) e->stepped = true, e->seq++, note_phase(n, JG_ROUTE_PHASE_DELIVERED);  // (the receiving half of the vote mail is this step too: it has a number on every node)
        continue;
      }
      const size_t o = rt.in_off[n];
      e->stepped = true;
      e->seq++;
      note_phase(n, JG_ROUTE_PHASE_DELIVERED);
      JgApplyJob j{};
      j.d = e->dev;
      // VoteRequest -> one VoteResponse; VoteResponse -> DROP + Heartbeat on elect() (candidate.rs:108-113): two slots per row
      const bool two = votes && jg_kinds_within(rt.kinds_in[n], (1u << JG_CMD_VOTE_REQUEST) | (1u << JG_CMD_VOTE_RESPONSE));
      if ((rc = prepare_rows(e, rt.n_in[n], rt.cols.group + o, rt.cols.kind + o, rt.cols.from + o, rt.cols.term + o, rt.cols.id + o,
                             rt.cols.aux + o, rt.cols.flag + o, nullptr, nullptr, 0, now_ms, &j.a, two ? 2u : 0u)))
        return rc;
      widest = std::max(widest, j.a.n);
      jobs.push_back(j);
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
      note_phase(n, JG_ROUTE_PHASE_INJECTED);
      JgApplyJob j{};
      j.d = e->dev;
      const uint64_t* none = (const uint64_t*)e->d_ones;  // (injected rows carry no blocks: checked above)
      if ((rc = prepare_rows(e, (uint32_t)b.n, b.group, b.kind, b.from, b.term, b.id, b.aux, b.flag, none, none, 0, now_ms, &j.a))) return rc;
      widest = std::max(widest, j.a.n);
      jobs.push_back(j);
    }
  }
  std::vector<JgFollowerJob> fjobs;
  if (c->any) {
    if ((rc = cluster_tables_any(c, now_ms, false))) return rc;  // (its own copy, behind the sparse steps' tables)
    for (uint32_t n = 0; n < R; n++) {  // (every node's two halves: seq - 1 and seq)
      const uint32_t step = c->nodes[n]->seq - seq_base[n];
      if (step < 8) phases[n] |= JG_ROUTE_PHASE_LEADER << (3u * (step - 1u)) | JG_ROUTE_PHASE_FOLLOWER << (3u * step);
    }
  } else {
    if ((rc = cluster_follower_jobs(c, now_ms, fjobs))) return rc;
    for (uint32_t n = 0; n < R; n++)
      if (n != c->lead) note_phase(n, JG_ROUTE_PHASE_FOLLOWER);
    // (the lead node's leader half is numbered by jg_step_dense_leader inside cluster_round_body: its next step)
    if (L->seq - seq_base[c->lead] + 1u < 8) phases[c->lead] |= JG_ROUTE_PHASE_LEADER << (3u * (L->seq - seq_base[c->lead] + 1u));
  }
  hipStream_t st = L->stream;
  uint32_t* d_cursor = rt.d_count + (size_t)R * ROUTE_WORDS;
  uint32_t* d_keep_n = d_cursor + JG_ROUTE_SEGS;
  // the staging in segments, one cursor each (jg_route_reserve)
  const uint32_t n_seg = JG_ROUTE_SEGS;
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
    const size_t bk_words = (size_t)n_tiles * JG_ROUTE_SCAN_TILE + bk.n_buckets + 1 + n_tiles + 1;  // hist (whole tiles) | cur | done | tile
    if (rt.bk_cap < bk_words) {
      if (rt.bk_hist) HIPCHK(hipFree(rt.bk_hist));
      rt.bk_cap = (uint32_t)bk_words;
      HIPCHK(hipMalloc((void**)&rt.bk_hist, bk_words * 4));
    }
    bk.hist = rt.bk_hist, bk.cur = bk.hist + (size_t)n_tiles * JG_ROUTE_SCAN_TILE, bk.done = bk.cur + bk.n_buckets, bk.tile = bk.done + 1;
  }
  const uint32_t bk_clear = n_tiles * JG_ROUTE_SCAN_TILE + bk.n_buckets + 1;  // counts, cursors and the scan's ticket
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
          j.t = t, j.n = r.n, j.per_row = r.msg_per_row, j.step = jg_route_phase(phases[s], r.seq - seq_base[s]);
          j.msg_cnt = r.d_msg_cnt, j.msg = r.d_msg, j.fsm_cnt = r.d_fsm_cnt;
          rjobs.push_back(j);
          widest_r = std::max(widest_r, r.n);
        }
      JgRouteXqJob j{};
      j.t = t, j.xq = e->dev.xq, j.xq_n = e->dev.xq_n, j.xq_cap = e->dev.xq_cap, j.seq_base = seq_base[s], j.phases = phases[s];
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
      j.d = e->dev, j.self = n, j.seq = seq_base[n] + 1u, j.step = JG_ROUTE_PHASE_DELIVERED, j.need = R - 1u, j.now = now_ms;
      if (e->seq < j.seq) return fail(JG_EDEVICE, "internal: routed round: the delivered step has no number");
    }
  }
  bool cleared = false;  // (the transport's tallies and bucket counters: zeroed with the round's mail, or by the first attempt's own launch)
  static const bool clear_early = std::getenv("JG_ROUTE_CLEAR_AT_HEAD") == nullptr;
  {  // slices 0-3 in one copy - by a kernel out of the pinned staging (a copy engine's start-up was the round's largest gap)
    if (!jobs_a.empty()) std::memcpy(slice_h(0), jobs_a.data(), jobs_a.size() * sizeof(JgApplyJob));
    if (!jobs_v.empty()) std::memcpy(slice_h(0) + jobs_a.size() * sizeof(JgApplyJob), jobs_v.data(), jobs_v.size() * sizeof(JgApplyJob));
    if (!jobs_b.empty()) std::memcpy(slice_h(1), jobs_b.data(), jobs_b.size() * sizeof(JgApplyJob));
    if (!fjobs.empty()) std::memcpy(slice_h(2), fjobs.data(), fjobs.size() * sizeof(JgFollowerJob));
    const uint32_t n8 = (uint32_t)(4 * jg_dense_cluster::Route::JOB_SLICE / 8);
    if (vwords) {
      // ... the kernel that clears this round's mail where it was written two rounds ago (a workgroup per chunk of the bitmaps) -
      // and the tallies with it (until round 6: three launches)
      // (the mail itself - this round's to fill - was cleared beside the injected rows' step of the round before, where the
      // receiving half had just read it: clear_early below; JG_ROUTE_CLEAR_AT_HEAD=1: here, as until the end of round 6)
      JgVoteMail m = vcur;
      if (clear_early) m.words = 0;
      hipLaunchKernelGGL(k_votes_clear, dim3(clear_early ? 32u : (vcur.words + JG_VOTE_CHUNK - 1) / JG_VOTE_CHUNK), dim3(JG_BLOCK), 0, st, m, rt.d_count, (uint32_t)words,
                         bk.hist, bk_clear, (uint64_t*)slice_d(0), (const uint64_t*)slice_h(0), n8);
      cleared = true;
    } else {
      hipLaunchKernelGGL(k_copy_words, dim3((n8 + JG_BLOCK - 1) / JG_BLOCK), dim3(JG_BLOCK), 0, st, (uint64_t*)slice_d(0), (const uint64_t*)slice_h(0), n8);
    }
    HIPCHK(hipGetLastError());
  }
  // -- 1. (launches) what the transport delivered last round, then this round's injected rows.  (With the vote mail the
  // delivered ROWS are other partitions' than the words', so their step could run BESIDE the receiving half on a stream of its
  // own: tried in round 6 and removed - the two cross-queue dependencies cost 12-14 us more than the 34 us they hide,
  // profiles/r06/ab_side_stream_sort_buckets.txt.)
  uint32_t vote_grid = 1;
  if (vwords) {
    uint32_t slots = L->count_slots;
    for (jg_engine* e : c->nodes) slots = std::min(slots, e->count_slots);
    const uint32_t n_chunks = (vprev.words + JG_VOTE_CHUNK - 1) / JG_VOTE_CHUNK;  // (a workgroup per chunk of the bitmap; its counter slot is blockIdx.x)
    vote_grid = std::max(1u, std::min(n_chunks, slots));
  }
  // (JG_ROUTE_SPLIT_HEAD=1: the receiving half and the delivered rows' step as launches of their own, one behind the other, as
  // until the end of round 6 - the A/B)
  static const bool split_head = std::getenv("JG_ROUTE_SPLIT_HEAD") != nullptr;
  if (vwords && !split_head && !jobs_a.empty() && jobs_v.empty() && small_tiles(widest_a)) {
    // the vote mail's receiving half and the delivered rows' step side by side in ONE launch (k_round_head_multi)
    hipLaunchKernelGGL(k_round_head_multi, dim3(std::max(vote_grid, run_grid(widest_a)), R + (uint32_t)jobs_a.size()), dim3(JG_BLOCK), 0, L->stream, vjobs, R, vprev,
                       vcur, (const JgApplyJob*)slice_d(0));
    HIPCHK(hipGetLastError());
  } else {
    if ((rc = apply_all(0, jobs_a, widest_a, L->stream))) return rc;
    if (!jobs_v.empty()) {  // (different nodes than jobs_a's: the two launches are independent of each other)
      hipLaunchKernelGGL(small_tiles(widest_v) ? k_apply_vote_runs_multi_small : k_apply_vote_runs_multi,
                         dim3(run_grid(widest_v), (uint32_t)jobs_v.size()), dim3(JG_BLOCK), 0, L->stream,
                         (const JgApplyJob*)slice_d(0) + jobs_a.size());
      HIPCHK(hipGetLastError());
    }
    if (vwords) {  // (partitions other than the rows': a partition's mail of a round is words or rows, never both)
      hipLaunchKernelGGL(k_vote_half_multi, dim3(vote_grid, R), dim3(JG_BLOCK), 0, L->stream, vjobs, vprev, vcur);
      HIPCHK(hipGetLastError());
    }
  }
  if (vwords && clear_early) {
    // the injected rows' step with the clearing of the mail the receiving half has just read beside it (the next round's to fill)
    const uint32_t n_chunks = (vprev.words + JG_VOTE_CHUNK - 1) / JG_VOTE_CHUNK;
    const uint32_t gx = jobs_b.empty() ? n_chunks : std::max<uint32_t>(grid_for(widest_b, L->count_slots), std::min<uint32_t>(n_chunks, 64u));
    hipLaunchKernelGGL(k_apply_rows_clear_multi, dim3(std::min<uint32_t>(gx, L->count_slots), (uint32_t)jobs_b.size() + 1u), dim3(JG_BLOCK), 0, L->stream,
                       (const JgApplyJob*)slice_d(1), (uint32_t)jobs_b.size(), vprev);
    HIPCHK(hipGetLastError());
  } else if ((rc = apply_all(1, jobs_b, widest_b, L->stream))) {
    return rc;
  }
  // -- 2. the dense round; ClientRequests only where the lead node (still) leads
  if (c->any) {  // (whoever owns a group reads `offered`: nothing to mask)
    if ((rc = cluster_launch_any(c))) return rc;
  } else {
    // (ClientRequests only where the lead node - still - leads: its own slot's column holds the offers as they are, and the
    // leader half does not look at a non-leader's - JgLeaderNode::mask_offers; until round 6 a launch of its own wrote a masked
    // copy of the column every round)
    L->cluster_mask_offers = true;
    rc = cluster_round_body(c, now_ms, true, false, slice_h(2), slice_d(2), &fjobs);
    L->cluster_mask_offers = false;
    if (rc) return rc;
  }
  T1 = clk();
  // -- 3. the transport, on the lead node's stream behind everybody's round
  for (uint32_t r = 0; r < R; r++)
    if (r != c->lead && (rc = jg_stream_wait(L, c->nodes[r]))) return rc;
  // The ordering pass (bucket by (destination, group tile), sort every bucket in LDS: jg_route.h) takes everything it
  // needs to know about the round's rows from the device - the segments' cursors - so it is launched BEHIND the
  // delivering pass before the host has seen the counts: the host then waits for the counts' copy only (an event),
  // with the ordering still queued, and does its bookkeeping and the next round's preparation while the device
  // works.  (Waiting first left the device idle for the wake-up and the five launches: profiles/r04/ab_route_order.txt.)
  // A pass that has to be repeated (staging too small, emission index too wide) repeats the ordering with it.
  const bool optimistic = rt.last_total != 0;
  if (!rt.ev_counts) HIPCHK(hipEventCreateWithFlags(&rt.ev_counts, hipEventDisableTiming));
  auto launch_order = [&](uint32_t fullest_seg) {
    bk.shift = ord_bits + 3 + JG_ROUTE_STEP_BITS + tile_bits;  // (its counters were cleared with the tallies, before the delivering pass)
    const uint32_t seg_cap = rt.cap / n_seg;
    const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>((std::min(fullest_seg, seg_cap) + JG_BLOCK - 1) / JG_BLOCK, 4096 / n_seg));
    hipLaunchKernelGGL(k_route_hist, dim3(grid, n_seg), dim3(JG_BLOCK), 0, st, (const uint32_t*)d_cursor, seg_cap, (const uint64_t*)rt.key, bk);
    JgRouteXqDone xd{};  // (the queues the delivering pass delivered whole are emptied by the scan's last workgroup: no launch of their own)
    xd.R = R, xd.route_words = ROUTE_WORDS, xd.n_seg = n_seg, xd.seg_cap = seg_cap, xd.count = rt.d_count, xd.cursor = d_cursor;
    for (uint32_t s = 0; s < R; s++) xd.xq_n[s] = c->nodes[s]->dev.xq_n;
    hipLaunchKernelGGL(k_route_scan_all, dim3(n_tiles), dim3(JG_BLOCK), 0, st, bk, xd);
    hipLaunchKernelGGL(k_route_scatter, dim3(grid, n_seg), dim3(JG_BLOCK), 0, st, (const uint32_t*)d_cursor, seg_cap, (const uint64_t*)rt.key,
                       (const uint32_t*)rt.idx, bk, rt.key_alt, rt.idx_alt);
    const uint32_t per_wg = JG_ROUTE_SORT_BUCKETS;
    hipLaunchKernelGGL(k_route_sort_build, dim3((bk.n_buckets + per_wg - 1) / per_wg), dim3(JG_BLOCK), 0, st, bk, rt.key_alt, rt.idx_alt,
                       (const jg_msg_row*)rt.row, rt.cols, per_wg);
  };
  bool ordered = false;
  if (vwords) {  // the census of everything the round emitted (once: a repeated delivering pass finds it done)
    hipLaunchKernelGGL(k_votes_census_multi, dim3(std::max<uint32_t>((widest_r + JG_BLOCK - 1) / JG_BLOCK, 64u), (uint32_t)(rjobs.size() + xjobs.size())), dim3(JG_BLOCK), 0, st,
                       (const JgRouteRecJob*)slice_d(3), (uint32_t)rjobs.size(), (const JgRouteXqJob*)(slice_d(3) + rb), vcur, R - 1u);
    // (the validation of the copies' counts - a launch of its own until round 6, k_votes_validate - rides on the census:
    // profiles/r06/routed_round_15_launches_ab.txt)
    HIPCHK(hipGetLastError());
  }
  for (int attempt = 0;; attempt++) {  // (repeated once when the staging turns out too small: the pass modifies nothing)
    if (!cleared) hipLaunchKernelGGL(k_route_clear, dim3(64), dim3(JG_BLOCK), 0, st, rt.d_count, (uint32_t)words, bk.hist, bk_clear);
    cleared = false;  // (a repeated attempt clears for itself)
    if (attempt) {  // (every attempt ends with a synchronisation - the counts - so the staging is free again)
      if ((rc = route_jobs())) return rc;
      HIPCHK(hipMemcpyAsync(slice_d(3), slice_h(3), rb + xb, hipMemcpyHostToDevice, st));
    }
    {  // the delivering pass: every sender's sparse steps, every sender's exceptional queue and (words) the answer words that must be rows after all, in ONE launch
      const uint32_t rec_x = (widest_r + JG_BLOCK * JG_ROUTE_ITEMS - 1) / (JG_BLOCK * JG_ROUTE_ITEMS);
      // (256 workgroups per queue: a round's queue holds a few ten thousand rows, and every workgroup - busy or not - pays the tally)
      const dim3 grid(std::max<uint32_t>(rec_x, 256u), (uint32_t)(rjobs.size() + xjobs.size() * (vwords ? 2 : 1)));
      if (vwords)
        hipLaunchKernelGGL(k_route_deliver_multi_words, grid, dim3(JG_BLOCK), 0, st, (const JgRouteRecJob*)slice_d(3), (uint32_t)rjobs.size(),
                           (const JgRouteXqJob*)(slice_d(3) + rb), (uint32_t)xjobs.size(), vcur);
      else
        hipLaunchKernelGGL(k_route_deliver_multi, grid, dim3(JG_BLOCK), 0, st, (const JgRouteRecJob*)slice_d(3), (uint32_t)rjobs.size(),
                           (const JgRouteXqJob*)(slice_d(3) + rb), (uint32_t)xjobs.size());
    }
    HIPCHK(hipGetLastError());
    ordered = false;
    // (the counts' copy stays on the round's own stream: on a stream of its own - tried in round 6 - the event that orders it
    // behind the delivering pass costs the round's stream as much as the copy did: 0.379-0.385 against 0.372-0.380 ms per round,
    // profiles/r06/ab_counts_on_a_side_stream.txt)
    HIPCHK(hipMemcpyAsync(rt.h_count, rt.d_count, words * 4, hipMemcpyDeviceToHost, st));
    if (optimistic) {
      HIPCHK(hipEventRecord(rt.ev_counts, st));
      launch_order(2 * rt.last_fullest_seg + JG_BLOCK);  // (a round's rows come in the numbers the last round's did; the kernels stride)
      HIPCHK(hipGetLastError());
      ordered = true;
    }
    T2 = clk();
    if (optimistic) HIPCHK(hipEventSynchronize(rt.ev_counts));
    else HIPCHK(hipStreamSynchronize(st));  // (the first round: nothing to size the ordering pass with yet)
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
                         e->dev.xq_cap, seq_base[s], phases[s], rt.xq_keep[s], d_keep_n + s);
      HIPCHK(hipMemcpyAsync(e->dev.xq, rt.xq_keep[s], (size_t)kx * sizeof(JgXqRec), hipMemcpyDeviceToDevice, st));
      HIPCHK(hipMemcpyAsync(e->dev.xq_n, d_keep_n + s, 4, hipMemcpyDeviceToDevice, st));
    } else {
      emptied.p[emptied.n++] = e->dev.xq_n;
    }
  }
  // (a round that has rows to order has emptied them already: the last workgroup of its k_route_scan - JgRouteXqDone)
  if (emptied.n && !ordered && !total) hipLaunchKernelGGL(k_route_clear_words, dim3(1), dim3(64), 0, st, emptied);
  // the staged rows in (destination, group, phase, emission index, sender) order -> the command columns of every
  // node's next round: bucket by (destination, group tile), sort every bucket in LDS (jg_route.h; round 2's library
  // radix sort took 240 us per 1.4 M rows)
  rt.last_total = total, rt.last_fullest_seg = fullest_seg;
  if (total && !ordered) launch_order(fullest_seg);
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
