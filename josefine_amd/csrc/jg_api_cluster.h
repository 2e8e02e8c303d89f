// jg_api_cluster.h - jg_dense_cluster: the closed loop of dense node ticks (single lead and per-partition leadership), its
// rounds replayed as hipGraphs.  Part of josefine_gpu.hip's one translation unit.
#pragma once
struct jg_dense_cluster {
  std::vector<jg_engine*> nodes;
  uint32_t G = 0, R = 0, lead = 0;
  uint32_t lead_id = 0;
  uint64_t *acks = nullptr, *hbr_commit = nullptr, *o_ae = nullptr;  // acks: the lead node's inbox answer words
  uint64_t* o_aec = nullptr;  // [G] the followers' AppendEntries word where it is the same for all of them (JgLeaderNode::o_aec)
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
      (rc = alloc(16 * G, (void**)&c->o_beat)) || (rc = alloc(8 * R * G, (void**)&c->o_ae)) || (rc = alloc(8 * G, (void**)&c->o_aec)) ||
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
      hipMemsetAsync(c->o_ae, 0xff, 8 * R * G, L->stream) != hipSuccess || hipMemsetAsync(c->o_aec, 0xff, 8 * G, L->stream) != hipSuccess) {
    jg_dense_cluster_destroy(c);
    return rc;
  }
  c->own_stream.assign(n_nodes, nullptr);
  for (uint32_t r = 0; r < n_nodes; r++) {
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
  if (out) {  // the cluster keeps ONE AppendEntries word per group where every follower's is the same: written out into the block's rows here
    jg_engine* L = c->nodes[c->lead];
    HIPCHK(hipSetDevice(L->device));
    hipLaunchKernelGGL(k_aec_expand, dim3((c->G + 255) / 256), dim3(256), 0, L->stream, c->G, c->R, c->any ? JG_OWNER_NONE : c->lead,
                       (const uint8_t*)c->owner, (const uint64_t*)c->o_aec, c->o_ae);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(L->stream));
  }
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
  j.a.beat = c->o_beat, j.a.ae = c->o_ae + (size_t)r * c->G, j.a.aec = c->o_aec;
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
  L->cluster_aec = c->o_aec;  // (the cluster's own mailboxes: the followers' AppendEntries words are one word where they agree)
  rc = jg_step_dense_leader(L, now_ms, &in, &out);
  L->cluster_aec = nullptr;
  if (rc) return rc;
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
    c->nodes[r]->cluster_aec = c->o_aec;
    rc = jg_step_dense_follower(c->nodes[r], now_ms, &fi, &fo, 1);
    c->nodes[r]->cluster_aec = nullptr;
    if (rc) return rc;
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
    nd.o_beat = c->o_beat, nd.o_ae = c->o_ae, nd.o_aec = c->o_aec;
    nd.now = now_ms;
    nd.owner = c->owner, nd.offered = c->offered;
    JgLeaderJob& j = lj[r];
    j.h = jg_dense_hot_of(e->dev), j.dp = e->d_dev, j.acks = c->acks, j.seq = e->seq - 1, j.us = (int)r, j.nd = nd;
    JgFollowerJob& f = fj[r];
    f = JgFollowerJob{};
    f.d = e->dev;
    f.a.clock = replay ? c->clock : nullptr, f.a.clock_slot = r, f.a.seq_off = 1;
    f.a.leader = nullptr, f.a.leader_id = 0;
    f.a.beat = c->o_beat, f.a.ae = c->o_ae + (size_t)r * G, f.a.aec = c->o_aec;
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
  ca.R = R, ca.G = c->G, ca.owner = c->owner, ca.answers = c->acks;
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
  // (how many rounds the long graph holds: 8 measured best - profiles/micro/ab_rounds_per_graph.txt; kernel timing brackets single launches)
  constexpr uint32_t per_graph = 8u;
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
