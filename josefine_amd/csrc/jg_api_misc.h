// jg_api_misc.h - Chain::compact, jg_sync / jg_stream_wait, the drain entry points, jg_read_state, counters, device memory,
// timers and the synthetic stream.  Part of josefine_gpu.hip's one translation unit.
#pragma once
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
  if (n) {  // group ascending, then the order of the walk (ids descending): off the tick, ordered on the host
    std::vector<JgCompactRow> raw(n);
    HIPCHK(hipMemcpy(raw.data(), e->d_compact, (size_t)n * sizeof(JgCompactRow), hipMemcpyDeviceToHost));
    std::sort(raw.begin(), raw.end(), [](const JgCompactRow& a, const JgCompactRow& b) {  // (pad: the block's position in its group's walk)
      return a.group != b.group ? a.group < b.group : a.pad < b.pad;
    });
    for (const JgCompactRow& r : raw) e->q_compacted.push_back(jg_compact_row{r.group, 0, r.id});
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
