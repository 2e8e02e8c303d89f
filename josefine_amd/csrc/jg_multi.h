// jg_multi.h — several devices behind one engine handle (SURVEY.md §8(b),(e)).
//
// The reference has exactly one caller of the Raft handle, `event_loop`
// (src/raft/server.rs:103-165): one loop receives every message and tick, calls `apply`, and the
// outputs leave on two channels.  A josefine process that drives D GPUs keeps that shape: ONE
// jg_engine whose groups are sharded over D single-device engines by contiguous ownership
// (shard d owns [d*S, (d+1)*S), S = ceil(G / D)).  Groups are independent (one Raft<T> per group,
// mod.rs:326-341), so nothing is ever exchanged between shards: the router below only
//   - buckets host command rows by owner (a stable partition: a group's rows keep their order),
//   - runs the per-shard host work (radix sort, staging, launches, drains) on one host thread per
//     shard, each shard on its own HIP stream of its own device,
//   - merges what the shards emitted back into the single-engine order: per step, groups
//     ascending (= shards ascending); steps in order.  Shards of one parent step share a step
//     sequence number; every output segment of a shard is tagged with it.
// Host marshalling only — no Raft logic lives here.  Included by josefine_gpu.hip (which defines
// jg_engine, collect() and the thread-local error string).
#pragma once
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace {

// One persistent host thread per shard beyond the first (the caller's thread serves shard 0).
struct JgPool {
  struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has = false, quit = false, done = false;
    int rc = 0;
    std::string err;
  };
  std::vector<std::unique_ptr<Worker>> w;

  void start(size_t n_extra) {
    for (size_t i = 0; i < n_extra; i++) {
      w.emplace_back(new Worker());
      Worker* k = w.back().get();
      k->th = std::thread([k] {
        std::unique_lock<std::mutex> lk(k->m);
        for (;;) {
          k->cv.wait(lk, [k] { return k->has || k->quit; });
          if (k->quit) return;
          k->has = false;
          lk.unlock();
          g_err.clear();
          const int rc = k->job();
          lk.lock();
          k->rc = rc;
          k->err = rc ? g_err : std::string();
          k->done = true;
          k->cv.notify_all();
        }
      });
    }
  }
  void stop() {
    for (auto& k : w) {
      {
        std::lock_guard<std::mutex> lk(k->m);
        k->quit = true;
      }
      k->cv.notify_all();
      if (k->th.joinable()) k->th.join();
    }
    w.clear();
  }
  // fn(d) for d in [0, n): d = 0 on the calling thread, the others on their workers, concurrently.
  // Returns the first failing status (its message becomes the caller's jg_last_error()).
  int run(size_t n, const std::function<int(size_t)>& fn) {
    for (size_t d = 1; d < n; d++) {
      Worker* k = w[d - 1].get();
      {
        std::lock_guard<std::mutex> lk(k->m);
        k->job = [&fn, d] { return fn(d); };
        k->done = false;
        k->has = true;
      }
      k->cv.notify_all();
    }
    int rc = fn(0);
    std::string err = rc ? g_err : std::string();
    size_t bad = 0;
    for (size_t d = 1; d < n; d++) {
      Worker* k = w[d - 1].get();
      std::unique_lock<std::mutex> lk(k->m);
      k->cv.wait(lk, [k] { return k->done; });
      if (!rc && k->rc) rc = k->rc, err = k->err, bad = d;
    }
    if (rc) g_err = "shard " + std::to_string(bad) + ": " + err;
    return rc;
  }
};

}  // namespace

struct JgRouter {
  std::vector<jg_engine*> sh;  // the shards: ordinary single-device engines
  std::vector<uint32_t> lo;    // lo[d] = first group of shard d; lo[D] = G
  uint32_t S = 0;              // groups per shard (the last one may own fewer)
  JgPool pool;
  // per-shard buckets of jg_submit (reused)
  struct Bucket {
    std::vector<uint8_t> kind, flag;
    std::vector<uint32_t> group, from;
    std::vector<uint64_t> term, id, aux, blk_id, blk_next;
    void clear() {
      kind.clear(), flag.clear(), group.clear(), from.clear(), term.clear(), id.clear(), aux.clear();
      blk_id.clear(), blk_next.clear();
    }
  };
  std::vector<Bucket> bk;
  std::vector<std::vector<uint64_t>> ack_stage;  // jg_step_dense_acks: [R][G_d] slices of the host block
  // merged outputs, single-engine order
  std::vector<jg_msg_row> msgs;
  std::vector<jg_fsm_row> fsm;
  std::vector<jg_fault_row> faults;
  // rows handed out by a *_view call live here, untouched, until that queue's next drain
  std::vector<jg_msg_row> msgs_view;
  std::vector<jg_fsm_row> fsm_view;

  size_t D() const { return sh.size(); }
  uint32_t owner(uint32_t g) const { return g / S; }
  int run(const std::function<int(size_t)>& fn) { return pool.run(D(), fn); }
};

namespace {

// Shards of one parent step carry the same sequence number: align before stepping.
inline void router_align_seq(jg_engine* p) {
  JgRouter& r = *p->router;
  uint32_t s = p->seq;
  for (jg_engine* e : r.sh)
    if ((int32_t)(e->seq - s) > 0) s = e->seq;
  for (jg_engine* e : r.sh) e->seq = s;
  p->seq = s;
}
inline void router_after_step(jg_engine* p) {
  JgRouter& r = *p->router;
  for (jg_engine* e : r.sh)
    if ((int32_t)(e->seq - p->seq) > 0) p->seq = e->seq;
  p->stepped = true;
}

int router_create(const jg_config* cfg, jg_engine** out) {
  if (cfg->n_devices > JG_MAX_DEVICES) return fail(JG_EINVAL, "n_devices out of range");
  if (cfg->n_groups == 0) return fail(JG_EINVAL, "n_groups cannot be 0");
  const uint32_t G = cfg->n_groups;
  const uint32_t S = (G + cfg->n_devices - 1) / cfg->n_devices;
  const uint32_t D = (G + S - 1) / S;  // trailing shards that would be empty are not created
  jg_engine* p = new jg_engine();
  p->cfg = *cfg;
  p->router = new JgRouter();
  JgRouter& r = *p->router;
  r.S = S;
  for (uint32_t d = 0; d < D; d++) {
    jg_config c = *cfg;
    c.n_devices = 0;
    c.device_id = cfg->device_ids[d];
    c.n_groups = std::min<uint32_t>(S, G - d * S);
    c.group_base = cfg->group_base + (uint64_t)d * S;
    jg_engine* e = nullptr;
    const int rc = jg_engine_create(&c, &e);
    if (rc) {
      const std::string msg = "shard " + std::to_string(d) + ": " + g_err;
      jg_engine_destroy(p);
      return fail(rc, msg);
    }
    e->parent = p;
    e->track_segs = true;
    r.sh.push_back(e);
    r.lo.push_back(d * S);
  }
  r.lo.push_back(G);
  r.bk.resize(D);
  r.ack_stage.resize(D);
  r.pool.start(D - 1);
  *out = p;
  return JG_OK;
}

void router_destroy(jg_engine* p) {
  JgRouter* r = p->router;
  r->pool.stop();
  for (jg_engine* e : r->sh) {
    e->parent = nullptr;
    jg_engine_destroy(e);
  }
  delete r;
  p->router = nullptr;
}

int router_set_self_slots(jg_engine* p, const uint8_t* slots) {
  JgRouter& r = *p->router;
  if (p->stepped) return fail(JG_EINVAL, "self slots are fixed after the first step");
  for (uint32_t g = 0; g < p->cfg.n_groups; g++)
    if (slots[g] >= p->cfg.n_replicas) return fail(JG_EINVAL, "self slot out of range");
  for (size_t d = 0; d < r.D(); d++) {
    const int rc = jg_set_self_slots(r.sh[d], slots + r.lo[d]);
    if (rc) return rc;
  }
  return JG_OK;
}

// jg_submit: validate the whole batch first (nothing is consumed on error), then a stable
// partition of the rows by owner; AppendEntries rows take their slice of the block side arrays along.
int router_submit(jg_engine* p, const jg_cmd_batch* b) {
  JgRouter& r = *p->router;
  int rc = validate_batch(p->cfg.n_groups, b);
  if (rc) return rc;
  for (auto& k : r.bk) k.clear();
  for (size_t i = 0; i < b->n; i++) {
    const uint32_t g = b->group[i];
    const uint32_t d = r.owner(g);
    JgRouter::Bucket& k = r.bk[d];
    k.kind.push_back(b->kind[i]);
    k.group.push_back(g - r.lo[d]);
    k.from.push_back(b->from ? b->from[i] : 0);
    k.term.push_back(b->term ? b->term[i] : 0);
    k.flag.push_back(b->flag ? b->flag[i] : 0);
    uint64_t id = b->id ? b->id[i] : 0, aux = b->aux ? b->aux[i] : 0;
    if (b->kind[i] == JG_CMD_APPEND_ENTRIES) {
      const uint64_t at = k.blk_id.size();
      k.blk_id.insert(k.blk_id.end(), b->blk_id + id, b->blk_id + id + aux);
      k.blk_next.insert(k.blk_next.end(), b->blk_next + id, b->blk_next + id + aux);
      id = at;
    }
    k.id.push_back(id);
    k.aux.push_back(aux);
  }
  for (size_t d = 0; d < r.D(); d++) {
    JgRouter::Bucket& k = r.bk[d];
    if (k.kind.empty()) continue;
    jg_cmd_batch s{};
    s.n = k.kind.size();
    s.kind = k.kind.data(), s.group = k.group.data(), s.id = k.id.data();
    // (a column the caller did not provide stays absent: the shard's node step does not upload it)
    s.from = b->from ? k.from.data() : nullptr, s.term = b->term ? k.term.data() : nullptr;
    s.aux = (b->aux || b->n_blocks) ? k.aux.data() : nullptr, s.flag = b->flag ? k.flag.data() : nullptr;
    s.n_blocks = k.blk_id.size(), s.blk_id = k.blk_id.data(), s.blk_next = k.blk_next.data();
    rc = jg_submit(r.sh[d], &s);  // (copies into the shard's pending columns)
    if (rc) return rc;
  }
  return JG_OK;
}

int router_step(jg_engine* p, uint64_t now_ms) {
  JgRouter& r = *p->router;
  router_align_seq(p);
  const int rc = r.run([&](size_t d) { return jg_step(r.sh[d], now_ms); });
  router_after_step(p);
  return rc;
}

int node_step(jg_engine* e, uint64_t now_ms, uint32_t flags);

// jg_step_node on every shard (each one's rows were bucketed by jg_submit)
int router_step_node(jg_engine* p, uint64_t now_ms, uint32_t flags) {
  JgRouter& r = *p->router;
  for (jg_engine* s : r.sh)
    if (s->inflight.phase) return fail(JG_EINVAL, "a drain is in transfer: jg_drain_wait first");
  router_align_seq(p);
  const int rc = r.run([&](size_t d) { return node_step(r.sh[d], now_ms, flags); });
  router_after_step(p);
  p->node.last_flags = flags;
  return rc;
}
// the shards' outbox columns, concatenated into the parent's group numbering
int router_node_outbox(jg_engine* p, jg_node_outbox* out) {
  JgRouter& r = *p->router;
  jg_engine::NodeStep& nd = p->node;
  if (!nd.last_flags) return fail(JG_EINVAL, "no jg_step_node yet");
  const size_t G = p->cfg.n_groups, R = p->cfg.n_replicas;
  *out = jg_node_outbox{};
  const bool lead = (nd.last_flags & JG_NODE_LEADER_HALF) && (nd.last_flags & JG_NODE_TICK), fol = (nd.last_flags & JG_NODE_FOLLOWER_HALF) != 0;
  if (lead) nd.cat_beat.resize(G), nd.cat_ae.resize(R * G);
  if (fol) nd.cat_answer.resize(G), nd.cat_hbc.resize(G);
  for (size_t d = 0; d < r.D(); d++) {
    jg_node_outbox o{};
    const int rc = jg_node_outbox_view(r.sh[d], &o);
    if (rc) return rc;
    const size_t lo = r.lo[d], n = r.lo[d + 1] - r.lo[d];
    if (lead) {
      std::memcpy(nd.cat_beat.data() + lo, o.beat, n * sizeof(jg_leader_beat));
      for (size_t q = 0; q < R; q++) std::memcpy(nd.cat_ae.data() + q * G + lo, o.ae + q * n, n * 8);
    }
    if (fol) {
      std::memcpy(nd.cat_answer.data() + lo, o.answer, n * 8);
      std::memcpy(nd.cat_hbc.data() + lo, o.hb_commit, n * 8);
    }
    out->rows += o.rows, out->rows_general += o.rows_general, out->bytes_h2d += o.bytes_h2d, out->bytes_d2h += o.bytes_d2h;
  }
  if (lead) out->beat = nd.cat_beat.data(), out->ae = nd.cat_ae.data();
  if (fol) out->answer = nd.cat_answer.data(), out->hb_commit = nd.cat_hbc.data();
  return JG_OK;
}

int router_step_dense_acks_shards(jg_engine* p, const uint64_t* const* acks_dev, uint32_t n_ticks) {
  JgRouter& r = *p->router;
  for (size_t d = 0; d < r.D(); d++)
    if (!acks_dev[d]) return fail(JG_EINVAL, "null ack block for a shard");
  router_align_seq(p);
  const int rc = r.run([&](size_t d) { return jg_step_dense_acks_device_n(r.sh[d], acks_dev[d], n_ticks); });
  router_after_step(p);
  return rc;
}

// host [R][G] block -> per-shard [R][G_d] blocks (each shard uploads and launches on its own thread)
int router_step_dense_acks(jg_engine* p, const uint64_t* acks_host) {
  JgRouter& r = *p->router;
  const size_t G = p->cfg.n_groups, R = p->cfg.n_replicas;
  router_align_seq(p);
  const int rc = r.run([&](size_t d) {
    const size_t n = r.lo[d + 1] - r.lo[d];
    std::vector<uint64_t>& st = r.ack_stage[d];
    st.resize(R * n);
    for (size_t q = 0; q < R; q++) std::memcpy(st.data() + q * n, acks_host + q * G + r.lo[d], n * 8);
    return jg_step_dense_acks(r.sh[d], st.data());
  });
  router_after_step(p);
  return rc;
}

// Merge what the shards have queued into the parent's queues, single-engine order: by step
// sequence number, and within one step shards ascending (= groups ascending).
template <typename Row, typename QOf, typename SegOf>
void router_merge(JgRouter& r, std::vector<Row>& out, QOf q_of, SegOf seg_of) {
  const size_t D = r.D();
  std::vector<size_t> si(D, 0), off(D, 0);
  for (;;) {
    bool any = false;
    uint32_t best = 0;
    for (size_t d = 0; d < D; d++) {
      const std::vector<JgSeg>& sg = seg_of(d);
      if (si[d] >= sg.size()) continue;
      if (!any || (int32_t)(sg[si[d]].seq - best) < 0) best = sg[si[d]].seq, any = true;
    }
    if (!any) break;
    for (size_t d = 0; d < D; d++) {
      const std::vector<JgSeg>& sg = seg_of(d);
      while (si[d] < sg.size() && sg[si[d]].seq == best) {
        const Row* src = q_of(d).p + off[d];
        const size_t n = sg[si[d]].n, at = out.size();
        out.insert(out.end(), src, src + n);
        for (size_t i = 0; i < n; i++) out[at + i].group += r.lo[d];
        off[d] += n;
        si[d]++;
      }
    }
  }
  for (size_t d = 0; d < D; d++) {
    q_of(d).n = 0;
    seg_of(d).clear();
  }
}

// pipelined drains (jg_drain_prefetch): the shards' batches cover the same steps, and the merge
// needs all of them: nothing is delivered until every shard's batch has landed
inline bool router_all_landed(jg_engine* p) {
  for (jg_engine* e : p->router->sh)
    if (!inflight_landed(e)) return false;
  return true;
}

int router_collect(jg_engine* p, int release_mask) {
  JgRouter& r = *p->router;
  if (p->pipelined && !router_all_landed(p)) return JG_OK;
  const int rc = r.run([&](size_t d) { return collect(r.sh[d], 3); });
  if (rc) return rc;
  if (release_mask & 1) r.msgs_view.clear();
  if (release_mask & 2) r.fsm_view.clear();
  router_merge<jg_msg_row>(
      r, r.msgs, [&](size_t d) -> PinnedQueue<jg_msg_row>& { return r.sh[d]->q_msgs; },
      [&](size_t d) -> std::vector<JgSeg>& { return r.sh[d]->seg_m; });
  router_merge<jg_fsm_row>(
      r, r.fsm, [&](size_t d) -> PinnedQueue<jg_fsm_row>& { return r.sh[d]->q_fsm; },
      [&](size_t d) -> std::vector<JgSeg>& { return r.sh[d]->seg_f; });
  {  // faults: each shard's list is ordered by (step, group) already
    const size_t D = r.D();
    std::vector<size_t> at(D, 0);
    for (;;) {
      bool any = false;
      uint32_t best = 0;
      for (size_t d = 0; d < D; d++) {
        jg_engine* e = r.sh[d];
        if (at[d] >= e->q_faults.size()) continue;
        if (!any || (int32_t)(e->q_fault_seq[at[d]] - best) < 0) best = e->q_fault_seq[at[d]], any = true;
      }
      if (!any) break;
      for (size_t d = 0; d < D; d++) {
        jg_engine* e = r.sh[d];
        while (at[d] < e->q_faults.size() && e->q_fault_seq[at[d]] == best) {
          jg_fault_row f = e->q_faults[at[d]++];
          f.group += r.lo[d];
          r.faults.push_back(f);
        }
      }
    }
    for (jg_engine* e : r.sh) e->q_faults.clear(), e->q_fault_seq.clear();
  }
  return JG_OK;
}

template <typename Row>
int router_drain(jg_engine* p, std::vector<Row>& q, int mask, Row* out, size_t cap, size_t* n) {
  if (!n) return fail(JG_EINVAL, "null argument");
  const int rc = router_collect(p, mask);
  if (rc) return rc;
  *n = q.size();
  if (!out) return JG_OK;
  if (cap < q.size()) return fail(JG_ECAPACITY, "output buffer too small");
  if (!q.empty()) std::memcpy(out, q.data(), q.size() * sizeof(Row));
  q.clear();
  return JG_OK;
}
template <typename Row>
int router_drain_view(jg_engine* p, std::vector<Row>& q, std::vector<Row>& view, int mask, const Row** rows, size_t* n) {
  if (!rows || !n) return fail(JG_EINVAL, "null argument");
  if (p->pipelined && !router_all_landed(p)) {
    // a batch is still in transfer: router_collect would not collect (and would not release the
    // outstanding view either), so the swap below would hand the view's rows back to `q` and they
    // would be delivered a second time behind the next batch.  Nothing new: the earlier view stays
    // valid and untouched, exactly like drain_view() of a single-device engine.
    *rows = view.data();
    *n = 0;
    return JG_OK;
  }
  const int rc = router_collect(p, mask);  // (releases this queue's previous view)
  if (rc) return rc;
  view.swap(q);
  *rows = view.data();
  *n = view.size();
  return JG_OK;
}

int router_read_state(jg_engine* p, int field, uint32_t replica, void* out, uint32_t g0, uint32_t n) {
  JgRouter& r = *p->router;
  if ((uint64_t)g0 + n > p->cfg.n_groups) return fail(JG_EINVAL, "group range out of bounds");
  if (field < 0 || field >= JG_FIELD__COUNT) return fail(JG_EINVAL, "unknown field");
  const size_t w = field_width(field);
  uint32_t g = g0;
  const uint32_t end = g0 + n;
  while (g < end) {
    const uint32_t d = r.owner(g);
    const uint32_t take = std::min(end, r.lo[d + 1]) - g;
    const int rc = jg_read_state(r.sh[d], field, replica, (char*)out + (size_t)(g - g0) * w, g - r.lo[d], take);
    if (rc) return rc;
    g += take;
  }
  return JG_OK;
}

int router_get_counters(jg_engine* p, uint64_t out[4]) {
  JgRouter& r = *p->router;
  out[0] = out[1] = out[2] = out[3] = 0;
  for (jg_engine* e : r.sh) {
    uint64_t c[4];
    const int rc = jg_get_counters(e, c);
    if (rc) return rc;
    for (int i = 0; i < 4; i++) out[i] += c[i];
  }
  return JG_OK;
}

}  // namespace
