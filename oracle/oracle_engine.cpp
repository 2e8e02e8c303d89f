// oracle_engine.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see raft_oracle.hpp).
//
// Batched facade over the single-group restatement: the same entry points as
// include/josefine_gpu.h with a `jo_` prefix, so tests/ can push one trace
// through both and compare every state column and every drained row.  It loops
// the oracle over groups exactly as josefine would run one `event_loop` per
// process (src/raft/server.rs:103-165); this is also the "port" CPU baseline
// that bench.py times (per ack: HashMap remove/insert + Vec sort, mirroring
// src/raft/progress.rs:42-60).
#include "raft_oracle.hpp"

#include <algorithm>
#include <cstring>
#include <string>
#include <deque>
#include <thread>

using namespace jo;

struct jo_engine {
  jg_config cfg;
  std::vector<Raft> groups;
  std::vector<uint8_t> self_slot;
  bool stepped = false;
  // queued commands, bucketed per group in submit order
  std::vector<std::vector<Cmd>> pending;
  std::vector<uint32_t> touched;
  std::vector<jg_msg_row> msgs;
  std::vector<jg_fsm_row> fsms;
  std::vector<jg_fault_row> faults;
  std::vector<jg_compact_row> compacted;
  uint64_t counters[4] = {0, 0, 0, 0};
  unsigned threads = 1;
  // jo_step_node: the step's outbox columns (host vectors) and what jo_node_outbox_view reports
  std::vector<uint64_t> n_o_ae, n_o_answer, n_o_hbc;
  std::vector<uint64_t> n_o_aec;  // JG_NODE_COMMON_AE: the partition's AppendEntries word where every addressee's is the same
  bool n_ae_individual = false;
  std::vector<jg_leader_beat> n_o_beat;
  std::vector<jg_fsm_row> n_fsm;  // fsm rows of the dense halves of the step in progress
  std::vector<uint64_t> n_in_answers, n_in_hbc;  // jo_node_inbox_columns: [R][G] columns handed out for the next step
  uint32_t n_col_mask = 0, n_col_hbc_mask = 0;
  std::vector<jg_msg_row> v_msgs;  // rows handed out by the *_view drains
  std::vector<jg_fsm_row> v_fsms;
  jg_node_outbox n_last{};
  uint32_t n_last_flags = 0;
  // JG_NODE_KEEP: a kept step's outputs - its outbox columns and every row it queued - wait here until its outbox is
  // viewed (jo_node_outbox_view serves the oldest); at most two
  struct KeptStep {
    std::vector<uint64_t> o_ae, o_answer, o_hbc, o_aec;
    std::vector<jg_leader_beat> o_beat;
    bool ae_individual = false;
    jg_node_outbox last{};
    uint32_t flags = 0;
    std::vector<jg_msg_row> msgs;
    std::vector<jg_fsm_row> fsms;
    std::vector<jg_fault_row> faults;
  };
  std::deque<KeptStep> kept;
};

static thread_local std::string g_err;
static int fail(int code, const char* msg) {
  g_err = msg;
  return code;
}

static void init_group(jo_engine* e, uint32_t g) {
  const jg_config& c = e->cfg;
  Timing t;
  t.heartbeat_timeout_ms = c.heartbeat_timeout_ms;
  t.election_min_ms = c.election_timeout_min_ms;
  t.election_max_ms = c.election_timeout_max_ms;
  t.seed = c.seed;
  t.separate_commit_key = (c.flags & JG_CFG_SEPARATE_COMMIT_KEY) != 0;
  std::vector<NodeId> peers;
  for (uint32_t r = 0; r < c.n_replicas; r++)
    if (r != e->self_slot[g]) peers.push_back(c.node_ids[r]);
  e->groups[g].init(c.node_ids[e->self_slot[g]], peers, c.group_base + g, t, 0);
}

extern "C" {

const char* jo_last_error(void) { return g_err.c_str(); }

int jo_engine_create(const jg_config* cfg, jo_engine** out) {
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
  jo_engine* e = new jo_engine();
  e->cfg = *cfg;
  e->groups.resize(cfg->n_groups);
  e->self_slot.assign(cfg->n_groups, 0);
  e->pending.resize(cfg->n_groups);
  for (uint32_t g = 0; g < cfg->n_groups; g++) init_group(e, g);
  *out = e;
  return JG_OK;
}

void jo_engine_destroy(jo_engine* e) { delete e; }

int jo_set_threads(jo_engine* e, unsigned n) {
  e->threads = n ? n : 1;
  return JG_OK;
}

int jo_set_self_slots(jo_engine* e, const uint8_t* slots) {
  if (e->stepped) return fail(JG_EINVAL, "self slots are fixed after the first step");
  for (uint32_t g = 0; g < e->cfg.n_groups; g++)
    if (slots[g] >= e->cfg.n_replicas) return fail(JG_EINVAL, "self slot out of range");
  for (uint32_t g = 0; g < e->cfg.n_groups; g++) {
    e->self_slot[g] = slots[g];
    init_group(e, g);
  }
  return JG_OK;
}

int jo_submit(jo_engine* e, const jg_cmd_batch* b) {
  for (size_t i = 0; i < b->n; i++) {
    if (b->group[i] >= e->cfg.n_groups) return fail(JG_EINVAL, "group out of range");
    if (b->kind[i] >= JG_CMD__COUNT) return fail(JG_EINVAL, "unknown command kind");
    if (b->kind[i] == JG_CMD_APPEND_ENTRIES && b->id[i] + b->aux[i] > b->n_blocks)
      return fail(JG_EINVAL, "block side-array range out of bounds");
  }
  for (size_t i = 0; i < b->n; i++) {
    Cmd c;
    c.kind = b->kind[i];
    c.from = b->from ? b->from[i] : 0;
    c.term = b->term ? b->term[i] : 0;
    c.id = b->id ? b->id[i] : 0;
    c.aux = b->aux ? b->aux[i] : 0;
    c.flag = b->flag ? b->flag[i] : 0;
    if (c.kind == JG_CMD_APPEND_ENTRIES)
      for (uint64_t k = 0; k < c.aux; k++) c.blocks.push_back(Block{b->blk_id[c.id + k], b->blk_next[c.id + k]});
    uint32_t g = b->group[i];
    if (e->pending[g].empty()) e->touched.push_back(g);
    e->pending[g].push_back(std::move(c));
  }
  return JG_OK;
}

static void collect(jo_engine* e, uint32_t g) {
  Raft& r = e->groups[g];
  for (const Msg& m : r.rpc) {
    jg_msg_row row;
    std::memset(&row, 0, sizeof row);
    row.group = g;
    row.kind = m.kind;
    row.to_kind = m.to_kind;
    row.flag = m.flag;
    row.to_id = m.to_id;
    row.from = m.from;
    row.term = m.term;
    row.id = m.id;
    row.aux = m.aux;
    e->msgs.push_back(row);
  }
  r.rpc.clear();
  for (const FsmRow& f : r.fsm) {
    jg_fsm_row row;
    std::memset(&row, 0, sizeof row);
    row.group = g;
    row.kind = f.kind;
    row.a = f.a;
    row.b = f.b;
    e->fsms.push_back(row);
  }
  r.fsm.clear();
  e->counters[1] += r.decisions;
  r.decisions = 0;
}

int jo_step(jo_engine* e, uint64_t now_ms) {
  e->stepped = true;
  std::sort(e->touched.begin(), e->touched.end());
  for (uint32_t g : e->touched) {
    Raft& r = e->groups[g];
    for (const Cmd& c : e->pending[g]) {
      int before = r.fault;
      r.apply(c, now_ms);
      if (r.fault && r.fault != before) e->faults.push_back(jg_fault_row{g, (uint32_t)r.fault});
      e->counters[0]++;
    }
    e->pending[g].clear();
    collect(e, g);
  }
  e->touched.clear();
  return JG_OK;
}

// one worker's share of a dense tick: commands applied, decisions taken, new faults (group order)
struct DenseShare {
  uint64_t ncmd = 0, decisions = 0;
  std::vector<jg_fault_row> faults;
};
static void dense_range(jo_engine* e, const uint64_t* acks, uint32_t g0, uint32_t g1, DenseShare* out) {
  const uint32_t G = e->cfg.n_groups, R = e->cfg.n_replicas;
  uint64_t n = 0, dec = 0;
  for (uint32_t g = g0; g < g1; g++) {
    Raft& r = e->groups[g];
    const int fault_before = r.fault;
    uint32_t s = e->self_slot[g];
    uint64_t n_append = acks[(size_t)s * G + g];
    if (!r.fault) {
      if (r.role != JG_ROLE_LEADER) {
        if (n_append) r.fault = JG_FAULT_ENGINE_DENSE_NONLEADER;
      } else if (n_append >= JG_MAX_DENSE_APPENDS) {
        r.fault = JG_FAULT_ENGINE_DENSE_APPENDS;  // own slot outside its domain: nothing of the tick is applied
      } else {
        Cmd c;
        c.kind = JG_CMD_CLIENT_REQUEST;
        for (uint64_t i = 0; i < n_append && !r.fault; i++) {
          r.apply(c, 0);
          n++;
        }
        c.kind = JG_CMD_APPEND_RESPONSE;
        c.flag = 1;
        for (uint32_t q = 0; q < R && !r.fault; q++) {
          if (q == s) continue;
          uint64_t h = acks[(size_t)q * G + g];
          if (h == JG_NO_ACK) continue;
          c.from = e->cfg.node_ids[q];
          c.id = h;
          r.apply(c, 0);
          n++;
        }
        r.rpc.clear();
        r.fsm.clear();  // dense path reports deltas, not rows
      }
    }
    if (r.fault && r.fault != fault_before) out->faults.push_back(jg_fault_row{g, (uint32_t)r.fault});
    dec += r.decisions;
    r.decisions = 0;
  }
  out->ncmd = n;
  out->decisions = dec;
}

int jo_step_dense_acks(jo_engine* e, const uint64_t* acks) {
  e->stepped = true;
  const uint32_t G = e->cfg.n_groups;
  unsigned T = std::min<unsigned>(e->threads, G ? G : 1);
  std::vector<DenseShare> share(T);
  if (T <= 1) {
    dense_range(e, acks, 0, G, &share[0]);
  } else {
    // one std::thread per share and tick.  (Measured on the 256-thread bench host: a persistent pool
    // woken through a mutex + condition variable, and one that spins on an atomic, both ran this
    // 0.2-ms-per-worker tick several times SLOWER than spawning: 0.9e7 and 0.3e7 decisions/s
    // against 3.6e7.)
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) {
      uint32_t g0 = (uint32_t)((uint64_t)G * t / T), g1 = (uint32_t)((uint64_t)G * (t + 1) / T);
      th.emplace_back(dense_range, e, acks, g0, g1, &share[t]);
    }
    for (auto& x : th) x.join();
  }
  for (unsigned t = 0; t < T; t++) {  // block partition: concatenation keeps the group order
    e->counters[0] += share[t].ncmd;
    e->counters[1] += share[t].decisions;
    e->faults.insert(e->faults.end(), share[t].faults.begin(), share[t].faults.end());
  }
  e->counters[2] += G;
  return JG_OK;
}

// ---- dense node tick (include/josefine_gpu.h "dense node tick") ---------------------------------
// The specification restated over host arrays: apply the commands the mailbox rows stand for,
// one by one, through Raft::apply, and sort what the group emits into mailbox columns or rows.
static int slot_of_id(const jo_engine* e, NodeId id) {
  for (uint32_t q = 0; q < e->cfg.n_replicas; q++)
    if (e->cfg.node_ids[q] == id) return (int)q;
  return -1;
}
static void push_row(jo_engine* e, uint32_t g, const Msg& m) {
  jg_msg_row row;
  std::memset(&row, 0, sizeof row);
  row.group = g;
  row.kind = m.kind;
  row.to_kind = m.to_kind;
  row.flag = m.flag;
  row.to_id = m.to_id;
  row.from = m.from;
  row.term = m.term;
  row.id = m.id;
  row.aux = m.aux;
  e->msgs.push_back(row);
}
static void note_fault(jo_engine* e, uint32_t g, int before) {
  Raft& r = e->groups[g];
  if (r.fault && r.fault != before) e->faults.push_back(jg_fault_row{g, (uint32_t)r.fault});
}

// jo_step_node: the fsm_tx rows a group pushed during a dense half, run-length encoded the way
// include/josefine_gpu.h specifies for that entry point: consecutive Apply ranges of one kind
// ([a,b] then [b,c]) are one range [a,c] (leader.rs:93 / follower.rs:204 ranges concatenate exactly).
static void node_take_fsm(jo_engine* e, uint32_t g) {
  Raft& r = e->groups[g];
  size_t first = e->n_fsm.size();
  for (const FsmRow& f : r.fsm) {
    if (f.kind != JG_FSM_NOTIFY && e->n_fsm.size() > first) {
      jg_fsm_row& last = e->n_fsm.back();
      if (last.kind == f.kind && last.b == f.a) {
        last.b = f.b;
        continue;
      }
    }
    jg_fsm_row row;
    std::memset(&row, 0, sizeof row);
    row.group = g, row.kind = f.kind, row.a = f.a, row.b = f.b;
    e->n_fsm.push_back(row);
  }
}

// Command::Tick of a leader (leader.rs:234-245) into the outbox: columns if the chain is in run form built by
// append only and every AppendEntries word can hold its range start key, rows otherwise.
static void leader_tick_out(jo_engine* e, uint32_t g, uint64_t now_ms, const jg_leader_outbox* out) {
  const uint32_t G = e->cfg.n_groups;
  Raft& r = e->groups[g];
  // columns stand for a Tick only when every AppendEntries word can hold its range start key: a
  // progress head at or above 2^56 - 1 (a forged AppendResponse: heads only grow, progress.rs:133-140)
  // does not fit the 56-bit field, so that leader's Tick travels as rows
  bool columns = r.chain.run_form();  // (the id set is a run [0, top], every parent id - 1: whatever head and id_gen are)
  for (const auto& kv : r.progress.progress)
    if (kv.first != r.id && kv.second.head >= JG_MAILBOX_NONE) columns = false;
  if (columns && r.chain.top() >= JG_MAILBOX_NONE) {  // 56-bit block ids in mailbox words: loud, never wrong
    r.fault = JG_FAULT_ENGINE_MAILBOX_RANGE;
    return;
  }
  Cmd c;
  c.kind = JG_CMD_TICK;
  r.apply(c, now_ms);
  e->counters[0]++;
  if (columns) {
    out->beat[g].term = r.current_term;
    for (const Msg& m : r.rpc) {
      if (m.kind == JG_CMD_HEARTBEAT) {
        out->beat[g].hb_commit = m.id;
      } else {
        const int q = slot_of_id(e, m.to_id);
        out->ae[(size_t)q * G + g] = JG_AE(m.id, m.aux);
      }
    }
  } else {
    for (const Msg& m : r.rpc) push_row(e, g, m);
  }
  r.rpc.clear();
}

// What a non-leader pushed on rpc_tx during its half: AppendResponse / HeartbeatResponse into the answer word
// (56-bit block ids in mailbox words: the callers raise the fault where the answer is produced), everything else as rows.
static void follower_answers_out(jo_engine* e, uint32_t g, const jg_follower_outbox* out) {
  Raft& r = e->groups[g];
  uint64_t o_ack = JG_MAILBOX_NONE;
  uint8_t o_has = JG_HB_NONE;
  for (const Msg& m : r.rpc) {
    if (m.kind == JG_CMD_APPEND_RESPONSE) {
      if (m.id < JG_MAILBOX_NONE) o_ack = m.id;
    } else if (m.kind == JG_CMD_HEARTBEAT_RESPONSE) {
      out->hb_commit[g] = m.id;
      o_has = m.flag;
    } else {
      push_row(e, g, m);
    }
  }
  out->answer[g] = JG_ANSWER(o_ack, o_has);
  r.rpc.clear();
}

int jo_step_dense_leader(jo_engine* e, uint64_t now_ms, const jg_leader_inbox* in, const jg_leader_outbox* out) {
  e->stepped = true;
  const uint32_t G = e->cfg.n_groups, R = e->cfg.n_replicas;
  // (the inbox block holds answer words: JG_ANSWER(AppendResponse.head, HeartbeatResponse code))
  const uint64_t* acks = in ? in->answers : nullptr;
  auto ack_of = [](uint64_t w) { return (w >> 8) == JG_MAILBOX_NONE ? JG_NO_ACK : w >> 8; };
  if (!acks && !out) return JG_OK;
  for (uint32_t g = 0; g < G; g++) {
    Raft& r = e->groups[g];
    const uint32_t s = e->self_slot[g];
    if (out) {  // defaults: "none"
      out->beat[g] = jg_leader_beat{0, JG_NO_ACK};
      for (uint32_t q = 0; q < R; q++) out->ae[(size_t)q * G + g] = JG_NO_ACK;
    }
    if (r.fault) continue;
    const int fault0 = r.fault;
    const uint64_t n_append = acks ? ack_of(acks[(size_t)s * G + g]) : 0;
    if (r.role != JG_ROLE_LEADER) {  // acks / responses are ignored (follower.rs:62, candidate.rs:194)
      if (n_append) r.fault = JG_FAULT_ENGINE_DENSE_NONLEADER;
      note_fault(e, g, fault0);
      continue;
    }
    if (n_append >= JG_MAX_DENSE_APPENDS) {  // own slot outside its domain: nothing of the tick is applied
      r.fault = JG_FAULT_ENGINE_DENSE_APPENDS;
      note_fault(e, g, fault0);
      continue;
    }
    Cmd c;
    // 1. HeartbeatResponses, ascending slot: extra AppendEntries are rows
    if (acks) {
      c.kind = JG_CMD_HEARTBEAT_RESPONSE;
      for (uint32_t q = 0; q < R && !r.fault; q++) {
        if (q == s) continue;
        const uint8_t has = (uint8_t)(acks[(size_t)q * G + g] & 0xffu);
        if (has == JG_HB_NONE) continue;
        c.from = e->cfg.node_ids[q];
        c.flag = has;
        c.id = has ? 0 : in->hbr_commit[(size_t)q * G + g];
        r.apply(c, now_ms);
        e->counters[0]++;
      }
      for (const Msg& m : r.rpc) push_row(e, g, m);
      r.rpc.clear();
    }
    // 2. appends, then acks in ascending slot order
    c = Cmd();
    c.kind = JG_CMD_CLIENT_REQUEST;
    for (uint64_t i = 0; i < n_append && !r.fault; i++) {
      r.apply(c, now_ms);
      e->counters[0]++;
    }
    c.kind = JG_CMD_APPEND_RESPONSE;
    c.flag = 1;
    for (uint32_t q = 0; acks && q < R && !r.fault; q++) {
      if (q == s) continue;
      const uint64_t h = ack_of(acks[(size_t)q * G + g]);
      if (h == JG_NO_ACK) continue;
      c.from = e->cfg.node_ids[q];
      c.id = h;
      r.apply(c, now_ms);
      e->counters[0]++;
    }
    r.rpc.clear();
    r.fsm.clear();  // dense steps report deltas, not rows
    // 3. Tick: columns if the chain is in run form built by append only, rows otherwise
    if (out && !r.fault) leader_tick_out(e, g, now_ms, out);
    note_fault(e, g, fault0);
    e->counters[1] += r.decisions;
    r.decisions = 0;
  }
  e->counters[2] += G;
  return JG_OK;
}

int jo_step_dense_follower(jo_engine* e, uint64_t now_ms, const jg_follower_inbox* in, const jg_follower_outbox* out,
                           int tick) {
  e->stepped = true;
  const uint32_t G = e->cfg.n_groups;
  for (uint32_t g = 0; g < G; g++) {
    Raft& r = e->groups[g];
    out->answer[g] = JG_NO_ACK;
    if (r.fault) continue;
    const bool was_leader = r.role == JG_ROLE_LEADER;
    const uint64_t in_term = in->beat[g].term, in_hbc = in->beat[g].hb_commit;
    const uint64_t in_from = in->ae[g] >> 8;
    const uint32_t in_n = (uint32_t)(in->ae[g] & 0xffu);
    const int fault0 = r.fault;
    const NodeId lead = in->leader ? in->leader[g] : in->leader_id;
    Cmd c;
    if (in_hbc != JG_NO_ACK) {
      c.kind = JG_CMD_HEARTBEAT;
      c.from = lead;
      c.term = in_term;
      c.id = in_hbc;
      r.apply(c, now_ms);
      e->counters[0]++;
    }
    if (in_n != JG_AE_NONE) {
      c = Cmd();
      c.kind = JG_CMD_APPEND_ENTRIES;
      c.from = lead;
      c.term = in_term;
      for (uint32_t k = 0; k < in_n; k++) c.blocks.push_back(Block{in_from + 1 + k, in_from + k});
      r.apply(c, now_ms);
      e->counters[0]++;
      // 56-bit block ids in mailbox words: raised where the answer is produced
      for (const Msg& m : r.rpc)
        if (m.kind == JG_CMD_APPEND_RESPONSE && m.id >= JG_MAILBOX_NONE && !r.fault) r.fault = JG_FAULT_ENGINE_MAILBOX_RANGE;
    }
    // the Tick of a group the leader half ticks is not this half's: a leader that steps down on the way (leader.rs:200-208)
    // has had its Tick for this round (role at entry, not the role after the inputs)
    if (tick && !was_leader && r.role != JG_ROLE_LEADER && !r.fault) {
      c = Cmd();
      c.kind = JG_CMD_TICK;
      r.apply(c, now_ms);
      e->counters[0]++;
    }
    follower_answers_out(e, g, out);
    r.fsm.clear();
    note_fault(e, g, fault0);
    e->counters[1] += r.decisions;
    r.decisions = 0;
  }
  e->counters[2] += G;
  return JG_OK;
}

int jo_chain_compact(jo_engine*, size_t n_trees, const uint64_t* off, const uint64_t* ids, const uint64_t* nexts,
                     const uint64_t* commits, uint8_t* removed) {
  for (size_t t = 0; t < n_trees; t++) {
    Chain c;
    // later entries overwrite earlier ones with the same id (sled insert = upsert)
    std::map<BlockId, size_t> last;
    for (uint64_t i = off[t]; i < off[t + 1]; i++) {
      c.db[ids[i]] = Block{ids[i], nexts[i]};
      last[ids[i]] = i;
      removed[i] = 0;
    }
    c.commit = commits[t];
    for (BlockId id : c.compact()) removed[last[id]] = 1;
  }
  return JG_OK;
}

// Chain::compact on every healthy group's own chain; removed blocks queued as (group, id) rows,
// group ascending, ids descending (the order of the walk).
int jo_chain_compact_resident(jo_engine* e, size_t* n_removed) {
  size_t n = 0;
  for (uint32_t g = 0; g < e->cfg.n_groups; g++) {
    Raft& r = e->groups[g];
    if (r.fault) continue;
    std::vector<BlockId> gone = r.chain.compact();
    std::sort(gone.begin(), gone.end(), [](BlockId a, BlockId b) { return a > b; });
    for (BlockId id : gone) e->compacted.push_back(jg_compact_row{g, 0, id});
    n += gone.size();
  }
  if (n_removed) *n_removed = n;
  return JG_OK;
}
int jo_drain_compacted(jo_engine* e, jg_compact_row* out, size_t cap, size_t* n) {
  *n = e->compacted.size();
  if (!out) return JG_OK;
  if (cap < *n) return fail(JG_ECAPACITY, "output buffer too small");
  if (*n) std::memcpy(out, e->compacted.data(), *n * sizeof(jg_compact_row));
  e->compacted.clear();
  return JG_OK;
}

#define DRAIN(vec, T)                                          \
  if (!n) return fail(JG_EINVAL, "null count");                \
  if (!out) {                                                  \
    *n = e->vec.size();                                        \
    return JG_OK;                                              \
  }                                                            \
  if (cap < e->vec.size()) {                                   \
    *n = e->vec.size();                                        \
    return fail(JG_ECAPACITY, "output buffer too small");      \
  }                                                            \
  *n = e->vec.size();                                          \
  if (*n) std::memcpy(out, e->vec.data(), *n * sizeof(T));     \
  e->vec.clear();                                              \
  return JG_OK;

int jo_drain_messages(jo_engine* e, jg_msg_row* out, size_t cap, size_t* n) { DRAIN(msgs, jg_msg_row) }
int jo_drain_applies(jo_engine* e, jg_fsm_row* out, size_t cap, size_t* n) { DRAIN(fsms, jg_fsm_row) }
int jo_drain_faults(jo_engine* e, jg_fault_row* out, size_t cap, size_t* n) { DRAIN(faults, jg_fault_row) }
// the view forms (tests/cpp drives the C++ host mirror against this library): rows stay put until the next view of the queue
int jo_drain_messages_view(jo_engine* e, const jg_msg_row** rows, size_t* n) {
  e->v_msgs.clear();
  e->v_msgs.swap(e->msgs);
  *rows = e->v_msgs.data(), *n = e->v_msgs.size();
  return JG_OK;
}
int jo_drain_applies_view(jo_engine* e, const jg_fsm_row** rows, size_t* n) {
  e->v_fsms.clear();
  e->v_fsms.swap(e->fsms);
  *rows = e->v_fsms.data(), *n = e->v_fsms.size();
  return JG_OK;
}

int jo_read_state(jo_engine* e, int field, uint32_t replica, void* out, uint32_t g0, uint32_t n) {
  if ((uint64_t)g0 + n > e->cfg.n_groups) return fail(JG_EINVAL, "group range out of bounds");
  if (field == JG_FIELD_MATCH && replica >= e->cfg.n_replicas) return fail(JG_EINVAL, "replica out of range");
  for (uint32_t i = 0; i < n; i++) {
    const Raft& r = e->groups[g0 + i];
    auto slot_of = [&](NodeId id) -> int {
      for (uint32_t q = 0; q < e->cfg.n_replicas; q++)
        if (e->cfg.node_ids[q] == id) return (int)q;
      return -1;
    };
    switch (field) {
      case JG_FIELD_TERM: ((uint64_t*)out)[i] = r.current_term; break;
      case JG_FIELD_VOTED_FOR: ((uint32_t*)out)[i] = r.has_voted ? r.voted_for : 0; break;
      case JG_FIELD_HAS_VOTED: ((uint8_t*)out)[i] = r.has_voted; break;
      case JG_FIELD_ROLE: ((uint8_t*)out)[i] = (uint8_t)r.role; break;
      case JG_FIELD_COMMIT: ((uint64_t*)out)[i] = r.chain.commit; break;
      case JG_FIELD_HEAD: ((uint64_t*)out)[i] = r.chain.head; break;
      case JG_FIELD_ID_GEN: ((uint64_t*)out)[i] = r.chain.id_gen; break;
      case JG_FIELD_MATCH: {
        uint64_t v = 0;
        if (r.role == JG_ROLE_LEADER) {
          auto it = r.progress.progress.find(e->cfg.node_ids[replica]);
          if (it != r.progress.progress.end()) v = it->second.head;
        }
        ((uint64_t*)out)[i] = v;
        break;
      }
      case JG_FIELD_REPL_STATE: {
        uint8_t m = 0;
        if (r.role == JG_ROLE_LEADER)
          for (auto& kv : r.progress.progress) {
            int s = slot_of(kv.first);
            if (s >= 0 && kv.second.replicate) m |= (uint8_t)(1u << s);
          }
        ((uint8_t*)out)[i] = m;
        break;
      }
      case JG_FIELD_VOTE_SEEN:
      case JG_FIELD_VOTE_GRANTED: {
        uint8_t m = 0;
        if (r.role == JG_ROLE_CANDIDATE)
          for (auto& kv : r.election.votes) {
            int s = slot_of(kv.first);
            if (s >= 0 && (field == JG_FIELD_VOTE_SEEN || kv.second)) m |= (uint8_t)(1u << s);
          }
        ((uint8_t*)out)[i] = m;
        break;
      }
      case JG_FIELD_FAULT: ((uint8_t*)out)[i] = (uint8_t)r.fault; break;
      case JG_FIELD_LEADER_ID: ((uint32_t*)out)[i] = (r.role == JG_ROLE_FOLLOWER && r.has_leader) ? r.leader_id : 0; break;
      case JG_FIELD_HAS_LEADER: ((uint8_t*)out)[i] = r.role == JG_ROLE_FOLLOWER && r.has_leader; break;
      case JG_FIELD_ELECTION_TIME: ((uint64_t*)out)[i] = r.election_time; break;
      case JG_FIELD_ELECTION_TIMEOUT: ((uint32_t*)out)[i] = r.election_timeout; break;
      case JG_FIELD_HEARTBEAT_TIME: ((uint64_t*)out)[i] = r.role == JG_ROLE_LEADER ? r.heartbeat_time : 0; break;
      case JG_FIELD_QUEUED_REQS: ((uint32_t*)out)[i] = r.queued_reqs; break;
      case JG_FIELD_SELF_SLOT: ((uint8_t*)out)[i] = e->self_slot[g0 + i]; break;
      default: return fail(JG_EINVAL, "unknown field");
    }
  }
  return JG_OK;
}

int jo_get_counters(jo_engine* e, uint64_t out[4]) {
  std::memcpy(out, e->counters, sizeof e->counters);
  return JG_OK;
}

// Synthetic ack-stream generator — specification in DESIGN.md "Synthetic traces";
// restated independently on the device (josefine_amd/csrc/synth.hip).
static inline uint64_t synth_hash(uint64_t seed, uint64_t tick, uint64_t gg, uint32_t r) {
  return mix64(mix64(seed + tick * 0x9e3779b97f4a7c15ull) ^ (gg * 8 + r));
}

int jo_synth_fill_acks(jo_engine* e, uint32_t mode, uint64_t tick, uint64_t* sim, uint64_t* acks) {
  const uint32_t G = e->cfg.n_groups, R = e->cfg.n_replicas;
  if (mode > 1) return fail(JG_EINVAL, "unknown synth mode");
  for (uint32_t g = 0; g < G; g++) {
    uint32_t s = e->self_slot[g];
    uint64_t gg = e->cfg.group_base + g;
    uint64_t lead = sim[(size_t)s * G + g];
    uint64_t n_append = mode == 0 ? 1 : synth_hash(e->cfg.seed, tick, gg, s) % 3;
    acks[(size_t)s * G + g] = n_append;
    sim[(size_t)s * G + g] = lead + n_append;
    for (uint32_t r = 0; r < R; r++) {
      if (r == s) continue;
      size_t k = (size_t)r * G + g;
      if (mode == 0) {
        acks[k] = lead;
        sim[k] = lead;
      } else {
        uint64_t u = synth_hash(e->cfg.seed, tick, gg, r);
        uint32_t p = (uint32_t)(u % 100);
        if (p < 5) {
          acks[k] = JG_NO_ACK;  // dropped
        } else if (p < 10) {
          acks[k] = sim[k];  // stale duplicate of the previous ack
        } else {
          uint64_t adv = (u >> 32) % (JG_MAX_INFLIGHT + 1);
          uint64_t v = std::min(lead, sim[k] + adv);
          acks[k] = v;
          sim[k] = v;
        }
      }
    }
  }
  return JG_OK;
}

// ---- jo_step_node: the specification of jg_step_node, restated over host vectors -------------------
// What server::event_loop does between two firings of its interval (src/raft/server.rs:120-161), for every
// partition the node hosts: Apply::apply of every queued row IN THE ORDER IT ARRIVED (mod.rs:471-479), then
// Command::Tick.  The classification below decides nothing about results - only about the REPRESENTATION
// the engine reports (which partitions count as rows_general, which messages leave as mailbox columns and
// which as rows); every partition's rows are applied one command at a time in stream order either way.
int jo_step_node(jo_engine* e, uint64_t now_ms, uint32_t flags) {
  if (!(flags & (JG_NODE_LEADER_HALF | JG_NODE_FOLLOWER_HALF)) || (flags & ~127u)) return fail(JG_EINVAL, "jg_step_node: bad flags");  // (JG_NODE_ASYNC: a matter of when the engine looks at its own counts)
  const bool keep = (flags & JG_NODE_KEEP) != 0;
  if (keep && !(flags & JG_NODE_ASYNC)) return fail(JG_EINVAL, "jg_step_node: JG_NODE_KEEP goes with JG_NODE_ASYNC");
  if (e->kept.size() >= (keep ? 2u : 1u)) return fail(JG_EINVAL, "jg_step_node: kept steps are outstanding (JG_NODE_KEEP): jg_node_outbox_view first");
  const size_t keep_m0 = e->msgs.size(), keep_f0 = e->fsms.size(), keep_q0 = e->faults.size();
  e->stepped = true;
  const uint32_t G = e->cfg.n_groups, R = e->cfg.n_replicas;
  const bool lead_half = flags & JG_NODE_LEADER_HALF, fol_half = flags & JG_NODE_FOLLOWER_HALF, tick = flags & JG_NODE_TICK;
  // column inbound (jo_node_inbox_columns): a sender that spoke a column this step may not also speak rows
  if (e->n_col_mask && !lead_half) return fail(JG_EINVAL, "jg_step_node: a column was handed out but the leader half does not run");
  const uint32_t col_mask = e->n_col_mask, col_hbc = e->n_col_hbc_mask;
  e->n_col_mask = e->n_col_hbc_mask = 0;
  if (col_mask)
    for (uint32_t g : e->touched)
      for (const Cmd& c : e->pending[g])
        if (c.kind == JG_CMD_APPEND_RESPONSE || c.kind == JG_CMD_HEARTBEAT_RESPONSE) {
          const int s = slot_of_id(e, c.from);
          if (s >= 0 && ((col_mask >> s) & 1u)) return fail(JG_EINVAL, "jg_step_node: a row names a sender whose answers arrived as a column");
        }
  // pass 1: which mailbox entries each partition's rows fill; what cannot be served in column form
  struct Cls {
    uint32_t seen = 0;  // bits 0-7 AppendResponse per slot, 8-15 HeartbeatResponse per slot, 16 Heartbeat, 17 AppendEntries, 18 ClientRequest
    bool general = false;
    uint64_t hb_term = 0, ae_term = 0;
    uint32_t hb_from = 0, ae_from = 0;
    size_t hb_at = 0, ae_at = 0;  // position in the partition's stream
  };
  std::vector<Cls> cls(G);
  uint64_t n_rows = 0, n_general = 0;
  std::sort(e->touched.begin(), e->touched.end());
  auto run_of = [](const Cmd& c) {  // AppendEntries blocks = a run (from, from + n]: what one JG_AE word can stand for?
    const size_t n = c.blocks.size();
    if (n > 0xfe) return false;
    if (n == 0) return true;
    const uint64_t id0 = c.blocks[0].id;
    if (id0 == 0 || id0 - 1 + n >= JG_MAILBOX_NONE) return false;
    for (size_t k = 0; k < n; k++)
      if (c.blocks[k].id != id0 + k || c.blocks[k].next != id0 + k - 1) return false;
    return true;
  };
  for (uint32_t g : e->touched) {
    Cls& k = cls[g];
    const Raft& r = e->groups[g];
    size_t at = 0;
    for (const Cmd& c : e->pending[g]) {
      n_rows++;
      at++;
      uint32_t bit = 0;
      bool general = false;
      switch (c.kind) {
        case JG_CMD_APPEND_RESPONSE:
        case JG_CMD_HEARTBEAT_RESPONSE: {
          const int s = slot_of_id(e, c.from);
          general = !lead_half || s < 0 || (uint32_t)s == e->self_slot[g] || (c.kind == JG_CMD_APPEND_RESPONSE && c.id >= JG_MAILBOX_NONE);
          if (s >= 0) bit = 1u << ((c.kind == JG_CMD_APPEND_RESPONSE ? 0 : 8) + s);
          break;
        }
        case JG_CMD_CLIENT_REQUEST:
          general = !lead_half || r.role != JG_ROLE_LEADER;
          bit = 1u << 18;
          break;
        case JG_CMD_HEARTBEAT:
          // (a leader's answer to these is a role change or nothing, leader.rs:200-208,263: never an answer word)
          general = !fol_half || c.id == JG_NO_ACK || c.from == 0 || r.role == JG_ROLE_LEADER;
          bit = 1u << 16;
          k.hb_term = c.term, k.hb_from = c.from, k.hb_at = at;
          break;
        case JG_CMD_APPEND_ENTRIES:
          general = !fol_half || c.from == 0 || !run_of(c) || r.role == JG_ROLE_LEADER;
          bit = 1u << 17;
          k.ae_term = c.term, k.ae_from = c.from, k.ae_at = at;
          break;
        default: general = true;
      }
      if (general || (k.seen & bit)) k.general = true;  // outside the vocabulary / a second row for one mailbox entry
      k.seen |= bit;
    }
    // one answer word holds the follower's two responses in the order HeartbeatResponse, AppendResponse (and one
    // beat word the term and the sender of both rows): an AppendEntries that arrived BEFORE the Heartbeat is answered
    // in the other order - rows
    if ((k.seen & (3u << 16)) == (3u << 16) && (k.hb_term != k.ae_term || k.hb_from != k.ae_from || k.ae_at < k.hb_at)) k.general = true;
  }
  // pass 2: the general partitions' rows go through jo_step (first); the others keep theirs for the halves
  std::vector<uint32_t> still;
  for (uint32_t g : e->touched) {
    if (!cls[g].general) continue;
    n_general += e->pending[g].size();
    still.push_back(g);
  }
  e->touched.swap(still);  // (`still` = every partition with rows, ascending)
  int rc = jo_step(e, now_ms);
  if (rc) return rc;
  e->n_fsm.clear();
  e->n_o_beat.assign(G, jg_leader_beat{0, JG_NO_ACK});
  e->n_o_ae.assign((size_t)R * G, JG_NO_ACK);
  e->n_o_answer.assign(G, JG_NO_ACK);
  e->n_o_hbc.assign(G, 0);
  const jg_leader_outbox l_out{e->n_o_beat.data(), e->n_o_ae.data()};
  const jg_follower_outbox f_out{e->n_o_answer.data(), e->n_o_hbc.data()};
  // the rows of a column-form partition, one command at a time in the order they arrived
  auto apply_rows = [&](uint32_t g) {
    Raft& r = e->groups[g];
    for (const Cmd& c : e->pending[g]) {
      r.apply(c, now_ms);
      e->counters[0]++;
      // 56-bit block ids in mailbox words: raised where the answer is produced
      if (c.kind == JG_CMD_APPEND_ENTRIES)
        for (const Msg& m : r.rpc)
          if (m.kind == JG_CMD_APPEND_RESPONSE && m.id >= JG_MAILBOX_NONE && !r.fault) r.fault = JG_FAULT_ENGINE_MAILBOX_RANGE;
    }
    e->pending[g].clear();
  };
  if (lead_half) {  // ---- the partitions this node leads: their rows, the peers' columns, the Tick
    for (uint32_t g = 0; g < G; g++) {
      Raft& r = e->groups[g];
      if (r.fault || r.role != JG_ROLE_LEADER) continue;
      const int fault0 = r.fault;
      const uint32_t s = e->self_slot[g];
      apply_rows(g);
      // a peer's column is its answers of the tick, after the rows: per slot HeartbeatResponse, AppendResponse
      // (a follower answers the Heartbeat before the AppendEntries of one Tick, leader.rs:234-245)
      for (uint32_t q = 0; q < R; q++) {
        if (!((col_mask >> q) & 1u) || q == s) continue;
        const uint64_t w = e->n_in_answers[(size_t)q * G + g];
        Cmd c;
        c.from = e->cfg.node_ids[q];
        if ((w & 0xffu) != JG_HB_NONE) {
          c.kind = JG_CMD_HEARTBEAT_RESPONSE;
          c.flag = (uint8_t)(w & 0xffu);
          c.id = c.flag ? 0 : (((col_hbc >> q) & 1u) ? e->n_in_hbc[(size_t)q * G + g] : 0);
          r.apply(c, now_ms);
          e->counters[0]++;
        }
        if ((w >> 8) != JG_MAILBOX_NONE) {
          c.kind = JG_CMD_APPEND_RESPONSE;
          c.flag = 1;
          c.id = w >> 8;
          r.apply(c, now_ms);
          e->counters[0]++;
        }
      }
      for (const Msg& m : r.rpc) push_row(e, g, m);  // (the extra AppendEntries of apply_heartbeat_response, leader.rs:222-231)
      r.rpc.clear();
      node_take_fsm(e, g);
      r.fsm.clear();
      if (tick && !r.fault) leader_tick_out(e, g, now_ms, &l_out);
      note_fault(e, g, fault0);
      e->counters[1] += r.decisions;
      r.decisions = 0;
    }
    e->counters[2] += G;
  }
  if (fol_half) {  // ---- everybody else: Heartbeat / AppendEntries rows, the Tick; the answers as words
    for (uint32_t g = 0; g < G; g++) {
      Raft& r = e->groups[g];
      if (r.fault || r.role == JG_ROLE_LEADER) continue;
      const int fault0 = r.fault;
      apply_rows(g);
      if (tick && r.role != JG_ROLE_LEADER && !r.fault) {
        Cmd c;
        c.kind = JG_CMD_TICK;
        r.apply(c, now_ms);
        e->counters[0]++;
      }
      follower_answers_out(e, g, &f_out);
      node_take_fsm(e, g);
      r.fsm.clear();
      note_fault(e, g, fault0);
      e->counters[1] += r.decisions;
      r.decisions = 0;
    }
    e->counters[2] += G;
  }
  // rows nobody applies: a faulted partition's (its process is gone), AppendResponse / HeartbeatResponse rows of a
  // partition this node does not lead when only the leader half runs (ignored: follower.rs:62, candidate.rs:194)
  for (uint32_t g : still) {
    e->counters[0] += e->pending[g].size();
    e->pending[g].clear();
  }
  // the dense halves' fsm rows of one step: partitions ascending (at most one half emits for a partition)
  std::stable_sort(e->n_fsm.begin(), e->n_fsm.end(), [](const jg_fsm_row& a, const jg_fsm_row& b) { return a.group < b.group; });
  if (flags & JG_NODE_FSM_FUSED) {
    // a leader partition's rows of the step - [Apply {c0, c1}] Notify {a, b} [Apply {c1, c2}] - as ONE JG_FSM_LEADER_STEP row
    // where the commit index is within 255 of the appended block (josefine_gpu.h)
    std::vector<jg_fsm_row> fused;
    for (size_t i = 0; i < e->n_fsm.size();) {
      size_t j = i;
      while (j < e->n_fsm.size() && e->n_fsm[j].group == e->n_fsm[i].group) j++;
      const jg_fsm_row* r = &e->n_fsm[i];
      const size_t n = j - i;
      size_t at = 0;
      const jg_fsm_row* before = (n > at && r[at].kind == JG_FSM_APPLY_LEADER) ? &r[at++] : nullptr;
      const jg_fsm_row* notify = (n > at && r[at].kind == JG_FSM_NOTIFY) ? &r[at++] : nullptr;
      const jg_fsm_row* after = (n > at && r[at].kind == JG_FSM_APPLY_LEADER) ? &r[at++] : nullptr;
      bool done = false;
      if (notify && at == n) {
        const uint64_t commit = e->groups[r[0].group].chain.commit;
        const uint64_t c0 = before ? before->a : (after ? after->a : commit);
        const uint64_t c1 = before ? before->b : c0;
        const uint64_t c2 = after ? after->b : c1;
        const uint64_t a = notify->a;
        if (a >= c2 && a - c0 <= 255u && c0 <= c1 && c1 <= c2 && (!after || after->a == c1)) {
          jg_fsm_row x;
          std::memset(&x, 0, sizeof x);
          x.group = r[0].group, x.kind = JG_FSM_LEADER_STEP, x.a = a, x.b = notify->b;
          x.pad[0] = (uint8_t)(a - c0), x.pad[1] = (uint8_t)(a - c1), x.pad[2] = (uint8_t)(a - c2);
          fused.push_back(x);
          done = true;
        }
      }
      if (!done) fused.insert(fused.end(), r, r + n);
      i = j;
    }
    e->n_fsm.swap(fused);
  }
  e->fsms.insert(e->fsms.end(), e->n_fsm.begin(), e->n_fsm.end());
  e->n_fsm.clear();
  e->n_ae_individual = false;
  if ((flags & JG_NODE_COMMON_AE) && lead_half && tick) {
    e->n_o_aec.assign(G, JG_NO_ACK);
    for (uint32_t g = 0; g < G; g++) {
      const uint32_t s = e->self_slot[g];
      uint64_t first = JG_NO_ACK;
      bool have = false, same = true;
      for (uint32_t q = 0; q < R; q++) {
        if (q == s) continue;
        const uint64_t w = e->n_o_ae[(size_t)q * G + g];
        if (!have) first = w, have = true;
        same = same && w == first;
      }
      e->n_o_aec[g] = same ? first : JG_AEC_INDIVIDUAL;
      e->n_ae_individual = e->n_ae_individual || !same;
    }
  }
  e->n_last = jg_node_outbox{};
  e->n_last.rows = n_rows, e->n_last.rows_general = n_general;
  e->n_last_flags = flags;
  if (keep) {  // everything the step produced leaves the queues again: due when its outbox is viewed
    e->kept.emplace_back();
    jo_engine::KeptStep& k = e->kept.back();
    k.o_ae.swap(e->n_o_ae), k.o_answer.swap(e->n_o_answer), k.o_hbc.swap(e->n_o_hbc), k.o_aec.swap(e->n_o_aec), k.o_beat.swap(e->n_o_beat);
    k.ae_individual = e->n_ae_individual, k.last = e->n_last, k.flags = flags;
    k.msgs.assign(e->msgs.begin() + keep_m0, e->msgs.end()), e->msgs.resize(keep_m0);
    k.fsms.assign(e->fsms.begin() + keep_f0, e->fsms.end()), e->fsms.resize(keep_f0);
    k.faults.assign(e->faults.begin() + keep_q0, e->faults.end()), e->faults.resize(keep_q0);
  }
  return JG_OK;
}

int jo_node_inbox_columns(jo_engine* e, uint32_t slot, uint64_t** answer, uint64_t** hb_commit) {
  const size_t G = e->cfg.n_groups, R = e->cfg.n_replicas;
  if (slot >= R) return fail(JG_EINVAL, "slot out of range");
  bool uniform = true;
  for (size_t g = 1; g < G; g++) uniform = uniform && e->self_slot[g] == e->self_slot[0];
  if (uniform && e->self_slot[0] == slot) return fail(JG_EINVAL, "jg_node_inbox_columns: the own slot's word carries the append count");
  e->n_in_answers.resize(R * G), e->n_in_hbc.resize(R * G);
  *answer = e->n_in_answers.data() + slot * G;
  e->n_col_mask |= 1u << slot;
  if (hb_commit) {
    *hb_commit = e->n_in_hbc.data() + slot * G;
    e->n_col_hbc_mask |= 1u << slot;
  } else {
    e->n_col_hbc_mask &= ~(1u << slot);
  }
  return JG_OK;
}

int jo_node_outbox_view(jo_engine* e, jg_node_outbox* out) {
  if (!e->n_last_flags) return fail(JG_EINVAL, "no jg_step_node yet");
  if (!e->kept.empty()) {  // JG_NODE_KEEP: the oldest kept step - its columns are the ones on view, its rows are due now
    jo_engine::KeptStep& k = e->kept.front();
    e->n_o_ae.swap(k.o_ae), e->n_o_answer.swap(k.o_answer), e->n_o_hbc.swap(k.o_hbc), e->n_o_aec.swap(k.o_aec), e->n_o_beat.swap(k.o_beat);
    e->n_ae_individual = k.ae_individual, e->n_last = k.last, e->n_last_flags = k.flags;
    e->msgs.insert(e->msgs.end(), k.msgs.begin(), k.msgs.end());
    e->fsms.insert(e->fsms.end(), k.fsms.begin(), k.fsms.end());
    e->faults.insert(e->faults.end(), k.faults.begin(), k.faults.end());
    e->kept.pop_front();
  }
  *out = e->n_last;
  if ((e->n_last_flags & JG_NODE_LEADER_HALF) && (e->n_last_flags & JG_NODE_TICK)) {
    out->beat = e->n_o_beat.data(), out->ae = e->n_o_ae.data();
    if (e->n_last_flags & JG_NODE_COMMON_AE) {
      out->aec = e->n_o_aec.data();
      if (!e->n_ae_individual) out->ae = nullptr;
    }
  }
  if (e->n_last_flags & JG_NODE_FOLLOWER_HALF) out->answer = e->n_o_answer.data(), out->hb_commit = e->n_o_hbc.data();
  return JG_OK;
}

uint32_t jo_abi_version(void) { return JG_ABI_VERSION; }

}  // extern "C"
