// raft_handle.hpp — C++ host mirror of josefine's Raft driver surface over the C ABI
// (include/josefine_gpu.h).  The reference is Rust and this image has no Rust
// toolchain, so the host side above the ABI is C++ with the reference's names and
// argument meaning; the Rust adapter a maintainer would add is in INTEGRATION.md.
//
//   reference (src/raft)                              here
//   enum Command { Tick, VoteRequest{..}, .. }        josefine::Command::{Tick(), VoteRequest(..), ..}   mod.rs:160-227
//   trait Apply { fn apply(self, Command) }           josefine::RaftHandle::apply / BatchedRaft::apply     mod.rs:471-489
//   RaftHandle::{is_follower,is_candidate,is_leader}  same                                                mod.rs:437-447
//   rpc_tx: UnboundedSender<Message>                  BatchedRaft::rpc_tx  (std::function sink)           mod.rs:337-338
//   fsm_tx: UnboundedSender<Instruction>              BatchedRaft::fsm_tx  (std::function sink)           mod.rs:339-340
//   Chain's sled tree (block payloads)                josefine::BlockStore (host, key-ordered map)        chain.rs:99-104
//
// What stays on the host, exactly as in the reference: block payloads, the queued
// client requests (follower.rs:22, candidate.rs:20) and the expansion of Apply
// ranges into blocks in key order (leader.rs:93, follower.rs:204).  No state-machine
// arithmetic happens here: every decision comes out of the device engine.
#pragma once
#include <algorithm>
#include <cstdint>
#include <chrono>
#include <deque>
#include <functional>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/josefine_gpu.h"

#include "types.hpp"
#include "chain_store.hpp"

namespace josefine {

// Host block store of one group: what sled holds in the reference (chain.rs:99-104) - in the reference's own
// byte layout (chain_store.hpp: 8-byte big-endian keys, bincode values, the "commit" key in the same tree).
using BlockStore = formats::ChainStore;

// Command rows in the shape of jg_cmd_batch (structure of arrays): what a batched event loop queues
// between two ticks instead of one heap object per message.
struct RowQueue {
  std::vector<uint8_t> kind, flag;
  std::vector<uint32_t> group, from;
  std::vector<uint64_t> term, id, aux, blk_id, blk_next;
  size_t size() const { return kind.size(); }
  bool empty() const { return kind.empty(); }
  void clear() {
    kind.clear(), flag.clear(), group.clear(), from.clear(), term.clear(), id.clear(), aux.clear(), blk_id.clear(), blk_next.clear();
  }
  void push(uint32_t g, uint8_t k, NodeId f = 0, Term t = 0, uint64_t i = 0, uint64_t a = 0, uint8_t fl = 0) {
    kind.push_back(k), group.push_back(g), from.push_back(f), term.push_back(t), id.push_back(i), aux.push_back(a), flag.push_back(fl);
  }
  // AppendEntries{term, leader_id, blocks}: the (id, next) pairs go to the side arrays (mod.rs:196-203)
  void push_append_entries(uint32_t g, NodeId leader, Term t, const Block* blocks, size_t n) {
    push(g, JG_CMD_APPEND_ENTRIES, leader, t, blk_id.size(), n, 0);
    for (size_t k = 0; k < n; k++) blk_id.push_back(blocks[k].id), blk_next.push_back(blocks[k].next);
  }
  // the blocks (first, first + n] of a run: next = id - 1 each (what a JG_AE mailbox word stands for)
  void push_append_run(uint32_t g, NodeId leader, Term t, BlockId from_id, uint32_t n) {
    push(g, JG_CMD_APPEND_ENTRIES, leader, t, blk_id.size(), n, 0);
    for (uint32_t k = 0; k < n; k++) blk_id.push_back(from_id + 1 + k), blk_next.push_back(from_id + k);
  }
  void append(const jg_cmd_batch& b) {  // bulk: a peer's whole frame
    const uint64_t shift = blk_id.size();
    const size_t at = size();
    kind.insert(kind.end(), b.kind, b.kind + b.n), group.insert(group.end(), b.group, b.group + b.n);
    auto put = [&](auto& v, auto* src) {
      if (src) v.insert(v.end(), src, src + b.n);
      else v.resize(at + b.n, 0);
    };
    put(from, b.from), put(term, b.term), put(id, b.id), put(aux, b.aux), put(flag, b.flag);
    if (shift)
      for (size_t i = 0; i < b.n; i++)
        if (b.kind[i] == JG_CMD_APPEND_ENTRIES) id[at + i] += shift;
    if (b.n_blocks) blk_id.insert(blk_id.end(), b.blk_id, b.blk_id + b.n_blocks), blk_next.insert(blk_next.end(), b.blk_next, b.blk_next + b.n_blocks);
  }
  jg_cmd_batch view() const {
    jg_cmd_batch b{};
    b.n = kind.size();
    b.kind = kind.data(), b.group = group.data(), b.from = from.data(), b.term = term.data(), b.id = id.data(), b.aux = aux.data();
    b.flag = flag.data(), b.n_blocks = blk_id.size(), b.blk_id = blk_id.data(), b.blk_next = blk_next.data();
    return b;
  }
};

class BatchedRaft;

class RaftHandle {  // mod.rs:417-468, a view of one group
 public:
  RaftHandle(BatchedRaft* e, uint32_t g) : e_(e), g_(g) {}
  RaftHandle apply(const Command& cmd, uint64_t now_ms = 0);
  bool is_follower() const { return role() == JG_ROLE_FOLLOWER; }
  bool is_candidate() const { return role() == JG_ROLE_CANDIDATE; }
  bool is_leader() const { return role() == JG_ROLE_LEADER; }
  Term current_term() const { return read64(JG_FIELD_TERM); }
  BlockId commit() const { return read64(JG_FIELD_COMMIT); }
  BlockId head() const { return read64(JG_FIELD_HEAD); }
  uint32_t fault() const { return read8(JG_FIELD_FAULT); }
  bool has_voted() const { return read8(JG_FIELD_HAS_VOTED) != 0; }
  NodeId voted_for() const { return read32(JG_FIELD_VOTED_FOR); }

 private:
  uint8_t role() const { return read8(JG_FIELD_ROLE); }
  uint64_t read64(int f) const;
  uint32_t read32(int f) const;
  uint8_t read8(int f) const;
  BatchedRaft* e_;
  uint32_t g_;
};

class BatchedRaft {
 public:
  // sinks: everything the groups push on the two channels, in per-group order
  std::function<void(const Message&)> rpc_tx;
  std::function<void(const Instruction&)> fsm_tx;
  // batch sinks: the same two channels, a whole step at a time and in wire form - the rows as the engine
  // drains them (views of its pinned queues: no copy, no per-row object) and, for step_node, the mailbox
  // columns.  A sink that is set replaces the per-row expansion of its channel; what the rows / columns
  // stand for is in include/josefine_gpu.h.
  std::function<void(const jg_msg_row*, size_t)> msg_rows_tx;
  std::function<void(const jg_fsm_row*, size_t)> fsm_rows_tx;
  std::function<void(const jg_node_outbox&)> columns_tx;

  // RaftHandle::new for n_groups groups (mod.rs:428-435)
  // `devices`: shard the groups over these HIP devices behind this one handle (jg_config.n_devices:
  // contiguous ownership, a device may repeat) — one event loop still owns the handle, as in the
  // reference (server.rs:103-165); empty = one shard on `device`.
  BatchedRaft(uint32_t n_groups, std::vector<NodeId> node_ids, int device = 0, uint64_t seed = 0,
              uint32_t flags = 0, std::vector<int> devices = {})
      : stores_(n_groups), queued_(n_groups), ids_(node_ids) {
    jg_config c{};
    c.n_devices = (uint32_t)devices.size();
    for (size_t d = 0; d < devices.size() && d < JG_MAX_DEVICES; d++) c.device_ids[d] = devices[d];
    c.abi_version = JG_ABI_VERSION;
    c.n_groups = n_groups;
    c.n_replicas = (uint32_t)node_ids.size();
    for (size_t r = 0; r < node_ids.size() && r < JG_MAX_REPLICAS; r++) c.node_ids[r] = node_ids[r];
    c.device_id = device;
    c.heartbeat_timeout_ms = 100;      // config.rs:104
    c.election_timeout_min_ms = 500;   // mod.rs:318
    c.election_timeout_max_ms = 1000;  // mod.rs:319
    c.seed = seed;
    c.flags = flags;
    check(jg_engine_create(&c, &e_));
    for (auto& s : stores_) s.insert(Block{0, 0, {}});  // genesis (chain.rs:139-153)
  }
  ~BatchedRaft() { jg_engine_destroy(e_); }
  BatchedRaft(const BatchedRaft&) = delete;
  BatchedRaft& operator=(const BatchedRaft&) = delete;

  RaftHandle handle(uint32_t g) { return RaftHandle(this, g); }
  const BlockStore& store(uint32_t g) const { return stores_[g]; }
  jg_engine* raw() { return e_; }

  // Apply::apply for one group: submit, step, forward the outputs to the sinks.
  RaftHandle apply(uint32_t g, const Command& cmd, uint64_t now_ms = 0) {
    submit(g, cmd);
    step(now_ms);
    return RaftHandle(this, g);
  }

  void submit(uint32_t g, const Command& cmd) {
    kind_.push_back(cmd.kind);
    group_.push_back(g);
    from_.push_back(cmd.from);
    term_.push_back(cmd.term);
    flag_.push_back(cmd.flag ? 1 : 0);
    if (cmd.kind == JG_CMD_APPEND_ENTRIES) {
      id_.push_back(blk_id_.size());
      aux_.push_back(cmd.blocks.size());
      for (const Block& b : cmd.blocks) {
        blk_id_.push_back(b.id);
        blk_next_.push_back(b.next);
        pending_blocks_.push_back({g, b});  // payload goes to the host store on extend
      }
    } else {
      id_.push_back(cmd.id);
      aux_.push_back(cmd.aux);
    }
    if (cmd.kind == JG_CMD_CLIENT_REQUEST) pending_reqs_[{g, cmd.id}] = cmd.proposal;
  }

  void step(uint64_t now_ms) {
    flush_rows();
    check(jg_step(e_, now_ms));
    after_step();
  }

  // a whole frame of rows at once (no payloads: block data and proposals travel through submit())
  void submit_rows(const jg_cmd_batch& b) {
    flush_rows();
    check(jg_submit(e_, &b));
  }
  // What event_loop does between two ticks, for every partition at once (jg_step_node): the queued rows
  // - the steady-state vocabulary through the dense kernels, anything else through the general state
  // machine - then Command::Tick (JG_NODE_TICK).  The Tick's messages and the followers' answers leave as
  // mailbox columns (columns_tx, or expanded into Messages for rpc_tx); `answers_to[g]`: the NodeId a
  // follower's dense answers are addressed to (the sender of the partition's Heartbeat / AppendEntries).
  void step_node(uint64_t now_ms, uint32_t flags, const std::vector<NodeId>* answers_to = nullptr) {
    step_node_begin(now_ms, flags);
    step_node_finish(answers_to);
  }
  // The node step in two halves, so that ONE loop overlaps with itself: begin() returns as soon as the rows are on the
  // device and classified (the engine's pinned input columns are free again) with the dense halves, the fsm build and the
  // downloads of the outputs still running; the caller decodes the NEXT tick's traffic meanwhile and calls finish() -
  // which waits for the outputs, feeds fsm_tx / rpc_tx / the column sink - right before the next begin().
  // keep (JG_NODE_KEEP, with async): TWO steps in flight - begin() may be called again before the step before has been
  // finished; finish() then serves the OLDER one (its outbox, its rows), while the device runs the newer.
  void step_node_begin(uint64_t now_ms, uint32_t flags, bool async = false, bool keep = false) {
    flush_rows();
    // async: no synchronisation at all inside the call (JG_NODE_ASYNC) - the engine settles the step (its general path, if
    // it has one) when step_node_finish asks for the outbox
    check(jg_step_node(e_, now_ms, flags | (async ? (uint32_t)JG_NODE_ASYNC : 0u) | (keep ? (uint32_t)JG_NODE_KEEP : 0u)));
    step_blocks_.emplace_back();
    step_blocks_.back().swap(pending_blocks_);  // (the payloads of THIS step's AppendEntries rows: stored when it is finished)
  }
  double ms_waited_for_outputs = 0;  // inside jg_node_outbox_view, summed over the steps finished so far (a loop's own accounting)
  bool node_step_open() const { return !step_blocks_.empty(); }
  size_t node_steps_open() const { return step_blocks_.size(); }
  void step_node_finish(const std::vector<NodeId>* answers_to = nullptr) {
    if (step_blocks_.empty()) return;
    jg_node_outbox o{};
    const auto w0 = std::chrono::steady_clock::now();
    check(jg_node_outbox_view(e_, &o));  // (the oldest open step's)
    ms_waited_for_outputs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    last_outbox_ = o;
    pending_blocks_.swap(step_blocks_.front());  // (nothing else is pending here: a plain step() is not taken between begin and finish)
    step_blocks_.pop_front();
    after_step();
    if (columns_tx) columns_tx(o);
    else if (rpc_tx) expand_columns(o, answers_to);
  }
  // zero-copy inbound rows: the engine's own pinned columns, filled in place by the transport's decoder
  jg_cmd_cols reserve_rows(size_t n, size_t n_blocks = 0) {
    flush_rows();
    jg_cmd_cols c{};
    check(jg_submit_reserve(e_, n, n_blocks, &c));
    return c;
  }
  void commit_rows(size_t n, size_t n_blocks, uint32_t optional_columns) { check(jg_submit_commit(e_, n, n_blocks, optional_columns)); }
  // column inbound (jg_node_inbox_columns): where member slot `slot`'s answer words (and HeartbeatResponse.commit
  // values) for the next step_node are to be written, in the engine's pinned memory
  void inbox_columns(uint32_t slot, uint64_t** answer, uint64_t** hb_commit) { check(jg_node_inbox_columns(e_, slot, answer, hb_commit)); }
  const jg_node_outbox& last_outbox() const { return last_outbox_; }
  // payload mirrors for rows that were queued in bulk (submit_rows): a ClientRequest's proposal, a block's data
  void note_proposal(uint32_t g, uint64_t request_id, std::vector<uint8_t> proposal) { pending_reqs_[{g, request_id}] = std::move(proposal); }
  void note_block(uint32_t g, const Block& b) { pending_blocks_.push_back({g, b}); }
  void store_block(uint32_t g, const Block& b) { stores_[g].insert(b); }
  // process restart of partition g's replica: the sled tree is re-opened from its bytes (ChainStore::from_raw)
  // and the engine restarts the instance on its own image of it (JG_CMD_RESTART = Raft::new + Chain::new on the
  // persisted tree, follower.rs:68-95, chain.rs:117-137); returns what Chain::new finds in the re-opened tree
  formats::ChainStore::Reopened restart(uint32_t g, uint64_t now_ms = 0) {
    stores_[g] = formats::ChainStore::from_raw(stores_[g].raw());
    extend_failed_.erase(g);
    const formats::ChainStore::Reopened r = stores_[g].reopen();
    queued_[g].clear();
    Command c;
    c.kind = JG_CMD_RESTART;
    submit(g, c);
    step(now_ms);
    return r;
  }
  // the replica of a partition that was RE-CREATED: an empty data directory (JG_CMD_RECREATE) - the host's block
  // store of the group starts over with genesis (chain.rs:139-153), like the engine's image of it
  void recreate(uint32_t g, uint64_t now_ms = 0) {
    stores_[g] = BlockStore{};
    stores_[g].insert(Block{0, 0, {}});
    extend_failed_.erase(g);
    queued_[g].clear();
    Command c;
    c.kind = JG_CMD_RECREATE;
    submit(g, c);
    step(now_ms);
  }
  NodeId self_id(uint32_t g) {
    uint8_t s = 0;
    check(jg_read_state(e_, JG_FIELD_SELF_SLOT, 0, &s, g, 1));
    return ids_[s];
  }

 private:
  friend class RaftHandle;
  void check(int rc) {
    if (rc != JG_OK) throw EngineError(rc, jg_last_error());
  }
  void flush_rows() {
    if (kind_.empty()) return;
    jg_cmd_batch b{};
    b.n = kind_.size();
    b.kind = kind_.data(), b.group = group_.data(), b.from = from_.data(), b.term = term_.data();
    b.id = id_.data(), b.aux = aux_.data(), b.flag = flag_.data();
    b.n_blocks = blk_id_.size(), b.blk_id = blk_id_.data(), b.blk_next = blk_next_.data();
    check(jg_submit(e_, &b));
    kind_.clear(), group_.clear(), from_.clear(), term_.clear(), id_.clear(), aux_.clear(), flag_.clear();
    blk_id_.clear(), blk_next_.clear();
  }
  void after_step() {
    // followers store the payloads of the blocks they were sent exactly as Chain::extend does (chain.rs:178-192): a
    // block whose parent is missing returns Err BEFORE db.insert, the event loop of that partition ends there
    // (server.rs:125-159 propagates with `?`) and nothing behind it is stored; a partition whose process was gone
    // before the AppendEntries (or that died of something else on the way: the stale-leader assert of
    // follower.rs:147-154 comes before the extends) stores nothing - the tree must stay byte for byte what sled holds.
    if (!pending_blocks_.empty()) {
      uint32_t lo = pending_blocks_.front().first, hi = lo;
      for (auto& pb : pending_blocks_) lo = std::min(lo, pb.first), hi = std::max(hi, pb.first);
      std::vector<uint8_t> fault(hi - lo + 1);
      check(jg_read_state(e_, JG_FIELD_FAULT, 0, fault.data(), lo, hi - lo + 1));
      for (auto& pb : pending_blocks_) {
        const uint32_t g = pb.first;
        uint8_t& f = fault[g - lo];
        if (f && (f != JG_FAULT_EXTEND_MISSING_PARENT || extend_failed_.count(g))) continue;
        if (f == JG_FAULT_EXTEND_MISSING_PARENT && !stores_[g].has(pb.second.next)) {
          extend_failed_.insert(g);  // the failing block: dropped, and everything of this partition behind it
          continue;
        }
        stores_[g].insert(pb.second);
      }
    }
    pending_blocks_.clear();
    pump();
  }
  // the mailbox columns as the Messages they stand for (include/josefine_gpu.h, "dense node tick")
  void expand_columns(const jg_node_outbox& o, const std::vector<NodeId>* answers_to) {
    const uint32_t G = (uint32_t)stores_.size(), R = (uint32_t)ids_.size();
    if (o.beat)
      for (uint32_t g = 0; g < G; g++) {
        jg_msg_row r{};
        r.group = g, r.term = o.beat[g].term;
        bool have_from = false;
        auto from = [&] {
          if (!have_from) r.from = self_id(g), have_from = true;
        };
        if (o.beat[g].hb_commit != JG_NO_ACK) {  // Heartbeat{term, commit, leader_id} to everybody (leader.rs:44-51)
          from();
          r.kind = JG_CMD_HEARTBEAT, r.to_kind = JG_TO_PEERS, r.to_id = 0, r.id = o.beat[g].hb_commit, r.aux = 0;
          emit(r, r.id);
        }
        // (JG_NODE_COMMON_AE: one word for every addressee unless the partition's words differ)
        const bool common = o.aec && o.aec[g] != JG_AEC_INDIVIDUAL;
        if (common && o.aec[g] == JG_NO_ACK) continue;
        if (common) from();
        for (uint32_t q = 0; q < R; q++) {  // AppendEntries per follower, ascending slot (leader.rs:124-174)
          if (common && ids_[q] == r.from) continue;
          const uint64_t w = common ? o.aec[g] : o.ae[(size_t)q * G + g];
          if (w == JG_NO_ACK) continue;
          from();
          r.kind = JG_CMD_APPEND_ENTRIES, r.to_kind = JG_TO_PEER, r.to_id = ids_[q], r.id = w >> 8, r.aux = w & 0xffu;
          emit(r, r.id);
        }
      }
    if (o.answer)
      for (uint32_t g = 0; g < G; g++) {
        const uint64_t w = o.answer[g];
        if (w == JG_NO_ACK) continue;
        jg_msg_row r{};
        r.group = g, r.from = self_id(g), r.to_kind = JG_TO_PEER, r.to_id = answers_to ? (*answers_to)[g] : 0;
        if ((w & 0xffu) != JG_HB_NONE) {  // HeartbeatResponse first: the Heartbeat was applied first (follower.rs:209-215)
          r.kind = JG_CMD_HEARTBEAT_RESPONSE, r.flag = (uint8_t)(w & 0xffu), r.term = 0, r.id = o.hb_commit[g], r.aux = 0;
          emit(r, r.id);
        }
        if ((w >> 8) != JG_MAILBOX_NONE) {  // AppendResponse{node_id, head, success} (follower.rs:163-172); its `term` is
          r.kind = JG_CMD_APPEND_RESPONSE, r.flag = 1, r.term = 0, r.id = w >> 8, r.aux = 0;  // not on the dense wire: no
          emit(r, r.id);                                                                    // leader reads it (leader.rs:211-219)
        }
      }
  }

  void pump() {
    size_t n = 0;
    if (fsm_rows_tx && msg_rows_tx) {  // batch sinks on both channels: the engine's pinned queues as they are
      const jg_fsm_row* fv = nullptr;
      check(jg_drain_applies_view(e_, &fv, &n));
      if (n) fsm_rows_tx(fv, n);
      const jg_msg_row* mv = nullptr;
      check(jg_drain_messages_view(e_, &mv, &n));
      if (n) msg_rows_tx(mv, n);
      return;
    }
    check(jg_drain_applies(e_, nullptr, 0, &n));
    std::vector<jg_fsm_row> fr(n);
    if (n) check(jg_drain_applies(e_, fr.data(), n, &n));
    if (fsm_rows_tx) {
      if (n) fsm_rows_tx(fr.data(), n);
      fr.clear();
    }
    auto deliver = [&](const jg_fsm_row& r) {
      BlockStore& st = stores_[r.group];
      if (r.kind == JG_FSM_NOTIFY) {
        // leader append (leader.rs:177-188): the block now exists with the request's payload
        Block b{r.a, r.a ? st.prev_key(r.a) : 0, {}};
        auto it = pending_reqs_.find({r.group, r.b});
        if (it != pending_reqs_.end()) {
          b.data = it->second;
          pending_reqs_.erase(it);
        }
        st.insert(b);
        if (fsm_tx) {
          Instruction ins;
          ins.kind = Instruction::Notify, ins.group = r.group, ins.request_id = r.b, ins.block_id = r.a;
          fsm_tx(ins);
        }
      } else {
        // range(a..=b).skip(1) for the leader (leader.rs:93), range(a..b) for a follower
        // (follower.rs:204): by key order over what is stored, not by parent pointers.
        st.set_commit(r.b);  // chain.commit(id) persists the "commit" key first (chain.rs:195-205)
        const BlockId hi = r.b;
        const std::vector<Block> blocks = st.range(r.a, &hi, r.kind == JG_FSM_APPLY_LEADER);
        bool skip = r.kind == JG_FSM_APPLY_LEADER;
        for (const Block& blk : blocks) {
          if (skip) {
            skip = false;
            continue;
          }
          if (fsm_tx) {
            Instruction ins;
            ins.kind = Instruction::Apply, ins.group = r.group, ins.block = blk;
            fsm_tx(ins);
          }
        }
      }
    };
    for (const jg_fsm_row& r : fr) {
      if (r.kind != JG_FSM_LEADER_STEP) {
        deliver(r);
        continue;
      }
      // JG_NODE_FSM_FUSED: a leader's step as one row - Apply {c0, c1}, Notify {a, b}, Apply {c1, c2} (josefine_gpu.h)
      const uint64_t c0 = r.a - r.pad[0], c1 = r.a - r.pad[1], c2 = r.a - r.pad[2];
      jg_fsm_row x{};
      x.group = r.group;
      if (c1 != c0) x.kind = JG_FSM_APPLY_LEADER, x.a = c0, x.b = c1, deliver(x);
      x.kind = JG_FSM_NOTIFY, x.a = r.a, x.b = r.b, deliver(x);
      if (c2 != c1) x.kind = JG_FSM_APPLY_LEADER, x.a = c1, x.b = c2, deliver(x);
    }
    check(jg_drain_messages(e_, nullptr, 0, &n));
    std::vector<jg_msg_row> mr(n);
    if (n) check(jg_drain_messages(e_, mr.data(), n, &n));
    if (msg_rows_tx) {
      if (n) msg_rows_tx(mr.data(), n);
      mr.clear();
    }
    for (const jg_msg_row& r : mr) {
      std::deque<uint64_t>& q = queued_[r.group];
      if (r.kind == JG_CMD_CLIENT_REQUEST && r.to_kind == JG_TO_QUEUE) {
        if (r.flag == JG_QUEUE_DROP) q.clear();  // follower.rs:295 / candidate.rs:216-238
        else q.push_back(r.id);                  // follower.rs:268, candidate.rs:191
        continue;
      }
      if (r.kind == JG_CMD_CLIENT_REQUEST && r.flag == JG_QUEUE_FLUSH) {  // follower.rs:190-197
        for (uint64_t tok : q) emit(r, tok);
        q.clear();
        continue;
      }
      emit(r, r.id);
    }
  }
  void emit(const jg_msg_row& r, uint64_t id) {
    if (!rpc_tx) return;
    Message m;
    m.group = r.group;
    m.from = Address{JG_TO_PEER, r.from};
    m.to = Address{r.to_kind, r.to_id};
    m.command.kind = r.kind;
    m.command.from = r.from;
    m.command.term = r.term;
    m.command.id = id;
    m.command.aux = r.aux;
    m.command.flag = r.flag != 0 && r.kind != JG_CMD_CLIENT_REQUEST;
    if (r.kind == JG_CMD_APPEND_ENTRIES) {  // leader.rs:124-174: range(id..).skip(1).take(aux)
      // (the engine has counted the blocks: the iterator is never pulled past them into the "commit" key, Q9)
      const std::vector<Block> blocks = stores_[r.group].range(r.id, nullptr, false, (size_t)r.aux + 1);
      for (size_t k = 1; k < blocks.size(); k++) m.command.blocks.push_back(blocks[k]);
    }
    if (r.kind == JG_CMD_CLIENT_REQUEST) {
      auto it = pending_reqs_.find({r.group, id});
      if (it != pending_reqs_.end()) m.command.proposal = it->second;
    }
    rpc_tx(m);
  }
  jg_engine* e_ = nullptr;
  jg_node_outbox last_outbox_{};
  std::vector<BlockStore> stores_;
  std::vector<std::deque<uint64_t>> queued_;
  std::vector<NodeId> ids_;
  std::vector<uint8_t> kind_, flag_;
  std::vector<uint32_t> group_, from_;
  std::vector<uint64_t> term_, id_, aux_, blk_id_, blk_next_;
  std::vector<std::pair<uint32_t, Block>> pending_blocks_;
  std::deque<std::vector<std::pair<uint32_t, Block>>> step_blocks_;  // one entry per node step begun and not finished (oldest first)
  std::set<uint32_t> extend_failed_;  // partitions whose process died in Chain::extend (until their restart)
  std::map<std::pair<uint32_t, uint64_t>, std::vector<uint8_t>> pending_reqs_;
};

// ---- dense node tick over device-resident mailboxes -------------------------------------------
// The steady-state loop of a batched event loop (server.rs:103-165 for many partitions): per
// TICK one leader half on the node that leads and one follower half on every other node, the
// messages being columns in HBM (include/josefine_gpu.h "dense node tick").  `DenseCluster`
// owns the mailbox columns of one leader node and its followers when all of them live in this
// process (one engine per node; e.g. the 3 brokers of examples/multi-node in one runtime) and
// chains the engines' streams with jg_stream_wait — no host synchronisation per round.
class DenseCluster {
 public:
  // nodes[r] hosts replica slot r of every group; nodes[lead] leads them
  DenseCluster(std::vector<jg_engine*> nodes, uint32_t n_groups, uint32_t lead, NodeId lead_id)
      : nodes_(std::move(nodes)), G_(n_groups), R_((uint32_t)nodes_.size()), lead_(lead), lead_id_(lead_id) {
    jg_engine* L = nodes_[lead_];
    answers_ = (uint64_t*)alloc(L, 8ull * R_ * G_);  // the leader's inbox: one answer word per slot and group
    hbr_commit_ = (uint64_t*)alloc(L, 8ull * R_ * G_);
    o_beat_ = (jg_leader_beat*)alloc(L, sizeof(jg_leader_beat) * (size_t)G_);
    o_ae_ = (uint64_t*)alloc(L, 8ull * R_ * G_);
    std::vector<uint64_t> a((size_t)R_ * G_, JG_NO_ACK);  // nothing from anybody; own slot: zero appends
    for (uint32_t g = 0; g < G_; g++) a[(size_t)lead_ * G_ + g] = JG_ANSWER(0, JG_HB_NONE);
    check(jg_device_upload(L, answers_, a.data(), a.size() * 8));
    std::fill(a.begin(), a.end(), JG_NO_ACK);  // (the lead node's own row of the AppendEntries block is never written: jg_leader_outbox)
    check(jg_device_upload(L, o_ae_, a.data(), a.size() * 8));
  }
  ~DenseCluster() {
    for (void* p : bufs_) (void)jg_device_free(nodes_[lead_], p);
  }
  // number of ClientRequests every group appends in the next round (leader.rs:177-197)
  void set_appends(const std::vector<uint64_t>& n) {
    std::vector<uint64_t> w(G_);  // the own slot's answer words
    for (uint32_t g = 0; g < G_; g++) w[g] = JG_ANSWER(n[g], JG_HB_NONE);
    check(jg_device_upload(nodes_[lead_], answers_ + (size_t)lead_ * G_, w.data(), (size_t)G_ * 8));
  }
  // one protocol round at logical time now_ms
  void round(uint64_t now_ms) {
    jg_engine* L = nodes_[lead_];
    for (uint32_t r = 0; r < R_; r++)
      if (r != lead_) check(jg_stream_wait(L, nodes_[r]));  // last round's answers are in
    jg_leader_inbox in{answers_, hbr_commit_};
    jg_leader_outbox out{o_beat_, o_ae_};
    check(jg_step_dense_leader(L, now_ms, &in, &out));
    for (uint32_t r = 0; r < R_; r++) {
      if (r == lead_) continue;
      check(jg_stream_wait(nodes_[r], L));
      jg_follower_inbox fi{};
      fi.leader = nullptr, fi.leader_id = lead_id_;
      fi.beat = o_beat_, fi.ae = o_ae_ + (size_t)r * G_;
      jg_follower_outbox fo{answers_ + (size_t)r * G_, hbr_commit_ + (size_t)r * G_};
      check(jg_step_dense_follower(nodes_[r], now_ms, &fi, &fo, 1));
    }
  }
  void sync() {
    for (jg_engine* e : nodes_) check(jg_sync(e));
  }

 private:
  static void check(int rc) {
    if (rc != JG_OK) throw EngineError(rc, jg_last_error());
  }
  void* alloc(jg_engine* e, size_t bytes) {
    void* p = nullptr;
    check(jg_device_alloc(e, bytes, &p));
    bufs_.push_back(p);
    return p;
  }
  std::vector<jg_engine*> nodes_;
  uint32_t G_, R_, lead_;
  NodeId lead_id_;
  uint64_t *answers_, *hbr_commit_, *o_ae_;
  jg_leader_beat* o_beat_;
  std::vector<void*> bufs_;
};

// The same closed loop driven from inside the library (jg_dense_cluster_*: the round replayed as a
// hipGraph), and - round_routed - with the library's device-side transport for everything outside the
// mailbox vocabulary: the votes of an election travel between the nodes' engines without a host in
// between (the in-process stand-in for rpc_tx -> tcp.rs -> the peer's event_loop, server.rs:127-137).
class LibraryCluster {
 public:
  LibraryCluster(const std::vector<jg_engine*>& nodes, uint32_t lead) : R_((uint32_t)nodes.size()) {
    check(jg_dense_cluster_create(nodes.data(), R_, lead, &c_));
  }
  ~LibraryCluster() { jg_dense_cluster_destroy(c_); }
  LibraryCluster(const LibraryCluster&) = delete;
  LibraryCluster& operator=(const LibraryCluster&) = delete;
  void set_appends(uint64_t per_group_and_round) { check(jg_dense_cluster_set_appends(c_, per_group_and_round, nullptr)); }
  void rounds(uint64_t now_ms, uint64_t dt_ms, uint32_t n) { check(jg_dense_cluster_rounds(c_, now_ms, dt_ms, n)); }
  // `inject`: per node a device batch (jg_step_device_rows' form) or n == 0; empty = nothing for anybody
  jg_route_stats round_routed(uint64_t now_ms, const std::vector<jg_cmd_batch>& inject = {}) {
    jg_route_stats st{};
    check(jg_dense_cluster_round_routed(c_, now_ms, inject.empty() ? nullptr : inject.data(), &st));
    return st;
  }

 private:
  static void check(int rc) {
    if (rc != JG_OK) throw EngineError(rc, jg_last_error());
  }
  uint32_t R_;
  jg_dense_cluster* c_ = nullptr;
};

// ---- fsm::Driver and server::event_loop for many partitions ---------------------------------------
// The two tasks on either side of Raft<T> in a josefine process, batched over every partition the
// process hosts.  Logical time: `run_until(now_ms)` plays the
// interval timer of server.rs:25,113 (one Command::Tick per partition every 100 ms).
//
//   reference                                              here
//   fsm::Driver::run  (fsm.rs:51-88)                       BatchedEventLoop::on_instruction
//     Notify -> notifications[block_id] = (addr, req id)     notifications_[{group, block_id}]
//     Apply  -> skip block 0; fsm.transition(data); if a     fsm(group, data); a pending notification
//               notification is pending, ClientResponse        completes the client's request
//               {id, res} to that address via rpc_tx           (the hop rpc_tx -> event_loop ->
//                                                              oneshot of server.rs:144-152 is direct)
//   server::event_loop (server.rs:103-165)                 BatchedEventLoop::run_until
//     step_interval.tick() => apply(Tick)                    one Tick per partition per TICK_MS
//     tcp_rx.recv()        => apply(msg.command)             tcp_rx(msg): queued, applied at the next step
//     rpc_rx.recv(): Peer / Peers => tcp_tx.send(msg)        tcp_tx sink
//                    Local        => apply(msg.command)      re-submitted in the same step
//                    Client       => requests.remove(id)     completes the client's request
//     client_rx.recv()     => apply(ClientRequest{new id})   propose(group, proposal, on_response)
class BatchedEventLoop {
 public:
  static constexpr uint64_t TICK_MS = 100;  // server.rs:25
  using Response = std::function<void(bool ok, const std::vector<uint8_t>& res)>;
  // Fsm::transition (fsm.rs:16): state machine of one partition; an exception is a ResponseError
  std::function<std::vector<uint8_t>(uint32_t group, const std::vector<uint8_t>& data)> fsm;
  std::function<void(const Message&)> tcp_tx;  // to the peer transport (tcp.rs), by partition
  // What the interval's Tick is (server.rs:125): JG_NODE_TICK through jg_step_node - ONE flag for all
  // partitions, the inbound rows classified and served by the dense kernels on the device - or, with
  // dense = false, one Tick row per partition through jg_submit + jg_step (the general state machine only:
  // round 2's loop, kept for the A/B in tests/cpp/bench_event_loop.cpp).
  bool dense = true;
  uint32_t halves = JG_NODE_LEADER_HALF | JG_NODE_FOLLOWER_HALF;
  // ONE loop that overlaps with itself: a step returns as soon as its rows are on the device and classified; what it
  // pushed on fsm_tx / rpc_tx is delivered at the start of the NEXT step (or by flush()) - the channels of the
  // reference are asynchronous too (mod.rs:337-340) -, so that the transport decodes tick t + 1 into the engine's
  // pinned columns while the device runs tick t's kernels and sends its outputs home.
  bool pipelined = false;
  // ... with TWO ticks in flight (2; needs pipelined): what step t pushed on fsm_tx / rpc_tx is delivered at the start of
  // step t + 2 - step t + 1, begun a tick ago, keeps the device busy meanwhile (JG_NODE_KEEP: the engine keeps a step's
  // outbox and rows in a set of its own until they have been viewed, and a view waits for ITS step only) - so that a tick's
  // whole chain (rows up, kernels, columns and fsm rows down: ~3 ms at 1 M x 5) has two periods to complete and the loop's
  // period is its slowest stage (the transport's decoding + the sinks, the bus one way, the bus the other way), not their
  // sum.  The channels of the reference promise an order, not a latency (mod.rs:337-340): every channel carries the same
  // rows in the same order per step, one step later than with 1.
  uint32_t in_flight = 1;
  // the node step's compact bus formats (ABI v7): JG_NODE_COMMON_AE - the Tick's AppendEntries words come home as one
  // word per partition where the followers' agree - and / or JG_NODE_FSM_FUSED - a leader's fsm_tx rows of a step as one
  // row; both are expanded again by BatchedRaft before rpc_tx / fsm_tx see them (a column / row sink sees the compact form)
  uint32_t bus = 0;

  explicit BatchedEventLoop(BatchedRaft& raft, uint32_t n_groups) : raft_(raft), G_(n_groups), answers_to_(n_groups, 0) {
    raft_.rpc_tx = [this](const Message& m) { on_message(m); };
    raft_.fsm_tx = [this](const Instruction& i) { on_instruction(i); };
  }
  // a message from a peer's event loop (tcp_rx, server.rs:126-137): queued as a ROW (structure of arrays),
  // applied at the next step; block payloads go to the partition's store when the step has extended the chain
  void tcp_rx(const Message& m) {
    const Command& c = m.command;
    if (c.kind == JG_CMD_APPEND_ENTRIES) {
      in_.push_append_entries(m.group, c.from, c.term, c.blocks.data(), c.blocks.size());
      for (const Block& b : c.blocks) in_blocks_.push_back({m.group, b});
    } else {
      // (a HeartbeatResponse does not name its sender in the Command; the Message does: rpc.rs:17-27)
      const NodeId from = c.kind == JG_CMD_HEARTBEAT_RESPONSE ? m.from.peer : c.from;
      in_.push(m.group, c.kind, from, c.term, c.id, c.aux, c.flag ? 1 : 0);
      if (c.kind == JG_CMD_CLIENT_REQUEST) proxied_[{m.group, c.id}] = c.proposal;
    }
    // (who a partition's answers go to - the sender of the Heartbeat / AppendEntries being answered - is read off the
    // row queue when the step that applies it begins: step(); a pipelined loop decodes tick t + 1 while tick t's
    // answers are still to be addressed)
  }
  // a peer's whole frame of rows at once (payload-free: a columnar transport ships block data separately)
  void tcp_rx_rows(const jg_cmd_batch& b) { in_.append(b); }
  // ... or decoded straight into the engine's pinned columns: no copy on the host at all (rows committed this
  // way are applied before the rows queued through tcp_rx / tcp_rx_rows of the same step)
  jg_cmd_cols tcp_rx_reserve(size_t n, size_t n_blocks = 0) { return raft_.reserve_rows(n, n_blocks); }
  void tcp_rx_commit(size_t n, size_t n_blocks, uint32_t optional_columns) {
    raft_.commit_rows(n, n_blocks, optional_columns);
    direct_rows_ += n;
  }
  // ... and a batched peer ships its answers as the COLUMN it produced (jg_node_outbox.answer): 8 bytes per partition
  // instead of two rows; filled in place, applied by the next step's leader half (hb_commit == nullptr: not needed)
  void tcp_rx_answer_column(uint32_t peer_slot, uint64_t** answer, uint64_t** hb_commit) {
    raft_.inbox_columns(peer_slot, answer, hb_commit);
    direct_rows_ += 1;  // (something to apply even without a Tick)
  }
  // RaftClient::propose (client.rs:35): returns the request id (Uuid::new_v4 -> a counter)
  uint64_t propose(uint32_t group, std::vector<uint8_t> proposal, Response on_response) {
    const uint64_t id = ++next_request_;
    requests_[id] = std::move(on_response);
    in_.push(group, JG_CMD_CLIENT_REQUEST, 0, 0, id, 0, 0);
    proposals_.push_back({group, id, std::move(proposal)});
    return id;
  }
  // Advance logical time to now_ms: everything that arrived is applied, a Tick per partition
  // whenever the interval fires (the first one immediately, like tokio::time::interval).
  void run_until(uint64_t now_ms) {
    for (;;) {
      const bool tick = next_tick_ <= now_ms;
      const uint64_t at = tick ? next_tick_ : now_ms;
      if (!tick && in_.empty() && !direct_rows_) break;
      step(at, tick);
      if (tick) next_tick_ += TICK_MS;
    }
  }
  // pipelined: deliver what the last step pushed on fsm_tx / rpc_tx (run_until does it at the start of the next step)
  void flush() {
    while (raft_.node_step_open()) finish_oldest();
    drain_local(last_at_);
  }
  size_t pending_requests() const { return requests_.size(); }
  size_t queued_rows() const { return in_.size(); }

 private:
  void step(uint64_t at, bool tick) {
    // (pipelined: the previous step's outputs first - its pinned queues and outbox are about to be reused, the blocks
    // noted below belong to THIS step (BatchedRaft::after_step stores a step's blocks where its extend succeeded), and
    // its answers go to the senders of ITS Heartbeat / AppendEntries rows, not to this step's)
    const bool two = dense && pipelined && in_flight >= 2;
    if (two) {
      // two ticks in flight: what the step BEFORE LAST pushed on the channels is delivered now - its outputs have had this
      // long to travel home - while the last step, begun a tick ago, keeps the device busy
      while (raft_.node_steps_open() >= 2) finish_oldest();
      if (!local_.empty()) flush();  // (Address::Local messages are applied by a step of their own: nothing may be in flight)
    } else if (dense) {
      flush();
    }
    // (who THIS step's answers go to: noted per step, applied when the step is finished - with two ticks in flight the
    // step before is finished after this one has begun, and its answers go to the senders of ITS rows)
    answers_of_step_.emplace_back();
    for (size_t i = 0; i < in_.size(); i++)
      if (in_.kind[i] == JG_CMD_HEARTBEAT || in_.kind[i] == JG_CMD_APPEND_ENTRIES) answers_of_step_.back().push_back({in_.group[i], in_.from[i]});
    // payload-carrying rows (client proposals, blocks) go through submit() so that BatchedRaft's request /
    // block mirrors see them; everything else is one bulk jg_submit of the row queue
    for (auto& p : proposals_) raft_.note_proposal(p.group, p.id, std::move(p.data));
    proposals_.clear();
    for (auto& kv : proxied_) raft_.note_proposal(kv.first.first, kv.first.second, kv.second);
    proxied_.clear();
    for (auto& pb : in_blocks_) raft_.note_block(pb.first, pb.second);
    in_blocks_.clear();
    direct_rows_ = 0;
    if (dense) {
      if (!in_.empty()) raft_.submit_rows(in_.view());
      in_.clear();
      raft_.step_node_begin(at, halves | bus | (tick ? (uint32_t)JG_NODE_TICK : 0u), pipelined, two);
      last_at_ = at;
      if (pipelined) return;
      finish_oldest();
    } else {
      answers_of_step_.pop_back();
      if (tick)
        for (uint32_t g = 0; g < G_; g++) in_.push(g, JG_CMD_TICK);
      if (!in_.empty()) raft_.submit_rows(in_.view());
      in_.clear();
      raft_.step(at);
    }
    drain_local(at);
  }
  void finish_oldest() {
    if (!answers_of_step_.empty()) {
      for (const auto& u : answers_of_step_.front()) answers_to_[u.first] = u.second;
      answers_of_step_.pop_front();
    }
    raft_.step_node_finish(&answers_to_);
  }
  // Address::Local messages (server.rs:143) are applied before anything new is accepted
  void drain_local(uint64_t at) {
    int guard = 0;
    while (!local_.empty() && guard++ < 64) {
      std::deque<Message> lo;
      lo.swap(local_);
      for (const Message& m : lo) raft_.submit(m.group, m.command);
      raft_.step(at);
    }
  }
  void on_message(const Message& m) {  // rpc_rx (server.rs:139-155)
    switch (m.to.kind) {
      case JG_TO_PEER:
      case JG_TO_PEERS:
        if (tcp_tx) tcp_tx(m);
        break;
      case JG_TO_LOCAL: local_.push_back(m); break;
      case JG_TO_CLIENT: complete(m.command.id, true, {}); break;
      default: throw EngineError(JG_EINVAL, "unexpected message");  // server.rs:154
    }
  }
  void on_instruction(const Instruction& i) {  // fsm.rs:56-84
    const std::pair<uint32_t, BlockId> key{i.group, i.kind == Instruction::Notify ? i.block_id : i.block.id};
    if (i.kind == Instruction::Notify) {
      notifications_[key] = i.request_id;
      return;
    }
    if (i.block.id == 0) return;  // fsm.rs:59-61
    bool ok = true;
    std::vector<uint8_t> res;
    try {
      if (fsm) res = fsm(i.group, i.block.data);
    } catch (...) {
      ok = false;
    }
    auto it = notifications_.find(key);
    if (it != notifications_.end()) {
      complete(it->second, ok, res);
      notifications_.erase(it);
    }
  }
  void complete(uint64_t id, bool ok, const std::vector<uint8_t>& res) {
    auto it = requests_.find(id);
    if (it == requests_.end()) return;  // a proxied request: its oneshot lives on another node (server.rs:129-133)
    Response cb = std::move(it->second);
    requests_.erase(it);
    if (cb) cb(ok, res);
  }

  struct Proposal {
    uint32_t group;
    uint64_t id;
    std::vector<uint8_t> data;
  };
  BatchedRaft& raft_;
  uint32_t G_;
  uint64_t next_tick_ = 0, next_request_ = 0, last_at_ = 0;
  size_t direct_rows_ = 0;                                // rows committed in place since the last step
  RowQueue in_;                                           // tcp_rx + client_rx since the last step, stream order
  std::vector<std::pair<uint32_t, Block>> in_blocks_;     // payloads of the AppendEntries rows in in_
  std::vector<Proposal> proposals_;                       // payloads of the ClientRequest rows in in_
  std::map<std::pair<uint32_t, uint64_t>, std::vector<uint8_t>> proxied_;
  std::vector<NodeId> answers_to_;                        // per partition: who its Heartbeat / AppendEntries came from
  std::deque<std::vector<std::pair<uint32_t, NodeId>>> answers_of_step_;  // ... as noted by the steps begun and not finished
  std::deque<Message> local_;
  std::map<uint64_t, Response> requests_;
  std::map<std::pair<uint32_t, BlockId>, uint64_t> notifications_;
};

inline RaftHandle RaftHandle::apply(const Command& cmd, uint64_t now_ms) { return e_->apply(g_, cmd, now_ms); }
inline uint64_t RaftHandle::read64(int f) const {
  uint64_t v = 0;
  e_->check(jg_read_state(e_->e_, f, 0, &v, g_, 1));
  return v;
}
inline uint32_t RaftHandle::read32(int f) const {
  uint32_t v = 0;
  e_->check(jg_read_state(e_->e_, f, 0, &v, g_, 1));
  return v;
}
inline uint8_t RaftHandle::read8(int f) const {
  uint8_t v = 0;
  e_->check(jg_read_state(e_->e_, f, 0, &v, g_, 1));
  return v;
}

}  // namespace josefine
