// raft_oracle.hpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the hot path of tychedelia/josefine's `src/raft` (the
// "Chained Raft" state machine), one group at a time, written to mirror the
// reference function by function.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may build or call this; the shipped engine
// (josefine_amd/csrc) never links it and has no CPU fallback.
//
// PARITY PINNING: the Rust reference cannot be compiled in this environment
// (no cargo/rustc, crates not vendored — SURVEY.md §8(c)), so there is no
// oracle/_ref.  This restatement is pinned against every known-answer test the
// reference's own test modules hold for this path (single-node cases, the
// chain tests incl. the only `compact` vector) in tests/test_reference_kats.py;
// every multi-replica result (majority, election, can_vote clauses …) is
// "parity unpinned" by the reference itself — those rest on fidelity to the
// cited lines plus the hand-derived vectors of SURVEY.md §8(c).
//
// Where the reference panics or returns Err the group records a sticky fault
// code (include/josefine_gpu.h JG_FAULT_*) and stops applying commands, which is
// what happens to the reference process (event_loop propagates with `?`,
// src/raft/server.rs:125-159).  Wall-clock and thread_rng are replaced by a
// logical clock `now_ms` and a counter-based RNG.
//
// All `file:line` citations are relative to /root/reference.
#pragma once
#include <cstdint>
#include <map>
#include <set>
#include <unordered_map>
#include <vector>
#include <algorithm>

#include "../include/josefine_gpu.h"

namespace jo {

using NodeId = uint32_t;   // src/raft/mod.rs:136
using Term = uint64_t;     // src/raft/mod.rs:139
using BlockId = uint64_t;  // src/raft/chain.rs:29-36: 8-byte BE bytes; Ord == u64 order

// splitmix64 finaliser — the counter-based RNG shared (by specification, not by
// code) with the device engine; see DESIGN.md "Logical time and randomness".
static inline uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

struct Block {  // src/raft/chain.rs:86-91 (payload `data` stays with the host store)
  BlockId id;
  BlockId next;
};

struct Msg {  // src/raft/rpc.rs:17-27 flattened; field use per include/josefine_gpu.h
  uint8_t kind, to_kind, flag;
  NodeId to_id, from;
  uint64_t term, id, aux;
};

struct FsmRow {  // src/raft/fsm.rs:20-29, Apply run-length encoded as a key range
  uint8_t kind;
  uint64_t a, b;
};

struct Cmd {  // src/raft/mod.rs:160-227
  uint8_t kind = JG_CMD_NOOP;
  NodeId from = 0;
  Term term = 0;
  uint64_t id = 0;
  uint64_t aux = 0;
  uint8_t flag = 0;
  std::vector<Block> blocks;  // AppendEntries.blocks
};

struct Timing {
  uint32_t heartbeat_timeout_ms = 100;  // src/raft/config.rs:104
  uint32_t election_min_ms = 500;       // src/raft/mod.rs:318
  uint32_t election_max_ms = 1000;      // src/raft/mod.rs:319
  uint64_t seed = 0;
  bool separate_commit_key = false;  // JG_CFG_SEPARATE_COMMIT_KEY (do not reproduce Q9)
};

// ---------------------------------------------------------------------------
// Chain — src/raft/chain.rs:99-254.  sled is an ordered map over byte keys; the
// block keys are 8-byte BE ids, so std::map<u64,…> iterates in the same order.
// The "commit" key lives in the same tree and sorts after every id below
// 0x636f6d6d69740000 (chain.rs:198) — modelled by `commit_key`.
struct Chain {
  std::map<BlockId, Block> db;
  bool commit_key = false;
  uint64_t id_gen = 0;
  BlockId commit = 0, head = 0;

  // Chain::new on an empty directory (chain.rs:117-137) + init (139-153).
  void open_fresh() {
    db.clear();
    commit_key = false;
    reopen();
  }
  // Chain::new on an existing tree: reload "commit", head = id_gen = commit,
  // re-run init() when commit == 0 (chain.rs:119-134).
  void reopen() {
    uint64_t c = commit_key ? commit : 0;
    id_gen = c;
    commit = c;
    head = c;
    if (c == 0) {
      uint64_t id = id_gen++;  // chain.rs:140 (assert_eq!(id, 0) holds)
      db[id] = Block{id, id};  // genesis: next == id == 0 (chain.rs:143-150)
    }
  }
  bool has(BlockId id) const { return db.count(id) != 0; }  // chain.rs:155-157

  // chain.rs:160-175.  Returns fault code or 0; *out = new id.
  int append(BlockId* out) {
    uint64_t id = id_gen++;                                  // :161
    if (!(id > head)) return JG_FAULT_APPEND_ID_NOT_ABOVE_HEAD;  // :163
    db[id] = Block{id, head};                                // :164-172 (sled insert = upsert)
    head = id;                                               // :173
    *out = id;
    return 0;
  }
  // chain.rs:178-192
  int extend(const Block& b) {
    if (!has(b.next)) return JG_FAULT_EXTEND_MISSING_PARENT;  // :180-185
    db[b.id] = b;                                             // :187-189
    head = b.id;                                              // :190
    return 0;
  }
  // chain.rs:195-205
  int do_commit(BlockId id) {
    if (has(id)) {
      commit_key = true;  // :198
      commit = id;        // :199
      return 0;
    }
    return JG_FAULT_COMMIT_MISSING_BLOCK;  // :200-202
  }
  // chain.rs:239-253 — returns the removed ids in walk order.
  std::vector<BlockId> compact() {
    std::vector<BlockId> removed;
    bool have_next = false;
    BlockId next_id = 0;
    // range(BlockId(0)..commit).rev()
    auto lo = db.lower_bound(0), hi = db.lower_bound(commit);
    std::vector<Block> walk;
    for (auto it = lo; it != hi; ++it) walk.push_back(it->second);
    for (auto it = walk.rbegin(); it != walk.rend(); ++it) {
      if (have_next && it->id != next_id) {  // :244
        removed.push_back(it->id);
        db.erase(it->id);  // :246
      }
      have_next = true;
      next_id = it->next;  // :249 — also for a block just removed (Q7)
    }
    return removed;
  }
  // Number of *items* an unbounded `range(from..)` yields before the iterator would
  // have to decode the "commit" key, i.e. block keys >= from.
  size_t blocks_from(BlockId from) const {
    return (size_t)std::distance(db.lower_bound(from), db.end());
  }
  // "Run form built by append only" (include/josefine_gpu.h, dense mailbox vocabulary): the id
  // set is exactly [0, head], every block's parent is id-1 and id_gen == head+1 — then the
  // blocks after any key are id-consecutive and an AppendEntries fits the (from, n) columns.
  // O(#blocks): test infrastructure.
  bool run_form_by_append() const {
    if (id_gen != head + 1 || db.size() != head + 1) return false;
    for (auto& kv : db)
      if (kv.second.next != (kv.first ? kv.first - 1 : 0)) return false;
    return !db.empty() && db.rbegin()->first == head;
  }
  // The id set is a run [0, top] and every block's parent its predecessor - whatever head and id_gen are (a restarted
  // replica's head is its commit index, below the top of what sled kept: chain.rs:117-137): blocks after any key are
  // id-consecutive, so an AppendEntries still fits the (from, n) mailbox word.
  bool run_form() const {
    if (db.empty() || db.size() != db.rbegin()->first + 1) return false;
    for (auto& kv : db)
      if (kv.second.next != (kv.first ? kv.first - 1 : 0)) return false;
    return true;
  }
  BlockId top() const { return db.empty() ? 0 : db.rbegin()->first; }
};

// ---------------------------------------------------------------------------
// Election — src/raft/election.rs:6-73
enum class ElectionStatus { Elected, Voting, Defeated };
struct Election {
  std::vector<NodeId> voter_ids;
  std::unordered_map<NodeId, bool> votes;
  void reset() { votes.clear(); }                          // :29-31
  void vote(NodeId id, bool v) { votes[id] = v; }          // :33-35 (insert overwrites)
  size_t quorum_size() const {                             // :66-73
    if (voter_ids.size() == 1) return 0;
    return voter_ids.size() / 2 + 1;
  }
  ElectionStatus status() const {                          // :37-57
    size_t yes = 0, total = 0;
    for (auto& kv : votes) {
      if (kv.second) yes++;
      total++;
    }
    if (yes >= quorum_size()) return ElectionStatus::Elected;
    if (total - yes == quorum_size()) return ElectionStatus::Defeated;
    return ElectionStatus::Voting;
  }
};

// ---------------------------------------------------------------------------
// ReplicationProgress — src/raft/progress.rs:9-232 (Snapshot state is never
// constructed in the reference, so only Probe / Replicate exist here).
struct NodeProgress {
  bool replicate = false;  // false = Probe (progress.rs:155-162 initial state)
  BlockId head = 0;
  // progress.rs:76-94 with Progress::increment (133-140)
  void advance(BlockId id) {
    bool inc = head < id;
    if (inc) head = id;
    replicate = inc;  // Probe+inc -> Replicate; Probe+!inc -> Probe; Repl+inc -> Repl; Repl+!inc -> Probe
  }
};
struct ReplicationProgress {
  std::unordered_map<NodeId, NodeProgress> progress;
  // progress.rs:42-46
  int advance(NodeId node, BlockId id) {
    auto it = progress.find(node);
    if (it == progress.end()) return JG_FAULT_PROGRESS_UNKNOWN_NODE;  // :43
    NodeProgress p = it->second;
    progress.erase(it);
    p.advance(id);
    progress[node] = p;
    return 0;
  }
  // progress.rs:48-60
  BlockId committed_index() const {
    std::vector<BlockId> idx;
    for (auto& kv : progress) idx.push_back(kv.second.head);
    std::sort(idx.begin(), idx.end(), [](BlockId a, BlockId b) { return b < a; });
    return idx[idx.size() / 2];
  }
};

// ---------------------------------------------------------------------------
// Raft<T> + RaftHandle — src/raft/mod.rs:326-341, 417-489.
struct Raft {
  // config (src/raft/config.rs): own id and `nodes` in config order.
  NodeId id = 1;
  std::vector<NodeId> nodes;
  uint64_t rng_key = 0;  // global group id: keys the timeout RNG
  Timing tm;

  // State (mod.rs:271-287)
  Term current_term = 0;
  bool has_voted = false;
  NodeId voted_for = 0;
  uint64_t election_time = 0;     // Some(..) always after init (follower.rs:93-95)
  uint32_t election_timeout = 0;
  uint32_t rng_draws = 0;

  int role = JG_ROLE_FOLLOWER;
  // Follower (follower.rs:19-23)
  bool has_leader = false;
  NodeId leader_id = 0;
  // queued client requests (follower.rs:22 / candidate.rs:20).  Request payloads stay
  // with the host adapter, which mirrors this queue from the QUEUE / FLUSH / DROP
  // rows below (include/josefine_gpu.h "client request queue rows"); the state
  // machine itself only needs the length.
  uint32_t queued_reqs = 0;
  // Candidate (candidate.rs:17-21)
  Election election;
  // Leader (leader.rs:24-30)
  ReplicationProgress progress;
  uint64_t heartbeat_time = 0;

  Chain chain;
  int fault = 0;

  // output "channels" (mod.rs:337-340)
  std::vector<Msg> rpc;
  std::vector<FsmRow> fsm;
  // accounting: SURVEY.md §8(d) "decision"
  uint64_t decisions = 0;

  // ---- construction: Raft::<Follower>::new (follower.rs:68-95) -------------
  void init(NodeId self_id, const std::vector<NodeId>& peer_ids, uint64_t key, const Timing& t,
            uint64_t now) {
    id = self_id;
    nodes = peer_ids;
    rng_key = key;
    tm = t;
    rng_draws = 0;
    chain.open_fresh();
    reset_volatile(now);
  }
  // State::default() + Follower role + set_election_timeout (follower.rs:78-95)
  void reset_volatile(uint64_t now) {
    current_term = 0;
    has_voted = false;
    voted_for = 0;
    role = JG_ROLE_FOLLOWER;
    has_leader = false;
    leader_id = 0;
    queued_reqs = 0;  // the process (and its in-memory queue) is gone: no DROP row
    election = Election{};
    progress = ReplicationProgress{};
    heartbeat_time = 0;
    fault = 0;
    set_election_timeout(now);
  }
  // process restart: Chain::new on the persisted tree + fresh volatile state
  // (empty_store: JG_CMD_RECREATE - Chain::new on an empty directory, chain.rs:117-153)
  void restart(uint64_t now, bool empty_store = false) {
    if (empty_store) chain.open_fresh();
    else chain.reopen();
    reset_volatile(now);
  }

  // follower.rs:103-113; thread_rng replaced by mix64(seed, group, draw#)
  void set_election_timeout(uint64_t now) {
    uint32_t span = tm.election_max_ms - tm.election_min_ms;
    uint64_t r = mix64(tm.seed ^ mix64(rng_key * 0xd1342543de82ef95ull + rng_draws));
    rng_draws++;
    election_timeout = tm.election_min_ms + (span ? (uint32_t)(r % span) : 0);
    election_time = now;
  }
  bool needs_election(uint64_t now) const {  // mod.rs:352-357
    return (now - election_time) > election_timeout;
  }
  // Raft::term (mod.rs:360-365) + Role::term per role
  // returns fault code (leader.rs:33-35 is unimplemented!())
  int set_term(Term t) {
    has_voted = false;
    voted_for = 0;
    current_term = t;
    switch (role) {
      case JG_ROLE_FOLLOWER: has_leader = false; leader_id = 0; return 0;  // follower.rs:27-29
      case JG_ROLE_CANDIDATE: election.reset(); return 0;                 // candidate.rs:161-163
      default: return JG_FAULT_LEADER_TERM_UNIMPLEMENTED;                 // leader.rs:33-35
    }
  }
  void send(uint8_t to_kind, NodeId to, const Msg& m0) {  // mod.rs:390-400
    Msg m = m0;
    m.to_kind = to_kind;
    m.to_id = to;
    rpc.push_back(m);
  }

  // ---- role transitions ------------------------------------------------------
  void become_candidate() {  // From<Raft<Follower>> for Raft<Candidate>, follower.rs:285-304
    election = Election{};
    election.voter_ids = nodes;
    election.voter_ids.push_back(id);
    role = JG_ROLE_CANDIDATE;
    // queued_reqs: Candidate starts with an empty Vec (follower.rs:295) — the
    // follower's queued requests are dropped.
    drop_queue();
    has_leader = false;
    leader_id = 0;
  }
  void drop_queue() {
    if (queued_reqs) {
      Msg m{};
      m.kind = JG_CMD_CLIENT_REQUEST;
      m.from = id;
      m.flag = JG_QUEUE_DROP;
      m.aux = queued_reqs;
      send(JG_TO_QUEUE, 0, m);
    }
    queued_reqs = 0;
  }
  void enqueue(uint64_t token) {
    Msg m{};
    m.kind = JG_CMD_CLIENT_REQUEST;
    m.from = id;
    m.id = token;
    send(JG_TO_QUEUE, 0, m);
    queued_reqs++;
  }
  void become_follower_from_candidate() {  // candidate.rs:198-214 (keeps queued_reqs)
    role = JG_ROLE_FOLLOWER;
    has_leader = false;
    leader_id = 0;
    election = Election{};
  }
  void become_follower_from_leader() {  // leader.rs:268-284
    role = JG_ROLE_FOLLOWER;
    has_leader = false;
    leader_id = 0;
    queued_reqs = 0;
    progress = ReplicationProgress{};
  }
  void become_leader(uint64_t now) {  // candidate.rs:216-238
    progress = ReplicationProgress{};
    for (NodeId n : nodes) progress.progress[n] = NodeProgress{};
    progress.progress[id] = NodeProgress{};
    heartbeat_time = now;
    role = JG_ROLE_LEADER;
    drop_queue();  // Leader has no queue; candidate's queued requests are dropped
    election = Election{};
  }

  // ---- Apply::apply dispatch (mod.rs:471-479) --------------------------------
  void apply(const Cmd& c, uint64_t now) {
    if (c.kind == JG_CMD_RESTART || c.kind == JG_CMD_RECREATE) {
      restart(now, c.kind == JG_CMD_RECREATE);
      return;
    }
    if (fault) return;  // the process is gone
    int f = 0;
    switch (role) {
      case JG_ROLE_FOLLOWER: f = follower_apply(c, now); break;
      case JG_ROLE_CANDIDATE: f = candidate_apply(c, now); break;
      default: f = leader_apply(c, now); break;
    }
    if (f && !fault) fault = f;
  }

  // ======================= Follower (follower.rs:36-64) =======================
  int follower_apply(const Cmd& c, uint64_t now) {
    switch (c.kind) {
      case JG_CMD_TICK: return follower_tick(now);
      case JG_CMD_APPEND_ENTRIES: return follower_append_entries(c, now);
      case JG_CMD_HEARTBEAT: return follower_heartbeat(c.from, c.term, c.id, now);
      case JG_CMD_VOTE_REQUEST: return follower_vote_request(c.from, c.aux, c.id);
      case JG_CMD_TIMEOUT: return follower_timeout(now);
      case JG_CMD_CLIENT_REQUEST: return follower_client_request(c.id);
      case JG_CMD_CLIENT_RESPONSE: {  // follower.rs:272-282
        Msg m{};
        m.kind = JG_CMD_CLIENT_RESPONSE;
        m.from = id;
        m.id = c.id;
        send(JG_TO_CLIENT, 0, m);
        return 0;
      }
      default: return 0;  // apply_self (follower.rs:62,115-117)
    }
  }
  int follower_tick(uint64_t now) {  // follower.rs:121-128
    if (needs_election(now)) return follower_timeout(now);
    return 0;
  }
  bool can_vote(Term last_term, BlockId head) const {  // follower.rs:97-101
    return !(has_voted || current_term > last_term || chain.commit > head);
  }
  int follower_append_entries(const Cmd& c, uint64_t now) {  // follower.rs:130-176
    NodeId leader = c.from;
    if (!has_voted && c.term >= current_term) {  // :137
      set_term(c.term);                          // :138
      election_time = now;                       // :141 (timeout duration unchanged)
      has_leader = true;                         // :142
      leader_id = leader;
      has_voted = true;                          // :143
      voted_for = leader;
    }
    if (has_voted) {  // :147-154
      if (voted_for != leader && c.term < current_term) return JG_FAULT_FOLLOWER_STALE_LEADER;
    }
    if (!c.blocks.empty()) {  // :157
      for (const Block& b : c.blocks) {
        int f = chain.extend(b);  // :159 (`?` — blocks before the failing one stay)
        if (f) return f;
      }
      Msg m{};  // :163-172
      m.kind = JG_CMD_APPEND_RESPONSE;
      m.from = id;
      m.term = current_term;
      m.id = chain.head;
      m.flag = 1;
      send(JG_TO_PEER, leader, m);
    }
    return 0;
  }
  int follower_heartbeat(NodeId leader, Term term, BlockId commit, uint64_t now) {  // follower.rs:178-217
    set_election_timeout(now);  // :184
    set_term(term);             // :185 — unconditional, even when lower (Q6)
    has_leader = true;          // :186
    leader_id = leader;
    has_voted = true;           // :187
    voted_for = leader;
    // :190-197 flush queued client requests to the leader, in order (one FLUSH row
    // = `aux` ClientRequest messages, expanded by the host adapter)
    if (queued_reqs) {
      Msg m{};
      m.kind = JG_CMD_CLIENT_REQUEST;
      m.from = id;
      m.flag = JG_QUEUE_FLUSH;
      m.aux = queued_reqs;
      send(JG_TO_PEER, leader, m);
    }
    queued_reqs = 0;
    bool has_committed = chain.has(commit);  // :200
    if (has_committed && commit > chain.commit) {  // :201
      BlockId prev = chain.commit;
      int f = chain.do_commit(commit);  // :203 (cannot fail: has() just held)
      if (f) return f;
      fsm.push_back(FsmRow{JG_FSM_APPLY_FOLLOWER, prev, commit});  // :204-206 range(prev..commit)
    }
    Msg m{};  // :209-215
    m.kind = JG_CMD_HEARTBEAT_RESPONSE;
    m.from = id;
    m.id = chain.commit;
    m.flag = has_committed ? 1 : 0;
    send(JG_TO_PEER, leader, m);
    return 0;
  }
  int follower_vote_request(NodeId cand, Term last_term, BlockId head) {  // follower.rs:219-246
    Msg m{};
    m.kind = JG_CMD_VOTE_RESPONSE;
    m.from = id;
    m.term = current_term;  // the follower's *unchanged* term (Q5)
    if (can_vote(last_term, head)) {
      m.flag = 1;
      send(JG_TO_PEER, cand, m);
      has_voted = true;  // :234
      voted_for = cand;
    } else {
      m.flag = 0;
      send(JG_TO_PEER, cand, m);
    }
    return 0;
  }
  int follower_timeout(uint64_t now) {  // follower.rs:248-256
    if (!has_voted) {
      set_election_timeout(now);
      become_candidate();
      return seek_election(now);
    }
    return 0;
  }
  int follower_client_request(uint64_t token) {  // follower.rs:258-270
    if (has_leader) {
      Msg m{};
      m.kind = JG_CMD_CLIENT_REQUEST;
      m.from = id;
      m.id = token;
      send(JG_TO_PEER, leader_id, m);
    } else {
      enqueue(token);
    }
    return 0;
  }

  // ======================= Candidate (candidate.rs) ==========================
  int seek_election(uint64_t now) {  // candidate.rs:24-45
    has_voted = true;  // :25
    voted_for = id;
    current_term += 1;  // :26
    for (size_t i = 0; i < nodes.size(); i++) {  // :30-37 — one *broadcast* per configured peer
      Msg m{};
      m.kind = JG_CMD_VOTE_REQUEST;
      m.from = id;
      m.term = current_term;
      m.aux = current_term;  // last_term: term (Q5)
      m.id = chain.head;
      send(JG_TO_PEERS, 0, m);
    }
    return candidate_vote_response(true, id, now);  // :40-44 self vote
  }
  int candidate_apply(const Cmd& c, uint64_t now) {  // candidate.rs:170-196
    switch (c.kind) {
      case JG_CMD_TICK: return candidate_tick(now);
      case JG_CMD_VOTE_REQUEST: return candidate_vote_request(c.from, c.term);
      case JG_CMD_VOTE_RESPONSE: return candidate_vote_response(c.flag != 0, c.from, now);
      case JG_CMD_APPEND_ENTRIES:  // :116-134
        if (c.term >= current_term) become_follower_from_candidate();
        return 0;
      case JG_CMD_HEARTBEAT: return candidate_heartbeat(c.term, c.from, c.id);
      case JG_CMD_CLIENT_REQUEST:  // :190-193
        enqueue(c.id);
        return 0;
      default: return 0;
    }
  }
  int candidate_tick(uint64_t now) {  // candidate.rs:48-68
    if (needs_election(now)) {
      switch (election.status()) {
        case ElectionStatus::Voting:
        case ElectionStatus::Defeated:
          has_voted = false;  // :53/:59
          voted_for = 0;
          become_follower_from_candidate();
          return follower_timeout(now);  // raft.apply(Command::Timeout)
        default: return JG_FAULT_CANDIDATE_TICK_ELECTED;  // :64
      }
    }
    return 0;
  }
  int candidate_vote_request(NodeId cand, Term term) {  // candidate.rs:71-88
    if (term > current_term) {
      set_term(term);
      become_follower_from_candidate();
      return 0;
    }
    Msg m{};
    m.kind = JG_CMD_VOTE_RESPONSE;
    m.from = id;
    m.term = current_term;
    m.flag = 0;
    send(JG_TO_PEER, cand, m);
    return 0;
  }
  int candidate_vote_response(bool granted, NodeId from, uint64_t now) {  // candidate.rs:91-98
    election.vote(from, granted);
    decisions++;
    switch (election.status()) {
      case ElectionStatus::Elected: {  // elect(): candidate.rs:108-113
        become_leader(now);
        Msg m{};  // heartbeat(): leader.rs:44-51
        m.kind = JG_CMD_HEARTBEAT;
        m.from = id;
        m.term = current_term;
        m.id = chain.commit;
        send(JG_TO_PEERS, 0, m);
        return 0;
      }
      case ElectionStatus::Voting: return 0;
      default:  // defeat(): candidate.rs:101-105
        has_voted = false;
        voted_for = 0;
        become_follower_from_candidate();
        return 0;
    }
  }
  int candidate_heartbeat(Term term, NodeId leader, BlockId commit) {  // candidate.rs:137-157
    bool has_committed = chain.has(commit);  // :144
    BlockId own_commit = chain.commit;       // :145
    set_term(term);                          // :146
    has_voted = true;                        // :147
    voted_for = leader;
    become_follower_from_candidate();        // :148 (leader_id stays None)
    Msg m{};
    m.kind = JG_CMD_HEARTBEAT_RESPONSE;
    m.from = id;
    m.id = own_commit;
    m.flag = has_committed ? 1 : 0;
    send(JG_TO_PEER, leader, m);
    return 0;
  }

  // ======================= Leader (leader.rs) =================================
  int leader_apply(const Cmd& c, uint64_t now) {  // leader.rs:248-266
    switch (c.kind) {
      case JG_CMD_TICK: return leader_tick(now);
      case JG_CMD_HEARTBEAT_RESPONSE:  // :222-231
        if (!c.flag && c.id > 0) return replicate();
        return 0;
      case JG_CMD_APPEND_RESPONSE: return leader_append_response(c.from, c.id);  // term, success ignored
      case JG_CMD_APPEND_ENTRIES:  // :200-208
        if (c.term > current_term) {
          int f = set_term(c.term);  // -> unimplemented!() (Q3)
          if (f) return f;
          become_follower_from_leader();
        }
        return 0;
      case JG_CMD_CLIENT_REQUEST: return leader_client_request(c.id);
      default: return 0;
    }
  }
  int leader_commit() {  // leader.rs:87-99
    decisions++;
    BlockId q = progress.committed_index();
    if (q > chain.commit) {
      BlockId prev = chain.commit;
      int f = chain.do_commit(q);
      if (f) return f;
      fsm.push_back(FsmRow{JG_FSM_APPLY_LEADER, prev, q});  // :93 range(prev..=new).skip(1)
    }
    return 0;
  }
  int leader_append_response(NodeId node, BlockId head) {  // leader.rs:211-219
    int f = progress.advance(node, head);
    if (f) return f;
    return leader_commit();
  }
  int leader_client_request(uint64_t token) {  // leader.rs:177-197
    BlockId bid = 0;
    int f = chain.append(&bid);
    if (f) return f;
    fsm.push_back(FsmRow{JG_FSM_NOTIFY, bid, token});  // :184-188
    return leader_append_response(id, chain.head);     // :190-196 self-ack
  }
  // leader.rs:124-174.  Item accounting for the unbounded ranges incl. Q9: see
  // DESIGN.md "replicate() and the commit key".
  int replicate() {
    for (NodeId n : nodes) {  // config order
      auto it = progress.progress.find(n);
      if (it == progress.progress.end()) continue;  // :126
      // is_active(): Probe -> !paused == true (progress.rs:163-165); Replicate ->
      // inflight.capacity() > len, always true since nothing is ever pushed (222-224)
      const NodeProgress& p = it->second;
      size_t want = p.replicate ? (size_t)JG_MAX_INFLIGHT + 1 : 2;  // items the iterator chain consumes
      size_t k = chain.blocks_from(p.head);
      bool key_in_range = chain.commit_key && !tm.separate_commit_key;
      if (k < want && key_in_range) return JG_FAULT_RANGE_HIT_COMMIT_KEY;  // chain.rs:219-226
      size_t n_blocks = std::min(k, want);
      n_blocks = n_blocks ? n_blocks - 1 : 0;  // nth(1) / skip(1)
      Msg m{};
      m.kind = JG_CMD_APPEND_ENTRIES;
      m.from = id;
      m.term = current_term;
      m.id = p.head;
      m.aux = n_blocks;
      send(JG_TO_PEER, n, m);
    }
    return 0;
  }
  int leader_tick(uint64_t now) {  // leader.rs:234-245 (write_state: debug dump, out of scope)
    if ((now - heartbeat_time) > tm.heartbeat_timeout_ms) {  // :78-80
      Msg m{};
      m.kind = JG_CMD_HEARTBEAT;
      m.from = id;
      m.term = current_term;
      m.id = chain.commit;
      send(JG_TO_PEERS, 0, m);
      heartbeat_time = now;  // :82-84
    }
    return replicate();
  }
};

}  // namespace jo
