// C++ host-adapter tests: the reference's own L1 tests, restated against
// josefine_amd/host/raft_handle.hpp (which drives the HIP engine through the C
// ABI), plus BASELINE.json config #1 — the examples/multi-node topology (ids 1,2,3)
// as three instances of one partition routed in-process.
// Built and run by tests/test_cpp_adapter.py (-m gpu).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <memory>

// CPU run of the host logic (tests/test_cpp_adapter.py, -m "not gpu"): the same C ABI as served by
// the oracle library (test infrastructure) stands in for the HIP engine; the dense mailbox test needs
// device memory and is left out.
#ifdef JG_TEST_AGAINST_ORACLE
#define jg_engine_create jo_engine_create
#define jg_engine_destroy jo_engine_destroy
#define jg_set_self_slots jo_set_self_slots
#define jg_submit jo_submit
#define jg_step jo_step
#define jg_drain_messages jo_drain_messages
#define jg_drain_applies jo_drain_applies
#define jg_drain_faults jo_drain_faults
#define jg_read_state jo_read_state
#define jg_last_error jo_last_error
#define jg_step_node jo_step_node
#define jg_node_outbox_view jo_node_outbox_view
#define jg_drain_messages_view jo_drain_messages_view
#define jg_drain_applies_view jo_drain_applies_view
#endif
#include "../../josefine_amd/host/formats.hpp"  // (includes raft_handle.hpp)

using namespace josefine;

static int g_failed = 0;
#define CHECK(cond)                                                     \
  do {                                                                  \
    if (!(cond)) {                                                      \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      g_failed++;                                                       \
    }                                                                   \
  } while (0)

// src/raft/leader.rs:299-327 apply_entry_single_node
static void apply_entry_single_node() {
  BatchedRaft raft(1, {1});
  std::vector<Instruction> fsm_rx;
  raft.fsm_tx = [&](const Instruction& i) { fsm_rx.push_back(i); };
  RaftHandle node = raft.handle(0).apply(Command::Timeout());
  CHECK(node.is_leader());
  const uint8_t magic_number = 123;
  node = node.apply(Command::ClientRequest(77, {magic_number}));
  node = node.apply(Command::Tick());
  CHECK(node.is_leader());
  // let block = leader.chain.range(..).take(2).last().unwrap();
  const std::vector<Block> two = raft.store(0).range(0, nullptr, false, 2);
  CHECK(two.size() == 2 && two[1].data == std::vector<uint8_t>{magic_number});
  CHECK(fsm_rx.size() == 2);
  CHECK(fsm_rx[0].kind == Instruction::Notify && fsm_rx[0].block_id == 1 && fsm_rx[0].request_id == 77);
  CHECK(fsm_rx[1].kind == Instruction::Apply && fsm_rx[1].block.data == std::vector<uint8_t>{magic_number});
}

// src/raft/follower.rs:338-358 apply_heartbeat
static void follower_apply_heartbeat() {
  BatchedRaft raft(1, {1});
  std::vector<Message> rpc_rx;
  raft.rpc_tx = [&](const Message& m) { rpc_rx.push_back(m); };
  RaftHandle follower = raft.handle(0).apply(Command::Heartbeat(12, 1, 11));
  CHECK(follower.is_follower());
  CHECK(follower.has_voted() && follower.voted_for() == 11);
  CHECK(follower.current_term() == 12);
  CHECK(rpc_rx.size() == 1);
  // but we don't have block 1 in our chain
  CHECK(rpc_rx[0].command.kind == JG_CMD_HEARTBEAT_RESPONSE && rpc_rx[0].command.id == 0 && !rpc_rx[0].command.flag);
  CHECK(rpc_rx[0].to.kind == JG_TO_PEER && rpc_rx[0].to.peer == 11);
}

// src/raft/candidate.rs:247-267 apply_heartbeat
static void candidate_apply_heartbeat() {
  BatchedRaft raft(1, {1, 2, 3});
  std::vector<Message> rpc_rx;
  raft.rpc_tx = [&](const Message& m) { rpc_rx.push_back(m); };
  RaftHandle candidate = raft.handle(0).apply(Command::Timeout());
  CHECK(candidate.is_candidate());
  rpc_rx.clear();
  RaftHandle follower = candidate.apply(Command::Heartbeat(11, 1, 6));
  CHECK(follower.is_follower() && follower.voted_for() == 6 && follower.current_term() == 11);
  CHECK(rpc_rx.size() == 1 && rpc_rx[0].command.kind == JG_CMD_HEARTBEAT_RESPONSE && rpc_rx[0].command.id == 0 &&
        !rpc_rx[0].command.flag);
}

// BASELINE.json config #1: 3-broker Chained-Raft group (examples/multi-node/node-{1,2,3}.toml),
// one partition; scripted: Timeout -> votes -> Elected -> 1 proposal -> acks -> commit 1.
static void multi_node_plumbing() {
  // three instances (= the three processes of the example) of one partition
  BatchedRaft raft(3, {1, 2, 3}, 0, 0, JG_CFG_SEPARATE_COMMIT_KEY);
  uint8_t slots[3] = {0, 1, 2};
  if (jg_set_self_slots(raft.raw(), slots) != JG_OK) {
    CHECK(!"jg_set_self_slots");
    return;
  }
  std::deque<Message> wire;
  std::vector<Instruction> fsm[3];
  raft.rpc_tx = [&](const Message& m) { wire.push_back(m); };
  raft.fsm_tx = [&](const Instruction& i) { fsm[i.group].push_back(i); };
  auto deliver_all = [&](uint64_t now) {
    int guard = 0;
    while (!wire.empty() && guard++ < 1000) {
      Message m = wire.front();
      wire.pop_front();
      for (uint32_t dst = 0; dst < 3; dst++) {
        NodeId dst_id = dst + 1;
        bool to_me = (m.to.kind == JG_TO_PEERS && dst_id != m.from.peer) || (m.to.kind == JG_TO_PEER && m.to.peer == dst_id);
        if (to_me) raft.apply(dst, m.command, now);
      }
    }
  };
  raft.apply(0, Command::Timeout());  // node 1 campaigns
  deliver_all(0);
  CHECK(raft.handle(0).is_leader());
  CHECK(raft.handle(1).is_follower() && raft.handle(2).is_follower());
  CHECK(raft.handle(1).voted_for() == 1 && raft.handle(1).current_term() == 1);  // via the leader's heartbeat
  raft.apply(0, Command::ClientRequest(1, {42}));  // propose
  raft.apply(0, Command::Tick(), 10);              // replicate(): Probe sends block 1
  deliver_all(10);
  CHECK(raft.handle(0).commit() == 1 && raft.handle(0).head() == 1);
  CHECK(raft.handle(1).head() == 1 && raft.handle(2).head() == 1);
  CHECK(raft.store(1).count(1) && raft.store(1).at(1).data == std::vector<uint8_t>{42});
  raft.apply(0, Command::Tick(), 150);  // heartbeat due: carries commit 1
  deliver_all(150);
  CHECK(raft.handle(1).commit() == 1 && raft.handle(2).commit() == 1);
  // leader applied (0,1] = block 1; followers applied range(0..1) = genesis only (Q6)
  CHECK(fsm[0].size() == 2 && fsm[0][1].kind == Instruction::Apply && fsm[0][1].block.id == 1);
  CHECK(fsm[1].size() == 1 && fsm[1][0].kind == Instruction::Apply && fsm[1][0].block.id == 0);
  for (uint32_t g = 0; g < 3; g++) CHECK(raft.handle(g).fault() == 0);
}

// The same 3-broker topology as three engines (one per broker process) hosting 64 partitions
// each, driven through the dense node tick: node 1 leads every partition, one ClientRequest per
// partition per round; the mailboxes never leave the device.
#ifndef JG_TEST_AGAINST_ORACLE
static void multi_node_dense_rounds() {
  const uint32_t G = 64;
  std::vector<std::unique_ptr<BatchedRaft>> nodes;
  std::vector<jg_engine*> raw;
  for (uint32_t r = 0; r < 3; r++) {
    nodes.emplace_back(new BatchedRaft(G, {1, 2, 3}, 0, 7 + r, JG_CFG_SEPARATE_COMMIT_KEY));
    std::vector<uint8_t> slots(G, (uint8_t)r);
    CHECK(jg_set_self_slots(nodes[r]->raw(), slots.data()) == JG_OK);
    raw.push_back(nodes[r]->raw());
  }
  // node 1 wins every election the reference's way: Timeout, then a granted vote from node 2
  for (uint32_t g = 0; g < G; g++) nodes[0]->submit(g, Command::Timeout());
  nodes[0]->step(0);
  for (uint32_t g = 0; g < G; g++) nodes[0]->submit(g, Command::VoteResponse(1, 2, true));
  nodes[0]->step(0);
  CHECK(nodes[0]->handle(5).is_leader());
  DenseCluster cl(raw, G, 0, 1);
  cl.set_appends(std::vector<uint64_t>(G, 1));
  const uint32_t T = 30;
  for (uint32_t t = 1; t <= T; t++) cl.round(100ull * t);
  cl.sync();
  for (uint32_t g : {0u, 17u, 63u}) {
    CHECK(nodes[0]->handle(g).head() == T && nodes[0]->handle(g).commit() + 3 >= T);
    for (uint32_t r = 1; r < 3; r++) {
      RaftHandle h = nodes[r]->handle(g);
      CHECK(h.is_follower() && h.fault() == 0 && h.voted_for() == 1 && h.current_term() == 1);
      CHECK(h.head() + 1 >= T && h.commit() + 5 >= T);
    }
  }
}
#endif


#ifndef JG_TEST_AGAINST_ORACLE
// Three nodes in one process, the library driving the rounds; then node 1's and node 3's replicas of every
// partition crash and restart, node 2 (restarted too: voted_for == None) times out
// and campaigns: its VoteRequests reach the others through the device-side transport, node 3 grants
// (can_vote, follower.rs:97-101), node 2 is elected (quorum 2 of 3 with its own vote) - no row of the
// election ever passes through the host.
static void library_cluster_routed_election() {
  const uint32_t G = 32;
  std::vector<std::unique_ptr<BatchedRaft>> nodes;
  std::vector<jg_engine*> raw;
  for (uint32_t r = 0; r < 3; r++) {
    nodes.emplace_back(new BatchedRaft(G, {1, 2, 3}, 0, 7 + r, JG_CFG_SEPARATE_COMMIT_KEY));
    std::vector<uint8_t> slots(G, (uint8_t)r);
    CHECK(jg_set_self_slots(nodes[r]->raw(), slots.data()) == JG_OK);
    raw.push_back(nodes[r]->raw());
  }
  for (uint32_t g = 0; g < G; g++) nodes[0]->submit(g, Command::Timeout());
  nodes[0]->step(0);
  for (uint32_t g = 0; g < G; g++) nodes[0]->submit(g, Command::VoteResponse(1, 2, true));
  nodes[0]->step(0);
  LibraryCluster cl(raw, 0);
  cl.set_appends(1);
  cl.rounds(100, 100, 10);  // steady state, replayed graph
  CHECK(nodes[0]->handle(3).head() == 10 && nodes[1]->handle(3).voted_for() == 1);
  // the crash: per node one device-resident, group-sorted batch
  auto upload = [&](jg_engine* e, const std::vector<uint8_t>& kinds) {  // the same rows for every group
    const size_t n = kinds.size() * G;
    std::vector<uint8_t> kind(n), flag(n, 0);
    std::vector<uint32_t> group(n), from(n, 0);
    std::vector<uint64_t> zero(n, 0);
    for (uint32_t g = 0; g < G; g++)
      for (size_t k = 0; k < kinds.size(); k++) kind[g * kinds.size() + k] = kinds[k], group[g * kinds.size() + k] = g;
    jg_cmd_batch b{};
    b.n = n;
    auto dev = [&](const void* src, size_t bytes) {
      void* p = nullptr;
      CHECK(jg_device_alloc(e, bytes, &p) == JG_OK && jg_device_upload(e, p, src, bytes) == JG_OK);
      return p;
    };
    b.kind = (const uint8_t*)dev(kind.data(), n), b.flag = (const uint8_t*)dev(flag.data(), n);
    b.group = (const uint32_t*)dev(group.data(), 4 * n), b.from = (const uint32_t*)dev(from.data(), 4 * n);
    b.term = (const uint64_t*)dev(zero.data(), 8 * n), b.id = (const uint64_t*)dev(zero.data(), 8 * n);
    b.aux = (const uint64_t*)dev(zero.data(), 8 * n);
    return b;
  };
  std::vector<jg_cmd_batch> inject(3);
  inject[0] = upload(raw[0], {JG_CMD_RESTART});
  inject[1] = upload(raw[1], {JG_CMD_RESTART, JG_CMD_TIMEOUT});
  inject[2] = upload(raw[2], {JG_CMD_RESTART});
  jg_route_stats st = cl.round_routed(1100, inject);
  CHECK(st.delivered[0] == 2ull * G && st.delivered[2] == 2ull * G && st.kept == 0);  // 2 copies of the VoteRequest each (candidate.rs:30-37)
  CHECK(nodes[1]->handle(3).is_candidate());
  st = cl.round_routed(1200);  // the others answer through can_vote
  CHECK(st.delivered[1] == 4ull * G);
  st = cl.round_routed(1300);  // the candidate counts: its own vote + the first grant = quorum
  for (uint32_t g : {0u, 3u, 31u}) {
    CHECK(nodes[1]->handle(g).is_leader() && nodes[1]->handle(g).fault() == 0);
    CHECK(nodes[0]->handle(g).is_follower() && nodes[2]->handle(g).is_follower());
  }
}
#endif

// src/raft/server.rs:185-216 event_loop: a single default-config node left alone for 2 s is leader
static void server_event_loop_single_node() {
  BatchedRaft raft(1, {1});
  BatchedEventLoop loop(raft, 1);
  loop.run_until(2000);
  CHECK(raft.handle(0).is_leader());
}

// Three processes (node-1..3 of examples/multi-node), each with its own engine hosting the same 8
// partitions, their event loops wired tcp_tx -> tcp_rx; timers are the only source of elections.
// A proposal through whoever leads a partition comes back to the client once it is committed and
// applied, and every node's state machine sees the same payload.
// `bus`: the node step's compact formats (JG_NODE_COMMON_AE | JG_NODE_FSM_FUSED) - BatchedRaft expands them again before
// rpc_tx / fsm_tx see anything: the same messages, the same transitions.
static void server_event_loops_three_nodes(uint32_t bus = 0) {
  const uint32_t G = 8, N = 3;
  std::vector<std::unique_ptr<BatchedRaft>> rafts;
  std::vector<std::unique_ptr<BatchedEventLoop>> loops;
  std::vector<std::vector<std::vector<uint8_t>>> applied(N, std::vector<std::vector<uint8_t>>(G));
  for (uint32_t n = 0; n < N; n++) {
    rafts.emplace_back(new BatchedRaft(G, {1, 2, 3}, 0, 100 + n, JG_CFG_SEPARATE_COMMIT_KEY));
    std::vector<uint8_t> slots(G, (uint8_t)n);
    CHECK(jg_set_self_slots(rafts[n]->raw(), slots.data()) == JG_OK);
    loops.emplace_back(new BatchedEventLoop(*rafts[n], G));
    loops[n]->bus = bus;
    loops[n]->fsm = [&applied, n](uint32_t g, const std::vector<uint8_t>& data) {
      applied[n][g].insert(applied[n][g].end(), data.begin(), data.end());
      return data;
    };
  }
  for (uint32_t n = 0; n < N; n++)
    loops[n]->tcp_tx = [&loops, n](const Message& m) {  // tcp.rs: Peer -> that peer, Peers -> everyone else
      for (uint32_t dst = 0; dst < 3; dst++) {
        if (dst == n) continue;
        if (m.to.kind == JG_TO_PEERS || m.to.peer == dst + 1) loops[dst]->tcp_rx(m);
      }
    };
  auto run_all = [&](uint64_t from, uint64_t to) {
    for (uint64_t t = from; t <= to; t += 10)
      for (uint32_t n = 0; n < N; n++) loops[n]->run_until(t);
  };
  run_all(0, 3000);
  uint32_t leaders = 0;
  for (uint32_t g = 0; g < G; g++) {
    int lead = -1;
    for (uint32_t n = 0; n < N; n++)
      if (rafts[n]->handle(g).is_leader()) lead = (int)n, leaders++;
    CHECK(lead >= 0);
    for (uint32_t n = 0; n < N; n++) CHECK(rafts[n]->handle(g).fault() == 0);
  }
  CHECK(leaders == G);  // exactly one leader per partition
  // one proposal per partition through its leader's event loop
  uint32_t answered = 0;
  for (uint32_t g = 0; g < G; g++)
    for (uint32_t n = 0; n < N; n++)
      if (rafts[n]->handle(g).is_leader())
        loops[n]->propose(g, {(uint8_t)(40 + g)}, [&answered, g](bool ok, const std::vector<uint8_t>& res) {
          answered += ok && res == std::vector<uint8_t>{(uint8_t)(40 + g)};
        });
  run_all(3010, 4000);
  CHECK(answered == G);
  for (uint32_t g = 0; g < G; g++)
    for (uint32_t n = 0; n < N; n++) {
      CHECK(rafts[n]->handle(g).head() == 1 && rafts[n]->handle(g).fault() == 0);
      if (rafts[n]->handle(g).is_leader()) {
        CHECK(rafts[n]->handle(g).commit() == 1);
        CHECK(applied[n][g] == std::vector<uint8_t>{(uint8_t)(40 + g)});
        CHECK(loops[n]->pending_requests() == 0);
      } else {
        // a follower commits 1 with the next heartbeat and applies range(0..1) = genesis only (Q6)
        CHECK(rafts[n]->handle(g).commit() == 1 && applied[n][g].empty());
        CHECK(rafts[n]->store(g).count(1) && rafts[n]->store(g).at(1).data == std::vector<uint8_t>{(uint8_t)(40 + g)});
      }
    }
}

// The same event loop over a MULTI-DEVICE engine (jg_config.n_devices: 3 shards, aliased onto device
// 0 here): one handle, one loop (server.rs:103-165), 10 single-node partitions; every partition elects
// itself from its own timer, a proposal per partition is committed, applied and answered, and the
// per-partition outcome is what the single-device engine gives.
static void server_event_loop_multi_device() {
  const uint32_t G = 10;
  std::vector<std::vector<uint8_t>> applied_one(G), applied_multi(G);
  for (int multi = 0; multi < 2; multi++) {
    BatchedRaft raft(G, {1}, 0, 7, 0, multi ? std::vector<int>{0, 0, 0} : std::vector<int>{});
    CHECK(jg_shard_count(raft.raw()) == (multi ? 3u : 1u));
    BatchedEventLoop loop(raft, G);
    auto& applied = multi ? applied_multi : applied_one;
    loop.fsm = [&applied](uint32_t g, const std::vector<uint8_t>& data) {
      applied[g].insert(applied[g].end(), data.begin(), data.end());
      return data;
    };
    loop.run_until(2000);
    uint32_t answered = 0;
    for (uint32_t g = 0; g < G; g++) {
      CHECK(raft.handle(g).is_leader());
      loop.propose(g, {(uint8_t)(g + 1), 9}, [&answered](bool ok, const std::vector<uint8_t>&) { answered += ok; });
    }
    loop.run_until(2300);
    CHECK(answered == G && loop.pending_requests() == 0);
    for (uint32_t g = 0; g < G; g++) CHECK(raft.handle(g).commit() == 1 && raft.handle(g).head() == 1);
  }
  CHECK(applied_one == applied_multi);
  for (uint32_t g = 0; g < G; g++) CHECK((applied_multi[g] == std::vector<uint8_t>{(uint8_t)(g + 1), 9}));
}

// SURVEY.md §8(f) rank 4 wired in: the host block store IS the reference's sled layout (ChainStore: 8-byte
// big-endian keys, bincode values, the "commit" key in the same tree).  A replica restarts: the tree is
// re-opened from its bytes and Chain::new (chain.rs:117-137) finds head = id_gen = commit - NOT the last stored
// block (Q8) - which is exactly what JG_CMD_RESTART makes of the engine's own image of the chain.
static void chain_store_restart() {
  auto read64 = [](BatchedRaft& r, int f, uint32_t g) {
    uint64_t v = 0;
    CHECK(jg_read_state(r.raw(), f, 0, &v, g, 1) == JG_OK);
    return v;
  };
  {  // a single-node leader with three committed blocks
    BatchedRaft raft(1, {1});
    raft.handle(0).apply(Command::Timeout());
    for (uint8_t k = 1; k <= 3; k++) raft.handle(0).apply(Command::ClientRequest(k, {k, k}));
    CHECK(raft.handle(0).head() == 3 && raft.handle(0).commit() == 3);
    CHECK(raft.store(0).entries() == 5 && raft.store(0).commit() == 3);  // blocks 0..3 + the "commit" key
    CHECK(raft.store(0).raw().rbegin()->first == formats::ChainStore::commit_key());  // ... which sorts after every block key
    CHECK(raft.store(0).at(2).data == (std::vector<uint8_t>{2, 2}) && raft.store(0).at(2).next == 1);
    const formats::ChainStore::Reopened r = raft.restart(0, 1000);
    CHECK(r.commit == 3 && r.head == 3 && r.id_gen == 3);
    RaftHandle h = raft.handle(0);
    CHECK(h.is_follower() && h.current_term() == 0 && !h.has_voted() && h.fault() == 0);
    CHECK(h.head() == r.head && h.commit() == r.commit && read64(raft, JG_FIELD_ID_GEN, 0) == r.id_gen);
    CHECK(raft.store(0).entries() == 5 && raft.store(0).at(3).data == (std::vector<uint8_t>{3, 3}));  // nothing was lost on disk
    // Q8: the restarted node wins its election again and dies on its first append (id_gen.fetch_add -> 3,
    // assert!(id > head) with head == 3: chain.rs:161-163)
    h = h.apply(Command::Timeout(), 2000);
    CHECK(h.is_leader());
    h = h.apply(Command::ClientRequest(9, {9}), 2100);
    CHECK(h.fault() == JG_FAULT_APPEND_ID_NOT_ABOVE_HEAD);
  }
  {  // a follower that holds blocks beyond its commit index: the restart puts its head BACK to the commit
    BatchedRaft raft(1, {1, 2, 3});
    auto blk = [](BlockId id) {
      Block b;
      b.id = id, b.next = id - 1, b.data = {(uint8_t)id};
      return b;
    };
    raft.handle(0).apply(Command::AppendEntries(1, 2, {blk(1), blk(2), blk(3), blk(4)}));
    raft.handle(0).apply(Command::Heartbeat(1, 2, 2));
    CHECK(raft.handle(0).head() == 4 && raft.handle(0).commit() == 2 && raft.store(0).commit() == 2);
    const formats::ChainStore::Reopened r = raft.restart(0, 500);
    CHECK(r.commit == 2 && r.head == 2 && r.id_gen == 2);
    CHECK(raft.handle(0).head() == 2 && raft.handle(0).commit() == 2 && read64(raft, JG_FIELD_ID_GEN, 0) == 2);
    CHECK(raft.store(0).has(3) && raft.store(0).has(4));  // still in the tree: extend() will meet them again
    raft.handle(0).apply(Command::AppendEntries(1, 2, {blk(3), blk(4), blk(5)}), 600);
    CHECK(raft.handle(0).head() == 5 && raft.handle(0).fault() == 0 && raft.store(0).at(5).data == std::vector<uint8_t>{5});
  }
  {  // a tree without a commit key: Chain::new initialises it (genesis, id_gen 1)
    BatchedRaft raft(1, {1, 2, 3});
    const formats::ChainStore::Reopened r = raft.restart(0);
    CHECK(r.commit == 0 && r.head == 0 && r.id_gen == 1 && raft.handle(0).head() == 0 && read64(raft, JG_FIELD_ID_GEN, 0) == 1);
  }
  {  // a partition that was RE-CREATED (JG_CMD_RECREATE): an empty data directory - the store starts over with the engine's
    // image of it, and unlike the restarted leader above this one can append again
    BatchedRaft raft(1, {1});
    std::vector<Instruction> fsm_rx;
    raft.fsm_tx = [&](const Instruction& i) { fsm_rx.push_back(i); };
    raft.handle(0).apply(Command::Timeout());
    for (uint8_t k = 1; k <= 3; k++) raft.handle(0).apply(Command::ClientRequest(k, {k, k}));
    CHECK(raft.store(0).entries() == 5 && raft.handle(0).commit() == 3);
    raft.recreate(0, 1000);
    RaftHandle h = raft.handle(0);
    CHECK(h.is_follower() && h.current_term() == 0 && !h.has_voted() && h.fault() == 0 && h.head() == 0 && h.commit() == 0);
    CHECK(read64(raft, JG_FIELD_ID_GEN, 0) == 1 && raft.store(0).entries() == 1 && raft.store(0).has(0) && !raft.store(0).has(1));
    fsm_rx.clear();
    h = h.apply(Command::Timeout(), 2000);
    h = h.apply(Command::ClientRequest(9, {9, 9, 9}), 2100);
    CHECK(h.is_leader() && h.fault() == 0 && h.head() == 1 && h.commit() == 1);
    CHECK(raft.store(0).at(1).data == (std::vector<uint8_t>{9, 9, 9}) && raft.store(0).at(1).next == 0 && raft.store(0).commit() == 1);
    CHECK(fsm_rx.size() == 2 && fsm_rx[0].kind == Instruction::Notify && fsm_rx[0].request_id == 9 && fsm_rx[1].kind == Instruction::Apply &&
          fsm_rx[1].block.data == (std::vector<uint8_t>{9, 9, 9}));
  }
}

// Three event loops that talk through BYTES: every message a loop emits is serde_json-encoded and framed the way
// tcp.rs frames it (4-byte big-endian length, tcp.rs:40-51,143-156), travels through a byte pipe per (peer,
// partition) connection, and is unframed and decoded on the other side - the reference's wire, end to end.
// Decoded leniently the cluster elects its leaders by the timers and commits a proposal, as it does in memory;
// decoded as the reference itself reads (`as_the_reference_reads`) every BlockId-bearing frame is refused, so
// no VoteRequest ever arrives and nobody is ever elected - which is what would happen between stock josefine peers.
static void event_loops_over_the_wire() {
  for (int strict = 0; strict < 2; strict++) {
    const uint32_t G = 4, N = 3;
    std::vector<std::unique_ptr<BatchedRaft>> rafts;
    std::vector<std::unique_ptr<BatchedEventLoop>> loops;
    std::map<std::pair<uint32_t, uint32_t>, formats::Bytes> pipe;  // (destination node, partition) -> bytes in flight
    uint64_t frames = 0, refused = 0, bytes = 0;
    for (uint32_t n = 0; n < N; n++) {
      rafts.emplace_back(new BatchedRaft(G, {1, 2, 3}, 0, 300 + n, JG_CFG_SEPARATE_COMMIT_KEY));
      std::vector<uint8_t> slots(G, (uint8_t)n);
      CHECK(jg_set_self_slots(rafts[n]->raw(), slots.data()) == JG_OK);
      loops.emplace_back(new BatchedEventLoop(*rafts[n], G));
    }
    for (uint32_t n = 0; n < N; n++)
      loops[n]->tcp_tx = [&, n](const Message& m) {
        const formats::Bytes f = formats::frame(formats::encode_message(m));
        for (uint32_t dst = 0; dst < N; dst++)
          if (dst != n && (m.to.kind == JG_TO_PEERS || m.to.peer == dst + 1)) pipe[{dst, m.group}] += f, frames++, bytes += f.size();
      };
    auto deliver = [&] {
      for (auto& kv : pipe) {
        std::string payload;
        while (formats::unframe(kv.second, &payload)) {
          try {
            Message m = formats::decode_message(payload, strict != 0);
            m.group = kv.first.second;
            loops[kv.first.first]->tcp_rx(m);
          } catch (const formats::FormatError&) {
            refused++;  // tcp.rs logs the error and drops the frame
          }
        }
      }
    };
    auto run_all = [&](uint64_t from, uint64_t to) {
      for (uint64_t t = from; t <= to; t += 10) {
        deliver();
        for (uint32_t n = 0; n < N; n++) loops[n]->run_until(t);
      }
    };
    run_all(0, 3000);
    uint32_t leaders = 0;
    for (uint32_t g = 0; g < G; g++)
      for (uint32_t n = 0; n < N; n++) leaders += rafts[n]->handle(g).is_leader();
    if (strict) {
      CHECK(leaders == 0 && refused > 0 && frames > 0);  // VoteRequests carry a BlockId (`head`): refused, every one
      continue;
    }
    CHECK(leaders == G && refused == 0 && bytes > 0);
    uint32_t answered = 0;
    for (uint32_t g = 0; g < G; g++)
      for (uint32_t n = 0; n < N; n++)
        if (rafts[n]->handle(g).is_leader())
          loops[n]->propose(g, {(uint8_t)(70 + g)}, [&answered](bool ok, const std::vector<uint8_t>&) { answered += ok; });
    run_all(3010, 4000);
    CHECK(answered == G);
    for (uint32_t g = 0; g < G; g++)
      for (uint32_t n = 0; n < N; n++) {
        CHECK(rafts[n]->handle(g).head() == 1 && rafts[n]->handle(g).commit() == 1 && rafts[n]->handle(g).fault() == 0);
        CHECK(rafts[n]->store(g).at(1).data == std::vector<uint8_t>{(uint8_t)(70 + g)});  // the payload crossed the wire as JSON
        CHECK(rafts[n]->store(g).commit() == 1);
      }
  }
}

// A PIPELINED loop (BatchedEventLoop::pipelined: step t's outputs are delivered at the start of step t + 1) must be the
// plain loop, only later: (1) a block whose Chain::extend fails in step t + 1 (missing parent, chain.rs:180-185) is not
// persisted - the store stays byte for byte what the plain loop's holds; (2) step t's HeartbeatResponse / AppendResponse go
// to the sender of the message they answer (follower.rs:163-172,209-215), not to whoever sent the partition's traffic of
// step t + 1.  Both loops are fed the same frames; every message on tcp_tx and the sled images are compared.
static void pipelined_loop_is_the_plain_loop_later() {
  const uint32_t G = 4;
  struct Out {
    uint32_t group, to, kind;
    uint64_t id;
    bool operator==(const Out& o) const { return group == o.group && to == o.to && kind == o.kind && id == o.id; }
  };
  std::vector<std::vector<Out>> sent(2);
  std::vector<std::unique_ptr<BatchedRaft>> rafts;
  std::vector<std::unique_ptr<BatchedEventLoop>> loops;
  for (int k = 0; k < 2; k++) {
    rafts.emplace_back(new BatchedRaft(G, {1, 2, 3}, 0, 7, JG_CFG_SEPARATE_COMMIT_KEY));
    std::vector<uint8_t> slots(G, 1);  // this process is node 2 of every partition
    CHECK(jg_set_self_slots(rafts[k]->raw(), slots.data()) == JG_OK);
    loops.emplace_back(new BatchedEventLoop(*rafts[k], G));
    loops[k]->pipelined = k == 1;
    loops[k]->tcp_tx = [&sent, k](const Message& m) { sent[k].push_back({m.group, m.to.peer, m.command.kind, m.command.id}); };
  }
  auto msg = [](uint32_t g, NodeId from, const Command& c) {
    Message m;
    m.group = g, m.from = Address{JG_TO_PEER, from}, m.to = Address{JG_TO_PEER, 2}, m.command = c;
    return m;
  };
  auto ae = [](Term term, NodeId leader, std::vector<Block> blocks) {
    Command c;
    c.kind = JG_CMD_APPEND_ENTRIES, c.term = term, c.from = leader, c.blocks = std::move(blocks);
    return c;
  };
  for (int k = 0; k < 2; k++) {
    BatchedEventLoop& L = *loops[k];
    // tick 0: node 1 leads everything at term 1; partitions 0-2 get block 1
    for (uint32_t g = 0; g < G; g++) L.tcp_rx(msg(g, 1, Command::Heartbeat(1, 0, 1)));
    for (uint32_t g = 0; g < 3; g++) L.tcp_rx(msg(g, 1, ae(1, 1, {Block{1, 0, {(uint8_t)(10 + g)}}})));
    L.run_until(0);
    // tick 1: partition 0: a block whose parent is missing (the process dies in extend: nothing stored);
    //         partition 1: the next block; partition 2: a NEW leader (node 3, term 2) sends the partition's traffic;
    //         partition 3: node 3's heartbeat at a higher term
    L.tcp_rx(msg(0, 1, ae(1, 1, {Block{5, 4, {55}}, Block{6, 5, {66}}})));
    L.tcp_rx(msg(1, 1, ae(1, 1, {Block{2, 1, {22}}})));
    L.tcp_rx(msg(2, 3, Command::Heartbeat(2, 0, 3)));
    L.tcp_rx(msg(2, 3, ae(2, 3, {Block{2, 1, {33}}})));
    L.tcp_rx(msg(3, 3, Command::Heartbeat(2, 0, 3)));
    L.run_until(100);
    // tick 2: quiet, except partition 1
    L.tcp_rx(msg(1, 1, ae(1, 1, {Block{3, 2, {44}}})));
    L.run_until(200);
    L.flush();
  }
  CHECK(!sent[0].empty() && sent[0].size() == sent[1].size());
  // per partition the same messages in the same order (the pipelined loop hands a step's messages over one step later)
  for (uint32_t g = 0; g < G; g++) {
    std::vector<Out> a, b;
    for (const Out& o : sent[0]) if (o.group == g) a.push_back(o);
    for (const Out& o : sent[1]) if (o.group == g) b.push_back(o);
    CHECK(a == b);
  }
  // step 0's answers went to node 1 although partition 2 / 3's next traffic came from node 3
  for (int k = 0; k < 2; k++) {
    uint32_t first_to[4] = {0, 0, 0, 0}, last_to[4] = {0, 0, 0, 0};
    for (const Out& o : sent[k]) {
      if (!first_to[o.group]) first_to[o.group] = o.to;
      last_to[o.group] = o.to;
    }
    CHECK(first_to[2] == 1 && first_to[3] == 1 && last_to[2] == 3 && last_to[3] == 3 && last_to[1] == 1);
  }
  for (uint32_t g = 0; g < G; g++) {
    CHECK(rafts[0]->store(g).raw() == rafts[1]->store(g).raw());
    CHECK(rafts[0]->handle(g).fault() == rafts[1]->handle(g).fault() && rafts[0]->handle(g).head() == rafts[1]->handle(g).head());
  }
  CHECK(rafts[1]->handle(0).fault() == JG_FAULT_EXTEND_MISSING_PARENT);
  CHECK(!rafts[1]->store(0).count(5) && !rafts[1]->store(0).count(6) && rafts[1]->store(0).count(1));
  CHECK(rafts[1]->store(1).count(3) && rafts[1]->store(2).count(2) && rafts[1]->store(2).at(2).data == std::vector<uint8_t>{33});
}

// fsm fan-out where KEY order is not PARENT order: a follower whose chain
// holds a dead branch.  range(prev..commit) (follower.rs:204) walks the keys, so the dead block is
// applied too, exactly as the reference's sled iterator would deliver it: ids 1 <- 2 <- 3 (dead) and
// 2 <- 4 <- 5; commit 5 applies keys [0, 5) = 0, 1, 2, 3, 4 (genesis is skipped by the driver, fsm.rs:59-61).
static void fsm_apply_walks_keys_not_parents() {
  BatchedRaft raft(1, {1, 2, 3});
  std::vector<uint8_t> seen;
  BatchedEventLoop loop(raft, 1);
  loop.fsm = [&seen](uint32_t, const std::vector<uint8_t>& data) {
    seen.insert(seen.end(), data.begin(), data.end());
    return data;
  };
  auto blk = [](BlockId id, BlockId next) {
    Block b;
    b.id = id, b.next = next, b.data = {(uint8_t)(10 * id)};
    return b;
  };
  Message m;
  m.group = 0;
  m.command = Command::AppendEntries(1, 2, {blk(1, 0), blk(2, 1), blk(3, 2), blk(4, 2), blk(5, 4)});
  loop.tcp_rx(m);
  m.command = Command::Heartbeat(1, 5, 2);
  loop.tcp_rx(m);
  loop.run_until(0);
  CHECK(raft.handle(0).head() == 5 && raft.handle(0).commit() == 5 && raft.handle(0).fault() == 0);
  CHECK((seen == std::vector<uint8_t>{10, 20, 30, 40}));  // keys 1..4 in key order, the dead block 3 included
}

// SURVEY.md §8(f) rank 3, FSM apply fan-out: what the host side of one step costs when every one of
// many partitions commits something - Notify + Apply rows drained, expanded against the block stores by
// key order and handed to fsm_tx.  Prints the rate; the check is only that everything arrives.
static void fsm_fanout_throughput() {
  const uint32_t G = 20000, rounds = 8;
  BatchedRaft raft(G, {1});  // single-node partitions: every ClientRequest commits at once
  uint64_t notifies = 0, applies = 0, bytes = 0;
  raft.fsm_tx = [&](const Instruction& i) {
    if (i.kind == Instruction::Notify) notifies++;
    else applies++, bytes += i.block.data.size();
  };
  for (uint32_t g = 0; g < G; g++) raft.submit(g, Command::Timeout());
  raft.step(0);
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t r = 0; r < rounds; r++) {
    for (uint32_t g = 0; g < G; g++) raft.submit(g, Command::ClientRequest(r * G + g, {1, 2, 3, 4, 5, 6, 7, 8}));
    raft.step(100 * (r + 1));
  }
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  CHECK(notifies == (uint64_t)G * rounds && applies == (uint64_t)G * rounds && bytes == 8ull * G * rounds);
  std::printf("fsm fan-out: %u partitions x %u rounds, %.2f M instructions/s through fsm_tx (submit + step + drain + expansion, one host thread)\n",
              G, rounds, (double)(notifies + applies) / s / 1e6);
}

int main() {
  try {
    apply_entry_single_node();
    follower_apply_heartbeat();
    candidate_apply_heartbeat();
    multi_node_plumbing();
#ifndef JG_TEST_AGAINST_ORACLE
    multi_node_dense_rounds();
    library_cluster_routed_election();
#endif
    server_event_loop_single_node();
    server_event_loops_three_nodes();
    server_event_loops_three_nodes(JG_NODE_COMMON_AE | JG_NODE_FSM_FUSED);
    chain_store_restart();
    event_loops_over_the_wire();
    pipelined_loop_is_the_plain_loop_later();
    fsm_apply_walks_keys_not_parents();
    fsm_fanout_throughput();
#ifndef JG_TEST_AGAINST_ORACLE
    server_event_loop_multi_device();
#endif
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 2;
  }
  if (g_failed) {
    std::fprintf(stderr, "%d check(s) failed\n", g_failed);
    return 1;
  }
  std::puts("cpp adapter ok");
  return 0;
}
