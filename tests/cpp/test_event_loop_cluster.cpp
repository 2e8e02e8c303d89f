// A whole cluster THROUGH THE APPLY SURFACE: R nodes x G partitions, one josefine::BatchedEventLoop per
// node (server::event_loop for many partitions, src/raft/server.rs:103-165), each over its own engine,
// wired tcp_tx -> tcp_rx through a host "transport" that carries what the loops emit - the mailbox
// columns of jg_step_node as the rows they stand for, and every other message row as it is.  Nothing is
// synthetic: every AppendResponse is a real follower's answer to a real AppendEntries.
//
// Built twice by tests/test_cpp_adapter.py: against libjosefine_gpu.so (the HIP engine: the dense
// kernels behind the loop) and, with -DJG_TEST_AGAINST_ORACLE, against the oracle library (the same host
// code, the CPU restatement underneath).  Both print one line with an FNV-1a hash over EVERY row both
// channels of every node emitted, tick by tick (fsm_tx rows, rpc_tx rows, outbox columns) and the final
// state; the test compares the two lines.
//
//   usage: test_event_loop_cluster G R T mode      mode: scripted | elect | failover
//     scripted  node 1 wins every election up front (Timeout + granted votes), then T ticks with one
//               ClientRequest per partition per tick at the leader: the steady state
//     elect     nobody is told anything: the timers elect (votes travel as rows through the transport),
//               proposals go to whoever leads, for T ticks
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

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
#define jg_get_counters jo_get_counters
#endif
#include "../../josefine_amd/host/raft_handle.hpp"

using namespace josefine;

struct Fnv {
  uint64_t h = 0xcbf29ce484222325ull;
  void bytes(const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001b3ull;
  }
  void u64(uint64_t v) { bytes(&v, 8); }
};

int main(int argc, char** argv) {
  const uint32_t G = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 2000;
  const uint32_t R = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 5;
  const uint32_t T = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 50;
  const bool scripted = argc > 4 ? std::string(argv[4]) == "scripted" : true;
  // failover: like elect, and at tick T / 3 every node's process "crashes and restarts" for the partitions it leads
  // (JG_CMD_RESTART = Raft::new on the persisted chain, follower.rs:68-95: term 0, no vote, a follower): those partitions
  // are leaderless until a follower's timer fires and a NEW leader - necessarily on another node - is elected
  const bool failover = argc > 4 && std::string(argv[4]) == "failover";
  try {
    std::vector<NodeId> ids;
    for (uint32_t r = 0; r < R; r++) ids.push_back(r + 1);
    std::vector<std::unique_ptr<BatchedRaft>> rafts;
    std::vector<std::unique_ptr<BatchedEventLoop>> loops;
    std::vector<Fnv> hash(R);
    std::vector<uint64_t> n_fsm(R, 0), n_msg(R, 0), n_cols(R, 0), n_general(R, 0), n_rows(R, 0);
    // What the transport delivers to node n before its next step: every sender's stream in its own order - all a network promises
    // (tcp.rs: one connection per peer pair) - the senders INTERLEAVED per partition: what every sender said first about a
    // partition arrives before anybody's second row about it, the device transport's order (jg_route.h).  (Sender after sender,
    // as until round 6, is a legal schedule too - under which an election of more than three nodes is never won: a candidate
    // broadcasts its VoteRequest once per peer, candidate.rs:30-37, a voter grants the first copy and refuses the rest, and the
    // later answer of a voter overwrites the earlier, election.rs:33-35.)
    struct Wire {
      std::vector<std::vector<jg_msg_row>> from;  // [sender]: its rows for this node, in the order it emitted them (AppendEntries: id = the run's start, aux = its length)
      std::vector<uint32_t> nth;                  // scratch: rows seen per partition
      void push(uint32_t src, uint32_t g, uint8_t kind, NodeId from_id, Term term, uint64_t id, uint64_t aux = 0, uint8_t flag = 0) {
        jg_msg_row r{};
        r.group = g, r.kind = kind, r.from = from_id, r.term = term, r.id = id, r.aux = aux, r.flag = flag;
        from[src].push_back(r);
      }
      bool empty() const {
        for (const auto& v : from)
          if (!v.empty()) return false;
        return true;
      }
      void clear() {
        for (auto& v : from) v.clear();
      }
      void deliver(RowQueue& q, uint32_t G) {  // (emission index within the partition, sender): a stable order
        struct Key {
          uint32_t k, src, at;
        };
        std::vector<Key> order;
        nth.assign(G, 0);
        for (uint32_t s = 0; s < from.size(); s++) {
          for (uint32_t i = 0; i < from[s].size(); i++) order.push_back({nth[from[s][i].group]++, s, i});
          for (const jg_msg_row& r : from[s]) nth[r.group] = 0;
        }
        std::stable_sort(order.begin(), order.end(), [](const Key& a, const Key& b) { return a.k != b.k ? a.k < b.k : a.src < b.src; });
        for (const Key& o : order) {
          const jg_msg_row& r = from[o.src][o.at];
          if (r.kind == JG_CMD_APPEND_ENTRIES) q.push_append_run(r.group, r.from, r.term, r.id, (uint32_t)r.aux);
          else q.push(r.group, r.kind, r.from, r.term, r.id, r.aux, r.flag);
        }
      }
    };
    std::vector<Wire> wire(R);
    for (Wire& w : wire) w.from.resize(R);
    RowQueue wire_rows;
    std::vector<std::vector<NodeId>> answers_to(R, std::vector<NodeId>(G, 0));
    for (uint32_t n = 0; n < R; n++) {
      rafts.emplace_back(new BatchedRaft(G, ids, 0, 1000 + n, JG_CFG_SEPARATE_COMMIT_KEY));
      std::vector<uint8_t> slots(G, (uint8_t)n);
      if (jg_set_self_slots(rafts[n]->raw(), slots.data()) != JG_OK) throw std::runtime_error("jg_set_self_slots");
      loops.emplace_back(new BatchedEventLoop(*rafts[n], G));
      // JG_CLUSTER_PIPELINED=1: every loop overlaps with itself (a step's outputs are delivered at the start of the next)
      loops[n]->pipelined = std::getenv("JG_CLUSTER_PIPELINED") != nullptr;
      // JG_CLUSTER_IN_FLIGHT=2 (with JG_CLUSTER_PIPELINED): two ticks in flight per loop (JG_NODE_KEEP) - a step's outputs
      // are delivered at the start of the step after the next
      if (const char* f = std::getenv("JG_CLUSTER_IN_FLIGHT")) loops[n]->in_flight = (uint32_t)std::atoi(f);
      // JG_CLUSTER_COMPACT=1: ABI v7's bus formats - the Tick's AppendEntries words as one word per partition where the
      // followers' agree, a leader's fsm_tx rows of a step as one row (the sinks below see the compact forms)
      if (std::getenv("JG_CLUSTER_COMPACT")) loops[n]->bus = JG_NODE_COMMON_AE | JG_NODE_FSM_FUSED;
    }
    for (uint32_t n = 0; n < R; n++) {
      BatchedRaft& raft = *rafts[n];
      raft.fsm_rows_tx = [&, n](const jg_fsm_row* rows, size_t k) {
        hash[n].bytes(rows, k * sizeof(jg_fsm_row));
        n_fsm[n] += k;
      };
      raft.msg_rows_tx = [&, n](const jg_msg_row* rows, size_t k) {  // rpc_rx: Peer / Peers -> the transport
        hash[n].bytes(rows, k * sizeof(jg_msg_row));
        n_msg[n] += k;
        for (size_t i = 0; i < k; i++) {
          const jg_msg_row& r = rows[i];
          if (r.kind == JG_CMD_CLIENT_REQUEST) continue;  // request-mirror instructions: not on the wire in this test
          for (uint32_t dst = 0; dst < R; dst++) {
            if (dst == n) continue;
            if (!(r.to_kind == JG_TO_PEERS || (r.to_kind == JG_TO_PEER && r.to_id == ids[dst]))) continue;
            wire[dst].push(n, r.group, r.kind, r.from, r.term, r.id, r.aux, r.flag);
            if (r.kind == JG_CMD_HEARTBEAT || r.kind == JG_CMD_APPEND_ENTRIES) answers_to[dst][r.group] = r.from;
          }
        }
      };
      raft.columns_tx = [&, n](const jg_node_outbox& o) {  // the mailbox columns, as the rows they stand for
        n_general[n] += o.rows_general, n_rows[n] += o.rows;
        if (std::getenv("JG_DEBUG_CLUSTER") && o.rows_general) std::fprintf(stderr, "node %u: %llu of %llu rows general\n", n, (unsigned long long)o.rows_general, (unsigned long long)o.rows);
        if (o.beat) {
          hash[n].bytes(o.beat, (size_t)G * sizeof(jg_leader_beat));
          // (compact: what the words STAND FOR is hashed - an engine may say JG_AEC_INDIVIDUAL where the words happen to agree)
          auto ae_word = [&](uint32_t q, uint32_t g) {
            return (o.aec && o.aec[g] != JG_AEC_INDIVIDUAL) ? (q == n ? (uint64_t)JG_NO_ACK : o.aec[g]) : o.ae[(size_t)q * G + g];
          };
          if (o.aec) {
            for (uint32_t q = 0; q < R; q++)
              for (uint32_t g = 0; g < G; g++) hash[n].u64(ae_word(q, g));
          } else {
            hash[n].bytes(o.ae, (size_t)R * G * 8);
          }
          for (uint32_t g = 0; g < G; g++) {
            const bool hb = o.beat[g].hb_commit != JG_NO_ACK;
            for (uint32_t q = 0; q < R; q++) {
              if (q == n) continue;
              const uint64_t w = ae_word(q, g);
              if (hb) wire[q].push(n, g, JG_CMD_HEARTBEAT, ids[n], o.beat[g].term, o.beat[g].hb_commit), n_cols[n]++;
              if (w != JG_NO_ACK) wire[q].push(n, g, JG_CMD_APPEND_ENTRIES, ids[n], o.beat[g].term, w >> 8, w & 0xffu), n_cols[n]++;
              if (hb || w != JG_NO_ACK) answers_to[q][g] = ids[n];
            }
          }
        }
        if (o.answer) {
          hash[n].bytes(o.answer, (size_t)G * 8);
          for (uint32_t g = 0; g < G; g++) {
            const uint64_t w = o.answer[g];
            if (w == JG_NO_ACK) continue;
            const NodeId to = answers_to[n][g];
            if (to == 0 || to > R) continue;
            if ((w & 0xffu) != JG_HB_NONE) {
              hash[n].u64(o.hb_commit[g]);
              wire[to - 1].push(n, g, JG_CMD_HEARTBEAT_RESPONSE, ids[n], 0, o.hb_commit[g], 0, (uint8_t)(w & 0xffu)), n_cols[n]++;
            }
            if ((w >> 8) != JG_MAILBOX_NONE) wire[to - 1].push(n, g, JG_CMD_APPEND_RESPONSE, ids[n], 0, w >> 8, 0, 1), n_cols[n]++;
          }
        }
      };
    }
    uint64_t now = 0;
    if (scripted) {  // node 1 wins everywhere: Timeout, then granted votes from the next R/2 nodes (candidate.rs:91-113)
      RowQueue q;
      for (uint32_t g = 0; g < G; g++) q.push(g, JG_CMD_TIMEOUT);
      rafts[0]->submit_rows(q.view());
      rafts[0]->step(0);
      for (uint32_t k = 1; k <= R / 2; k++) {
        q.clear();
        for (uint32_t g = 0; g < G; g++) q.push(g, JG_CMD_VOTE_RESPONSE, ids[k], 1, 0, 0, 1);
        rafts[0]->submit_rows(q.view());
        rafts[0]->step(0);
      }
      for (uint32_t n = 0; n < R; n++) wire[n].clear();  // the campaign's own broadcasts are not part of the run
    }
    // the interval's first tick fires immediately (tokio::time::interval, server.rs:113): every loop takes it
    // at logical time 0, so that each run_until below covers exactly one firing
    for (uint32_t n = 0; n < R; n++) loops[n]->run_until(0);
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t proposals = 0;
    std::vector<uint8_t> role(G);
    // the last quarter of the run, when the elections have settled: leadership lies where the timers put it (on
    // every node), and the partitions replicate under THOSE leaders - in the dense kernels, nothing on the general path
    const uint32_t tail_from = T - T / 4;
    uint64_t general_before_tail = 0;
    std::vector<std::vector<uint64_t>> commit_at_tail(R, std::vector<uint64_t>(G, 0));
    std::vector<std::vector<uint8_t>> led_before(R, std::vector<uint8_t>(G, 0));
    uint64_t restarted = 0, moved = 0;
    for (uint32_t t = 0; t < T; t++) {
      if (failover && t == T / 3) {
        for (uint32_t n = 0; n < R; n++) loops[n]->flush();
        for (uint32_t n = 0; n < R; n++) {
          if (jg_read_state(rafts[n]->raw(), JG_FIELD_ROLE, 0, role.data(), 0, G) != JG_OK) throw std::runtime_error("read role");
          RowQueue q;
          for (uint32_t g = 0; g < G; g++)
            if (role[g] == JG_ROLE_LEADER) q.push(g, JG_CMD_RESTART), led_before[n][g] = 1, restarted++;
          if (!q.empty()) {
            rafts[n]->submit_rows(q.view());
            rafts[n]->step(now);
          }
        }
      }
      if (t == tail_from) {
        for (uint32_t n = 0; n < R; n++) {
          general_before_tail += n_general[n];
          if (jg_read_state(rafts[n]->raw(), JG_FIELD_COMMIT, 0, commit_at_tail[n].data(), 0, G) != JG_OK) throw std::runtime_error("read commit");
        }
      }
      now += BatchedEventLoop::TICK_MS;
      for (uint32_t n = 0; n < R; n++) {
        BatchedEventLoop& loop = *loops[n];
        if (!wire[n].empty()) {
          wire_rows.clear();
          wire[n].deliver(wire_rows, G);
          loop.tcp_rx_rows(wire_rows.view());
        }
        wire[n].clear();
        // client_rx: one request per partition this node leads
        if (scripted) {
          if (n == 0) {
            RowQueue cr;
            for (uint32_t g = 0; g < G; g++) cr.push(g, JG_CMD_CLIENT_REQUEST, 0, 0, (uint64_t)t * G + g);
            loop.tcp_rx_rows(cr.view());
            proposals += G;
          }
        } else {
          if (jg_read_state(rafts[n]->raw(), JG_FIELD_ROLE, 0, role.data(), 0, G) != JG_OK) throw std::runtime_error("read role");
          RowQueue cr;
          for (uint32_t g = 0; g < G; g++)
            if (role[g] == JG_ROLE_LEADER) cr.push(g, JG_CMD_CLIENT_REQUEST, 0, 0, ((uint64_t)t * R + n) * G + g), proposals++;
          if (!cr.empty()) loop.tcp_rx_rows(cr.view());
        }
      }
      // every node takes its step with what had been delivered before the tick (the outputs of this tick
      // travel while the interval runs: they are next tick's input)
      for (uint32_t n = 0; n < R; n++) loops[n]->run_until(now);
    }
    for (uint32_t n = 0; n < R; n++) loops[n]->flush();  // (pipelined: the last tick's outputs)
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // final state into the hash; invariants
    Fnv all;
    uint64_t leaders = 0, faults = 0, decisions = 0;
    std::vector<uint64_t> col(G);
    std::vector<uint8_t> b8(G);
    uint64_t min_commit = ~0ull, max_head = 0;
    std::vector<uint64_t> leads(R, 0);
    uint64_t tail_committing = 0;  // partitions whose LEADER's commit index advanced during the tail
    for (uint32_t n = 0; n < R; n++) {
      all.u64(hash[n].h);
      for (int f : {JG_FIELD_TERM, JG_FIELD_HEAD, JG_FIELD_COMMIT}) {
        if (jg_read_state(rafts[n]->raw(), f, 0, col.data(), 0, G) != JG_OK) throw std::runtime_error("read");
        all.bytes(col.data(), (size_t)G * 8);
        if (f == JG_FIELD_HEAD)
          for (uint32_t g = 0; g < G; g++) max_head = std::max(max_head, col[g]);
        if (f == JG_FIELD_COMMIT && n == 0)
          for (uint32_t g = 0; g < G; g++) min_commit = std::min(min_commit, col[g]);
        if (f == JG_FIELD_COMMIT) {
          if (jg_read_state(rafts[n]->raw(), JG_FIELD_ROLE, 0, b8.data(), 0, G) != JG_OK) throw std::runtime_error("read");
          for (uint32_t g = 0; g < G; g++)
            if (b8[g] == JG_ROLE_LEADER) leads[n]++, tail_committing += col[g] > commit_at_tail[n][g], moved += failover && !led_before[n][g];
        }
      }
      for (int f : {JG_FIELD_ROLE, JG_FIELD_FAULT}) {
        if (jg_read_state(rafts[n]->raw(), f, 0, b8.data(), 0, G) != JG_OK) throw std::runtime_error("read");
        all.bytes(b8.data(), G);
        for (uint32_t g = 0; g < G; g++) {
          if (f == JG_FIELD_ROLE) leaders += b8[g] == JG_ROLE_LEADER;
          else faults += b8[g] != 0;
        }
      }
      uint64_t c[4];
      if (jg_get_counters(rafts[n]->raw(), c) != JG_OK) throw std::runtime_error("counters");
      decisions += c[1];
    }
    uint64_t fsm = 0, msg = 0, cols = 0, general = 0, rows = 0;
    for (uint32_t n = 0; n < R; n++) fsm += n_fsm[n], msg += n_msg[n], cols += n_cols[n], general += n_general[n], rows += n_rows[n];
    bool ok = faults == 0;
    // (pipelined loops deliver a step's outputs at the start of the next: the round trip is two ticks longer)
    const bool two_in_flight = std::getenv("JG_CLUSTER_PIPELINED") && std::getenv("JG_CLUSTER_IN_FLIGHT") && std::atoi(std::getenv("JG_CLUSTER_IN_FLIGHT")) >= 2;
    // (two in flight: another tick each way - and a round trip of six ticks is longer than the leader's window of MAX_INFLIGHT = 5
    // blocks per follower (progress.rs:117): it replicates 5 blocks per 6 ticks while the clients propose one per tick, so the
    // commit index falls behind by a sixth of the run - the reference's flow control, identically on the oracle-backed loops)
    const uint64_t lag = two_in_flight ? 16 + T / 5 : std::getenv("JG_CLUSTER_PIPELINED") ? 8 : 4;
    if (scripted) ok = ok && leaders == G && max_head == T && min_commit + lag >= T && general == 0;
    else ok = ok && leaders <= G;
    std::string by_node;
    for (uint32_t n = 0; n < R; n++) by_node += (n ? "/" : "") + std::to_string(leads[n]);
    std::printf("cluster %s hash=%016llx G=%u R=%u T=%u mode=%s leaders=%llu faults=%llu proposals=%llu fsm_rows=%llu msg_rows=%llu "
                "column_messages=%llu rows_in=%llu rows_general=%llu decisions=%llu max_head=%llu min_commit_node1=%llu "
                "leaders_by_node=%s tail_ticks=%u tail_rows_general=%llu tail_partitions_committing=%llu restarted_leaders=%llu "
                "led_by_another_node_now=%llu\n",
                ok ? "ok" : "FAILED", (unsigned long long)all.h, G, R, T, scripted ? "scripted" : failover ? "failover" : "elect", (unsigned long long)leaders,
                (unsigned long long)faults, (unsigned long long)proposals, (unsigned long long)fsm, (unsigned long long)msg,
                (unsigned long long)cols, (unsigned long long)rows, (unsigned long long)general, (unsigned long long)decisions,
                (unsigned long long)max_head, (unsigned long long)min_commit, by_node.c_str(), T - tail_from,
                (unsigned long long)(general - general_before_tail), (unsigned long long)tail_committing, (unsigned long long)restarted,
                (unsigned long long)moved);
    std::fprintf(stderr, "[%.2f s for %u ticks of %u nodes x %u partitions: %.3g decisions/s through the loops incl. the host transport]\n", secs, T, R,
                 G, (double)decisions / secs);
    return ok ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 2;
  }
}
