// bench_event_loop — decisions/s THROUGH the reference's driver surface (bench.py --event-loop).
//
// One node that leads G partitions of R replicas, driven the way server::event_loop drives Raft<T>
// (src/raft/server.rs:103-165) - inbound messages as rows, one Tick per 100 ms of logical time, outputs
// on the two channels - but for every partition at once: josefine::BatchedRaft::step_node (jg_step_node)
// under josefine::BatchedEventLoop's row queues.  Per tick and partition the node receives what a leader
// in steady state receives: one ClientRequest, one AppendResponse from every follower (acknowledging the
// previous tick's block) and, every other tick (the heartbeat period), one HeartbeatResponse from every
// follower; rows arrive shuffled across partitions.  The followers are synthetic (their answers are
// computed on the host, like the device-resident ack stream of the headline bench); a full cluster of
// real loops is tests/cpp/test_event_loop_cluster.cpp.
//
// modes
//   inplace   the "transport" writes the rows straight into the engine's pinned columns
//             (jg_submit_reserve / jg_submit_commit): no host copy at all
//   columns   the followers are batched peers: each ships its answers as the column it produced (one JG_ANSWER word
//             per partition: jg_node_inbox_columns), written in place; only the ClientRequests are rows
//   copy      rows handed over as a jg_cmd_batch (BatchedEventLoop::tcp_rx_rows + run_until): two host copies
//   general   round 2's loop: every row and one Tick ROW per partition through jg_submit + jg_step (the
//             general state machine, host radix sort) - the A/B
// Both channels are consumed by batch sinks that read every byte they are handed (a 64-bit sum).
//   pipe / pipecolumns            inplace / columns with ONE loop that overlaps with itself (BatchedEventLoop::pipelined)
//   pipetasks / pipetaskscolumns  the same loop with the work the reference does NOT do on the event loop's task taken
//             off its thread: frames are decoded by the per-connection read tasks (src/raft/tcp.rs:139-170 - the loop
//             receives Commands from a channel, server.rs:120-137), fsm_tx is consumed by the driver task (src/raft/fsm.rs)
//             and rpc_tx by the per-peer senders (tcp.rs:87-137).  Here: `helpers` (argument 8, default R - 1) threads
//             beside the loop's own, fork-join - the per-connection decoders fill their slices of the pinned columns,
//             the consumers read their slices of the output batches; the loop thread works along and goes on when all
//             are done.  Everything that touches the engine stays on the loop thread.
//   pipetaskswire  pipetasks with the peers' traffic as what a stock josefine peer puts on the wire: every AppendResponse /
//             HeartbeatResponse is a LengthDelimitedCodec frame around serde_json(Message) (src/raft/tcp.rs:40-51,139-170;
//             host/formats.hpp), one byte stream per connection - encoded before the timed region (the SENDERS' work) - and the
//             connection tasks run formats::decode_message on every frame before they write the row: what the transport's
//             decoder costs when it decodes the reference's bytes (the other modes' "decoder" writes rows it computes).  The
//             ClientRequests stay rows: they reach the loop through client_rx, an in-process channel (server.rs:156-160).
//
// loops (argument 7, default 1): the process hosts the G partitions on that many event loops, one thread and
// one engine (its own HIP stream) each, G / loops partitions per loop - the reference runs one event_loop task
// per partition on tokio's worker threads (src/raft/server.rs:103, src/lib.rs); here a loop is a BATCH of
// partitions.  While one loop waits for its step, the others decode, commit and consume: host work, PCIe
// transfers (both directions) and kernels of different loops overlap.  The timed region starts when every loop
// has finished its warm-up ticks and ends when the last loop has finished its last tick.
// Prints one JSON object.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>

#include "formats.hpp"  // (includes raft_handle.hpp)

using namespace josefine;
using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

static uint64_t sum_words(const void* p, size_t bytes) {
  const uint64_t* w = (const uint64_t*)p;
  uint64_t s = 0;
  for (size_t i = 0; i < bytes / 8; i++) s += w[i];
  return s;
}

// The tasks beside the event loop (decoders, channel consumers) as a fork-join pool: run(n, f) executes f(0..n-1) on the
// helpers and the caller (jobs handed out by an atomic counter) and returns when all of them are done.
class Tasks {
 public:
  explicit Tasks(uint32_t helpers) {
    for (uint32_t i = 0; i < helpers; i++) th_.emplace_back([this] { work(); });
  }
  ~Tasks() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (std::thread& t : th_) t.join();
  }
  uint32_t threads() const { return (uint32_t)th_.size() + 1; }
  template <class F>
  void run(uint32_t n, F&& f) {
    if (th_.empty() || n <= 1) {
      for (uint32_t i = 0; i < n; i++) f(i);
      return;
    }
    Run r;  // (lives until every helper that picked it up has let go of it)
    r.job = f, r.n = n;
    {
      std::lock_guard<std::mutex> lk(m_);
      cur_ = &r, left_ = n, users_ = 0, epoch_++;
    }
    cv_.notify_all();
    drain(r);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return left_ == 0 && users_ == 0; });
    cur_ = nullptr;
  }

 private:
  struct Run {
    std::function<void(uint32_t)> job;
    uint32_t n = 0;
    std::atomic<uint32_t> next{0};
  };
  void drain(Run& r) {
    for (;;) {
      const uint32_t i = r.next.fetch_add(1);
      if (i >= r.n) return;
      r.job(i);
      std::lock_guard<std::mutex> lk(m_);
      if (--left_ == 0) done_.notify_all();
    }
  }
  void work() {
    uint64_t seen = 0;
    for (;;) {
      Run* r;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
        if (stop_) return;
        seen = epoch_;
        r = cur_;
        if (!r) continue;  // (woke up after the run was over)
        users_++;
      }
      drain(*r);
      std::lock_guard<std::mutex> lk(m_);
      if (--users_ == 0) done_.notify_all();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  Run* cur_ = nullptr;
  uint32_t left_ = 0, users_ = 0;
  uint64_t epoch_ = 0;
  bool stop_ = false;
};

// every loop arrives; the last one to arrive stamps the time all of them then share
struct Rendezvous {
  std::mutex m;
  std::condition_variable cv;
  uint32_t n, here = 0, round = 0;
  Clock::time_point stamp{};
  explicit Rendezvous(uint32_t n_) : n(n_) {}
  Clock::time_point arrive() {
    std::unique_lock<std::mutex> lk(m);
    const uint32_t r = round;
    if (++here == n) {
      here = 0, round++, stamp = Clock::now();
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return round != r; });
    }
    return stamp;
  }
};
struct LoopResult {
  bool ok = false;
  std::string error;
  uint64_t decisions = 0, rows_in = 0, general = 0, fsm_rows = 0, msg_rows = 0, up_bytes = 0, down_bytes = 0, sink = 0, wire_bytes = 0;
  double wall_ms = 0, t_fill = 0, t_submit = 0, t_step = 0, t_sinks = 0, t_wait = 0;
  float k_us = 0;
  uint32_t k_n = 0;
};

static void run_loop(uint32_t G, uint32_t R, uint32_t T, uint32_t W, const std::string& mode_arg, int device, uint64_t seed,
                     uint32_t helpers, bool compact, Rendezvous& rv, LoopResult& out) {
  // "pipe" / "pipecolumns": the loop overlaps with itself (BatchedEventLoop::pipelined) and its rows are validated on the
  // device (JG_COL_UNCHECKED) - otherwise exactly "inplace" / "columns"; "pipetasks…": with the decoder / consumer tasks
  const bool pipe = mode_arg.rfind("pipe", 0) == 0;
  const bool with_tasks = mode_arg.rfind("pipetasks", 0) == 0;
  const bool cols_in = mode_arg.size() >= 7 && mode_arg.compare(mode_arg.size() - 7, 7, "columns") == 0;
  const bool wire = mode_arg.size() >= 4 && mode_arg.compare(mode_arg.size() - 4, 4, "wire") == 0;
  const std::string mode = !pipe ? mode_arg : (cols_in ? "columns" : "inplace");  // ("pipetaskswire": inplace rows, decoded from frames)
  // (with the tasks: ... and the committed batch leaves for the device at once, JG_COL_UPLOAD_NOW, while the previous step's
  // outputs travel the other way: 8.0 -> 11.1 x 10^8 decisions/s with row inbound.  A loop that decodes and consumes on its
  // own thread is bound by that thread, and the copy engine reading the host's memory beside it costs it 10-20 %: it
  // uploads at the head of the step as before.  JG_BENCH_EARLY_UPLOAD=0 / 1 overrides: the A/B)
  const char* early_env = std::getenv("JG_BENCH_EARLY_UPLOAD");
  const bool early = early_env ? early_env[0] == '1' : with_tasks;
  const uint32_t unchecked = pipe ? (uint32_t)JG_COL_UNCHECKED | (early ? (uint32_t)JG_COL_UPLOAD_NOW : 0u) : 0u;
  // compact (ABI v7's bus formats, pipelined modes): the transport's decoder writes kind | sender slot << 4 | flag << 7 into
  // ONE byte (no from / flag columns: 13 bytes per row, not 18), the Tick's AppendEntries words come home as one word per
  // partition (8 bytes, not 8 (R - 1)) and a leader's Apply + Notify of a tick as one fsm row (24 bytes, not 48)
  const bool packed = compact && pipe && !cols_in;
  // ... and the ids (block ids, commit indices, request tokens: all below 2^32 here) as 32-bit values (JG_COL_ID32: one commit per tick)
  const bool id32 = compact && pipe;
  auto put_id = [id32](uint64_t* id, size_t i, uint64_t v) {
    if (id32) ((uint32_t*)id)[i] = (uint32_t)v;
    else id[i] = v;
  };
  Tasks tasks(with_tasks ? helpers : 0u);  // (no helpers: run() is a plain loop on the calling thread)
  constexpr uint32_t SPLIT = 8;            // jobs per batch and connection: the helpers draw them as they come free
  auto sum_split = [&](const void* p, size_t items, size_t item_bytes) {
    uint64_t part[SPLIT] = {};
    tasks.run(SPLIT, [&](uint32_t j) {
      const size_t a = items * j / SPLIT, b = items * (j + 1) / SPLIT;
      part[j] = sum_words((const uint8_t*)p + a * item_bytes, (b - a) * item_bytes);
    });
    uint64_t s = 0;
    for (uint64_t v : part) s += v;
    return s;
  };
  bool started = false, finished = false;  // (a loop that fails still arrives: the others must not wait for ever)
  try {
    std::vector<NodeId> ids;
    for (uint32_t r = 0; r < R; r++) ids.push_back(r + 1);
    BatchedRaft raft(G, ids, device, seed, JG_CFG_SEPARATE_COMMIT_KEY);
    BatchedEventLoop loop(raft, G);
    loop.halves = JG_NODE_LEADER_HALF;  // this node leads every partition
    loop.dense = mode != "general";
    loop.pipelined = pipe;
    // JG_BENCH_IN_FLIGHT=2: two ticks in flight (BatchedEventLoop::in_flight, JG_NODE_KEEP) - tick t + 1 is begun before tick
    // t's outputs are read; the same rows reach the same sinks in the same order, the device is never idle in between
    if (const char* f = std::getenv("JG_BENCH_IN_FLIGHT")) loop.in_flight = pipe ? (uint32_t)std::max(1, std::atoi(f)) : 1u;
    if (compact) loop.bus = JG_NODE_COMMON_AE | JG_NODE_FSM_FUSED;
    uint64_t sink = 0, fsm_rows = 0, msg_rows = 0, col_bytes = 0, up_bytes = 0, general = 0;
    double t_sinks = 0;  // (inside the sinks: the consumers' reading of what came home - the rest of a step's time is the engine's calls and their waits)
    raft.fsm_rows_tx = [&](const jg_fsm_row* r, size_t n) {
      const auto a = Clock::now();
      sink += sum_split(r, n, sizeof(jg_fsm_row)), fsm_rows += n;
      t_sinks += ms_since(a);
    };
    raft.msg_rows_tx = [&](const jg_msg_row* r, size_t n) {
      const auto a = Clock::now();
      sink += sum_split(r, n, sizeof(jg_msg_row)), msg_rows += n;
      t_sinks += ms_since(a);
    };
    raft.columns_tx = [&](const jg_node_outbox& o) {
      const auto a = Clock::now();
      if (o.beat) sink += sum_split(o.beat, G, 16) + (o.aec ? sum_split(o.aec, G, 8) : 0) + (o.ae ? sum_split(o.ae, (size_t)R * G, 8) : 0);
      col_bytes += o.bytes_d2h, up_bytes += o.bytes_h2d, general += o.rows_general;
      t_sinks += ms_since(a);
    };
    {  // this node wins every election the reference's way: Timeout, then granted votes until quorum
      RowQueue q;
      for (uint32_t g = 0; g < G; g++) q.push(g, JG_CMD_TIMEOUT);
      raft.submit_rows(q.view());
      raft.step(0);
      for (uint32_t k = 1; k <= R / 2; k++) {
        q.clear();
        for (uint32_t g = 0; g < G; g++) q.push(g, JG_CMD_VOTE_RESPONSE, ids[k], 1, 0, 0, 1);
        raft.submit_rows(q.view());
        raft.step(0);
      }
    }
    // a fixed shuffle of the partitions: rows arrive in no particular order
    std::vector<uint32_t> perm(G);
    std::iota(perm.begin(), perm.end(), 0u);
    uint64_t x = 88172645463325252ull + seed;
    // (JG_BENCH_SHUFFLE_WINDOW=w: the shuffle only moves a partition within its window of w - how much of the
    //  classification's cost is the disorder: 1 = rows in partition order)
    const char* win_env = std::getenv("JG_BENCH_SHUFFLE_WINDOW");
    const uint32_t win = win_env ? (uint32_t)std::max(1, std::atoi(win_env)) : 0u;
    if (win)
      for (uint32_t lo = 0; lo < G; lo += win)
        for (uint32_t i = std::min(G, lo + win) - 1; i > lo; i--) {
          x ^= x << 13, x ^= x >> 7, x ^= x << 17;
          std::swap(perm[i], perm[lo + (uint32_t)(x % (i - lo + 1))]);
        }
    for (uint32_t i = G - 1; i > 0 && !win; i--) {
      x ^= x << 13, x ^= x >> 7, x ^= x << 17;
      std::swap(perm[i], perm[(uint32_t)(x % (i + 1))]);
    }
    // tick t's inbound rows, written by `put(i, kind, group, from, id, flag)` - the transport's decoder
    auto rows_of_tick = [&](uint32_t t) { return (size_t)G * (1 + (R - 1) * ((t & 1) ? 2 : 1)); };
    // connection s (0: the clients, r >= 1: peer r) delivers its rows of tick t into its own stretch of the batch; job j of
    // SPLIT covers the j-th part of the connection's partitions
    auto fill_part = [&](uint32_t t, uint32_t s, uint32_t j, uint8_t* kind, uint32_t* group, uint32_t* from, uint64_t* id, uint8_t* flag) {
      const bool hb = t & 1;  // a follower answers the heartbeat it got with the previous tick's messages
      const uint32_t k0 = (uint32_t)((uint64_t)G * j / SPLIT), k1 = (uint32_t)((uint64_t)G * (j + 1) / SPLIT);
      if (s == 0) {
        for (uint32_t k = k0; k < k1; k++) {
          const uint32_t g = perm[k];
          kind[k] = JG_CMD_CLIENT_REQUEST, group[k] = g, put_id(id, k, (uint64_t)t * G + g);
          if (!packed) from[k] = 0, flag[k] = 0;
        }
        return;
      }
      const size_t per = hb ? 2 : 1;
      size_t i = (size_t)G + (size_t)(s - 1) * G * per + (size_t)k0 * per;
      uint32_t at = (uint32_t)((k0 + 7919ull * s) % G);  // (peer s's rows: the shuffle, rotated)
      for (uint32_t k = k0; k < k1; k++) {
        const uint32_t g = perm[at];
        at = at + 1 == G ? 0 : at + 1;
        if (packed) {  // (sender slot s and flag 1 in the kind byte)
          const uint8_t hi = (uint8_t)(s << 4 | 0x80u);
          if (hb) kind[i] = JG_CMD_HEARTBEAT_RESPONSE | hi, group[i] = g, put_id(id, i, t ? t - 1 : 0), i++;
          kind[i] = JG_CMD_APPEND_RESPONSE | hi, group[i] = g, put_id(id, i, t), i++;
          continue;
        }
        if (hb) kind[i] = JG_CMD_HEARTBEAT_RESPONSE, group[i] = g, from[i] = ids[s], id[i] = t ? t - 1 : 0, flag[i] = 1, i++;
        kind[i] = JG_CMD_APPEND_RESPONSE, group[i] = g, from[i] = ids[s], id[i] = t, flag[i] = 1, i++;
      }
    };
    // wire: connection s's byte stream of tick t - the frames of its rows in fill_part's order - and where job j's part begins
    std::vector<std::vector<formats::Bytes>> wire_buf;
    std::vector<std::vector<std::vector<size_t>>> wire_cut;
    uint64_t wire_bytes = 0;
    if (wire) {
      wire_buf.assign(W + T, std::vector<formats::Bytes>(R));
      wire_cut.assign(W + T, std::vector<std::vector<size_t>>(R, std::vector<size_t>(SPLIT + 1, 0)));
      tasks.run((W + T) * (R - 1), [&](uint32_t job) {  // (the senders' work: outside the timed region)
        const uint32_t t = job / (R - 1), sc = 1 + job % (R - 1);
        const bool hb = t & 1;
        formats::Bytes& b = wire_buf[t][sc];
        Message m;
        m.from = Address{JG_TO_PEER, ids[sc]}, m.to = Address{JG_TO_PEER, ids[0]};
        for (uint32_t j = 0; j < SPLIT; j++) {
          wire_cut[t][sc][j] = b.size();
          const uint32_t k0 = (uint32_t)((uint64_t)G * j / SPLIT), k1 = (uint32_t)((uint64_t)G * (j + 1) / SPLIT);
          for (uint32_t k = k0; k < k1; k++) {
            if (hb) m.command = Command::HeartbeatResponse(t ? t - 1 : 0, true), b += formats::frame(formats::encode_message(m));
            m.command = Command::AppendResponse(ids[sc], 0, t, true), b += formats::frame(formats::encode_message(m));
          }
        }
        wire_cut[t][sc][SPLIT] = b.size();
      });
    }
    // ... and its connection task: every frame through the reference's codec, the row written from what it says (the
    // connection says which partition: a stock josefine process hosts one, its peers' connections are that partition's)
    auto decode_part = [&](uint32_t t, uint32_t sc, uint32_t j, uint8_t* kind, uint32_t* group, uint32_t* from, uint64_t* id, uint8_t* flag) {
      const bool hb = t & 1;
      const uint32_t k0 = (uint32_t)((uint64_t)G * j / SPLIT);
      const size_t per = hb ? 2 : 1;
      size_t i = (size_t)G + (size_t)(sc - 1) * G * per + (size_t)k0 * per;
      uint32_t at = (uint32_t)((k0 + 7919ull * sc) % G), seen = 0;
      const formats::Bytes& b = wire_buf[t][sc];
      std::string payload;
      for (size_t p = wire_cut[t][sc][j], end = wire_cut[t][sc][j + 1]; p < end;) {
        uint32_t n = 0;
        for (int q = 0; q < 4; q++) n = (n << 8) | (uint8_t)b[p + q];
        payload.assign(b, p + 4, n);
        p += 4 + (size_t)n;
        const Message m = formats::decode_message(payload);
        const Command& c = m.command;
        const NodeId sender = c.kind == JG_CMD_HEARTBEAT_RESPONSE ? m.from.peer : c.from;  // (rpc.rs:17-27: the Message names the sender)
        group[i] = perm[at], put_id(id, i, c.id);
        if (packed) {
          uint32_t slot = 7;  // (nobody: reads NodeId 0)
          for (uint32_t q = 0; q < R; q++) slot = ids[q] == sender ? q : slot;
          kind[i] = (uint8_t)(c.kind | slot << 4 | (c.flag ? 0x80u : 0u));
        } else {
          kind[i] = c.kind, flag[i] = c.flag ? 1 : 0, from[i] = sender;
        }
        i++;
        if (++seen == per) seen = 0, at = at + 1 == G ? 0 : at + 1;
      }
    };
    auto fill = [&](uint32_t t, uint8_t* kind, uint32_t* group, uint32_t* from, uint64_t* id, uint8_t* flag) {
      tasks.run(R * SPLIT, [&](uint32_t job) {
        if (wire && job / SPLIT) decode_part(t, job / SPLIT, job % SPLIT, kind, group, from, id, flag);
        else fill_part(t, job / SPLIT, job % SPLIT, kind, group, from, id, flag);
      });
      if (wire)
        for (uint32_t sc = 1; sc < R; sc++) wire_bytes += wire_buf[t][sc].size();
      return rows_of_tick(t);
    };
    uint64_t c0[4], c1[4];
    double t_fill = 0, t_submit = 0, t_step = 0;
    uint64_t rows_in = 0;
    RowQueue staged;
    Clock::time_point t_begin{};
    if (jg_kernel_timing(raft.raw(), 1) != JG_OK) throw std::runtime_error("jg_kernel_timing");
    for (uint32_t t = 0; t < W + T; t++) {
      if (t == W) {
        loop.flush();
        if (jg_sync(raft.raw()) != JG_OK || jg_get_counters(raft.raw(), c0) != JG_OK) throw std::runtime_error("counters");
        sink = fsm_rows = msg_rows = col_bytes = up_bytes = general = rows_in = wire_bytes = 0;
        t_fill = t_submit = t_step = t_sinks = 0;
        raft.ms_waited_for_outputs = 0;
        t_begin = rv.arrive(), started = true;
      }
      const uint64_t now = 100ull * (t + 1);
      const size_t n = rows_of_tick(t);
      auto a = Clock::now();
      if (mode == "columns") {
        const bool hb = t & 1;
        uint64_t* ans[JG_MAX_REPLICAS] = {};
        for (uint32_t r = 1; r < R; r++) loop.tcp_rx_answer_column(r, &ans[r], nullptr);
        const jg_cmd_cols c = loop.tcp_rx_reserve(G);
        const uint64_t w = JG_ANSWER((uint64_t)t, hb ? 1u : JG_HB_NONE);
        tasks.run(R * SPLIT, [&](uint32_t job) {
          const uint32_t s = job / SPLIT, j = job % SPLIT;
          const uint32_t k0 = (uint32_t)((uint64_t)G * j / SPLIT), k1 = (uint32_t)((uint64_t)G * (j + 1) / SPLIT);
          if (s) {  // peer s's answers to last tick's Heartbeat / AppendEntries: one word per partition
            for (uint32_t g = k0; g < k1; g++) ans[s][g] = w;
          } else {
            for (uint32_t k = k0; k < k1; k++) c.kind[k] = JG_CMD_CLIENT_REQUEST, c.group[k] = perm[k], put_id(c.id, k, (uint64_t)t * G + perm[k]);
          }
        });
        t_fill += ms_since(a), a = Clock::now();
        loop.tcp_rx_commit(G, 0, unchecked | (id32 ? (uint32_t)JG_COL_ID32 : 0u));
        t_submit += ms_since(a), a = Clock::now();
        loop.run_until(now);
        t_step += ms_since(a);
        rows_in += G - n;  // (n is added below: count the rows actually submitted)
      } else if (mode == "inplace") {
        static const bool tick_trace = std::getenv("JG_BENCH_TICK_TRACE") != nullptr;  // (one line per tick on stderr: where the loop's thread was)
        const double w_before = raft.ms_waited_for_outputs, s_before = t_sinks;
        const double at0 = t >= W ? std::chrono::duration<double, std::milli>(Clock::now() - t_begin).count() : 0;
        const auto r0 = Clock::now();
        const jg_cmd_cols c = loop.tcp_rx_reserve(n);
        const double ms_reserve = ms_since(r0);
        const size_t k = fill(t, c.kind, c.group, c.from, c.id, c.flag);
        const double ms_f = ms_since(a);
        t_fill += ms_f, a = Clock::now();
        loop.tcp_rx_commit(k, 0, (packed ? (uint32_t)JG_COL_PACKED_KIND : (uint32_t)(JG_COL_FROM | JG_COL_FLAG)) | (id32 ? (uint32_t)JG_COL_ID32 : 0u) | unchecked);
        const double ms_c = ms_since(a);
        t_submit += ms_c, a = Clock::now();
        loop.run_until(now);
        const double ms_s = ms_since(a);
        t_step += ms_s;
        if (tick_trace && t >= W)
          std::fprintf(stderr, "[tick %u] at %.3f ms: %zu rows, reserve %.3f, fill %.3f, commit %.3f, step %.3f (waiting for outputs %.3f, sinks %.3f)\n", t, at0, n, ms_reserve,
                       ms_f, ms_c, ms_s, raft.ms_waited_for_outputs - w_before, t_sinks - s_before);
      } else {
        staged.kind.resize(n), staged.group.resize(n), staged.from.resize(n), staged.id.resize(n), staged.flag.resize(n);
        fill(t, staged.kind.data(), staged.group.data(), staged.from.data(), staged.id.data(), staged.flag.data());
        t_fill += ms_since(a), a = Clock::now();
        jg_cmd_batch b{};
        b.n = n, b.kind = staged.kind.data(), b.group = staged.group.data(), b.from = staged.from.data(), b.id = staged.id.data();
        b.flag = staged.flag.data();
        loop.tcp_rx_rows(b);
        t_submit += ms_since(a), a = Clock::now();
        loop.run_until(now);
        t_step += ms_since(a);
      }
      rows_in += n;
    }
    loop.flush();  // (pipelined: the last tick's outputs)
    const Clock::time_point t_end = rv.arrive();
    finished = true;
    out.wall_ms = std::chrono::duration<double, std::milli>(t_end - t_begin).count();
    if (jg_sync(raft.raw()) != JG_OK || jg_get_counters(raft.raw(), c1) != JG_OK) throw std::runtime_error("counters");
    (void)jg_kernel_timing_read(raft.raw(), &out.k_us, &out.k_n);
    // the closed form of this stream: every leader appended one block per tick and committed the previous one
    std::vector<uint64_t> head(G), commit(G);
    std::vector<uint8_t> fault(G);
    jg_read_state(raft.raw(), JG_FIELD_HEAD, 0, head.data(), 0, G);
    jg_read_state(raft.raw(), JG_FIELD_COMMIT, 0, commit.data(), 0, G);
    jg_read_state(raft.raw(), JG_FIELD_FAULT, 0, fault.data(), 0, G);
    bool ok = true;
    for (uint32_t g = 0; g < G; g++) ok = ok && head[g] == W + T && commit[g] == W + T - 1 && fault[g] == 0;
    out.decisions = c1[1] - c0[1];
    out.down_bytes = col_bytes + fsm_rows * sizeof(jg_fsm_row) + msg_rows * sizeof(jg_msg_row);
    out.rows_in = rows_in, out.general = general, out.fsm_rows = fsm_rows, out.msg_rows = msg_rows, out.up_bytes = up_bytes, out.sink = sink;
    out.t_fill = t_fill, out.t_submit = t_submit, out.t_step = t_step, out.t_sinks = t_sinks, out.t_wait = raft.ms_waited_for_outputs;
    out.wire_bytes = wire_bytes;
    out.ok = ok;
  } catch (const std::exception& e) {
    out.ok = false, out.error = e.what();
    if (!started) rv.arrive();
    if (!finished) rv.arrive();
  }
}

// `bench_event_loop tasks-selftest`: the fork-join pool alone (no engine, no GPU): every job of every run exactly once
static int tasks_selftest() {
  Tasks tasks(4);
  uint64_t x = 1;
  for (uint32_t round = 0; round < 20000; round++) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    const uint32_t n = (uint32_t)(x >> 33) % 41;
    std::vector<std::atomic<uint32_t>> hit(n);
    for (auto& h : hit) h.store(0);
    std::atomic<uint64_t> spun{0};
    tasks.run(n, [&](uint32_t i) {
      uint64_t v = i;
      for (uint32_t k = 0; k < (round % 7) * 400u; k++) v = v * 6364136223846793005ull + k;  // (long enough for the helpers to join in)
      spun.fetch_add(v);
      hit[i].fetch_add(1);
    });
    for (uint32_t i = 0; i < n; i++)
      if (hit[i].load() != 1) {
        std::fprintf(stderr, "run %u: job %u of %u ran %u times\n", round, i, n, hit[i].load());
        return 1;
      }
  }
  std::printf("{\"ok\": true, \"threads\": %u}\n", tasks.threads());
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "tasks-selftest") return tasks_selftest();
  const uint32_t G = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 100000;
  const uint32_t R = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 5;
  const uint32_t T = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 50;
  const uint32_t W = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 10;
  const std::string mode = argc > 5 ? argv[5] : "inplace";
  const int device = argc > 6 ? std::atoi(argv[6]) : 0;
  const uint32_t L = argc > 7 ? (uint32_t)std::max(1, std::atoi(argv[7])) : 1;
  const uint32_t helpers = argc > 8 ? (uint32_t)std::max(0, std::atoi(argv[8])) : R - 1;
  const bool compact = argc > 9 && std::string(argv[9]) == "compact";  // ABI v7's bus formats (pipelined modes)
  if (G % L) {
    std::fprintf(stderr, "the partitions do not divide over %u loops\n", L);
    return 2;
  }
  Rendezvous rv(L);
  std::vector<LoopResult> res(L);
  std::vector<std::thread> th;
  for (uint32_t l = 1; l < L; l++) th.emplace_back([&, l] { run_loop(G / L, R, T, W, mode, device, 42 + l, helpers, compact, rv, res[l]); });
  run_loop(G / L, R, T, W, mode, device, 42, helpers, compact, rv, res[0]);
  for (std::thread& t : th) t.join();
  LoopResult a;
  a.ok = true;
  double k_us = 0;
  for (const LoopResult& r : res) {
    if (!r.ok) std::fprintf(stderr, "loop failed: %s\n", r.error.empty() ? "the closed form of the stream is violated" : r.error.c_str());
    a.ok = a.ok && r.ok;
    a.decisions += r.decisions, a.rows_in += r.rows_in, a.general += r.general, a.fsm_rows += r.fsm_rows, a.msg_rows += r.msg_rows;
    a.up_bytes += r.up_bytes, a.down_bytes += r.down_bytes, a.sink += r.sink, a.wire_bytes += r.wire_bytes;
    a.wall_ms = std::max(a.wall_ms, r.wall_ms);
    a.t_fill += r.t_fill / L, a.t_submit += r.t_submit / L, a.t_step += r.t_step / L, a.t_sinks += r.t_sinks / L, a.t_wait += r.t_wait / L;  // (per loop: they run side by side)
    k_us += r.k_us / L, a.k_n += r.k_n;
  }
  const uint32_t in_flight = mode.rfind("pipe", 0) == 0 && std::getenv("JG_BENCH_IN_FLIGHT") ? (uint32_t)std::max(1, std::atoi(std::getenv("JG_BENCH_IN_FLIGHT"))) : 1u;
  std::printf("{\"ok\": %s, \"mode\": \"%s\", \"ticks_in_flight\": %u, \"bus\": \"%s\", \"G\": %u, \"R\": %u, \"loops\": %u, \"task_threads_beside_each_loop\": %u, \"ticks\": %u, \"warmup\": %u, \"decisions\": %llu, \"wall_ms\": %.3f, "
              "\"decisions_per_s\": %.6g, \"ms_per_tick\": %.4f, \"ms_fill\": %.4f, \"ms_submit\": %.4f, \"ms_step_and_drain\": %.4f, \"ms_in_the_sinks\": %.4f, \"ms_waiting_for_outputs\": %.4f, "
              "\"rows_in_per_tick\": %.1f, \"rows_general\": %llu, \"fsm_rows_per_tick\": %.1f, \"msg_rows_per_tick\": %.1f, "
              "\"pcie_h2d_bytes_per_tick\": %.1f, \"pcie_d2h_bytes_per_tick\": %.1f, \"leader_kernel_us\": %.3f, \"leader_kernel_launches\": %u, "
              "\"wire_bytes_decoded_per_tick\": %.1f, \"sink\": %llu}\n",
              a.ok ? "true" : "false", mode.c_str(), in_flight, compact ? "compact (packed kind byte, 32-bit ids, common AppendEntries word, fused fsm row)" : "plain", G, R, L, mode.rfind("pipetasks", 0) == 0 ? helpers : 0u, T, W, (unsigned long long)a.decisions, a.wall_ms, a.decisions / (a.wall_ms / 1e3),
              a.wall_ms / T, a.t_fill / T, a.t_submit / T, a.t_step / T, a.t_sinks / T, a.t_wait / T, (double)a.rows_in / T, (unsigned long long)a.general,
              (double)a.fsm_rows / T, (double)a.msg_rows / T, (double)a.up_bytes / T, (double)a.down_bytes / T, k_us, a.k_n,
              (double)a.wire_bytes / T, (unsigned long long)a.sink);
  return a.ok ? 0 : 1;
}
